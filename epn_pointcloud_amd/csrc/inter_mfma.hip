// Fused InterSO3Conv kernels for gfx950 (f32 MFMA, v_mfma_f32_16x16x4_f32).
//
// Work decomposition ("a wavefront owns 16 output columns"): a column is one (b, p, a) triple,
// col = (b*p2 + p)*na + a, so the output tensor out_cl[col][o] is a plain row-major matrix.  For
// every chunk of 16 input channels a wave
//   1. regenerates the kernel-influence weights of each of its columns with ONE MFMA per 16x16 tile:
//        S[n][k] = alpha_n + beta_k + (2/sigma) g_n . (R_a kappa_k),    w = max(S, 0)
//      (= relu(1 - |g_n - R_a kappa_k|^2 / sigma), vgtk/vgtk/so3conv/functional.py:190-200, expanded),
//      whose C/D fragment IS the A fragment of the next MFMA, so w never leaves registers;
//   2. contracts over the K neighbours:  G[k][c] = sum_n w[k][n] F[idx[n], a, c]  (spconv/functional.py:384)
//      with the feature rows read straight from L2 in channels-last layout (64-byte segments);
//   3. transposes G through a wave-private LDS tile so that the 16 columns become the N dimension;
//   4. contracts over (c,k) with the BasicSO3Conv weight (so3conv/modules.py:52), W staged through LDS
//      and shared by the 4 waves of the workgroup:  out[o][col] += sum_ck W[o][ck] G[ck][col].
// Nothing of size [.., ks, K] or [.., cin, ks] is ever written to HBM.
//
// Fragment layouts (verified on hardware by tools/mfma_probe.hip): lane l, x = l & 15, j = l >> 4:
//   A[m = x][k = j],  B[k = j][n = x],  D[m = 4j + r][n = x]  (r = register 0..3).
#include <cstdlib>
#include <type_traits>

#include "conv_internal.h"
#include "gemm.h"
#include "inter_device.h"


namespace epn {

bool inter_mfma_available() { return true; }

namespace {

// ------------------------------------------------------------------------------------ forward
constexpr int WPF = 12;  // float4 registers per thread used to prefetch one W sub-chunk (cout*wk/4 <= 256*WPF)

template <int NT, int KT>
__global__ __launch_bounds__(64 * NW) void inter_fwd_kernel(InterArgs A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x = lane & 15, j = lane >> 4;
    const int ckl = 16 * A.ks;  // contraction length of one 16-channel chunk
    const int gss = ckl + 4;
    float *Gs = smem + (size_t)wave * 16 * gss;
    float *Ws = smem + (size_t)NW * 16 * gss;
    const int wss = A.wk + 4;
    const int MT = A.cout >> 4;
    const long long col0 = ((long long)blockIdx.x * NW + wave) * 16;
    const int CK = A.cin * A.ks;
    const bool fast = A.na >= 16;
    const int nsub = ckl / A.wk;
    const int nchunk = A.cin >> 4;
    const int vec_per_row = A.wk >> 2;
    const int nvec = A.cout * vec_per_row;   // float4 elements of one W sub-chunk (<= 256 * WPF)

    Seg<NT> s0, s1;
    if (fast) make_segments<NT>(A, col0, x, j, s0, s1);

    f32x4 acc[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};

    // W sub-chunk prefetch registers: element i = threadIdx.x + 256*u  ->  (o, v) = (i / vec_per_row, i % vec_per_row)
    f32x4 wpre[WPF];
    auto fetch_w = [&](int step) {   // step = ct*nsub + sub
        const int ct = step / nsub, sub = step - ct * nsub;
        const float *src = A.W + (size_t)ct * ckl + (size_t)sub * A.wk;
#pragma unroll
        for (int u = 0; u < WPF; ++u) {
            const int i = threadIdx.x + 256 * u;
            if (i < nvec) {
                const int o = i / vec_per_row, v = i - o * vec_per_row;
                wpre[u] = *reinterpret_cast<const f32x4 *>(src + (size_t)o * CK + 4 * v);
            }
        }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int u = 0; u < WPF; ++u) {
            const int i = threadIdx.x + 256 * u;
            if (i < nvec) {
                const int o = i / vec_per_row, v = i - o * vec_per_row;
                *reinterpret_cast<f32x4 *>(Ws + o * wss + 4 * v) = wpre[u];
            }
        }
    };

    fetch_w(0);
    for (int ct = 0; ct < nchunk; ++ct) {
        if (col0 < A.ncol) {
            if (fast) {
                group_segment<NT, KT>(A, s0, ct, x, j, Gs, gss);
                group_segment<NT, KT>(A, s1, ct, x, j, Gs, gss);
            } else {
                group_chunk_generic<NT, KT>(A, col0, ct, x, j, Gs, gss);
            }
        }
        for (int sub = 0; sub < nsub; ++sub) {
            __syncthreads();  // previous sub-chunk fully consumed (and this wave's Gs writes are visible)
            store_w();
            const int next = ct * nsub + sub + 1;
            if (next < nchunk * nsub) fetch_w(next);   // lands while this sub-chunk's MFMAs (and the next grouping) run
            __syncthreads();
            for (int g = 0; g < (A.wk >> 4); ++g) {
                const f32x4 bf = *reinterpret_cast<const f32x4 *>(Gs + x * gss + sub * A.wk + 16 * g + 4 * j);
#pragma unroll
                for (int m = 0; m < 16; ++m) {
                    if (m < MT) {
                        const f32x4 af = *reinterpret_cast<const f32x4 *>(Ws + (16 * m + x) * wss + 16 * g + 4 * j);
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[m] = mfma4(af[r], bf[r], acc[m]);
                    }
                }
            }
        }
    }
    const long long col = col0 + x;
    if (col < A.ncol) {
#pragma unroll
        for (int m = 0; m < 16; ++m)
            if (m < MT) *reinterpret_cast<f32x4 *>(A.out + col * A.cout + 16 * m + 4 * j) = acc[m];
    }
}

// Data-gradient tail for the columns of one segment: T[n][c] = sum_k w[n][k] dG[c,k], then
// dF[b, idx[n], a, c] += T[n][c].  dG of the wave's 16 columns sits in Gs[col][c_local*ks + k].
// DET: instead of the atomic scatter, the per-slot contributions T[b][p][n][a][c] (element type TG) are stored to the
// slab `A.out` (deterministic data gradient: inter_reduce_slots_kernel then sums them per destination in a fixed order).
template <int NT, int KT, typename TG = float, bool DET = false>
__device__ __forceinline__ void scatter_segment(const InterArgs &A, const Seg<NT> &sg, int ct, int x, int j,
                                                const TG *Gs, int gss, long long pt = 0) {
    if (sg.cnt <= 0) return;
    // transposed weights: S'[k][n] = beta_k + alpha_n + (2/sigma)(R_a kappa_k).g_n  (A = rk4 row incl. beta, B = (g,1))
    float gB[NT], alphaN[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        alphaN[t] = __shfl(sg.h.gA[t], 48 + x, 64);   // lane (x, 3) holds alpha of neighbour 16t + x
        gB[t] = j == 3 ? 1.0f : sg.h.gA[t];
    }
    float *dbase = const_cast<float *>(sg.fbase) + 16 * ct + x;   // fbase aliases grad_feats_cl in this kernel
    for (int i = 0; i < sg.cnt; ++i) {
        const int a = sg.a0 + i;
        const int jc = sg.jc0 + i;
        float rk[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) rk[kt] = A.rk4[((size_t)a * EPN_KS_MAX + 16 * kt + x) * 4 + j];
        float *drow = dbase + (size_t)a * A.cin;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f32x4 tt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                f32x4 s = {alphaN[t], alphaN[t], alphaN[t], alphaN[t]};
                s = mfma4(rk[kt], gB[t], s);  // D[m = k_local = 4j + r][n = x]
                // B operand of the contraction over k: dG[k = 16kt + 4j + r][c = x]; rows past ks carry w = 0
                const int jj = 16 * kt + 4 * j < A.ks ? j : 0;
                const f32x4 dgc = ld4f(Gs + jc * gss + x * A.ks + 16 * kt + 4 * jj);
#pragma unroll
                for (int r = 0; r < 4; ++r) tt = mfma4(relu_f(s[r]), dgc[r], tt);
            }
            // tt: lane (x = c, j), register r -> n = 16t + 4j + r
            if constexpr (DET) {
                TG *slab = reinterpret_cast<TG *>(A.out);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = 16 * t + 4 * j + r;
                    if (n < A.nn)
                        slab[(((size_t)pt * A.nn + n) * A.na + a) * A.cin + 16 * ct + x] =
                            (TG)(sg.h.ok[t][r] ? tt[r] : 0.0f);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (sg.h.mul[t][r] != 0.0f) atomicAdd(drow + sg.h.q[t][r], tt[r] * sg.h.mul[t][r]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------ backward (data)
// dG[ck][col] = sum_o W[o][ck] dOut[col][o]   (M = ck, N = col, contraction o; WT staged by o-groups)
// T[n][c]     = sum_k w[n][k] dG[c,k]         per column, then  dF[b, idx[n], a, c] += T[n][c]
template <int NT, int KT>
__global__ __launch_bounds__(64 * NW) void inter_bwd_data_kernel(InterArgs A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x = lane & 15, j = lane >> 4;
    const int ckl = 16 * A.ks;
    const int gss = ckl + 4;
    float *Gs = smem + (size_t)wave * 16 * gss;   // dG tile of this wave: [col][c_local*ks + k]
    float *Ws = smem + (size_t)NW * 16 * gss;     // WT rows [ckl][16(+4)] for one group of 16 output channels
    const int wss = 20;
    const int MTK = ckl >> 4;                     // 16-row tiles of the chunk (24 for ks = 24)
    const long long col0 = ((long long)blockIdx.x * NW + wave) * 16;
    const bool active = col0 < A.ncol;
    long long colx = col0 + x;
    colx = colx < A.ncol ? colx : A.ncol - 1;
    const bool fast = A.na >= 16;
    Seg<NT> s0, s1;
    if (fast) {
        InterArgs B = A;
        B.feats = A.out;   // segment base pointers address grad_feats_cl
        make_segments<NT>(B, col0, x, j, s0, s1);
    }

    // WT group prefetch: ckl rows x 16 floats = ckl*4 float4, element i = threadIdx.x + 256u -> (row, v) = (i>>2, i&3)
    constexpr int WTP = 8;   // ckl*4 <= 256*WTP  (ks <= 32)
    f32x4 wpre[WTP];
    f32x4 bnext;
    const int nog = A.cout >> 4, nchunk = A.cin >> 4;
    auto fetch = [&](int step) {   // step = ct*nog + og
        const int ct = step / nog, og = step - ct * nog;
#pragma unroll
        for (int u = 0; u < WTP; ++u) {
            const int i = threadIdx.x + 256 * u;
            if (i < ckl * 4)
                wpre[u] = *reinterpret_cast<const f32x4 *>(A.W + ((size_t)ct * ckl + (i >> 2)) * A.cout + 16 * og +
                                                           4 * (i & 3));
        }
        bnext = *reinterpret_cast<const f32x4 *>(A.gout + colx * A.cout + 16 * og + 4 * j);
    };
    fetch(0);
    for (int ct = 0; ct < nchunk; ++ct) {
        f32x4 dg[32];
#pragma unroll
        for (int m = 0; m < 32; ++m) dg[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int og = 0; og < nog; ++og) {
            __syncthreads();
#pragma unroll
            for (int u = 0; u < WTP; ++u) {
                const int i = threadIdx.x + 256 * u;
                if (i < ckl * 4) *reinterpret_cast<f32x4 *>(Ws + (i >> 2) * wss + 4 * (i & 3)) = wpre[u];
            }
            const f32x4 bf = bnext;
            if (ct * nog + og + 1 < nchunk * nog) fetch(ct * nog + og + 1);
            __syncthreads();
#pragma unroll
            for (int m = 0; m < 32; ++m) {
                if (m < MTK) {
                    const f32x4 af = *reinterpret_cast<const f32x4 *>(Ws + (16 * m + x) * wss + 4 * j);
#pragma unroll
                    for (int r = 0; r < 4; ++r) dg[m] = mfma4(af[r], bf[r], dg[m]);
                }
            }
        }
        // dg[m]: lane (x = col, j), register r -> ck_local = 16m + 4j + r
#pragma unroll
        for (int m = 0; m < 32; ++m)
            if (m < MTK) *reinterpret_cast<f32x4 *>(Gs + x * gss + 16 * m + 4 * j) = dg[m];
        __builtin_amdgcn_wave_barrier();
        if (!active) continue;
        if (fast) {
            scatter_segment<NT, KT>(A, s0, ct, x, j, Gs, gss);
            scatter_segment<NT, KT>(A, s1, ct, x, j, Gs, gss);
        } else {
            Seg<NT> sg;
            int last_pt = -1;
            for (int jc = 0; jc < 16; ++jc) {
                const long long col = col0 + jc;
                if (col >= A.ncol) break;
                const int a = (int)(col % A.na);
                const int pt = (int)(col / A.na);
                const int bb = pt / A.p2, pp = pt - bb * A.p2;
                if (pt != last_pt) {
                    load_hood<NT>(A, bb, pp, x, j, sg.h);
                    last_pt = pt;
                }
                sg.fbase = A.out + ((size_t)bb * A.p1) * A.na * A.cin;
                sg.a0 = a; sg.jc0 = jc; sg.cnt = 1;
                scatter_segment<NT, KT>(A, sg, ct, x, j, Gs, gss);
            }
        }
    }
}

// ------------------------------------------------------------------------------------ backward (weight)
// dW[o][ck] = sum_col dOut[col][o] G[ck][col].  A workgroup owns one 16-channel chunk (ckl columns of dW)
// and one block of up to 128 output channels, walks `col_tiles_per_wg` tiles of 64 columns, keeps its
// dW block in registers (wave w owns ck tiles w, w+4, ...) and adds it to grad_W once at the end.
template <int NT, int KT>
__global__ __launch_bounds__(64 * NW) void inter_bwd_weight_kernel(InterArgs A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x = lane & 15, j = lane >> 4;
    const int ckl = 16 * A.ks;
    const int gss = ckl + 4;
    float *Gs = smem + (size_t)wave * 16 * gss;
    const int ct = blockIdx.y;
    const int o0 = blockIdx.z * 128;
    const int MO = (A.cout - o0 < 128 ? A.cout - o0 : 128) >> 4;  // 16-row tiles of this block (<= 8)
    const int NTK = ckl >> 4;                                       // ck tiles in the chunk
    const int my_tiles = (NTK - wave + NW - 1) / NW;                // tiles wave, wave+4, ...  (<= 8)

    f32x4 acc[8][8];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 8; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    const long long tile0 = (long long)blockIdx.x * A.col_tiles_per_wg;
    for (int it = 0; it < A.col_tiles_per_wg; ++it) {
        const long long wg_col0 = (tile0 + it) * (16 * NW);
        if (wg_col0 >= A.ncol) break;
        const long long col0 = wg_col0 + wave * 16;
        __syncthreads();  // everyone done reading the previous Gs tiles
        if (col0 < A.ncol) {
            if (A.na >= 16) {
                Seg<NT> s0, s1;
                make_segments<NT>(A, col0, x, j, s0, s1);
                group_segment<NT, KT>(A, s0, ct, x, j, Gs, gss);
                group_segment<NT, KT>(A, s1, ct, x, j, Gs, gss);
            } else {
                group_chunk_generic<NT, KT>(A, col0, ct, x, j, Gs, gss);
            }
        }
        __syncthreads();
        for (int wsrc = 0; wsrc < NW; ++wsrc) {
            const long long c0 = wg_col0 + wsrc * 16;
            if (c0 >= A.ncol) break;
            const float *Gsrc = smem + (size_t)wsrc * 16 * gss;
#pragma unroll
            for (int s = 0; s < 4; ++s) {  // contraction step: columns c0 + 4s + j
                const long long col = c0 + 4 * s + j;
                const bool okc = col < A.ncol;
                float af[8];
#pragma unroll
                for (int m = 0; m < 8; ++m)
                    af[m] = (m < MO && okc) ? A.gout[col * A.cout + o0 + 16 * m + x] : 0.0f;
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    if (n < my_tiles) {
                        const float bfv = okc ? Gsrc[(4 * s + j) * gss + 16 * (wave + NW * n) + x] : 0.0f;
#pragma unroll
                        for (int m = 0; m < 8; ++m)
                            if (m < MO) acc[m][n] = mfma4(af[m], bfv, acc[m][n]);
                    }
                }
            }
        }
    }
    // acc[m][n]: lane (x = ck within tile, j), register r -> o = o0 + 16m + 4j + r
    const int CK = A.cin * A.ks;
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 8; ++n)
            if (m < MO && n < my_tiles) {
                const int ck = ct * ckl + 16 * (wave + NW * n) + x;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    atomicAdd(A.out + (size_t)(o0 + 16 * m + 4 * j + r) * CK + ck, acc[m][n][r]);
            }
}

// =====================================================================================================
// 8-wave variants (forward, weight gradient) for na >= 16, 16 < ks <= 32, nn <= 32.
// The 16-channel chunk is consumed in two passes over the kernel-point axis -- k in [0,16) and [16,ks) --
// so the wave-private transposition tile shrinks from 16 x (16*ks) to 16 x 256 floats (16.6 KB): 8 waves
// (2 per SIMD) fit beside the shared W tile in the 160 KB LDS.  The second pass's grouped features wait in
// registers (R1) while the first pass runs.  The 16 columns are statically unrolled, so the compiler
// overlaps one column's feature-row gathers with the previous columns' MFMAs.
// =====================================================================================================
constexpr int NW8 = 8;
constexpr int GS0 = 16 * 16 + 4;   // pass-0 tile row stride (floats)

template <int NT>
__device__ __forceinline__ void group16(const InterArgs &A, const Seg<NT> &s0, const Seg<NT> &s1, int ct, int x,
                                        int j, float *Gs, f32x4 (&R1)[16]) {
    int n0 = s0.cnt;
    // Opaque to the optimiser on purpose: otherwise every per-column scalar (anchor, offsets, table addresses
    // of all 16 unrolled columns) is hoisted out of the caller's channel-chunk loop and spilled.
    asm volatile("" : "+s"(n0));
    // Feature rows are gathered with buffer loads: descriptor = the segment's cloud slab (wave-uniform),
    // voffset = the neighbour's row offset + this lane's channel (column-independent, lives in the
    // neighbourhood registers), soffset = anchor*cin + chunk (wave-uniform).  Nothing per column is left for
    // the compiler to precompute and keep alive across the 16 unrolled columns.
    const unsigned slab = (unsigned)A.p1 * A.na * A.cin * 4u;
    const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(s0.fbase), 0, slab, 0x00020000);
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(s1.fbase), 0, slab, 0x00020000);
    int v0[NT][4], v1[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v0[t][r] = (s0.h.q[t][r] + x) * 4;
            v1[t][r] = (s1.h.q[t][r] + x) * 4;
        }
    // masked neighbours read row 0 and meet w == 0 (alpha = -1e30), so no select on the feature value
    auto gather = [&](int jc, float (&f)[NT][4]) {
        const bool first = jc < n0;   // wave-uniform
        const int a = first ? s0.a0 + jc : jc - n0;
        const int soff = (a * A.cin + 16 * ct) * 4;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                f[t][r] = __uint_as_float(first ? __builtin_amdgcn_raw_buffer_load_b32(r0, v0[t][r], soff, 0)
                                                : __builtin_amdgcn_raw_buffer_load_b32(r1, v1[t][r], soff, 0));
    };
    auto table = [&](int jc, float (&rk)[2], float (&beta)[2]) {
        const bool first = jc < n0;
        const int a = first ? s0.a0 + jc : jc - n0;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const float *e = A.rk4 + ((size_t)a * EPN_KS_MAX + 16 * kt + x) * 4;
            rk[kt] = e[j];
            beta[kt] = e[3];
        }
    };
    // Software pipeline of depth PD over the 16 statically unrolled columns: the feature rows and the rotated
    // kernel-point table of column jc + PD are requested before column jc is computed (L2 latency is several
    // columns' worth of MFMAs).  The arrays are indexed by compile-time constants only.
    constexpr int PD = 3;
    float f[16][NT][4], rk[16][2], beta[16][2];
#pragma unroll
    for (int jc = 0; jc < PD; ++jc) {
        gather(jc, f[jc]);
        table(jc, rk[jc], beta[jc]);
    }
#pragma unroll
    for (int jc = 0; jc < 16; ++jc) {
        if (jc + PD < 16) {
            gather(jc + PD, f[jc + PD]);
            table(jc + PD, rk[jc + PD], beta[jc + PD]);
        }
        const bool first = jc < n0;
        f32x4 g[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const float rkv = j == 3 ? 1.0f : rk[jc][kt];
            const float bt = beta[jc][kt];
            g[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                f32x4 sw = {bt, bt, bt, bt};
                sw = mfma4(first ? s0.h.gA[t] : s1.h.gA[t], rkv, sw);
#pragma unroll
                for (int r = 0; r < 4; ++r) g[kt] = mfma4(relu_f(sw[r]), f[jc][t][r], g[kt]);
            }
        }
        *reinterpret_cast<f32x4 *>(Gs + jc * GS0 + x * 16 + 4 * j) = g[0];   // k = 4j + r of channel x
        R1[jc] = g[1];                                                      // k = 16 + 4j + r (valid while < ks)
        __builtin_amdgcn_sched_barrier(0);   // keep the compiler from hoisting all 16 columns' gathers (spills)
    }
}

__device__ __forceinline__ void spill_pass1(float *Gs, const f32x4 (&R1)[16], int kw1, int gs1, int x, int j) {
    if (4 * j < kw1) {
#pragma unroll
        for (int jc = 0; jc < 16; ++jc) *reinterpret_cast<f32x4 *>(Gs + jc * gs1 + x * kw1 + 4 * j) = R1[jc];
    }
}

template <int NT, int MTMAX>
__global__ __launch_bounds__(64 * NW8) void inter_fwd8_kernel(InterArgs A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x = lane & 15, j = lane >> 4;
    float *Gs = smem + (size_t)wave * 16 * GS0;
    float *Ws = smem + (size_t)NW8 * 16 * GS0;
    const int wss = A.wk + 4;
    constexpr int MT = MTMAX;   // the launcher only takes cout == 16 * MTMAX: straight-line MFMA code, no per-tile branch
    const int kw1 = A.ks - 16, gs1 = 16 * kw1 + 4;
    const long long col0 = ((long long)epn_xcd_tile(blockIdx.x, gridDim.x) * NW8 + wave) * 16;
    const int CK = A.cin * A.ks;
    const int n0s = 256 / A.wk, n1s = (16 * kw1) / A.wk, spc = n0s + n1s;   // W sub-chunks per pass / per chunk
    const int nchunk = A.cin >> 4;
    const int vpr = A.wk >> 2, nvec = A.cout * vpr;

    Seg<NT> s0, s1;
    make_segments<NT>(A, col0, x, j, s0, s1);

    f32x4 acc[MTMAX];
#pragma unroll
    for (int m = 0; m < MTMAX; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr int WPF8 = 4;   // cout*wk/4 <= 512*WPF8
    f32x4 wpre[WPF8];
    auto fetch_w = [&](int step) {
        const int ct = step / spc, rem = step - ct * spc;
        const int pass = rem >= n0s ? 1 : 0;
        const int sub = pass ? rem - n0s : rem;
        const int kw = pass ? kw1 : 16;
#pragma unroll
        for (int u = 0; u < WPF8; ++u) {
            const int i = threadIdx.x + 64 * NW8 * u;
            if (i < nvec) {
                const int o = i / vpr, v = i - o * vpr;
                const int L = sub * A.wk + 4 * v;          // position in the pass's contraction axis
                const int cl = L / kw, kk = L - cl * kw;   // (channel, kernel point) of that position
                wpre[u] = *reinterpret_cast<const f32x4 *>(A.W + (size_t)o * CK + (size_t)(16 * ct + cl) * A.ks +
                                                           16 * pass + kk);
            }
        }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int u = 0; u < WPF8; ++u) {
            const int i = threadIdx.x + 64 * NW8 * u;
            if (i < nvec) {
                const int o = i / vpr, v = i - o * vpr;
                *reinterpret_cast<f32x4 *>(Ws + o * wss + 4 * v) = wpre[u];
            }
        }
    };
    auto contract = [&](int nsub, int gstride, int step0) {
        for (int sub = 0; sub < nsub; ++sub) {
            __syncthreads();   // Ws free again; this wave's tile writes are ordered before its reads
            store_w();
            if (step0 + sub + 1 < nchunk * spc) fetch_w(step0 + sub + 1);
            __syncthreads();
            // software-pipelined over the (g, m) sequence: the W fragment of the next step (and the G fragment of
            // the next g) are read from LDS while the current step's four MFMAs issue
            const int ng = A.wk >> 4;
            const float *gsrc = Gs + x * gstride + sub * A.wk + 4 * j;
            const float *wsrc = Ws + x * wss + 4 * j;
            f32x4 bf = *reinterpret_cast<const f32x4 *>(gsrc);
            f32x4 af = *reinterpret_cast<const f32x4 *>(wsrc);
            for (int g = 0; g < ng; ++g) {
                const int gn = g + 1 < ng ? g + 1 : g;
                const f32x4 bfn = *reinterpret_cast<const f32x4 *>(gsrc + 16 * gn);
#pragma unroll
                for (int m = 0; m < MTMAX; ++m) {
                    if (m < MT) {
                        const bool last = m + 1 == MT;
                        const f32x4 afn = *reinterpret_cast<const f32x4 *>(
                            wsrc + (last ? 16 * gn : 16 * g) + (last ? 0 : 16 * (m + 1)) * wss);
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[m] = mfma4(af[r], bf[r], acc[m]);
                        af = afn;
                    }
                }
                bf = bfn;
            }
        }
    };

    fetch_w(0);
    for (int ct = 0; ct < nchunk; ++ct) {
        f32x4 R1[16];
        group16<NT>(A, s0, s1, ct, x, j, Gs, R1);
        contract(n0s, GS0, ct * spc);
        spill_pass1(Gs, R1, kw1, gs1, x, j);   // own tile: LDS ops of one wave execute in order
        contract(n1s, gs1, ct * spc + n0s);
    }
    const long long col = col0 + x;
    if (col < A.ncol) {
#pragma unroll
        for (int m = 0; m < MTMAX; ++m)
            if (m < MT) *reinterpret_cast<f32x4 *>(A.out + col * A.cout + 16 * m + 4 * j) = acc[m];
    }
}

// dW[o][ck] = sum_col dOut[col][o] G[ck][col]: workgroup = (one 16-channel chunk, one block of <= 128 output
// channels, a run of 128-column tiles).  Pass 0 (k < 16): 16 ck tiles, wave w owns channels 2w, 2w+1; pass 1
// (k >= 16): 16*kw1/16 tiles, wave w owns tile w (if it exists).
template <int NT, int MO>   // MO = row tiles of the output-channel block (2, 4 or 8); ks == 24 (launcher)
__global__ __launch_bounds__(64 * NW8) void inter_bwd_weight8_kernel(InterArgs A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x = lane & 15, j = lane >> 4;
    float *Gs = smem + (size_t)wave * 16 * GS0;
    constexpr int kw1 = 8, gs1 = 16 * kw1 + 4;
    const int ct = blockIdx.y;
    const int o0 = blockIdx.z * 128;
    // Tile ownership: wave = (one row tile mi, one group ng of ck positions), so ONE dOut fragment per contraction
    // step feeds all of the wave's MFMAs, and the G fragments of four tiles come from one ds_read_b128: tile t of a
    // 64-position quad q covers the positions {64q + 4i + t}.
    constexpr int msplit = MO, nsplit = NW8 / MO;         // nsplit in {1, 2, 4}
    const int mi = wave % msplit, ng = wave / msplit;
    constexpr int q0n = 4 / nsplit;                       // pass-0 quads per wave (4, 2, 1)
    constexpr int nq1 = (16 * kw1) / 64;                  // pass-1 quads in total (2 for ks = 24)
    constexpr int q1n = (nq1 + nsplit - 1) / nsplit;      // per wave

    f32x4 acc0[q0n][4], acc1[q1n][4];   // [quad][tile in quad]
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int q = 0; q < q0n; ++q) acc0[q][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < q1n; ++q) acc1[q][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const long long tile0 = (long long)blockIdx.x * A.col_tiles_per_wg;
    for (int it = 0; it < A.col_tiles_per_wg; ++it) {
        const long long wg_col0 = (tile0 + it) * (16 * NW8);
        if (wg_col0 >= A.ncol) break;
        const long long col0 = wg_col0 + wave * 16;
        Seg<NT> s0, s1;
        make_segments<NT>(A, col0, x, j, s0, s1);
        f32x4 R1[16];
        __syncthreads();   // all waves finished reading the previous tiles
        group16<NT>(A, s0, s1, ct, x, j, Gs, R1);
        __syncthreads();
        // dOut fragments of source tile wsrc + 1 are requested while tile wsrc is contracted (global latency ~ one tile)
        auto load_af = [&](int wsrc, float (&af)[4]) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const long long col = wg_col0 + wsrc * 16 + 4 * s + j;
                af[s] = col < A.ncol ? A.gout[col * A.cout + o0 + 16 * mi + x] : 0.0f;
            }
        };
        float afc[4], afn[4];
        load_af(0, afc);
        for (int wsrc = 0; wsrc < NW8; ++wsrc) {
            const long long c0 = wg_col0 + wsrc * 16;
            if (c0 >= A.ncol) break;
            load_af(wsrc + 1 < NW8 ? wsrc + 1 : wsrc, afn);
            const float *Gsrc = smem + (size_t)wsrc * 16 * GS0;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bool okc = c0 + 4 * s + j < A.ncol;
#pragma unroll
                for (int q = 0; q < q0n; ++q) {
                    const f32x4 bf = *reinterpret_cast<const f32x4 *>(Gsrc + (4 * s + j) * GS0 +
                                                                      64 * (ng * q0n + q) + 4 * x);
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc0[q][t] = mfma4(afc[s], okc ? bf[t] : 0.0f, acc0[q][t]);
                }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) afc[s] = afn[s];
        }
        __syncthreads();
        spill_pass1(Gs, R1, kw1, gs1, x, j);
        __syncthreads();
        load_af(0, afc);
        for (int wsrc = 0; wsrc < NW8; ++wsrc) {
            const long long c0 = wg_col0 + wsrc * 16;
            if (c0 >= A.ncol) break;
            load_af(wsrc + 1 < NW8 ? wsrc + 1 : wsrc, afn);
            const float *Gsrc = smem + (size_t)wsrc * 16 * GS0;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bool okc = c0 + 4 * s + j < A.ncol;
#pragma unroll
                for (int q = 0; q < q1n; ++q) {
                    const int qq = ng * q1n + q;
                    if (qq < nq1) {
                        const f32x4 bf = *reinterpret_cast<const f32x4 *>(Gsrc + (4 * s + j) * gs1 + 64 * qq + 4 * x);
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc1[q][t] = mfma4(afc[s], okc ? bf[t] : 0.0f, acc1[q][t]);
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) afc[s] = afn[s];
        }
    }
    // lane (x, j), register r -> o = o0 + 16*mi + 4j + r;  tile (q, t), lane x -> position p = 64*quad + 4x + t;
    // pass 0: channel p / 16, k = p % 16;   pass 1: channel p / kw1, k = 16 + p % kw1
    const int CK = A.cin * A.ks;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int q = 0; q < q0n; ++q) {
            const int p = 64 * (ng * q0n + q) + 4 * x + t;
            const int ck0 = (16 * ct + (p >> 4)) * A.ks + (p & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                atomicAdd(A.out + (size_t)(o0 + 16 * mi + 4 * j + r) * CK + ck0, acc0[q][t][r]);
        }
#pragma unroll
        for (int q = 0; q < q1n; ++q) {
            const int qq = ng * q1n + q;
            if (qq < nq1) {
                const int p = 64 * qq + 4 * x + t;
                const int cl = p / kw1, kk = p - cl * kw1;
                const int ck1 = (16 * ct + cl) * A.ks + 16 + kk;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    atomicAdd(A.out + (size_t)(o0 + 16 * mi + 4 * j + r) * CK + ck1, acc1[q][t][r]);
            }
        }
    }
}

// Data gradient, 8 waves: a workgroup owns 4 column tiles, TWO waves per tile.  Each wave computes half of the
// dG rows of its tile (dG = W^T dOut), the pair meets at a barrier, then each wave runs the per-column tail
// (regenerated weights, contraction over k, atomic scatter) for 8 of the 16 columns.  Same work as the 4-wave
// kernel, but 2 waves per SIMD and W^T staged once for 64 columns.
template <int NT, int KT, int MH>   // MH = dG row tiles per wave = ks/2 (launcher: ks even), compile-time -> branch-free MFMA block
__global__ __launch_bounds__(64 * NW8) void inter_bwd_data8_kernel(InterArgs A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = wave >> 1, half = wave & 1;
    const int x = lane & 15, j = lane >> 4;
    const int ckl = 16 * A.ks;
    const int gss = ckl + 4;
    float *Gs = smem + (size_t)tile * 16 * gss;
    float *Ws = smem + (size_t)4 * 16 * gss;
    const int wss = 20;
    const int m0 = half * MH;
    const long long col0 = ((long long)epn_xcd_tile(blockIdx.x, gridDim.x) * 4 + tile) * 16;
    const bool active = col0 < A.ncol;
    long long colx = col0 + x;
    colx = colx < A.ncol ? colx : A.ncol - 1;

    Seg<NT> s0, s1;
    {
        InterArgs B = A;
        B.feats = A.out;   // segment base pointers address grad_feats_cl
        make_segments<NT>(B, col0, x, j, s0, s1);
    }
    // this wave's 8 columns [8*half, 8*half + 8) as (at most) two sub-segments
    const int lo = 8 * half, hi = lo + 8;
    {
        const int e0 = s0.cnt < hi ? s0.cnt : hi;          // s0 covers [0, s0.cnt)
        const int b0 = lo < e0 ? lo : e0;
        const int c0n = e0 - b0;
        const int tot = s0.cnt + s1.cnt;
        const int b1 = lo > s0.cnt ? lo : s0.cnt;          // s1 covers [s0.cnt, tot)
        const int e1 = hi < tot ? hi : tot;
        const int c1n = e1 > b1 ? e1 - b1 : 0;
        s1.a0 = b1 - s0.cnt; s1.jc0 = b1; s1.cnt = c1n;
        s0.a0 = s0.a0 + b0; s0.jc0 = b0; s0.cnt = c0n > 0 ? c0n : 0;
    }

    constexpr int WTP8 = 4;   // ckl*4 <= 512*WTP8
    f32x4 wpre[WTP8];
    f32x4 bnext;
    const int nog = A.cout >> 4, nchunk = A.cin >> 4;
    auto fetch = [&](int step) {
        const int ct = step / nog, og = step - ct * nog;
#pragma unroll
        for (int u = 0; u < WTP8; ++u) {
            const int i = threadIdx.x + 64 * NW8 * u;
            if (i < ckl * 4)
                wpre[u] = *reinterpret_cast<const f32x4 *>(A.W + ((size_t)ct * ckl + (i >> 2)) * A.cout + 16 * og +
                                                           4 * (i & 3));
        }
        bnext = *reinterpret_cast<const f32x4 *>(A.gout + colx * A.cout + 16 * og + 4 * j);
    };
    fetch(0);
    for (int ct = 0; ct < nchunk; ++ct) {
        f32x4 dg[MH];
#pragma unroll
        for (int m = 0; m < MH; ++m) dg[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int og = 0; og < nog; ++og) {
            __syncthreads();
#pragma unroll
            for (int u = 0; u < WTP8; ++u) {
                const int i = threadIdx.x + 64 * NW8 * u;
                if (i < ckl * 4) *reinterpret_cast<f32x4 *>(Ws + (i >> 2) * wss + 4 * (i & 3)) = wpre[u];
            }
            const f32x4 bf = bnext;
            if (ct * nog + og + 1 < nchunk * nog) fetch(ct * nog + og + 1);
            __syncthreads();
#pragma unroll
            for (int m = 0; m < MH; ++m) {
                const f32x4 af = *reinterpret_cast<const f32x4 *>(Ws + (16 * (m0 + m) + x) * wss + 4 * j);
#pragma unroll
                for (int r = 0; r < 4; ++r) dg[m] = mfma4(af[r], bf[r], dg[m]);
            }
        }
#pragma unroll
        for (int m = 0; m < MH; ++m) *reinterpret_cast<f32x4 *>(Gs + x * gss + 16 * (m0 + m) + 4 * j) = dg[m];
        __syncthreads();   // both halves of every tile are in LDS
        if (active) {
            scatter_segment<NT, KT>(A, s0, ct, x, j, Gs, gss);
            scatter_segment<NT, KT>(A, s1, ct, x, j, Gs, gss);
        }
        // the next chunk's first barrier (og = 0) orders these reads before the tile is overwritten
    }
}

// ------------------------------------------------------------------------------------ grouping only
// "Split" path: the grouped features G[col][c*ks + k] go to HBM once (kept for the backward pass) and the weight
// contraction is a plain [cols x cin*ks] x [cin*ks x cout] GEMM for the BLAS library (measured 110-150 TFLOP/s fp32 on
// these shapes in round 1 -- CHANGELOG 3.2 -- against 67-84 for the fused kernels above).  A wave owns one 16-column tile; no
// LDS, no barriers: weight generation + neighbour contraction exactly as in the fused kernels, the D fragment of
// the contraction (4 consecutive kernel points of one channel) is one 16-byte global store.
template <int NT, int KT, typename TF>
__global__ __launch_bounds__(64 * NW) void inter_group_kernel(InterArgs A) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x = lane & 15, j = lane >> 4;
    const long long col0 = ((long long)epn_xcd_tile(blockIdx.x, gridDim.x) * NW + wave) * 16;
    if (col0 >= A.ncol) return;
    const int gss = A.cin * A.ks;
    TF *G = reinterpret_cast<TF *>(A.out) + (size_t)col0 * gss;
    // blockIdx.y walks the 16-channel chunks in groups of col_tiles_per_wg (here: chunks per workgroup): with one
    // chunk per launch row, all tiles of a cloud touch the same 64-byte slice of every feature row at about the same
    // time, so the working set per cloud (p1*na*64 B) fits the XCD's L2
    const int ct0 = blockIdx.y * A.col_tiles_per_wg;
    const int ct1 = min(ct0 + A.col_tiles_per_wg, A.cin >> 4);
    if (A.na >= 16) {
        Seg<NT> s0, s1;
        make_segments<NT, TF>(A, col0, x, j, s0, s1);
        for (int ct = ct0; ct < ct1; ++ct) {
            group_segment<NT, KT, TF>(A, s0, ct, x, j, G + 16 * ct * A.ks, gss);
            group_segment<NT, KT, TF>(A, s1, ct, x, j, G + 16 * ct * A.ks, gss);
        }
    } else {
        for (int ct = ct0; ct < ct1; ++ct)
            group_chunk_generic<NT, KT, TF>(A, col0, ct, x, j, G + 16 * ct * A.ks, gss);
    }
}

// Wide-gather form of the grouping (na >= 16, cin % (16 CG) == 0).  In inter_group_kernel a lane owns ONE channel of a
// 16-channel chunk: every gather instruction moves 4 (fp32) or 2 (bf16) bytes per lane -- 64-byte segments, one address
// per lane through the texture addresser, K of them per column and chunk -- and the kernel-influence weights (S-MFMAs,
// table reads, relu, packing) are regenerated for every chunk.  Here a lane owns CG CONSECUTIVE channels of a group of
// 16 CG channels (CG = 4: 64 channels): one dwordx4 (fp32) / dwordx2 (bf16) load per neighbour row and lane -- 256 / 128
// contiguous bytes per 16 lanes -- serves CG chunks, and the weights of a column are generated once for all of them.
// "Chunk" e of the group is the channel set {16 CG g + CG x + e}: only the lane -> channel map changes, the layout of G
// does not (a D fragment is still four consecutive kernel points of one channel = one 16 / 8-byte store).
// amdgpu_waves_per_eu(2): a register budget of 256 (no instance with K <= 64 needs more) -- with the 512 that
// __launch_bounds__(256) alone allows, hipcc places the MFMA results in AGPRs and copies every one of them back
// (v_accvgpr_read_b32: 64 per column at K = 64, the largest single VALU item of the kernel)
#ifndef EPN_G_STORE_AUX
#define EPN_G_STORE_AUX 2     // cache policy of the stores of G (gfx940+: bit 0 = sc0, bit 1 = nt, bit 4 = sc1)
#endif
template <int NT, int KT, typename TF, int CG>
#ifdef EPN_GRP_NO_WPE
__global__ __launch_bounds__(64 * NW)
#else
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NT <= 4 ? 2 : 1)))
#endif
void inter_group_wide_kernel(InterArgs A) {
    constexpr bool BF = sizeof(TF) == 2;
    typedef unsigned uvec __attribute__((ext_vector_type(BF ? CG / 2 : CG)));   // one lane's CG channels of one row
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x = lane & 15, j = lane >> 4;
    const long long col0 = ((long long)epn_xcd_tile(blockIdx.x, gridDim.x) * NW + wave) * 16;
    if (col0 >= A.ncol) return;
    const int gss = A.cin * A.ks;
    TF *G = reinterpret_cast<TF *>(A.out) + (size_t)col0 * gss;
    const int coff = (int)blockIdx.y * 16 * CG + CG * x;       // first of this lane's channels
    Seg<NT> seg[2];
    make_segments<NT, TF>(A, col0, x, j, seg[0], seg[1]);
    for (int si = 0; si < 2; ++si) {
        const Seg<NT> &sg = seg[si];
        if (sg.cnt <= 0) continue;
        // store offsets (bytes) of the lane's D fragments: [kt] = lane part (0x80000000 = no such kernel points),
        // est[kt] = step from chunk e to e + 1
        unsigned vst[KT];
        int est[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            int o = coff * A.ks + 16 * kt + 4 * j;
            est[kt] = A.ks * (int)sizeof(TF);
            if (A.packed) {
                const int s0 = (int)blockIdx.y * 16 * CG + x, w0 = A.ks < 16 ? A.ks : 16;   // slot of the lane's channel e = 0
                o = kt == 0 ? s0 * w0 + 4 * j : w0 * A.cin + s0 * (A.ks - 16) + 4 * j;
                est[kt] = 16 * (kt == 0 ? w0 : A.ks - 16) * (int)sizeof(TF);
            }
            vst[kt] = 16 * kt + 4 * j < A.ks ? (unsigned)o * (unsigned)sizeof(TF) : 0x80000000u;
        }
        // Gathers, table reads and stores as BUFFER instructions (round 4): descriptor = the cloud's feature block / the
        // wave's 16 rows of G in SGPRs, a per-lane 32-bit byte offset that does not change from column to column (neighbour
        // row + channel; kernel points of the lane), a wave-uniform scalar offset per column (anchor; row of G + chunk).
        // The flat form paid one 64-bit address add per gather and per store, and an exec-mask branch around every store of
        // the lanes without kernel points (ks = 24: lane groups j >= 2 of the second 16) -- ~45 % of the kernel's VALU + SALU
        // instructions at K = 64, where only two waves fit a SIMD; a lane without kernel points now stores out of range,
        // which the buffer bounds check drops.
        const unsigned fbytes = (unsigned)A.p1 * (unsigned)A.na * (unsigned)A.cin * (unsigned)sizeof(TF);
        const __amdgpu_buffer_rsrc_t rF = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(sg.fbase), 0, (int)fbytes, 0x00020000);
        const unsigned gbytes = 16u * (unsigned)gss * (unsigned)sizeof(TF);
        const __amdgpu_buffer_rsrc_t rG = __builtin_amdgcn_make_buffer_rsrc(G, 0, (int)gbytes, 0x00020000);
        unsigned vq[NT][4];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) vq[t][r] = ((unsigned)sg.h.q[t][r] + (unsigned)coff) * (unsigned)sizeof(TF);
        auto gather = [&](int a, uvec (&f)[NT][4]) {
            const unsigned soff = (unsigned)a * (unsigned)A.cin * (unsigned)sizeof(TF);       // wave-uniform
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr (sizeof(uvec) == 16)
                        f[t][r] = __builtin_bit_cast(uvec, __builtin_amdgcn_raw_buffer_load_b128(rF, vq[t][r], soff, 0));
                    else if constexpr (sizeof(uvec) == 8)
                        f[t][r] = __builtin_bit_cast(uvec, __builtin_amdgcn_raw_buffer_load_b64(rF, vq[t][r], soff, 0));
                    else
                        f[t][r] = __builtin_bit_cast(uvec, __builtin_amdgcn_raw_buffer_load_b32(rF, vq[t][r], soff, 0));
                }
        };
        const __amdgpu_buffer_rsrc_t rT = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A.rk4), 0,
                                                                            A.na * EPN_KS_MAX * 16, 0x00020000);
        auto load_rk = [&](int a, RkRow<KT> &r) {
            const unsigned soff = (unsigned)a * (EPN_KS_MAX * 16u);
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                const unsigned v = (unsigned)(16 * kt + x) * 16u;
                r.rk[kt] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rT, v + 4u * (unsigned)j, soff, 0));
                r.beta[kt] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rT, v + 12u, soff, 0));
            }
        };
        auto store_g = [&](unsigned voff, unsigned soff, f32x4 g) {
#ifdef EPN_GRP_FLAT_ST
            if (voff != 0x80000000u) st4f(reinterpret_cast<TF *>(reinterpret_cast<char *>(G) + soff + voff), g);
            return;
#endif
            // The whole offset goes into the VGPR offset, the scalar offset stays the literal 0.  With the row offset in an
            // SGPR (`buffer_store_dwordx4 v[66:69], v59, s[4:7], s52 offen`) hipcc assumes the hardware interlocks a later
            // write of the data registers (GCNHazardRecognizer: "this hazard only exists if the instruction is not using a
            // register in the soffset field") and scheduled `v_cndmask_b32 v66, ...` two instructions behind the store; on
            // gfx950 the store then delivered the NEW v66 in lanes 12-15 / 28-31 of ~0.3 % of the rows of a full-size layer
            // (found with the VGPR-form MFMAs of this round, whose register allocation put such a pair together; the ISA scan
            // is in tools/isa_hazards.py).  One v_add_u32 per store buys the documented wait state.
            const unsigned off = voff + soff;
            if constexpr (BF) {
                typedef unsigned u32x2_s __attribute__((ext_vector_type(2)));
                const bf16x4_t b = {(__bf16)g[0], (__bf16)g[1], (__bf16)g[2], (__bf16)g[3]};
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_s, b), rG, off, 0, EPN_G_STORE_AUX);
            } else {
                typedef unsigned u32x4_s __attribute__((ext_vector_type(4)));
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_s, g), rG, off, 0, EPN_G_STORE_AUX);
            }
        };
        // One column: request column i + 1's feature rows and kernel-table entries into (fnext, rnext), then compute column
        // i from (fcur, rcur).  The two register sets swap ROLES between consecutive columns (the loop below is unrolled by
        // two): a copy "cur = next" at the end of an iteration would have to wait for the loads it copies -- with the
        // in-order counter that is s_waitcnt vmcnt(0), i.e. for every store of the column as well.  (Measured further:
        // branch-free buffer stores + counted dummy stores in the preheader bring hipcc's waits to the exact vmcnt(8..15) --
        // no store ever waited for -- and change nothing: 7.72-7.81 vs 7.78-7.79 ms per cls step, the other waves of the
        // SIMD already cover what is left.)
        auto column = [&](int i, const uvec (&fcur)[NT][4], const RkRow<KT> &rcur, uvec (&fnext)[NT][4], RkRow<KT> &rnext) {
            const int a = sg.a0 + i;
            const int an = i + 1 < sg.cnt ? a + 1 : a;          // last column re-reads its own rows (cache hit, unused)
            load_rk(an, rnext);
#ifdef EPN_TUNING
            if (!(A.wk & 2) || i == 0)
#endif
            gather(an, fnext);
            f32x4 w[KT][NT];
            make_weights_from<NT, KT>(rcur, j, sg.h, w);         // once per column, for all CG chunks
            const unsigned rowoff = (unsigned)(sg.jc0 + i) * (unsigned)gss * (unsigned)sizeof(TF);   // wave-uniform
            // bf16: the A fragments (weights rounded to bf16) once per column -- round 3 packed them again for every chunk e
            // (ISA: 64 v_cvt_pk_bf16_f32 per column instead of 16) -- and the B fragments assembled with byte permutes: the
            // fragment of chunk e holds channel e of FOUR gathered rows (r = 0..3), i.e. one 16-bit half of two dwords each:
            // v_perm_b32 picks both halves in one instruction (was: shift / and / select / or per value, ~14 VALU per fragment,
            // now 2 + 2 for the validity mask).  Masked slots also carry weight 0 (alpha = -1e30 in load_hood); the AND keeps
            // a non-finite value in feature row 0 from turning 0 * x into NaN.
            typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
            bf16x4_t wp[BF ? KT : 1][BF ? NT : 1];
            unsigned okm[BF ? NT : 1][2];
            if constexpr (BF) {
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                    for (int t = 0; t < NT; ++t) wp[kt][t] = pack4(w[kt][t]);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    okm[t][0] = (sg.h.ok[t][0] ? 0xffffu : 0u) | (sg.h.ok[t][1] ? 0xffff0000u : 0u);
                    okm[t][1] = (sg.h.ok[t][2] ? 0xffffu : 0u) | (sg.h.ok[t][3] ? 0xffff0000u : 0u);
                }
            }
            f32x4 gprev[KT], gcur[KT];
#pragma unroll
            for (int e = 0; e < CG; ++e) {
                if constexpr (BF) {
                    bf16x4_t fb4[NT];
                    constexpr unsigned SEL_LO = 0x05040100u, SEL_HI = 0x07060302u;   // D = {S1.half, S0.half}: S1 = low result half
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const unsigned sel = (e & 1) ? SEL_HI : SEL_LO;
                        const unsigned p01 = __builtin_amdgcn_perm(fcur[t][1][e >> 1], fcur[t][0][e >> 1], sel) & okm[t][0];
                        const unsigned p23 = __builtin_amdgcn_perm(fcur[t][3][e >> 1], fcur[t][2][e >> 1], sel) & okm[t][1];
                        fb4[t] = __builtin_bit_cast(bf16x4_t, u32x2_t{p01, p23});
                    }
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) {
                        f32x4 g = {0.f, 0.f, 0.f, 0.f};
                        if constexpr (NT % 2 == 0) {
#pragma unroll
                            for (int t = 0; t < NT; t += 2)
                                g = mfma_bf16_k32(wp[kt][t], wp[kt][t + 1], fb4[t], fb4[t + 1], g);
                        } else {
#pragma unroll
                            for (int t = 0; t < NT; ++t) g = mfma_bf16_k16(wp[kt][t], fb4[t], g);
                        }
                        gcur[kt] = g;
                    }
                } else {
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) {
                        f32x4 g = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int t = 0; t < NT; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
#ifdef EPN_TUNING
                                if (A.wk & 4) { g[r] += w[kt][t][r] + __uint_as_float(fcur[t][r][e]); continue; }
#endif
                                g = mfma4(w[kt][t][r], sg.h.ok[t][r] ? __uint_as_float(fcur[t][r][e]) : 0.0f, g);
                            }
                        gcur[kt] = g;
                    }
                }
                // The stores of a chunk are issued one chunk LATE, behind the next chunk's MFMAs (and 32 idle cycles after the
                // last one).  Measured on gfx950 with this toolchain: a buffer_store whose data registers are the direct
                // (VGPR-form) result of the MFMA just before it reads dword 0 of lanes 12-15 / 28-31 -- the last
                // write-back pass -- too early in ~0.3 % of the rows of a full-size layer (tools/scratch history);
                // the AGPR form (a v_accvgpr_read in between) and flat stores under a branch never showed it.  Distance is
                // the cure the ISA manual prescribes for XDL-write -> VMEM-read hazards; here it is made explicit.
                __builtin_amdgcn_sched_barrier(0);
                if (e > 0) {
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt)
#ifdef EPN_TUNING      // tools/group_ablation.py: 1 = no stores (kept alive by an impossible value), 2 = gathers once, 4 = no MFMAs
                        if (!(A.wk & 1) || gprev[kt][0] == 12345.678f)
#endif
                        store_g(vst[kt], rowoff + (unsigned)((e - 1) * est[kt]), gprev[kt]);   // vst = 0x80000000: out of range, dropped
                    asm volatile("s_nop 4" ::: "memory");     // nothing rewrites the data registers for five more cycles
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) gprev[kt] = gcur[kt];
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#ifdef EPN_TUNING
                if (!(A.wk & 1) || gprev[kt][0] == 12345.678f)
#endif
                store_g(vst[kt], rowoff + (unsigned)((CG - 1) * est[kt]), gprev[kt]);
            asm volatile("s_nop 4" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        uvec f0[NT][4], f1[NT][4];
        RkRow<KT> r0, r1;
        load_rk(sg.a0, r0);
        gather(sg.a0, f0);
        for (int i = 0; i < sg.cnt; i += 2) {
            column(i, f0, r0, f1, r1);
            if (i + 1 < sg.cnt) column(i + 1, f1, r1, f0, r0);
        }
    }
}

// Transpose of the grouping: dF[b, idx[n], a, c] += sum_k w[k][n] dG[col][c*ks + k]  (scatter_segment reads dG from HBM).
template <int NT, int KT, typename TG>   // dG in TG (float / bf16); the scatter target grad_feats is always fp32
__global__ __launch_bounds__(64 * NW) void inter_ungroup_kernel(InterArgs A) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x = lane & 15, j = lane >> 4;
    const long long col0 = ((long long)epn_xcd_tile(blockIdx.x, gridDim.x) * NW + wave) * 16;
    if (col0 >= A.ncol) return;
    const int gss = A.cin * A.ks;
    const TG *dG = reinterpret_cast<const TG *>(A.gout) + (size_t)col0 * gss;
    const int ct0 = blockIdx.y * A.col_tiles_per_wg;          // chunk-major launch order, see inter_group_kernel
    const int ct1 = min(ct0 + A.col_tiles_per_wg, A.cin >> 4);
    InterArgs B = A;
    B.feats = A.out;   // segment base pointers address grad_feats_cl
    if (A.na >= 16) {
        Seg<NT> s0, s1;
        make_segments<NT>(B, col0, x, j, s0, s1);
        for (int ct = ct0; ct < ct1; ++ct) {
            scatter_segment<NT, KT, TG>(A, s0, ct, x, j, dG + 16 * ct * A.ks, gss);
            scatter_segment<NT, KT, TG>(A, s1, ct, x, j, dG + 16 * ct * A.ks, gss);
        }
    } else {
        Seg<NT> sg;
        int last_pt = -1;
        for (int jc = 0; jc < 16; ++jc) {
            const long long col = col0 + jc;
            if (col >= A.ncol) break;
            const int a = (int)(col % A.na);
            const int pt = (int)(col / A.na);
            const int bb = pt / A.p2, pp = pt - bb * A.p2;
            if (pt != last_pt) {
                load_hood<NT>(A, bb, pp, x, j, sg.h);
                last_pt = pt;
            }
            sg.fbase = A.out + ((size_t)bb * A.p1) * A.na * A.cin;
            sg.a0 = a; sg.jc0 = jc; sg.cnt = 1;
            for (int ct = ct0; ct < ct1; ++ct) scatter_segment<NT, KT, TG>(A, sg, ct, x, j, dG + 16 * ct * A.ks, gss);
        }
    }
}

// Spatial order of the output points of each cloud (Morton code of the centre, 10 bits per axis over the cloud's
// bounding box): consecutive points in this order have heavily overlapping neighbourhoods, which is what
// inter_ungroup_shared_kernel exploits.  One workgroup per cloud, rank sort in LDS (p2 <= MORTON_MAX).
constexpr int MORTON_MAX = 4096;
__device__ __forceinline__ unsigned spread3(unsigned v) {   // 10 bits -> every third bit
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
__global__ __launch_bounds__(1024) void morton_order_kernel(const float *__restrict__ new_xyz, int p2,
                                                            int32_t *__restrict__ order) {
    __shared__ unsigned code[MORTON_MAX];
    __shared__ float red[6][16];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float *c = new_xyz + (size_t)b * 3 * p2;
    float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
    for (int i = tid; i < p2; i += 1024)
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            const float v = c[ax * p2 + i];
            lo[ax] = fminf(lo[ax], v); hi[ax] = fmaxf(hi[ax], v);
        }
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            lo[ax] = fminf(lo[ax], __shfl_xor(lo[ax], o, 64));
            hi[ax] = fmaxf(hi[ax], __shfl_xor(hi[ax], o, 64));
        }
        if ((tid & 63) == 0) { red[ax][tid >> 6] = lo[ax]; red[3 + ax][tid >> 6] = hi[ax]; }
    }
    __syncthreads();
    float scale[3];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        float l = red[ax][0], h = red[3 + ax][0];
        for (int w = 1; w < 16; ++w) { l = fminf(l, red[ax][w]); h = fmaxf(h, red[3 + ax][w]); }
        lo[ax] = l;
        scale[ax] = h > l ? 1023.0f / (h - l) : 0.0f;
    }
    for (int i = tid; i < p2; i += 1024) {
        unsigned m = 0;
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            const float v = (c[ax * p2 + i] - lo[ax]) * scale[ax];
            const unsigned q = (unsigned)fminf(fmaxf(v, 0.0f), 1023.0f);   // NaN -> 0
            m |= spread3(q) << ax;
        }
        code[i] = m;
    }
    __syncthreads();
    for (int i = tid; i < p2; i += 1024) {
        const unsigned mine = code[i];
        int rank = 0;
        for (int k = 0; k < p2; ++k) {
            const unsigned o = code[k];
            rank += (o < mine) || (o == mine && k < i);
        }
        order[(size_t)b * p2 + rank] = i;
    }
}

// Transpose of the grouping with the scatter pre-reduced in LDS.  A workgroup takes GP output points that are adjacent
// in the Morton order (one wave each).  Their GP*K neighbour slots name far fewer DISTINCT input points (ModelNet
// schedule: 2.9-3.4x fewer for GP = 8), so per anchor the waves store their per-slot contributions T[n][c] to an LDS
// tile (plain stores), and after one barrier the workgroup sums the slots of each distinct destination (lists built
// once per workgroup) and issues ONE global atomic per (destination, anchor, channel) instead of one per slot.  The
// tile is double-buffered over anchors: one barrier per column.  (ds_add_f32 into a shared accumulator was measured at
// ~1 lane per clock on gfx950 -- 2x slower than the global atomics it replaced; hence stores + a gather-sum.)
constexpr int USH_TAB = 4096;   // destinations are de-duplicated through a direct-address table: p1 <= USH_TAB
// DET: atomic-free, bitwise repeatable.  The slots of a destination are summed in ascending slot order and the sum is
// STORED to the slab row of the destination's first slot (slab[b][p][n][a][c], as inter_ungroup_slots_kernel's, but
// only ~1/3 of its rows are written); `canon` marks those rows for inter_reduce_slots_kernel, which adds them per
// destination in the fixed order of the inverse neighbour list.
// CW: 16-channel chunks handled per anchor step.  The kernel-influence weights of an (output point, anchor) pair do not
// depend on the channel: with CW > 1 they are generated ONCE per anchor (S-MFMAs, table reads, relu, packing) for CW
// chunks, and one barrier pair covers CW chunks' worth of contributions (the LDS tile row is 16 CW channels wide).
// CS: chunk groups handled one after the other per anchor step with the SAME weights (the tile stays 16 CW wide).
template <int NT, int KT, typename TG, int GP, int NB = 2, bool DET = false, int CW = 1, int CS = 1>   // NB tile buffers: 2 = one barrier per column, 1 = two (half the LDS)
__global__ __launch_bounds__(64 * GP) void inter_ungroup_shared_kernel(InterArgs A, const int32_t *__restrict__ order,
                                                                       unsigned char *__restrict__ canon = nullptr) {
    constexpr int EW = 16 * NT;        // neighbour slots per point (padded)
    constexpr int E = GP * EW;         // slots of the workgroup (128 or 256)
    // floats per slot row: 16 CW channels + 4 (rows 4 apart fall on distinct banks).  CW = 2 (the K = 64 instances): + 2
    // spreads the four lane groups over the banks just as evenly, and with 16-bit index arrays the tile is 78 KB -- TWO
    // workgroups per CU (at 86 KB, one: 1.68-1.78 -> 1.76-1.89 ms per layer)
    constexpr int SS = 16 * CW + (CW == 2 ? 2 : 4);
    // CW = 2 (row pitch 34 floats): a `ds_write_b32` of the per-slot stores is serviced in two 32-lane groups, lane groups
    // j = 0, 1 (rows 4 apart: 4 x 34 = 136 = 8 mod 32 banks) overlapped on 8 of their 16 banks -- a 2-way conflict on every
    // store, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.30-0.34 since round 3's "padding of 2 floats" (review, round 5; the
    // 16-point instances with pitch 20 / 68 sit at 0.002).  The tile ROW of slot n = 16 t + 4 j + r is free as long as writers
    // and readers agree: with bits 2 and 3 of the row swapped the two lane groups of a store are 8 rows = 272 = 16 banks apart
    // (disjoint), at the same 78 KB.  The readers' slot words (pk[], and list[] for destinations with more than three slots)
    // carry the permuted row.
#ifndef EPN_UNG_ROWPERM
#define EPN_UNG_ROWPERM 1
#endif
    constexpr bool RP = EPN_UNG_ROWPERM && CW == 2 && !DET;
    auto rowp = [](unsigned e) -> unsigned { return RP ? ((e & ~12u) | ((e & 4u) << 1) | ((e & 8u) >> 1)) : e; };
    typedef typename std::conditional<CW == 2, short, int>::type ix_t;   // destinations < 32768 (p1 <= USH_TAB), slots < 1024
    static_assert((CW == 1 && CS == 1) || !DET, "the deterministic form handles one chunk per step");
    constexpr int NTH = 64 * GP;
    constexpr int CH = E / 64;         // 64-slot chunks
    constexpr int EPT = (E + NTH - 1) / NTH;   // slots per thread in the set-up passes
    __shared__ ix_t qlist[E];          // destination of each slot, -1 when masked or a cyclic repeat
    __shared__ int slot_of[E];         // slot -> distinct-destination number
    __shared__ ix_t uq[E];             // distinct destination -> input point
    __shared__ int cnt[E];
    __shared__ ix_t off[E + 1];        // CSR over distinct destinations ...
    __shared__ ix_t list[E];           // ... of the slots that feed them
    __shared__ int chunk_cnt[CH];
    constexpr int BS = (E + 1) * SS;   // floats per tile buffer: E slot rows + one row of zeros (slot id E, see `ent`)
    __shared__ __attribute__((aligned(16))) float Tb[NB * BS];
    int *tab = reinterpret_cast<int *>(Tb);   // set-up only: input point -> first slot naming it
    static_assert(NB * BS >= USH_TAB, "direct-address table must fit the tile buffers");
    static_assert(E < 1024, "slot ids are packed in 10 bits");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x = lane & 15, j = lane >> 4;
    const int groups = A.p2 / GP;
    const int blk = epn_xcd_tile(blockIdx.x, gridDim.x);
    const int bb = blk / groups, grp = blk - bb * groups;
    const int pp = order[(size_t)bb * A.p2 + grp * GP + wave];
    const int ct0 = blockIdx.y * A.col_tiles_per_wg;                 // the set-up below serves this many 16-channel chunks
    const int ct1 = min(ct0 + A.col_tiles_per_wg, A.cin >> 4);
    const int gss = A.cin * A.ks;

    Hood<NT> h;
    load_hood<NT>(A, bb, pp, x, j, h);
    if (x == 0) {
        const int32_t *row = A.idx + ((size_t)bb * A.p2 + pp) * A.nn;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = 16 * t + 4 * j + r;
                qlist[wave * EW + n] = h.mul[t][r] != 0.0f ? row[n] : -1;
            }
    }
    for (int e = tid; e < E; e += NTH) cnt[e] = 0;
    __syncthreads();
    int myq[EPT], myr[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int e = tid + k * NTH;
        myq[k] = e < E ? qlist[e] : -1;
        if (myq[k] >= 0) tab[myq[k]] = 0x7fffffff;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < EPT; ++k)
        if (myq[k] >= 0) atomicMin(&tab[myq[k]], tid + k * NTH);
    __syncthreads();
    // number the distinct destinations in slot order: ballot per 64-slot chunk, then chunk prefix
    for (int c = wave; c < CH; c += GP) {
        const int e = c * 64 + lane;
        const int q = qlist[e];
        const bool leader = q >= 0 && tab[q] == e;
        const unsigned long long m = __ballot(leader);
        if (leader) slot_of[e] = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) chunk_cnt[c] = __popcll(m);
    }
    __syncthreads();
    int U = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) U += chunk_cnt[c];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int e = tid + k * NTH;
        if (myq[k] >= 0 && tab[myq[k]] == e) {
            int base = 0;
            for (int c = 0; c < (e >> 6); ++c) base += chunk_cnt[c];
            const int sl = base + slot_of[e];
            slot_of[e] = sl;
            uq[sl] = myq[k];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int e = tid + k * NTH;
        myr[k] = 0;
        if (myq[k] >= 0) {
            const int sl = slot_of[tab[myq[k]]];
            slot_of[e] = sl;                    // a leader rewrites its own value
            myr[k] = atomicAdd(&cnt[sl], 1);
        }
    }
    __syncthreads();
    if (wave == 0) {                            // exclusive scan of cnt[0..E) by one wave: E/64 consecutive slots per lane
        int loc[CH], sum = 0;
#pragma unroll
        for (int c = 0; c < CH; ++c) { loc[c] = sum; sum += cnt[lane * CH + c]; }
        int incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o, 64);
            incl += lane >= o ? v : 0;
        }
        const int excl = incl - sum;
#pragma unroll
        for (int c = 0; c < CH; ++c) off[lane * CH + c] = excl + loc[c];
        if (lane == 63) off[E] = incl;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < EPT; ++k)
        if (myq[k] >= 0) list[off[slot_of[tid + k * NTH]] + myr[k]] = tid + k * NTH;
    __syncthreads();                            // also: tab (aliasing Tb) is dead from here on
    if constexpr (DET) {
        __shared__ int wpp[GP];
        if (lane == 0) wpp[wave] = pp;
        for (int u = tid; u < U; u += NTH) {    // ascending slot order per destination: a fixed summation order
            const int k0 = off[u], k1 = off[u + 1];
            for (int k = k0 + 1; k < k1; ++k) {
                const int v = list[k];
                int m = k - 1;
                while (m >= k0 && list[m] > v) { list[m + 1] = list[m]; --m; }
                list[m + 1] = v;
            }
        }
        __syncthreads();
        // cnt[] (dead) <- slab row of each distinct destination = (point, neighbour slot) of its first slot
        for (int u = tid; u < U; u += NTH) {
            const int e0 = list[off[u]];
            cnt[u] = ((bb * A.p2 + wpp[e0 / EW]) * A.nn + (e0 % EW));
        }
        if (blockIdx.y == 0)
            for (int e = tid; e < E; e += NTH) {
                const int n = e % EW;
                if (n < A.nn)
                    canon[((size_t)bb * A.p2 + wpp[e / EW]) * A.nn + n] = (qlist[e] >= 0 && list[off[slot_of[e]]] == e) ? 1 : 0;
            }
        __syncthreads();
    }

    // The gather-sum of a step used to walk off[] -> list[] -> tile row per (destination, channel): 2 + 2 len dependent
    // LDS round trips.  pk[u] (slot_of[], dead by now) holds the first three slots of destination u, 10 bits each (unused
    // ones name the zero row) and whether the list goes on: one read, then the tile reads back to back.
    int *pk = slot_of;                          // (its readers finished before the barrier above)
    for (int u = tid; u < U; u += NTH) {
        const int k0 = off[u], len = off[u + 1] - k0;
        const unsigned s0 = rowp(list[k0]), s1 = len > 1 ? rowp(list[k0 + 1]) : E, s2 = len > 2 ? rowp(list[k0 + 2]) : E;
        pk[u] = (int)(s0 | (s1 << 10) | (s2 << 20) | (len > 3 ? 1u << 30 : 0u));
        // element offset of the destination's gradient row inside the cloud (cnt[] is dead by now unless DET keeps its slab
        // rows there): the atomic's address becomes a wave-uniform 64-bit base + this 32-bit offset -- round 3 rebuilt it per
        // atomic as ((size_t)uq * na + a) * cin: two quarter-rate v_mul_lo_u32 and two 64-bit multiply-adds per fp32 added
        if constexpr (!DET) cnt[u] = (int)uq[u] * A.na * A.cin;
    }
    for (int i = tid; i < NB * SS; i += NTH) Tb[(i / SS) * BS + E * SS + i % SS] = 0.0f;

    float gB[NT], alphaN[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        alphaN[t] = __shfl(h.gA[t], 48 + x, 64);
        gB[t] = j == 3 ? 1.0f : h.gA[t];
    }
    for (int ct = ct0; ct < ct1; ct += CW * CS) {
    const TG *dG = reinterpret_cast<const TG *>(A.gout) + ((size_t)bb * A.p2 + pp) * A.na * gss + (size_t)(16 * ct + x) * A.ks;
    float *dcloud = A.out + ((size_t)bb * A.p1) * A.na * A.cin + 16 * ct;
    // dG fragments and kernel-table rows of the NEXT anchor are requested before this anchor's MFMAs: a step is ~1000 cycles
    // of work per wave, an HBM round trip as long again -- without the prefetch every step of every wave waited it out
    typedef typename std::conditional<sizeof(TG) == 2, bf16x4_t, f32x4>::type frag_t;
    auto load_dg = [&](int a, int cs, frag_t (&d)[CW][KT]) {
#pragma unroll
        for (int cw = 0; cw < CW; ++cw)
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                const int jj = 16 * kt + 4 * j < A.ks ? j : 0;
                const TG *src = dG + (size_t)a * gss + (size_t)(16 * (cs * CW + cw)) * A.ks + 16 * kt + 4 * jj;
#ifdef EPN_UNG_NT
                if constexpr (sizeof(TG) == 2) d[cw][kt] = __builtin_nontemporal_load(reinterpret_cast<const bf16x4_t *>(src));
                else d[cw][kt] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(src));
#else
                if constexpr (sizeof(TG) == 2) d[cw][kt] = *reinterpret_cast<const bf16x4_t *>(src);
                else d[cw][kt] = ld4f(src);
#endif
            }
    };
    auto load_rk = [&](int a, float (&r)[KT]) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) r[kt] = A.rk4[((size_t)a * EPN_KS_MAX + 16 * kt + x) * 4 + j];
    };
    frag_t dnext[CW][KT];
    float rknext[KT];
    if constexpr (CS == 1) load_dg(0, 0, dnext);
    load_rk(0, rknext);
    for (int a = 0; a < A.na; ++a) {
        float rk[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) rk[kt] = rknext[kt];
        const int an = a + 1 < A.na ? a + 1 : a;       // last anchor re-reads its own rows (cache hits, unused)
        load_rk(an, rknext);
        // weights of this anchor: generated once, used by every chunk of the group
        typename std::conditional<sizeof(TG) == 2, bf16x4_t, f32x4>::type wgt[NT][KT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                f32x4 sk = {alphaN[t], alphaN[t], alphaN[t], alphaN[t]};
                sk = mfma4(rk[kt], gB[t], sk);
                if constexpr (sizeof(TG) == 2) {
                    wgt[t][kt] = relu_pack4(sk);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) sk[r] = relu_f(sk[r]);
                    wgt[t][kt] = sk;
                }
            }
#pragma unroll
        for (int cs = 0; cs < CS; ++cs) {
            const int ph = (((ct - ct0) / (CW * CS)) * A.na + a) * CS + cs;     // tile buffers alternate across steps
            float *buf = Tb + (ph & (NB - 1)) * BS + wave * EW * SS + x;
            frag_t dgc[CW][KT];
            if constexpr (CS == 1) {
#pragma unroll
                for (int cw = 0; cw < CW; ++cw)
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) dgc[cw][kt] = dnext[cw][kt];
#ifdef EPN_TUNING
                if (!(A.wk & 16) || a == 0)
#endif
                load_dg(an, 0, dnext);
            } else {
                load_dg(a, cs, dgc);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int cw = 0; cw < CW; ++cw) {
                    f32x4 tt = {0.f, 0.f, 0.f, 0.f};
#ifdef EPN_TUNING
                    if (A.wk & 8) {
                        tt[0] = (float)dgc[cw][0][0] + (float)wgt[t][0][0]; tt[1] = (float)dgc[cw][KT - 1][1]; tt[2] = (float)wgt[t][KT - 1][2];
                    } else
#endif
                    if constexpr (sizeof(TG) == 2) {
                        // bf16 dG: the contraction over the kernel points on the bf16 MFMA (both 16-point tiles in one K = 32)
                        if constexpr (KT == 2) tt = mfma_bf16_k32(wgt[t][0], wgt[t][1], dgc[cw][0], dgc[cw][1], tt);
                        else tt = mfma_bf16_k16(wgt[t][0], dgc[cw][0], tt);
                    } else {
#pragma unroll
                        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) tt = mfma4(wgt[t][kt][r], dgc[cw][kt][r], tt);
                    }
                    // tt: lane (x = c, j), register r -> slot n = 16t + 4j + r.  Every slot row is written, also the masked
                    // ones (cyclic repeats, shadow indices: mul = 0): no destination list names them, so nobody reads those
                    // rows -- round 3 skipped them with `if (mul != 0)`, i.e. an exec-mask save / branch / restore around each
                    // of the 16 NT CW LDS stores of a step (149 s_and_saveexec + 106 s_cbranch_execz in the K = 64 instance)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#ifdef EPN_TUNING
                        if (!(A.wk & 4) || tt[r] == 12345.678f)
#endif
                        buf[(16 * t + (RP ? 8 * (j & 1) + 4 * (j >> 1) : 4 * j) + r) * SS + 16 * cw] = tt[r] * h.mul[t][r];
                }
// (round 6, measured: an LDS-only barrier here -- no wait for the previous step's atomics or the prefetched dG fragments --
// changes nothing on the bf16 networks and costs the classification step 1.2 %: 530.6 / 529.0 against 536.6 / 536.5
// point-clouds/s, A/B of two builds on one box, profiles/r06_ab_rawbar_onchip.txt.  The wait is not what the waves wait for.)
#ifndef EPN_UNG_RAWBAR
#define EPN_UNG_RAWBAR 0
#endif
            if constexpr (EPN_UNG_RAWBAR) lds_barrier();     // (inter_device.h: no wait for the atomics / the prefetched dG)
            else __syncthreads();
            const float *rb = Tb + (ph & (NB - 1)) * BS;
            float *dstep = dcloud + (size_t)a * A.cin + 16 * cs * CW;      // wave-uniform
#ifdef EPN_TUNING
            if (!(A.wk & 2))
#endif
            for (int i = tid; i < U * 16 * CW; i += NTH) {
                const int u = i / (16 * CW), c = i % (16 * CW);
                const unsigned e = (unsigned)pk[u];
                float sum = rb[(e & 1023u) * SS + c];
                sum += rb[((e >> 10) & 1023u) * SS + c];
                sum += rb[((e >> 20) & 1023u) * SS + c];
                if (e >> 30) {
                    const int k1 = off[u + 1];
                    for (int k = off[u] + 3; k < k1; ++k) sum += rb[rowp((unsigned)list[k]) * SS + c];
                }
                if constexpr (DET) {
                    TG *srow = reinterpret_cast<TG *>(A.out) + ((size_t)cnt[u] * A.na + a) * A.cin + 16 * ct + c;
                    if constexpr (sizeof(TG) == 2) *srow = (__bf16)sum; else *srow = sum;
                } else {
#ifdef EPN_TUNING
                    if (!(A.wk & 1) || sum == 12345.678f)
#endif
                    atomicAdd(dstep + ((unsigned)cnt[u] + (unsigned)c), sum);
                }
            }
            if constexpr (NB == 1) {
                if constexpr (EPN_UNG_RAWBAR) lds_barrier();
                else __syncthreads();
            }
        }
    }
    }
}

// Deterministic transpose of the grouping, step 1: slab[b][p][n][a][c] = sum_k w[k][n] dG[col][c*ks + k] (no atomics).
template <int NT, int KT, typename TG>
__global__ __launch_bounds__(64 * NW) void inter_ungroup_slots_kernel(InterArgs A) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x = lane & 15, j = lane >> 4;
    const long long col0 = ((long long)epn_xcd_tile(blockIdx.x, gridDim.x) * NW + wave) * 16;
    if (col0 >= A.ncol) return;
    const int gss = A.cin * A.ks;
    const TG *dG = reinterpret_cast<const TG *>(A.gout) + (size_t)col0 * gss;
    const int ct0 = blockIdx.y * A.col_tiles_per_wg;
    const int ct1 = min(ct0 + A.col_tiles_per_wg, A.cin >> 4);
    Seg<NT> s0, s1;
    make_segments<NT, TG>(A, col0, x, j, s0, s1);     // only the neighbourhood fragments are used (fbase is not)
    const long long pt0 = col0 / A.na;
    for (int ct = ct0; ct < ct1; ++ct) {
        scatter_segment<NT, KT, TG, true>(A, s0, ct, x, j, dG + 16 * ct * A.ks, gss, pt0);
        scatter_segment<NT, KT, TG, true>(A, s1, ct, x, j, dG + 16 * ct * A.ks, gss, pt0 + 1);
    }
}

// Inverse neighbour list of one cloud as CSR, entries in increasing (p, n) order (what makes the sums below
// order-deterministic): off[b][q] .. off[b][q+1] index ent[b][.] = p*nn + n with idx[b][p][n] == q.  One workgroup per
// cloud; thread q scans the cloud's index rows, staged through LDS in pieces (broadcast reads), twice: count, then fill.
constexpr int INV_PIECE = 8192;
__global__ __launch_bounds__(1024) void inverse_list_kernel(const int32_t *__restrict__ idx, int p1, int entries,
                                                            int32_t *__restrict__ off, int32_t *__restrict__ ent) {
    __shared__ int32_t piece[INV_PIECE];
    __shared__ int32_t scan[1024];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int32_t *row = idx + (size_t)b * entries;
    int32_t *o = off + (size_t)b * (p1 + 1);
    int32_t *e = ent + (size_t)b * entries;
    // destinations are handled 1024 at a time (p1 <= 1024 for every shipped schedule: one round)
    int base_total = 0;
    for (int q0 = 0; q0 < p1; q0 += 1024) {
        const int q = q0 + tid;
        int cnt = 0;
        for (int e0 = 0; e0 < entries; e0 += INV_PIECE) {
            const int ne = min(INV_PIECE, entries - e0);
            __syncthreads();
            for (int i = tid; i < ne; i += 1024) piece[i] = row[e0 + i];
            __syncthreads();
            if (q < p1)
                for (int i = 0; i < ne; ++i) cnt += piece[i] == q;
        }
        // exclusive scan of the 1024 counts (Hillis-Steele in LDS)
        __syncthreads();
        scan[tid] = cnt;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            const int v = tid >= d ? scan[tid - d] : 0;
            __syncthreads();
            scan[tid] += v;
            __syncthreads();
        }
        const int start = base_total + scan[tid] - cnt;
        if (q < p1) o[q] = start;
        int w = start;
        for (int e0 = 0; e0 < entries; e0 += INV_PIECE) {
            const int ne = min(INV_PIECE, entries - e0);
            __syncthreads();
            for (int i = tid; i < ne; i += 1024) piece[i] = row[e0 + i];
            __syncthreads();
            if (q < p1)
                for (int i = 0; i < ne; ++i)
                    if (piece[i] == q) e[w++] = e0 + i;
        }
        base_total += scan[1023];
        __syncthreads();
    }
    if (tid == 0) o[p1] = base_total;
}

// The same list for clouds whose entries fit the LDS (p2*nn <= 32768, p1 <= 4096: every ModelNet / rotation layer): count
// per destination with LDS atomics, scan, fill in arrival order, then sort every destination's short segment -- the
// result is the same increasing (p, n) order whatever the arrival order was.  ~40 us instead of ~500 (the scanning
// kernel above compares every destination with every entry).
constexpr int INV_LDS_ENT = 32768, INV_LDS_P1 = 4096;
__global__ __launch_bounds__(1024) void inverse_list_lds_kernel(const int32_t *__restrict__ idx, int p1, int entries,
                                                                int32_t *__restrict__ off, int32_t *__restrict__ ent) {
    __shared__ int32_t lst[INV_LDS_ENT];
    __shared__ int cur[INV_LDS_P1];
    __shared__ int wsum[16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int32_t *row = idx + (size_t)b * entries;
    int32_t *o = off + (size_t)b * (p1 + 1);
    int32_t *e = ent + (size_t)b * entries;
    for (int q = tid; q < p1; q += 1024) cur[q] = 0;
    __syncthreads();
    for (int i = tid; i < entries; i += 1024) {
        const int q = row[i];
        if (q >= 0 && q < p1) atomicAdd(&cur[q], 1);
    }
    __syncthreads();
    // exclusive scan: four consecutive destinations per thread
    int c[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int q = 4 * tid + k;
        c[k] = q < p1 ? cur[q] : 0;
        sum += c[k];
    }
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d, 64);
        incl += lane >= d ? v : 0;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int start = incl - sum;
    for (int w = 0; w < wave; ++w) start += wsum[w];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int q = 4 * tid + k;
        if (q < p1) { o[q] = start; cur[q] = start; }
        start += c[k];
    }
    if (tid == 1023) o[p1] = start;
    __syncthreads();
    for (int i = tid; i < entries; i += 1024) {
        const int q = row[i];
        if (q >= 0 && q < p1) lst[atomicAdd(&cur[q], 1)] = i;
    }
    __syncthreads();                            // cur[q] = end of segment q = start of segment q + 1
    for (int q = tid; q < p1; q += 1024) {
        const int k0 = q ? cur[q - 1] : 0, k1 = cur[q];
        for (int k = k0 + 1; k < k1; ++k) {
            const int v = lst[k];
            int m = k - 1;
            while (m >= k0 && lst[m] > v) { lst[m + 1] = lst[m]; --m; }
            lst[m + 1] = v;
        }
    }
    __syncthreads();
    const int total = cur[p1 - 1];
    for (int i = tid; i < total; i += 1024) e[i] = lst[i];
}

// Step 2: dF[b][q][a][c] = sum over the inverse list of q, in list order, of slab[b][entry][a][c]; fp32 accumulation,
// output in TO.  One thread per 4 consecutive (a, c) elements of one destination row: fully coalesced slab reads.
template <typename TG, typename TO>
__global__ __launch_bounds__(256) void inter_reduce_slots_kernel(const TG *__restrict__ slab, const int32_t *__restrict__ off,
                                                                 const int32_t *__restrict__ ent, TO *__restrict__ dF,
                                                                 int b, int p1, int entries, int rowlen,
                                                                 const unsigned char *__restrict__ canon = nullptr) {
    const int v4 = rowlen >> 2;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)b * p1 * v4) return;
    const int c4 = (int)(i % v4);
    const long long bq = i / v4;
    const int bb = (int)(bq / p1), q = (int)(bq - (long long)bb * p1);
    const int32_t *o = off + (size_t)bb * (p1 + 1);
    const int32_t *e = ent + (size_t)bb * entries;
    const TG *sl = slab + (size_t)bb * entries * rowlen + 4 * c4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int e1 = o[q + 1];
    const unsigned char *cf = canon ? canon + (size_t)bb * entries : nullptr;
    for (int k = o[q]; k < e1; ++k) {
        if (cf && !cf[e[k]]) continue;          // pre-reduced slab: only the marked rows were written
        const f32x4 v = ld4f(sl + (size_t)e[k] * rowlen);
        acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
    }
    st4f(dF + (size_t)bq * rowlen + 4 * c4, acc);
}

// rk4[a][k] = ((2/sigma) R_a kappa_k, beta_k = -|R_a kappa_k|^2 / sigma), zero / -1e30 padded to EPN_KS_MAX kernel points --
// straight from the anchors (the rotation is rk_table_kernel's expression, functional.py:190): one table launch per call
__global__ void rk4_table_kernel(const float *__restrict__ anchors, const float *__restrict__ kernels, int na, int ks,
                                 float sigma_inv, float *__restrict__ rk4) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= na * EPN_KS_MAX) return;
    const int k = i % EPN_KS_MAX, a = i / EPN_KS_MAX;
    f32x4 v = {0.f, 0.f, 0.f, -1e30f};
    if (k < ks) {
        const float *kp = kernels + k * 3;
        float r[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float *R = anchors + a * 9 + d * 3;
            r[d] = R[0] * kp[0] + R[1] * kp[1] + R[2] * kp[2];
        }
        v[0] = 2.0f * sigma_inv * r[0];
        v[1] = 2.0f * sigma_inv * r[1];
        v[2] = 2.0f * sigma_inv * r[2];
        v[3] = -(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]) * sigma_inv;
    }
    *reinterpret_cast<f32x4 *>(rk4 + (size_t)i * 4) = v;
}

__global__ void transpose_kernel(const float *__restrict__ src, int rows, int cols, float *__restrict__ dst) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // dst[c][r] = src[r][c]
    if (i >= (size_t)rows * cols) return;
    const int r = i % rows;
    const int c = i / rows;
    dst[i] = src[(size_t)r * cols + c];
}

InterArgs make_args(const epn_inter_desc *d, const float *rk4) {
    InterArgs A;
    A.xyz = d->xyz; A.new_xyz = d->new_xyz; A.idx = d->ball_idx; A.rk4 = rk4;
    A.feats = nullptr; A.W = nullptr; A.gout = nullptr; A.out = nullptr;
    A.sigma_inv = 1.0f / d->sigma;
    A.b = d->b; A.p1 = d->p1; A.p2 = d->p2; A.nn = d->nn; A.na = d->na; A.ks = d->ks; A.cin = d->cin;
    A.cout = d->cout; A.wk = 0; A.packed = 0;
    A.ncol = (long long)d->b * d->p2 * d->na;
    A.col_tiles_per_wg = 1;
    return A;
}

size_t gs_bytes(const epn_inter_desc *d) { return (size_t)NW * 16 * (16 * d->ks + 4) * sizeof(float); }

template <typename K>
int set_lds(K kern, size_t bytes) {
    EPN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)bytes));
    return 0;
}

#define EPN_DISPATCH_NT_KT(FN, ...)                                              \
    do {                                                                         \
        const int nt_ = (d->nn + 15) / 16, kt_ = (d->ks + 15) / 16;              \
        if (kt_ == 1) {                                                          \
            if (nt_ <= 1) FN(1, 1, __VA_ARGS__);                                 \
            else if (nt_ <= 2) FN(2, 1, __VA_ARGS__);                            \
            else if (nt_ <= 4) FN(4, 1, __VA_ARGS__);                            \
            else FN(8, 1, __VA_ARGS__);                                          \
        } else {                                                                 \
            if (nt_ <= 1) FN(1, 2, __VA_ARGS__);                                 \
            else if (nt_ <= 2) FN(2, 2, __VA_ARGS__);                            \
            else if (nt_ <= 4) FN(4, 2, __VA_ARGS__);                            \
            else FN(8, 2, __VA_ARGS__);                                          \
        }                                                                        \
    } while (0)

}  // namespace

int launch_inter_tables_mfma(const epn_inter_desc *d, const float *rk, float *rk4, float *beta, hipStream_t st) {
    (void)beta; (void)rk;
    const int n = d->na * EPN_KS_MAX;
    EPN_LAUNCH_AUX(rk4_table_kernel, dim3(epn_cdiv(n, 256)), dim3(256), 0, st, d->anchors, d->kernels, d->na, d->ks,
                   1.0f / d->sigma, rk4);
    EPN_CHECK_LAUNCH();
    return 0;
}

static bool use8(const epn_inter_desc *d) {
    return d->na >= 16 && d->ks > 16 && d->ks <= 32 && d->nn <= 32;
}

int launch_inter_fwd_mfma(const epn_inter_desc *d, const float *rk4, const float *beta, const float *feats,
                          const float *W, float *out, hipStream_t st) {
    (void)beta;
    InterArgs A = make_args(d, rk4);
    A.feats = feats; A.W = W; A.out = out;
    const int mt_ = d->cout / 16;
    if (use8(d) && (mt_ == 1 || mt_ == 2 || mt_ == 4 || mt_ == 8 || mt_ == 16)) {
        const int kw1 = d->ks - 16;
        const size_t gsb = (size_t)NW8 * 16 * GS0 * sizeof(float);
        int wk = 16;
        for (int cand = 16; cand <= 256; cand += 16)
            if (256 % cand == 0 && (16 * kw1) % cand == 0 &&
                gsb + (size_t)d->cout * (cand + 4) * sizeof(float) <= 160 * 1024 && d->cout * cand <= 512 * 4 * 4)
                wk = cand;
        A.wk = wk;
        const size_t lds = gsb + (size_t)d->cout * (wk + 4) * sizeof(float);
        const unsigned grid = (unsigned)((A.ncol + 16 * NW8 - 1) / (16 * NW8));
#define EPN_FWD8(NT_, MT_)                                                                              \
    do {                                                                                                \
        int rc_ = set_lds(inter_fwd8_kernel<NT_, MT_>, lds);                                            \
        if (rc_) return rc_;                                                                            \
        EPN_LAUNCH((inter_fwd8_kernel<NT_, MT_>), dim3(grid), dim3(64 * NW8), lds, st, A);      \
    } while (0)
        const int mt = d->cout / 16;
        if (d->nn <= 16) {
            if (mt == 4) EPN_FWD8(1, 4); else if (mt == 8) EPN_FWD8(1, 8); else if (mt == 16) EPN_FWD8(1, 16);
            else if (mt == 2) EPN_FWD8(1, 2); else EPN_FWD8(1, 1);
        } else {
            if (mt == 4) EPN_FWD8(2, 4); else if (mt == 8) EPN_FWD8(2, 8); else if (mt == 16) EPN_FWD8(2, 16);
            else if (mt == 2) EPN_FWD8(2, 2); else EPN_FWD8(2, 1);
        }
#undef EPN_FWD8
        EPN_CHECK_LAUNCH();
        return 0;
    }
    const int ckl = 16 * d->ks;
    // W sub-chunk width: largest divisor of the chunk length (multiple of 16) that keeps Ws <= ~56 KB
    int wk = 16;
    for (int cand = 16; cand <= ckl; cand += 16)
        if (ckl % cand == 0 && (size_t)d->cout * (cand + 4) * sizeof(float) <= 56 * 1024 &&
            d->cout * cand <= 256 * WPF * 4)
            wk = cand;
    A.wk = wk;
    const size_t lds = gs_bytes(d) + (size_t)d->cout * (wk + 4) * sizeof(float);
    const unsigned grid = (unsigned)((A.ncol + 16 * NW - 1) / (16 * NW));
#define EPN_FWD(NT_, KT_, dummy)                                                                      \
    do {                                                                                              \
        int rc_ = set_lds(inter_fwd_kernel<NT_, KT_>, lds);                                           \
        if (rc_) return rc_;                                                                          \
        EPN_LAUNCH((inter_fwd_kernel<NT_, KT_>), dim3(grid), dim3(64 * NW), lds, st, A);      \
    } while (0)
    EPN_DISPATCH_NT_KT(EPN_FWD, 0);
#undef EPN_FWD
    EPN_CHECK_LAUNCH();
    return 0;
}

int launch_inter_bwd_data_mfma(const epn_inter_desc *d, const float *rk4, const float *WT, const float *dOut,
                               const float *W, float *dF, hipStream_t st) {
    // WT: scratch [cin*ks][cout] (the caller passes the workspace slot in the `beta` position)
    float *wt = const_cast<float *>(WT);
    const size_t nW = (size_t)d->cout * d->cin * d->ks;
    EPN_LAUNCH_AUX(transpose_kernel, dim3((unsigned)((nW + 255) / 256)), dim3(256), 0, st, W, d->cout,
                       d->cin * d->ks, wt);
    EPN_CHECK_LAUNCH();
    InterArgs A = make_args(d, rk4);
    A.W = wt; A.gout = dOut; A.out = dF;
    {
        if (d->na >= 16 && d->nn <= 32 && (d->ks == 24 || d->ks == 32 || d->ks == 16)) {
            const size_t lds8 = (size_t)4 * 16 * (16 * d->ks + 4) * sizeof(float) + (size_t)16 * d->ks * 20 * sizeof(float);
            const unsigned grid8 = (unsigned)((A.ncol + 63) / 64);
#define EPN_BD8(NT_, KT_, MH_)                                                                                 \
    do {                                                                                                       \
        int rc_ = set_lds(inter_bwd_data8_kernel<NT_, KT_, MH_>, lds8);                                        \
        if (rc_) return rc_;                                                                                   \
        EPN_LAUNCH((inter_bwd_data8_kernel<NT_, KT_, MH_>), dim3(grid8), dim3(64 * NW8), lds8, st, A); \
    } while (0)
            if (d->nn <= 16) {
                if (d->ks == 16) EPN_BD8(1, 1, 8); else if (d->ks == 24) EPN_BD8(1, 2, 12); else EPN_BD8(1, 2, 16);
            } else {
                if (d->ks == 16) EPN_BD8(2, 1, 8); else if (d->ks == 24) EPN_BD8(2, 2, 12); else EPN_BD8(2, 2, 16);
            }
#undef EPN_BD8
            EPN_CHECK_LAUNCH();
            return 0;
        }
    }
    const size_t lds = gs_bytes(d) + (size_t)16 * d->ks * 20 * sizeof(float);
    const unsigned grid = (unsigned)((A.ncol + 16 * NW - 1) / (16 * NW));
#define EPN_BD(NT_, KT_, dummy)                                                                       \
    do {                                                                                              \
        int rc_ = set_lds(inter_bwd_data_kernel<NT_, KT_>, lds);                                      \
        if (rc_) return rc_;                                                                          \
        EPN_LAUNCH((inter_bwd_data_kernel<NT_, KT_>), dim3(grid), dim3(64 * NW), lds, st, A); \
    } while (0)
    EPN_DISPATCH_NT_KT(EPN_BD, 0);
#undef EPN_BD
    EPN_CHECK_LAUNCH();
    return 0;
}

int launch_inter_bwd_weight_mfma(const epn_inter_desc *d, const float *rk4, const float *beta, const float *feats,
                                 const float *dOut, float *dW, hipStream_t st) {
    (void)beta;
    InterArgs A = make_args(d, rk4);
    A.feats = feats; A.gout = dOut; A.out = dW;
    // 8-wave kernel: every block of output channels must hold the same power-of-two number of row tiles
    const int mo8 = d->cout >= 128 ? 8 : d->cout / 16;
    if (use8(d) && d->ks == 24 && (d->cout % 128 == 0 || d->cout == 64 || d->cout == 32)) {
        const long long tiles8 = (A.ncol + 16 * NW8 - 1) / (16 * NW8);
        const int chunks8 = d->cin / 16, oblocks8 = (d->cout + 127) / 128;
        long long splits8 = (256 * 2 + chunks8 * oblocks8 - 1) / (chunks8 * oblocks8);
        if (splits8 > tiles8) splits8 = tiles8;
        if (splits8 < 1) splits8 = 1;
        A.col_tiles_per_wg = (int)((tiles8 + splits8 - 1) / splits8);
        const unsigned gx8 = (unsigned)((tiles8 + A.col_tiles_per_wg - 1) / A.col_tiles_per_wg);
        const size_t lds8 = (size_t)NW8 * 16 * GS0 * sizeof(float);
#define EPN_BW8(NT_, MO_)                                                                                        \
    do {                                                                                                         \
        int rc_ = set_lds(inter_bwd_weight8_kernel<NT_, MO_>, lds8);                                             \
        if (rc_) return rc_;                                                                                     \
        EPN_LAUNCH((inter_bwd_weight8_kernel<NT_, MO_>), dim3(gx8, chunks8, oblocks8), dim3(64 * NW8),   \
                           lds8, st, A);                                                                         \
    } while (0)
        if (d->nn <= 16) {
            if (mo8 == 8) EPN_BW8(1, 8); else if (mo8 == 4) EPN_BW8(1, 4); else EPN_BW8(1, 2);
        } else {
            if (mo8 == 8) EPN_BW8(2, 8); else if (mo8 == 4) EPN_BW8(2, 4); else EPN_BW8(2, 2);
        }
#undef EPN_BW8
        EPN_CHECK_LAUNCH();
        return 0;
    }
    const long long tiles = (A.ncol + 16 * NW - 1) / (16 * NW);
    const int chunks = d->cin / 16, oblocks = (d->cout + 127) / 128;
    // enough workgroups to fill 256 CUs a few times over, each walking a contiguous run of column tiles
    long long splits = (256 * 4 + chunks * oblocks - 1) / (chunks * oblocks);
    if (splits > tiles) splits = tiles;
    if (splits < 1) splits = 1;
    A.col_tiles_per_wg = (int)((tiles + splits - 1) / splits);
    const unsigned gx = (unsigned)((tiles + A.col_tiles_per_wg - 1) / A.col_tiles_per_wg);
    const size_t lds = gs_bytes(d);
#define EPN_BW(NT_, KT_, dummy)                                                                          \
    do {                                                                                                 \
        int rc_ = set_lds(inter_bwd_weight_kernel<NT_, KT_>, lds);                                       \
        if (rc_) return rc_;                                                                             \
        EPN_LAUNCH((inter_bwd_weight_kernel<NT_, KT_>), dim3(gx, chunks, oblocks), dim3(64 * NW), \
                           lds, st, A);                                                                  \
    } while (0)
    EPN_DISPATCH_NT_KT(EPN_BW, 0);
#undef EPN_BW
    EPN_CHECK_LAUNCH();
    return 0;
}

// 16-channel chunks per workgroup of the grouping kernels: the per-point set-up (index row, offsets, weights operand) is
// amortised over them, while the launch stays chunk-major enough for the gathered slice of a cloud to live in L2.
// Measured on the ModelNet schedule (sum over the six layers): fp32 1 / 2 / 4 chunks 10.3 / 9.5 / 9.8 ms, bf16 6.9 / - / 6.1.
static int chunks_per_row(int bf16 = 0) { return bf16 ? 4 : 2; }

int launch_inter_group_mfma(const epn_inter_desc *d, const float *rk4, const void *feats, void *G, int bf16,
                            hipStream_t st, int packed) {
    InterArgs A = make_args(d, rk4);
    A.feats = static_cast<const float *>(feats); A.out = static_cast<float *>(G);
    const unsigned grid = (unsigned)((A.ncol + 16 * NW - 1) / (16 * NW));
    // wide-gather form (a lane owns 4 or 2 consecutive channels).  Measured per layer, B=32/64 schedules: bf16 0.85-0.89 vs
    // 1.35 ms (K = 64), 1.18 vs 1.49 ms (K = 32) -- the bf16 kernel is bound by its gather instructions; fp32 1.18-1.28 vs
    // 1.32 ms at K = 32 but 2.05-2.44 vs 1.97 ms at K = 16 in the PLAIN column order, where its stores are 64-byte pieces
    // 384 bytes apart.  Packed order (contiguous stores): 1.53-1.72 ms at K = 16, 1.04-1.09 at K = 32 (plain 1.21-1.34);
    // bf16 1.00 / 0.78-0.81 ms (K = 32 / 64; plain 1.24-1.26 / 0.84-0.88) -- the wide kernel serves every packed call
    A.packed = packed;
#ifdef EPN_TUNING
    if ((kernel_policy() & 0xf00) == 0x800) A.wk = kernel_policy() & 0xff;   // ablation bits of inter_group_wide_kernel
#endif
    if (packed && !inter_group_packed_ok(d)) return EPN_EINVAL;
    if (d->na >= 16 && d->cin % 32 == 0 && d->nn <= 64 && (bf16 || d->nn > 16 || packed)) {
        const int cg = d->cin % 64 == 0 ? 4 : 2;
        const unsigned gyw = (unsigned)(d->cin / (16 * cg));
#define EPN_GRPW(NT_, KT_, dummy)                                                                                              \
    do {                                                                                                                       \
        if (bf16 && cg == 4) EPN_LAUNCH((inter_group_wide_kernel<NT_, KT_, __bf16, 4>), dim3(grid, gyw), dim3(64 * NW), 0, st, A); \
        else if (bf16) EPN_LAUNCH((inter_group_wide_kernel<NT_, KT_, __bf16, 2>), dim3(grid, gyw), dim3(64 * NW), 0, st, A);    \
        else if (cg == 4) EPN_LAUNCH((inter_group_wide_kernel<NT_, KT_, float, 4>), dim3(grid, gyw), dim3(64 * NW), 0, st, A);  \
        else EPN_LAUNCH((inter_group_wide_kernel<NT_, KT_, float, 2>), dim3(grid, gyw), dim3(64 * NW), 0, st, A);               \
    } while (0)
        EPN_DISPATCH_NT_KT(EPN_GRPW, 0);
#undef EPN_GRPW
        EPN_CHECK_LAUNCH();
        return 0;
    }
    A.col_tiles_per_wg = chunks_per_row(bf16);
    const unsigned gy = (unsigned)(((d->cin >> 4) + A.col_tiles_per_wg - 1) / A.col_tiles_per_wg);
#define EPN_GRP(NT_, KT_, dummy)                                                                                      \
    do {                                                                                                              \
        if (bf16) EPN_LAUNCH((inter_group_kernel<NT_, KT_, __bf16>), dim3(grid, gy), dim3(64 * NW), 0, st, A); \
        else EPN_LAUNCH((inter_group_kernel<NT_, KT_, float>), dim3(grid, gy), dim3(64 * NW), 0, st, A);      \
    } while (0)
    EPN_DISPATCH_NT_KT(EPN_GRP, 0);
#undef EPN_GRP
    EPN_CHECK_LAUNCH();
    return 0;
}

// Column permutation of the weights for packed grouped features.  One thread per element of the PLAIN matrix (coalesced on
// the plain side; the packed side moves in runs of ks - 16 .. 16 elements -- the matrices are a few MB, read from L2).
template <typename TO, bool UNPACK>
__global__ __launch_bounds__(256) void inter_pack_cols_kernel(const float *__restrict__ src, TO *__restrict__ dst,
                                                              int rows, int cin, int ks) {
    const int ck = cin * ks;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)rows * ck) return;
    const int r = (int)(i / ck), q = (int)(i - (long long)r * ck);
    const int c = q / ks, k = q - c * ks;
    const size_t pk = (size_t)r * ck + inter_packed_position(c, k, cin, ks);
    if constexpr (UNPACK) dst[i] = (TO)src[pk];
    else dst[pk] = (TO)src[i];
}

int launch_inter_pack_weights(const float *W, int cout, int cin, int ks, void *Wp, int bf16, hipStream_t st) {
    const long long n = (long long)cout * cin * ks;
    if (n == 0) return 0;
    const unsigned grid = (unsigned)((n + 255) / 256);
    if (bf16) EPN_LAUNCH((inter_pack_cols_kernel<__bf16, false>), dim3(grid), dim3(256), 0, st, W, static_cast<__bf16 *>(Wp), cout, cin, ks);
    else EPN_LAUNCH((inter_pack_cols_kernel<float, false>), dim3(grid), dim3(256), 0, st, W, static_cast<float *>(Wp), cout, cin, ks);
    EPN_CHECK_LAUNCH();
    return 0;
}

int launch_inter_unpack_weight_grad(const float *gWp, int cout, int cin, int ks, float *gW, hipStream_t st) {
    const long long n = (long long)cout * cin * ks;
    if (n == 0) return 0;
    EPN_LAUNCH((inter_pack_cols_kernel<float, true>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, gWp, gW, cout, cin, ks);
    EPN_CHECK_LAUNCH();
    return 0;
}

int launch_inverse_list(const int32_t *idx, int b, int p1, int p2, int nn, int32_t *off, int32_t *ent, hipStream_t st) {
    if ((long long)p2 * nn <= INV_LDS_ENT && p1 <= INV_LDS_P1)
        EPN_LAUNCH(inverse_list_lds_kernel, dim3(b), dim3(1024), 0, st, idx, p1, p2 * nn, off, ent);
    else
        EPN_LAUNCH(inverse_list_kernel, dim3(b), dim3(1024), 0, st, idx, p1, p2 * nn, off, ent);
    EPN_CHECK_LAUNCH();
    return 0;
}

// deterministic data gradient: slab (element type = dG's) <- per-slot contributions, then the ordered reduction
// output points per workgroup of the LDS-reduced scatter (see launch_inter_ungroup_mfma)
static int ungroup_group_points(const epn_inter_desc *d, int nt) {
    return nt <= 1 ? 8 : (nt <= 2 ? (d->p2 % 16 == 0 ? 16 : 8) : (nt <= 4 ? (d->p2 % 8 == 0 ? 8 : 4) : 2));
}

int launch_inter_ungroup_det_mfma(const epn_inter_desc *d, const float *rk4, const void *dG, void *dF, void *slab,
                                  const int32_t *off, const int32_t *ent, int bf16, hipStream_t st, int32_t *order,
                                  unsigned char *canon) {
    InterArgs A = make_args(d, rk4);
    A.gout = static_cast<const float *>(dG); A.out = static_cast<float *>(slab);
    const int rowlen = d->na * d->cin, entries = d->p2 * d->nn;
    const long long nred = (long long)d->b * d->p1 * (rowlen >> 2);
    const dim3 g2((unsigned)((nred + 255) / 256));
    {
        // pre-reduced form: the LDS-reduced scatter (inter_ungroup_shared_kernel, DET) stores one row per (workgroup,
        // distinct destination) -- about a third of the per-slot slab -- and the ordered reduction skips the rest
        const int nt = (d->nn + 15) / 16;
        const int gp = ungroup_group_points(d, nt);
        if (order && canon && d->p2 % gp == 0 && d->p2 <= MORTON_MAX && d->p1 <= USH_TAB) {
            EPN_LAUNCH_AUX(morton_order_kernel, dim3(d->b), dim3(1024), 0, st, d->new_xyz, d->p2, order);
            EPN_CHECK_LAUNCH();
            A.col_tiles_per_wg = 1;
            const dim3 grid((unsigned)(d->b * (d->p2 / gp)), (unsigned)(d->cin >> 4));
#define EPN_USHD(NT_, KT_, GP_)                                                                                           \
    do {                                                                                                                  \
        if (bf16) EPN_LAUNCH((inter_ungroup_shared_kernel<NT_, KT_, __bf16, GP_, ((NT_ == 1 && GP_ == 8) ? 2 : 1), true>), grid, dim3(64 * GP_), 0, st, A, order, canon); \
        else EPN_LAUNCH((inter_ungroup_shared_kernel<NT_, KT_, float, GP_, ((NT_ == 1 && GP_ == 8) ? 2 : 1), true>), grid, dim3(64 * GP_), 0, st, A, order, canon);      \
    } while (0)
            const int kt = (d->ks + 15) / 16;
            if (kt == 1) {
                if (nt <= 1) EPN_USHD(1, 1, 8);
                else if (nt <= 2) { if (gp == 16) EPN_USHD(2, 1, 16); else EPN_USHD(2, 1, 8); }
                else if (nt <= 4) { if (gp == 8) EPN_USHD(4, 1, 8); else EPN_USHD(4, 1, 4); }
                else EPN_USHD(8, 1, 2);
            } else {
                if (nt <= 1) EPN_USHD(1, 2, 8);
                else if (nt <= 2) { if (gp == 16) EPN_USHD(2, 2, 16); else EPN_USHD(2, 2, 8); }
                else if (nt <= 4) { if (gp == 8) EPN_USHD(4, 2, 8); else EPN_USHD(4, 2, 4); }
                else EPN_USHD(8, 2, 2);
            }
#undef EPN_USHD
            EPN_CHECK_LAUNCH();
            if (bf16)
                EPN_LAUNCH_AUX((inter_reduce_slots_kernel<__bf16, __bf16>), g2, dim3(256), 0, st, static_cast<const __bf16 *>(slab),
                                   off, ent, static_cast<__bf16 *>(dF), d->b, d->p1, entries, rowlen, canon);
            else
                EPN_LAUNCH_AUX((inter_reduce_slots_kernel<float, float>), g2, dim3(256), 0, st, static_cast<const float *>(slab),
                                   off, ent, static_cast<float *>(dF), d->b, d->p1, entries, rowlen, canon);
            EPN_CHECK_LAUNCH();
            return 0;
        }
    }
    const unsigned grid = (unsigned)((A.ncol + 16 * NW - 1) / (16 * NW));
    // every wave walks ALL 16-channel chunks of its 16 columns: the [16 anchors][cin] block of a slot is then written by
    // one wave within a few hundred cycles and leaves L2 as whole lines (chunk-major launch order, which the gather
    // kernels use for their reads, writes each 128-byte line in 32/64-byte pieces seconds apart: 2x slower, measured)
    A.col_tiles_per_wg = d->cin >> 4;
    const unsigned gy = 1;
#define EPN_USLOT(NT_, KT_, dummy)                                                                                       \
    do {                                                                                                                 \
        if (bf16) EPN_LAUNCH((inter_ungroup_slots_kernel<NT_, KT_, __bf16>), dim3(grid, gy), dim3(64 * NW), 0, st, A); \
        else EPN_LAUNCH((inter_ungroup_slots_kernel<NT_, KT_, float>), dim3(grid, gy), dim3(64 * NW), 0, st, A);      \
    } while (0)
    EPN_DISPATCH_NT_KT(EPN_USLOT, 0);
#undef EPN_USLOT
    EPN_CHECK_LAUNCH();
    if (bf16)
        EPN_LAUNCH_AUX((inter_reduce_slots_kernel<__bf16, __bf16>), g2, dim3(256), 0, st, static_cast<const __bf16 *>(slab),
                           off, ent, static_cast<__bf16 *>(dF), d->b, d->p1, entries, rowlen);
    else
        EPN_LAUNCH_AUX((inter_reduce_slots_kernel<float, float>), g2, dim3(256), 0, st, static_cast<const float *>(slab),
                           off, ent, static_cast<float *>(dF), d->b, d->p1, entries, rowlen);
    EPN_CHECK_LAUNCH();
    return 0;
}

// Morton order of the output points of every cloud (order[b][p2]); shared with inter_bwd_f2.hip
int launch_morton_order(const float *new_xyz, int b, int p2, int32_t *order, hipStream_t st) {
    if (p2 > MORTON_MAX) return EPN_EINVAL;
    EPN_LAUNCH_AUX(morton_order_kernel, dim3(b), dim3(1024), 0, st, new_xyz, p2, order);
    EPN_CHECK_LAUNCH();
    return 0;
}

int launch_inter_ungroup_mfma(const epn_inter_desc *d, const float *rk4, const void *dG, float *dF, int32_t *order,
                              int bf16, hipStream_t st) {
    InterArgs A = make_args(d, rk4);
    A.gout = static_cast<const float *>(dG); A.out = dF;
#ifdef EPN_TUNING
    if ((kernel_policy() & 0xf00) == 0x800) A.wk = kernel_policy() & 0xff;   // ablation bits of inter_ungroup_shared_kernel
#endif
    // LDS pre-reduced scatter over Morton-adjacent output points (see inter_ungroup_shared_kernel); kernel policy
    // 0x400 | 1 keeps the per-slot atomic scatter for A/B measurements
    const int nt = (d->nn + 15) / 16;
    // output points per workgroup (= waves): more points share more destinations (16: 4-5x fewer distinct destinations
    // than slots, 8: 3x) but cost LDS and a wider barrier.  Measured per layer (ModelNet / rotation / 3DMatch schedules):
    // K = 16: 8 points, two tile buffers (16: +7 %); K = 32: 16 points, one buffer (-5 % fp32, -12 % bf16);
    // K = 64: 8 points, one buffer; K = 128: 2.  K = 32 / 64 fall back to half the group when p2 does not divide.
    int gp = ungroup_group_points(d, nt);
    // the 32-channel K = 32 layers run two chunks per step (below) in a 39 KB tile with 8 points: four workgroups per CU beat
    // the 16-point form there (1.64 -> 1.43 ms); with 64-channel groups 16 points stay ahead (1.26 vs 1.75 fp32, 1.39 vs 1.45 bf16)
    if (nt == 2 && d->cin % 64 != 0 && d->cin % 32 == 0 && d->p2 % 8 == 0) gp = 8;
    if (order && d->p2 % gp == 0 && d->p2 <= MORTON_MAX && d->p1 <= USH_TAB && kernel_policy() != (0x400 | 1)) {
        EPN_LAUNCH_AUX(morton_order_kernel, dim3(d->b), dim3(1024), 0, st, d->new_xyz, d->p2, order);
        EPN_CHECK_LAUNCH();
        // Chunks per anchor step: the weights of an anchor are generated once for CW * CS chunks; CW of them share one
        // barrier pair (LDS tile 16 CW channels wide), CS such groups follow each other on the same tile.
        // Measured per layer (ms, one chunk per step -> grouped): CW = 4: fp32 K = 16 1.93 -> 1.77, K = 32 1.36 -> 1.26;
        // bf16 K = 32 1.60 -> 1.39; at K = 64 (8 waves per workgroup) the 150 KB tile of CW = 4 leaves one workgroup per CU
        // (1.5 -> 1.9-2.6) and sequential groups (CS = 4) lose to independent workgroups (1.50 -> 1.73): one chunk per step there.
        const bool wide = d->cin % 64 == 0 && nt <= 2;
        // K = 64 (nt = 4, 8 points per workgroup): two chunks per step -- the weights of an anchor generated once per 32 channels
        // (and the 32-channel K = 32 layers, which the four-chunk form does not take)
        const bool wide2 = !wide && d->cin % 32 == 0 && ((nt > 2 && nt <= 4 && gp == 8) || nt == 2);
        A.col_tiles_per_wg = wide ? 4 : (wide2 ? 2 : 1);   // (CS = 4 at K = 64 measured 1.50 -> 1.73 ms: not instantiated)
        const dim3 grid((unsigned)(d->b * (d->p2 / gp)), (unsigned)((d->cin >> 4) / A.col_tiles_per_wg));
#define EPN_USH(NT_, KT_, GP_)                                                                                            \
    do {                                                                                                                  \
        constexpr int nb_ = ((NT_ == 1 && GP_ == 8) ? 2 : 1);                                                             \
        if (wide) {                                                                                                       \
            if constexpr (NT_ <= 2) {                                                                                     \
                if (bf16) EPN_LAUNCH((inter_ungroup_shared_kernel<NT_, KT_, __bf16, GP_, nb_, false, 4, 1>), grid, dim3(64 * GP_), 0, st, A, order); \
                else EPN_LAUNCH((inter_ungroup_shared_kernel<NT_, KT_, float, GP_, nb_, false, 4, 1>), grid, dim3(64 * GP_), 0, st, A, order);      \
            }                                                                                                             \
        } else if (wide2) {                                                                                               \
            if constexpr ((NT_ == 4 && GP_ == 8) || NT_ == 2) {                                                           \
                if (bf16) EPN_LAUNCH((inter_ungroup_shared_kernel<NT_, KT_, __bf16, GP_, nb_, false, 2, 1>), grid, dim3(64 * GP_), 0, st, A, order); \
                else EPN_LAUNCH((inter_ungroup_shared_kernel<NT_, KT_, float, GP_, nb_, false, 2, 1>), grid, dim3(64 * GP_), 0, st, A, order);      \
            }                                                                                                             \
        } else {                                                                                                          \
            if (bf16) EPN_LAUNCH((inter_ungroup_shared_kernel<NT_, KT_, __bf16, GP_, nb_>), grid, dim3(64 * GP_), 0, st, A, order); \
            else EPN_LAUNCH((inter_ungroup_shared_kernel<NT_, KT_, float, GP_, nb_>), grid, dim3(64 * GP_), 0, st, A, order);      \
        }                                                                                                                 \
    } while (0)
        const int kt = (d->ks + 15) / 16;
        if (kt == 1) {
            if (nt <= 1) EPN_USH(1, 1, 8);
            else if (nt <= 2) { if (gp == 16) EPN_USH(2, 1, 16); else EPN_USH(2, 1, 8); }
            else if (nt <= 4) { if (gp == 8) EPN_USH(4, 1, 8); else EPN_USH(4, 1, 4); }
            else EPN_USH(8, 1, 2);
        } else {
            if (nt <= 1) EPN_USH(1, 2, 8);
            else if (nt <= 2) { if (gp == 16) EPN_USH(2, 2, 16); else EPN_USH(2, 2, 8); }
            else if (nt <= 4) { if (gp == 8) EPN_USH(4, 2, 8); else EPN_USH(4, 2, 4); }
            else EPN_USH(8, 2, 2);
        }
#undef EPN_USH
        EPN_CHECK_LAUNCH();
        return 0;
    }
    const unsigned grid = (unsigned)((A.ncol + 16 * NW - 1) / (16 * NW));
    A.col_tiles_per_wg = 1;
    const unsigned gy = (unsigned)(((d->cin >> 4) + A.col_tiles_per_wg - 1) / A.col_tiles_per_wg);
#define EPN_UGRP(NT_, KT_, dummy)                                                                                       \
    do {                                                                                                                \
        if (bf16) EPN_LAUNCH((inter_ungroup_kernel<NT_, KT_, __bf16>), dim3(grid, gy), dim3(64 * NW), 0, st, A); \
        else EPN_LAUNCH((inter_ungroup_kernel<NT_, KT_, float>), dim3(grid, gy), dim3(64 * NW), 0, st, A);      \
    } while (0)
    EPN_DISPATCH_NT_KT(EPN_UGRP, 0);
#undef EPN_UGRP
    EPN_CHECK_LAUNCH();
    return 0;
}

}  // namespace epn
