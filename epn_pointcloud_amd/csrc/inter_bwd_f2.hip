// Data gradient of InterSO3Conv with the gradient of the grouped features kept ON CHIP (round 6; review item 1a of round 5).
//
//   dF[b, idx[b,p,n], a, c] += sum_k w[b,p,a,k,n] * dG[col][c,k],     dG[col][c,k] = sum_o dOut[col][o] W[o][c*ks + k]
//
// replaces  autograd's transpose of  BasicSO3Conv.forward (vgtk/vgtk/so3conv/modules.py:48-55: dG = dOut . W)  chained with the
//           transpose of  inter_zpconv_grouping_naive (vgtk/vgtk/spconv/functional.py:372-390: gather backward = scatter-add)
// which the split form runs as a GEMM that WRITES dG[cols][cin*ks] (27 GB per cls step, B = 32) and a transpose kernel that
// reads it back (csrc/gemm_x3.hip + inter_ungroup_shared_kernel: 9.7 + 8.2 ms of a 60 ms step).  Here dG lives in registers.
//
// Structure = inter_ungroup_shared_kernel (csrc/inter_mfma.hip: a workgroup takes GP output points adjacent in Morton order, one
// wave each; per anchor the waves store per-slot contributions T[n][c] to an LDS tile, one barrier, one fp32 atomic per
// distinct (destination, anchor, channel)) with the load of the dG fragments replaced by the contraction that produces them:
//
//   * two-piece fp16 form (csrc/gemm.h): dOut scaled by 2^s from the device scalar max|dOut| and split in registers, W
//     pre-split ONCE per call into planes[2][cin*ks][cout] (tensor-wide power-of-two scale); hh + hl + lh on
//     v_mfma_f32_16x16x32_f16, fp32 accumulate; both scales are undone by folding 2^-a 2^-b into the per-slot multipliers.
//   * per (anchor tile of 16, 16-channel chunk): D_k[m = anchor][n = channel] = sum_o dOut[anchor][o] W[o][c, k], one MFMA
//     triple per kernel point k and 32 output channels: A = dOut fragments (resident per anchor tile), B = W planes staged
//     through a two-stage LDS ring by direct-to-LDS loads (48 KB per stage: 24 kernel points x 16 channels x 32 o x 2 planes;
//     16-byte slots swizzled so that the ds_read_b128 of a B fragment is conflict-free).  96 accumulator registers.
//   * the D fragment has lane = channel -- what the tail's contraction over the kernel points needs -- but register = anchor
//     where the tail wants register = kernel point: ONE in-register 4 x 4 transpose across the four 16-lane rows per (kernel
//     point slot, register) with v_permlane16_swap + v_permlane32_swap (96 swaps per tile and chunk: one per value and lane).
//     No LDS round trip for dG at all.
//   * the kernel-point order of the tail is free (both MFMA operands agree): the 24 points are dealt as 6 per lane group
//     (rows 16 + 4j + r of the rotated-kernel table hold point 16 + 2j + r for r < 2), so the second weight tile costs two
//     contraction MFMAs instead of four and every lane group owns six of the 24 D_k.
#include "inter_device.h"
#include "gemm.h"

namespace epn {
namespace {
EPN_F2_SENTINEL_DECL

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;
__device__ __forceinline__ void bf_glds16(const void *g, char *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((glb_void *)g, (lds_void *)lds_wave_base, 16, 0, 0);
}

constexpr int BF_KS = 24;            // kernel points (the shipped kernel set; other sets keep the split form)
constexpr int BF_TAB = 4096;         // destinations are de-duplicated through a direct-address table: p1 <= BF_TAB
constexpr int BF_SLAB = 2 * BF_KS * 1024;   // bytes of one stage: [plane][k][16 channels][4 slots of 16 bytes]

// kernel point held by (lane group j, slot s) of the tail's contraction: slots 0..3 = first weight tile (rows 4 j + s), slots
// 4, 5 = second tile (rows 16 + 4 j + r, r < 2  <->  point 16 + 2 j + r)
__host__ __device__ constexpr int bf_kidx(int j, int s) { return s < 4 ? 4 * j + s : 16 + 2 * j + (s - 4); }

// rotated-kernel table in that order: rk4p[a][row][4] = ((2/sigma) R_a kappa_k, beta_k) of the point row `row` stands for
__global__ void bf_rk4p_table_kernel(const float *__restrict__ anchors, const float *__restrict__ kernels, int na,
                                     float sigma_inv, float *__restrict__ rk4p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= na * EPN_KS_MAX) return;
    const int row = i % EPN_KS_MAX, a = i / EPN_KS_MAX;
    int k = -1;
    if (row < 16) k = row;
    else if (((row - 16) & 3) < 2) k = 16 + 2 * ((row - 16) >> 2) + ((row - 16) & 3);
    f32x4 v = {0.f, 0.f, 0.f, -1e30f};
    if (k >= 0 && k < BF_KS) {
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
    *reinterpret_cast<f32x4 *>(rk4p + (size_t)i * 4) = v;
}

// planes[p][ck][cout] (fp16) = two-piece split of W[o][ck] * 2^s, s from the device scalar *wmax (tensor-wide)
__global__ __launch_bounds__(256) void bf_wt_planes_kernel(const float *__restrict__ W, int cout, int ck, const float *__restrict__ wmax,
                                                          _Float16 *__restrict__ planes) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    const int c0 = blockIdx.x * 32, o0 = blockIdx.y * 32;
    const float s = f2_scale_of(*wmax);
#pragma unroll
    for (int i = ty; i < 32; i += 8) tile[i][tx] = W[(size_t)(o0 + i) * ck + c0 + tx] * s;
    __syncthreads();
    const size_t plane = (size_t)ck * cout;
#pragma unroll
    for (int i = ty; i < 32; i += 8) {
        const float v = tile[tx][i];
        const _Float16 h = (_Float16)v;
        const _Float16 l = (_Float16)(v - (float)h);
        const size_t at = (size_t)(c0 + i) * cout + o0 + tx;
        planes[at] = h;
        planes[plane + at] = l;
    }
}

struct BwdF2Args {
    InterArgs A;                 // gout = dOut [ncol][cout], out = dF (fp32, accumulated into), rk4 = the permuted table
    const _Float16 *planes;      // [2][cin*ks][cout]
    const float *go_amax, *w_amax;
    const int32_t *order;
    int chunks_per_wg;
};

__device__ __forceinline__ f32x4 mfma_f16_k32(gemm_f16x8 a, gemm_f16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// 4 x 4 transpose across the four 16-lane rows: on return lane row j of r[i] holds what lane row i of r[j] held
__device__ __forceinline__ void rows_transpose4(float &r0, float &r1, float &r2, float &r3) {
    auto p01 = __builtin_amdgcn_permlane16_swap(__float_as_uint(r0), __float_as_uint(r1), false, false);
    auto p23 = __builtin_amdgcn_permlane16_swap(__float_as_uint(r2), __float_as_uint(r3), false, false);
    auto q02 = __builtin_amdgcn_permlane32_swap(p01[0], p23[0], false, false);
    auto q13 = __builtin_amdgcn_permlane32_swap(p01[1], p23[1], false, false);
    r0 = __uint_as_float(q02[0]); r2 = __uint_as_float(q02[1]);
    r1 = __uint_as_float(q13[0]); r3 = __uint_as_float(q13[1]);
}

template <int NT, int GP, int NS, int NB>
__global__ __launch_bounds__(64 * GP) void inter_bwd_data_f2_kernel(BwdF2Args P) {
    const InterArgs &A = P.A;
    constexpr int EW = 16 * NT;        // neighbour slots per point (padded)
    constexpr int E = GP * EW;         // slots of the workgroup
    constexpr int SS = 20;             // floats per slot row: 16 channels + 4 (rows 4 apart fall on distinct banks)
    constexpr int NTH = 64 * GP;
    constexpr int CH = E / 64;
    constexpr int EPT = (E + NTH - 1) / NTH;
    constexpr int BS = (E + 1) * SS;   // floats per tile buffer: E slot rows + one row of zeros
    static_assert(E < 1024, "slot ids are packed in 10 bits");
    // ONE shared array (a second __shared__ object makes hipcc drain vmcnt in front of every LDS read of a direct-to-LDS
    // pipeline): [ring of two W stages | tile buffers | index arrays]
    constexpr int OFF_T = 2 * BF_SLAB;
    constexpr int OFF_I = OFF_T + NB * BS * 4;
    constexpr int NINT = ((6 * E + 1 + CH + 3) + 3) & ~3;
    // the rotated-kernel table of the launch (na <= 64 anchors x 32 rows x 4 floats): every anchor step of every chunk reads two
    // of its rows -- from global memory (v1) that was a ~1 us dependent load in front of each of the 60 x chunks steps of a
    // workgroup, the largest single item of the tail (profiles/r06_bwd_onchip_ablation_v1.txt)
    constexpr int OFF_R = OFF_I + NINT * 4;
    constexpr int TOTAL = OFF_R + 64 * EPN_KS_MAX * 4 * 4;
    static_assert(TOTAL <= 160 * 1024, "LDS");
    static_assert(2 * BF_SLAB >= BF_TAB * 4, "direct-address table aliases the W ring");
    __shared__ __attribute__((aligned(1024))) char smem[TOTAL];
    float *Tb = reinterpret_cast<float *>(smem + OFF_T);
    int *qlist = reinterpret_cast<int *>(smem + OFF_I);
    int *slot_of = qlist + E, *uq = slot_of + E, *cnt = uq + E, *off = cnt + E, *list = off + E + 1, *chunk_cnt = list + E;
    int *tab = reinterpret_cast<int *>(smem);   // set-up only
    float *rkl = reinterpret_cast<float *>(smem + OFF_R);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x = lane & 15, j = lane >> 4;
    const int groups = A.p2 / GP;
    const int blk = epn_xcd_tile(blockIdx.x, gridDim.x);
    const int bb = blk / groups, grp = blk - bb * groups;
    const int pp = P.order[(size_t)bb * A.p2 + grp * GP + wave];
    const int nchunk = A.cin >> 4;
    const int ct0 = blockIdx.y * P.chunks_per_wg;
    const int ct1 = min(ct0 + P.chunks_per_wg, nchunk);
    const int nct = ct1 - ct0;
    const int CK = A.cin * BF_KS;

    // ---- set-up: the distinct destinations of the workgroup's slots (as inter_ungroup_shared_kernel)
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
    for (int i = tid; i < A.na * EPN_KS_MAX; i += NTH)
        reinterpret_cast<f32x4 *>(rkl)[i] = reinterpret_cast<const f32x4 *>(A.rk4)[i];
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
            slot_of[e] = sl;
            myr[k] = atomicAdd(&cnt[sl], 1);
        }
    }
    __syncthreads();
    if (wave == 0) {
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
    __syncthreads();                            // also: tab (aliasing the W ring) is dead from here on
    int *pk = slot_of;
    for (int u = tid; u < U; u += NTH) {
        const int k0 = off[u], len = off[u + 1] - k0;
        const unsigned s0 = list[k0], s1 = len > 1 ? list[k0 + 1] : E, s2 = len > 2 ? list[k0 + 2] : E;
        pk[u] = (int)(s0 | (s1 << 10) | (s2 << 20) | (len > 3 ? 1u << 30 : 0u));
        cnt[u] = uq[u] * A.na * A.cin;          // element offset of the destination's gradient row inside the cloud
    }
    for (int i = tid; i < NB * SS; i += NTH) Tb[(i / SS) * BS + E * SS + i % SS] = 0.0f;

    // ---- scales of the two-piece form: both undone through the per-slot multipliers (powers of two: exact)
    const float a_scale = f2_scale_of(*P.go_amax);
    const float unscale = f2_inverse(a_scale) * f2_inverse(f2_scale_of(*P.w_amax));
    float hmul[NT][4];
    float gB[NT], alphaN[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        alphaN[t] = __shfl(h.gA[t], 48 + x, 64);
        gB[t] = j == 3 ? 1.0f : h.gA[t];
#pragma unroll
        for (int r = 0; r < 4; ++r) hmul[t][r] = h.mul[t][r] * unscale;
    }

    // ---- staging of the W planes: stage (ct, s) = [plane][k][channel 16 ct + c][o = 32 s .. 32 s + 31]; one 1 KiB wave
    // instruction per (plane, k): lane -> (c = lane / 4, slot = lane % 4); the slot of o-octet jj is (c / 4) ^ (-jj & 3), which
    // spreads the four lane groups of a ds_read_b128 fragment read over the bank row (a fragment = lane (x = c, j) <- octet j)
    const int sc = lane >> 2, sslot = lane & 3;
    const int sj = (-(sslot ^ (sc >> 2))) & 3;
    const _Float16 *sbase = P.planes + (size_t)(sc * BF_KS) * A.cout + 8 * sj;
    auto stage = [&](int stepi) {
        const int ci = (stepi / NS) % nct, s = stepi % NS;
        const int ct = ct0 + ci;
        char *dst = smem + (stepi & 1) * BF_SLAB;
#pragma unroll
        for (int i = 0; i < (2 * BF_KS + GP - 1) / GP; ++i) {
            const int q = wave + i * GP;
            if ((2 * BF_KS) % GP == 0 || q < 2 * BF_KS) {
                const int plane = q / BF_KS, k = q - plane * BF_KS;
                bf_glds16(sbase + ((size_t)plane * CK + (size_t)ct * 16 * BF_KS + k) * A.cout + 32 * s, dst + q * 1024);
            }
        }
    };
    const int rklane = x * 4 + j;
    const int rslot = ((x >> 2) ^ ((-j) & 3)) * 16;           // byte offset of this lane's slot inside a 64-byte row
    const int roff = x * 64 + rslot;

    const int NAT = (A.na + 15) >> 4;
    const int nsteps = NAT * nct * NS;
    const float *dout_pt = A.gout + ((size_t)bb * A.p2 + pp) * A.na * A.cout;
    float *dcloud = A.out + ((size_t)bb * A.p1) * A.na * A.cin;
    // dOut fragment of a step: A operand, lane (x = anchor of the tile, j) <- o = 32 s + 8 j .. + 7 of that anchor's row.  Loaded
    // one step ahead as raw fp32 (8 registers) and split when its step starts: resident fragments of a whole tile (cout / 4
    // registers) do not fit beside the 96 accumulators at cout = 256
    auto load_a = [&](int stepi_, f32x4 &u, f32x4 &v) {
        const int at_ = stepi_ / (NS * nct), s_ = stepi_ % NS;
        int arow = 16 * at_ + x;
        arow = arow < A.na ? arow : A.na - 1;
        const float *src = dout_pt + (size_t)arow * A.cout + 32 * s_ + 8 * j;
        u = *reinterpret_cast<const f32x4 *>(src);
        v = *reinterpret_cast<const f32x4 *>(src + 4);
    };
    __syncthreads();                            // set-up reads of tab are over: the ring may be written
    constexpr int NC = NT == 1 ? 2 : 1;         // cached gather-sum items per thread (512 threads x 16 channels: 32 destinations each)
    unsigned cpk[NC], coff[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        const int i = tid + q * NTH;
        const bool ok = i < U * 16;
        cpk[q] = ok ? (unsigned)pk[i >> 4] : 0u;
        coff[q] = ok ? (unsigned)cnt[i >> 4] + (unsigned)(i & 15) : 0u;
    }
    float rkn[2] = {0.f, 0.f};
    int rkn_a = -1;                             // anchor whose table rows rkn holds
    stage(0);
    f32x4 au, av;
    load_a(0, au, av);
    int stepi = 0;
    unsigned phase = 0;                         // tile buffers alternate across anchor steps
    float chk = 0.0f;
    for (int at = 0; at < NAT; ++at) {
        for (int ci = 0; ci < nct; ++ci) {
            const int ct = ct0 + ci;
            f32x4 acc[BF_KS];
#pragma unroll
            for (int k = 0; k < BF_KS; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();                // stage stepi has landed for every wave; the other buffer is free
                gemm_f16x8 ah, al;
                {
                    const float xx[8] = {au[0], au[1], au[2], au[3], av[0], av[1], av[2], av[3]};
                    f2_split8(xx, a_scale, ah, al);
                }
#ifdef EPN_TUNING
                if (!(A.wk & 4))
#endif
                if (stepi + 1 < nsteps) {
                    stage(stepi + 1);
                    load_a(stepi + 1, au, av);
                }
                const char *slab = smem + (stepi & 1) * BF_SLAB + roff;
#ifdef EPN_TUNING
                const bool skip_mfma = A.wk & 2;
#else
                constexpr bool skip_mfma = false;
#endif
                // Fragment reads one pair of kernel points ahead of their MFMAs, as inline assembly with hand-counted waits: left to
                // hipcc each ds_read_b128 sits directly in front of its three DEPENDENT MFMAs (read latency + three accumulate
                // latencies per kernel point, v1: 2400 cycles per step for 1152 cycles of matrix work).  Two kernel points form a
                // group: term by term across the pair, so that no MFMA waits for its predecessor's result.
                if (!skip_mfma) {
                    const unsigned sl = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char *)slab;
                    gemm_f16x8 fh[3][2], fl[3][2];          // [buffer][kernel point of the pair]: two pairs in flight ahead
#define BF_RD(buf_, q_, k_)                                                                                                    \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fh[buf_][q_]) : "v"(sl), "n"((k_) * 1024));                            \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fl[buf_][q_]) : "v"(sl), "n"((BF_KS + (k_)) * 1024))
#define BF_RD2(buf_, g_) do { BF_RD(buf_, 0, 2 * (g_)); BF_RD(buf_, 1, 2 * (g_) + 1); } while (0)
                    BF_RD2(0, 0); BF_RD2(1, 1);
#pragma unroll
                    for (int g = 0; g < BF_KS / 2; ++g) {
                        const int cb = g % 3, nb = (g + 2) % 3;
                        if (g + 2 < BF_KS / 2) {
                            if (nb == 0) BF_RD2(0, g + 2); else if (nb == 1) BF_RD2(1, g + 2); else BF_RD2(2, g + 2);
                            asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");      // two younger pairs may be outstanding
                        } else if (g + 1 < BF_KS / 2) {
                            asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                        } else {
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        }
                        if (cb == 0) { asm volatile("" : "+v"(fh[0][0]), "+v"(fl[0][0]), "+v"(fh[0][1]), "+v"(fl[0][1])); }
                        else if (cb == 1) { asm volatile("" : "+v"(fh[1][0]), "+v"(fl[1][0]), "+v"(fh[1][1]), "+v"(fl[1][1])); }
                        else { asm volatile("" : "+v"(fh[2][0]), "+v"(fl[2][0]), "+v"(fh[2][1]), "+v"(fl[2][1])); }
                        acc[2 * g] = mfma_f16_k32(ah, fl[cb][0], acc[2 * g]);
                        acc[2 * g + 1] = mfma_f16_k32(ah, fl[cb][1], acc[2 * g + 1]);
                        acc[2 * g] = mfma_f16_k32(al, fh[cb][0], acc[2 * g]);
                        acc[2 * g + 1] = mfma_f16_k32(al, fh[cb][1], acc[2 * g + 1]);
                        acc[2 * g] = mfma_f16_k32(ah, fh[cb][0], acc[2 * g]);
                        acc[2 * g + 1] = mfma_f16_k32(ah, fh[cb][1], acc[2 * g + 1]);
                    }
#undef BF_RD2
#undef BF_RD
                }
                __builtin_amdgcn_sched_barrier(0);
                ++stepi;
            }
            // acc[k]: lane (x = channel, j'), register r' = dG[anchor 16 at + 4 j' + r'][c][k] (scaled).  From here on the values
            // are used one register at a time: scalars, so that the swaps below do not drag 128-bit register tuples along
            float dg[BF_KS][4];
#pragma unroll
            for (int k = 0; k < BF_KS; ++k)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dg[k][r] = acc[k][r];
                    chk = fmaf(dg[k][r], 0.0f, chk);
                }
            // rows <-> kernel-point slots: afterwards dg[bf_kidx(ja, s)][ra] in lane group j = dG[anchor (ja, ra)][c][bf_kidx(j, s)]
#pragma unroll
            for (int s = 0; s < 6; ++s)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    rows_transpose4(dg[bf_kidx(0, s)][r], dg[bf_kidx(1, s)][r], dg[bf_kidx(2, s)][r], dg[bf_kidx(3, s)][r]);

            // ---- per-anchor tail: regenerated weights, contraction over the kernel points, LDS-reduced scatter
#pragma unroll
            for (int ra = 0; ra < 4; ++ra)
#pragma unroll
                for (int ja = 0; ja < 4; ++ja) {
                    const int a = 16 * at + 4 * ja + ra;
                    if (a >= A.na) continue;                               // wave-uniform (and workgroup-uniform)
#ifdef EPN_TUNING
                    if (A.wk & 1) {                                        // ablation: no tail (the values are still consumed)
                        float q = 0.f;
                        for (int r = 0; r < 4; ++r) q += dg[bf_kidx(ja, r)][ra];
                        for (int r = 0; r < 2; ++r) q += dg[bf_kidx(ja, 4 + r)][ra];
                        if (q == 12345.678f) atomicAdd(dcloud, q);
                        continue;
                    }
#endif
                    // rows of the rotated-kernel table: requested one anchor step ahead (an LDS round trip in front of the first
                    // MFMA of every step otherwise)
                    float rk[2];
                    if (rkn_a != a) {            // (first step of a tile, or the step before was past the last anchor)
                        rk[0] = rkl[a * (EPN_KS_MAX * 4) + rklane];
                        rk[1] = rkl[a * (EPN_KS_MAX * 4) + 64 + rklane];
                    } else {
                        rk[0] = rkn[0]; rk[1] = rkn[1];
                    }
                    float *buf = Tb + (phase & (NB - 1)) * BS + wave * EW * SS + x;
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        f32x4 w0 = {alphaN[t], alphaN[t], alphaN[t], alphaN[t]}, w1 = w0;
                        w0 = mfma4(rk[0], gB[t], w0);
                        w1 = mfma4(rk[1], gB[t], w1);
                        // two accumulation chains of three (a single chain of six waits for each predecessor's result)
                        f32x4 ta = {0.f, 0.f, 0.f, 0.f}, tb = ta;
                        ta = mfma4(relu_f(w0[0]), dg[bf_kidx(ja, 0)][ra], ta);
                        tb = mfma4(relu_f(w0[1]), dg[bf_kidx(ja, 1)][ra], tb);
                        ta = mfma4(relu_f(w0[2]), dg[bf_kidx(ja, 2)][ra], ta);
                        tb = mfma4(relu_f(w0[3]), dg[bf_kidx(ja, 3)][ra], tb);
                        ta = mfma4(relu_f(w1[0]), dg[bf_kidx(ja, 4)][ra], ta);
                        tb = mfma4(relu_f(w1[1]), dg[bf_kidx(ja, 5)][ra], tb);
                        // lane (x = c, j), register r -> slot n = 16 t + 4 j + r
#pragma unroll
                        for (int r = 0; r < 4; ++r) buf[(16 * t + 4 * j + r) * SS] = (ta[r] + tb[r]) * hmul[t][r];
                    }
                    lds_barrier();               // (not __syncthreads: that would wait for the previous step's atomics)
                    if (!(ra == 3 && ja == 3)) {
                        const int an = 16 * at + (ja == 3 ? ra + 1 : 4 * (ja + 1) + ra);       // next anchor of the tile
                        rkn[0] = rkl[an * (EPN_KS_MAX * 4) + rklane];
                        rkn[1] = rkl[an * (EPN_KS_MAX * 4) + 64 + rklane];
                        rkn_a = an;
                    }
                    const float *rb = Tb + (phase & (NB - 1)) * BS;
                    float *dstep = dcloud + (size_t)a * A.cin + 16 * ct;   // wave-uniform
                    // a thread's first NC (destination, channel) items are the same in every step: their slot words and row
                    // offsets live in registers (two dependent LDS round trips less per item and step)
#pragma unroll
                    for (int q = 0; q < NC; ++q) {
                        if (tid + q * NTH < U * 16) {
                            const unsigned e = cpk[q];
                            float sum = rb[(e & 1023u) * SS + (tid & 15)];
                            sum += rb[((e >> 10) & 1023u) * SS + (tid & 15)];
                            sum += rb[((e >> 20) & 1023u) * SS + (tid & 15)];
                            if (e >> 30) {
                                const int u = (tid + q * NTH) >> 4, k1 = off[u + 1];
                                for (int k = off[u] + 3; k < k1; ++k) sum += rb[list[k] * SS + (tid & 15)];
                            }
#ifdef EPN_TUNING
                            if (!(A.wk & 8) || sum == 12345.678f)
#endif
                            atomicAdd(dstep + coff[q], sum);
                        }
                    }
                    for (int i = tid + NC * NTH; i < U * 16; i += NTH) {
                        const int u = i >> 4, c = i & 15;
                        const unsigned e = (unsigned)pk[u];
                        float sum = rb[(e & 1023u) * SS + c];
                        sum += rb[((e >> 10) & 1023u) * SS + c];
                        sum += rb[((e >> 20) & 1023u) * SS + c];
                        if (e >> 30) {
                            const int k1 = off[u + 1];
                            for (int k = off[u] + 3; k < k1; ++k) sum += rb[list[k] * SS + c];
                        }
#ifdef EPN_TUNING
                        if (!(A.wk & 8) || sum == 12345.678f)
#endif
                        atomicAdd(dstep + ((unsigned)cnt[u] + (unsigned)c), sum);
                    }
                    if constexpr (NB == 1) lds_barrier();
                    ++phase;
                }
        }
    }
    EPN_F2_CHECK(chk);
}

size_t planes_bytes(const epn_inter_desc *d) { return ((size_t)4 * d->cin * d->ks * d->cout + 255) & ~(size_t)255; }

}  // namespace

long long f2_nonfinite_take_bwd(bool reset) {
    unsigned v = 0;
    hipError_t e = hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_f2_nonfinite), sizeof(v), 0, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return -(long long)e;
    if (reset && v) {
        const unsigned zero = 0;
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_f2_nonfinite), &zero, sizeof(zero), 0, hipMemcpyHostToDevice);
        if (e != hipSuccess) return -(long long)e;
    }
    return (long long)v;
}

bool inter_bwd_f2_ok(const epn_inter_desc *d) {
    if (!inter_group_mfma_ok(d)) return false;
    const int nt = (d->nn + 15) / 16;
    return d->ks == BF_KS && d->na >= 16 && d->na <= 64 && nt <= 2 && d->p2 % 8 == 0 && d->p2 <= 4096 && d->p1 <= BF_TAB &&
           (d->cout == 64 || d->cout == 128 || d->cout == 256) && d->cin % 16 == 0 &&
           (long long)d->p1 * d->na * d->cin < (1LL << 31);
}

// workspace (bytes) behind the rotated-kernel tables: the fp16 planes of W, then 256 bytes for max|W|
size_t inter_bwd_f2_extra_bytes(const epn_inter_desc *d) { return planes_bytes(d) + 256; }

int launch_inter_bwd_f2(const epn_inter_desc *d, float *rk4p, int32_t *order, const float *dOut, const float *W,
                        const float *go_amax, float *dF, void *extra, hipStream_t st) {
    _Float16 *planes = static_cast<_Float16 *>(extra);
    float *wmax = reinterpret_cast<float *>(static_cast<char *>(extra) + planes_bytes(d));
    const int ck = d->cin * d->ks;
    int rc = launch_absmax(W, ck, 1, (long long)d->cout * ck, wmax, st);
    if (rc) return rc;
    EPN_LAUNCH_AUX(bf_wt_planes_kernel, dim3(ck / 32, d->cout / 32), dim3(256), 0, st, W, d->cout, ck, wmax, planes);
    EPN_CHECK_LAUNCH();
    EPN_LAUNCH_AUX(bf_rk4p_table_kernel, dim3(epn_cdiv(d->na * EPN_KS_MAX, 256)), dim3(256), 0, st, d->anchors, d->kernels, d->na,
                   1.0f / d->sigma, rk4p);
    EPN_CHECK_LAUNCH();
    rc = launch_morton_order(d->new_xyz, d->b, d->p2, order, st);
    if (rc) return rc;
    BwdF2Args P;
    InterArgs &A = P.A;
    A.xyz = d->xyz; A.new_xyz = d->new_xyz; A.idx = d->ball_idx; A.rk4 = rk4p;
    A.feats = nullptr; A.W = nullptr; A.gout = dOut; A.out = dF;
    A.sigma_inv = 1.0f / d->sigma;
    A.b = d->b; A.p1 = d->p1; A.p2 = d->p2; A.nn = d->nn; A.na = d->na; A.ks = d->ks; A.cin = d->cin; A.cout = d->cout;
    A.wk = 0; A.packed = 0; A.ncol = (long long)d->b * d->p2 * d->na; A.col_tiles_per_wg = 1;
#ifdef EPN_TUNING
    if ((kernel_policy() & 0xf00) == 0x900) A.wk = kernel_policy() & 0xff;   // ablation bits (tools/bwd_onchip_probe.py)
#endif
    P.planes = planes; P.go_amax = go_amax; P.w_amax = wmax; P.order = order;
    constexpr int GP = 8;
    const int nchunk = d->cin >> 4;
    const int gx = d->b * (d->p2 / GP);
    // chunks per workgroup: the set-up and the dOut fragments are paid once per workgroup; keep >= ~2048 workgroups in flight
    int cpw = 4;
    while (cpw > 1 && ((long long)gx * ((nchunk + cpw - 1) / cpw) < 2048 || nchunk % cpw)) cpw >>= 1;
    P.chunks_per_wg = cpw;
    const dim3 grid((unsigned)gx, (unsigned)((nchunk + cpw - 1) / cpw));
    const int nt = (d->nn + 15) / 16, ns = d->cout / 32;
#define EPN_BF2(NT_, NS_)                                                                                                  \
    EPN_LAUNCH((inter_bwd_data_f2_kernel<NT_, GP, NS_, (NT_ == 1 ? 2 : 1)>), grid, dim3(64 * GP), 0, st, P)
    if (nt <= 1) {
        if (ns == 2) EPN_BF2(1, 2); else if (ns == 4) EPN_BF2(1, 4); else EPN_BF2(1, 8);
    } else {
        if (ns == 2) EPN_BF2(2, 2); else if (ns == 4) EPN_BF2(2, 4); else EPN_BF2(2, 8);
    }
#undef EPN_BF2
    EPN_CHECK_LAUNCH();
    return 0;
}

}  // namespace epn
