// InterSO3Conv with the grouped features kept ON CHIP: the grouping is the A-tile producer of the weight contraction.
//
//   out[col][o] = sum_{c,k} W[o, c*ks + k] * G[col][c, k],     G[col][c, k] = sum_n w[col][k][n] F[idx[n], a, c]
//   (vgtk/vgtk/spconv/functional.py:372-390 -> vgtk/vgtk/so3conv/modules.py:48-55,157-174)
//
// The split form (inter_group_kernel + gemm_nt) writes G[cols][cin*ks] to HBM (3-6 GB per layer) and reads it back; here a
// workgroup owns PT whole output points (64 tile rows per point: rows a >= na are padding), and for every chunk of 16
// input channels its eight waves
//   1. regenerate the kernel-influence weights of their own columns by S-MFMA and contract over the neighbours
//      (inter_device.h: exact-f32 v_mfma_f32_16x16x4_f32 for fp32 features, v_mfma_f32_16x16x32_bf16 for bf16 features),
//   2. deposit the D fragments -- split into three bf16 planes h + m + l for fp32 features (gemm_x3.hip: lossless) -- as
//      rows of an LDS tile whose positions follow the fragment layout (the weights are permuted to match, once per call),
//   3. run the weight contraction on v_mfma_f32_32x32x16_bf16 (six piece products per multiply for fp32 features) against
//      W streamed through a two-stage LDS ring by global_load_lds, exactly as gemm_nt_x3_kernel does.
// A chunk is consumed as "sub-quanta" of 128 (or 256) contraction positions: kernel points 0..15 (D registers {0,1} then
// {2,3} for fp32 features, whose three planes would not fit otherwise: registers {2,3} wait in VGPRs), then kernel points
// 16..ks-1.  Nothing of size [cols, cin*ks] exists anywhere.
#include "conv_internal.h"
#include "gemm.h"
#include "inter_device.h"

namespace epn {
namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void glds16(const void *g, char *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((glb_void *)g, (lds_void *)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ unsigned pack_rne(float a, float b) {   // v_cvt_pk_bf16_f32
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float lo_f(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float hi_f(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// two fp32 values -> one packed dword per bf16 plane (h, m, l), lossless (gemm_x3.hip)
__device__ __forceinline__ void split_pair(float a, float b, unsigned &h, unsigned &m, unsigned &l) {
    h = pack_rne(a, b);
    const float r0 = a - lo_f(h), r1 = b - hi_f(h);
    m = pack_rne(r0, r1);
    l = pack_rne(r0 - lo_f(m), r1 - hi_f(m));
}

struct FxArgs {
    InterArgs I;
    const void *Wp;     // [NP][cout][KP] bf16 planes of W in the kernel's contraction order (fx_prep_w_kernel)
    void *out;          // [ncol][cout] fp32 / bf16
    int KP;             // contraction positions = (cin / 16) * (256 + kq1)
    int kq1;            // positions of the second kernel-point block of a chunk: 16 * (ks - 16)
    int npts;           // b * p2
};

// Contraction position -> (channel, kernel point).  A chunk of 16 channels occupies QC = 256 + kq1 positions:
//   HALF (fp32 features):  [0,128)   lane (x, j) registers {0,1}: p = 2 (16j + x) + r        k = 4j + r
//                          [128,256) registers {2,3}:             p = 2 (16j + x) + (r - 2)  k = 4j + r
//   FULL (bf16 features):  [0,256)   p = 4 (16j + x) + r                                     k = 4j + r
//   both:                  [256, 256 + kq1)  p = 4 (16j + x) + r                              k = 16 + 4j + r
// channel = 16 ct + x.
__device__ __forceinline__ void fx_position(int half, int pos, int &x, int &k) {
    if (pos >= 256) {
        const int p = pos - 256;
        const int l = p >> 2, r = p & 3;
        x = l & 15; k = 16 + 4 * (l >> 4) + r;
    } else if (half) {
        const int p = pos & 127, hi = pos >> 7;
        const int l = p >> 1, r = (p & 1) + 2 * hi;
        x = l & 15; k = 4 * (l >> 4) + r;
    } else {
        const int l = pos >> 2, r = pos & 3;
        x = l & 15; k = 4 * (l >> 4) + r;
    }
}

// W[cout][cin*ks] fp32 -> planes [NP][cout][KP] bf16 in position order (NP = 3: lossless split; NP = 1: rounded)
__global__ void fx_prep_w_kernel(const float *__restrict__ W, int cout, int cin, int ks, int kq1, int KP, int np,
                                 unsigned *__restrict__ planes) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // pair index
    const int kp2 = KP >> 1;
    if (i >= (long long)cout * kp2) return;
    const int o = (int)(i / kp2), pos0 = 2 * (int)(i % kp2);
    const int qc = 256 + kq1;
    float v[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int pos = pos0 + e;
        const int ct = pos / qc;
        int x, k;
        fx_position(np == 3, pos - ct * qc, x, k);
        v[e] = k < ks ? W[(size_t)o * cin * ks + (size_t)(16 * ct + x) * ks + k] : 0.0f;
    }
    const size_t plane = (size_t)cout * kp2;
    if (np == 3) {
        unsigned h, m, l;
        split_pair(v[0], v[1], h, m, l);
        planes[i] = h; planes[plane + i] = m; planes[2 * plane + i] = l;
    } else {
        planes[i] = pack_rne(v[0], v[1]);
    }
}

// One production item = one column (anchor) of this wave for one kernel-point block kt of one 16-channel chunk: its
// gathers / table reads are ISSUED one stage before its MFMAs run (the stage barrier's vmcnt(0) lands them), and its D
// fragment waits in registers until the phase that consumes it starts.  Per chunk (ks = 24):
//   fp32 features:  A (4 stages: kernel points 0..15, D registers {0,1})   produces kt = 1 of this chunk, columns 0 .. CPW/2
//                   B (4 stages: D registers {2,3})                        produces kt = 1, columns CPW/2 .. CPW
//                   C (4 stages: kernel points 16..23)                     produces kt = 0 of the NEXT chunk
//   bf16 features:  A (8 stages: kernel points 0..15)                      produces kt = 1 of this chunk
//                   C (4 stages)                                           produces kt = 0 of the next chunk
template <typename TF, int NT, int PT, int WGM, int WGN, int TN, int KS>
__global__ __launch_bounds__(512) void inter_fx_fwd_kernel(FxArgs F) {
    constexpr bool X3 = std::is_same<TF, float>::value;
    constexpr int NP = X3 ? 3 : 1;
    constexpr bool HALF = X3;
    constexpr int NWV = 8;
    static_assert(WGM * WGN * KS == NWV, "eight waves");
    constexpr int BM = 64 * PT, TM = BM / (32 * WGM), BN = WGN * TN * 32, CPW = BM / NWV;
    static_assert(TM >= 1 && TM * 32 * WGM == BM, "row tiles");
    constexpr int PITCH = HALF ? 256 : 512;
    constexpr int A_PLANE = BM * PITCH, A_BYTES = NP * A_PLANE;
    constexpr int P_BYTES = BN * 64, W_STAGE = NP * P_BYTES;
    constexpr int NGB = NP * BN / 16;                 // 1 KiB wave-level load instructions per W stage
    constexpr int GPW = (NGB + NWV - 1) / NWV;
    constexpr int NACC = (X3 && TM * TN == 1) ? 2 : 1;   // a lone tile: two accumulators break the six-MFMA dependency chain
    static_assert(A_BYTES + 2 * W_STAGE <= 160 * 1024, "LDS");
    static_assert(KS == 1 || BM * BN * 4 <= A_BYTES, "K-split reduction buffer");
    __shared__ __attribute__((aligned(1024))) char smem[A_BYTES + 2 * W_STAGE];
    char *const atile = smem;
    char *const wring = smem + A_BYTES;

    const InterArgs &A = F.I;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x = lane & 15, j = lane >> 4;
    const unsigned tile = epn_xcd_tile(blockIdx.x, gridDim.x);
    const int pt0 = (int)tile * PT;

    // ---- producer set-up: this wave's CPW rows belong to ONE output point
    const int prow0 = wave * CPW;                       // first tile row of this wave
    const int pl = prow0 >> 6, a_base = prow0 & 63;
    int ptw = pt0 + pl;
    ptw = ptw < F.npts ? ptw : F.npts - 1;
    const int bb = ptw / A.p2, pp = ptw - bb * A.p2;
    Hood<NT> h;
    load_hood<NT>(A, bb, pp, x, j, h);
    const TF *fslab = reinterpret_cast<const TF *>(A.feats) + ((size_t)bb * A.p1) * A.na * A.cin;
    unsigned okmask[NT][2];                              // bf16: 0xffff per valid neighbour slot, packed like the values
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        okmask[t][0] = (h.ok[t][0] ? 0xffffu : 0u) | (h.ok[t][1] ? 0xffff0000u : 0u);
        okmask[t][1] = (h.ok[t][2] ? 0xffffu : 0u) | (h.ok[t][3] ? 0xffff0000u : 0u);
    }
    // Gathers and table reads use buffer loads: descriptor = this cloud's feature slab / the (R_a kappa_k) table
    // (wave-uniform), voffset = neighbour row + this lane's channel (column-independent), soffset = anchor * cin + chunk
    // (wave-uniform): nothing per column is left for the compiler to precompute and keep alive across the unrolled columns.
    const __amdgpu_buffer_rsrc_t rF = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<TF *>(fslab), 0, (unsigned)A.p1 * A.na * A.cin * (unsigned)sizeof(TF), 0x00020000);
    const __amdgpu_buffer_rsrc_t rT = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(A.rk4), 0, (unsigned)A.na * EPN_KS_MAX * 16u, 0x00020000);
    int vq[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) vq[t][r] = (h.q[t][r] + x) * (int)sizeof(TF);
    const int vt_rk = (x * 4 + (j < 3 ? j : 3)) * 4, vt_beta = (x * 4 + 3) * 4;

    struct Pend {                                        // loads of one production item in flight
        float f[NT][4];
        float e, beta;
    };
    auto issue = [&](Pend &P, int a, int ct, int kt) {
        const int soff = (a * A.cin + 16 * ct) * (int)sizeof(TF);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if constexpr (X3) P.f[t][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rF, vq[t][r], soff, 0));
                else P.f[t][r] = __uint_as_float((unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rF, vq[t][r], soff, 0));
            }
        const int toff = (a * EPN_KS_MAX + 16 * kt) * 16;
        P.e = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rT, vt_rk, toff, 0));
        P.beta = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rT, vt_beta, toff, 0));
    };
    // weights by S-MFMA (make_weights of inter_device.h, one kernel-point block) + neighbour contraction -> D fragment:
    // lane (x, j), register r -> kernel point 16kt + 4j + r, channel x
    auto compute = [&](const Pend &P) -> f32x4 {
        const float rk = j == 3 ? 1.0f : P.e;
        f32x4 w[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f32x4 sv = {P.beta, P.beta, P.beta, P.beta};
            sv = mfma4(h.gA[t], rk, sv);
#pragma unroll
            for (int r = 0; r < 4; ++r) sv[r] = relu_f(sv[r]);
            w[t] = sv;
        }
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        if constexpr (X3) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) g = mfma4(w[t][r], h.ok[t][r] ? P.f[t][r] : 0.0f, g);
        } else {
            bf16x4_t fb4[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                u32x2 pk;
                pk[0] = (__float_as_uint(P.f[t][0]) | (__float_as_uint(P.f[t][1]) << 16)) & okmask[t][0];
                pk[1] = (__float_as_uint(P.f[t][2]) | (__float_as_uint(P.f[t][3]) << 16)) & okmask[t][1];
                fb4[t] = __builtin_bit_cast(bf16x4_t, pk);
            }
            if constexpr (NT % 2 == 0) {
#pragma unroll
                for (int t = 0; t < NT; t += 2) g = mfma_bf16_k32(pack4(w[t]), pack4(w[t + 1]), fb4[t], fb4[t + 1], g);
            } else {
#pragma unroll
                for (int t = 0; t < NT; ++t) g = mfma_bf16_k16(pack4(w[t]), fb4[t], g);
            }
        }
        return g;
    };
    // D fragments waiting for their phase: fp32 features keep fp32 (split at the deposit), bf16 features keep packed pairs
    constexpr int GR = X3 ? 4 : 2;
    float gk0[CPW][GR], gk1[CPW][GR];
    auto keep = [&](float (&dst)[GR], const f32x4 g) {
        if constexpr (X3) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[r] = g[r];
        } else {
            dst[0] = __uint_as_float(pack_rne(g[0], g[1]));
            dst[1] = __uint_as_float(pack_rne(g[2], g[3]));
        }
    };
#pragma unroll
    for (int q = 0; q < CPW; ++q)
#pragma unroll
        for (int r = 0; r < GR; ++r) gk0[q][r] = gk1[q][r] = 0.f;

    // deposits (rows of padding anchors carry zeros)
    auto deposit_pair = [&](int q, float v0, float v1) {   // HALF: two D registers -> positions 2 (16j + x) + {0, 1}
        unsigned hh, mm, ll;
        split_pair(v0, v1, hh, mm, ll);
        const int row = prow0 + q;
        char *arow = atile + row * PITCH;
        const int off = (((4 * j + (x >> 2)) ^ (row & 15)) << 4) + (x & 3) * 4;
        *reinterpret_cast<unsigned *>(arow + off) = hh;
        *reinterpret_cast<unsigned *>(arow + A_PLANE + off) = mm;
        *reinterpret_cast<unsigned *>(arow + 2 * A_PLANE + off) = ll;
    };
    auto deposit_quad = [&](int q, const float (&g)[GR], bool lanes) {   // four D registers -> positions 4 (16j + x) + r
        const int row = prow0 + q;
        char *arow = atile + row * PITCH;
        const int off = (((8 * j + (x >> 1)) ^ (row & 15)) << 4) + (x & 1) * 8;
        if constexpr (X3) {
            unsigned h0, m0, l0, h1, m1, l1;
            split_pair(g[0], g[1], h0, m0, l0);
            split_pair(g[X3 ? 2 : 0], g[X3 ? 3 : 1], h1, m1, l1);
            if (lanes) {
                *reinterpret_cast<u32x2 *>(arow + off) = u32x2{h0, h1};
                *reinterpret_cast<u32x2 *>(arow + A_PLANE + off) = u32x2{m0, m1};
                *reinterpret_cast<u32x2 *>(arow + 2 * A_PLANE + off) = u32x2{l0, l1};
            }
        } else {
            if (lanes) *reinterpret_cast<u32x2 *>(arow + off) = u32x2{__float_as_uint(g[0]), __float_as_uint(g[1])};
        }
    };

    // ---- W ring: source pointers of this wave's load instructions (as gemm_nt_x3_kernel, planes only)
    const char *wsrc[GPW];
#pragma unroll
    for (int i = 0; i < GPW; ++i) {
        const int g = wave + i * NWV;
        const int rb = 16 * g + lane / 4;               // plane * BN + weight row of the tile
        const int plane = rb / BN, n = rb % BN;
        const int slot = (lane % 4) ^ ((rb >> 2) & 3);
        int gn = n < A.cout ? n : A.cout - 1;
        wsrc[i] = reinterpret_cast<const char *>(static_cast<const __bf16 *>(F.Wp) + ((size_t)(plane < NP ? plane : 0) * A.cout + gn) * F.KP + slot * 8);
    }
    auto wstage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < GPW; ++i) {
            const int g = wave + i * NWV;
            if (NGB % NWV == 0 || g < NGB) {
                glds16(wsrc[i], wring + buf * W_STAGE + g * 1024);
                wsrc[i] += 64;
            }
        }
    };

    // ---- consumer set-up
    const int kh = wave / (WGM * WGN);                   // K split: this wave's 16-position half of every stage
    const int wmn = wave % (WGM * WGN);
    const int wm = wmn / WGN, wn = wmn % WGN;
    const int li = lane & 31, lj = lane >> 5;
    const int fswB = (li >> 2) & 3;
    int aoff[TM], boff[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) aoff[i] = ((wm * TM + i) * 32 + li) * PITCH;
#pragma unroll
    for (int i = 0; i < TN; ++i) boff[i] = ((wn * TN + i) * 32 + li) * 64;
    const int fswA = li & 15;                            // row & 15 (tile rows of a fragment are 32-aligned + li)

    f32x16 acc[NACC][TM][TN];
#pragma unroll
    for (int c = 0; c < NACC; ++c)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][i][jn][r] = 0.f;

    const int nstage = F.KP / 32;
    const int nchunk = A.cin >> 4;
    int gs = 0;                                          // W stage counter
    wstage(0);

    constexpr int CSMAX = CPW / 4 > 0 ? CPW / 4 : 1;
    Pend pend[CSMAX];

    // One phase = NST stages over the A tile as deposited.  Stage s runs the MFMAs of the CS production items whose loads
    // are pending (kernel-point block KT of chunk ct_job, columns Q0 + s CS ..) and issues the loads of the following
    // items: the next stage's, or (last stage) the first CSN columns of the next phase's job (KTN of chunk ct_next).
    auto phase = [&](auto nst_c, auto kt_c, auto q0_c, auto cs_c, auto ktn_c, auto q0n_c, auto csn_c, int ct_job, int ct_next,
                     float (&gdst)[CPW][GR]) {
        constexpr int NST = decltype(nst_c)::value, KT = decltype(kt_c)::value, Q0 = decltype(q0_c)::value;
        constexpr int CS = decltype(cs_c)::value, KTN = decltype(ktn_c)::value, CSN = decltype(csn_c)::value;
        constexpr int Q0N = decltype(q0n_c)::value;
        int ab = a_base;
        asm volatile("" : "+s"(ab));                     // opaque: per-column scalars are not hoisted out of the chunk loop
#pragma unroll
        for (int s = 0; s < NST; ++s) {
            // explicit: hipcc's own count lets a direct-to-LDS load issued before other loads ride on their partial
            // vmcnt(N) waits -- measured on gfx950: the ring slot is then read before it landed
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                             // deposits visible, W stage gs landed, ring slot (gs+1)&1 free
            if (gs + 1 < nstage) wstage((gs + 1) & 1);
            // ---- production: this stage's items
            if (ct_job < nchunk) {
#pragma unroll
                for (int c = 0; c < CS; ++c) {
                    const int q = Q0 + s * CS + c;
                    if (ab + q < A.na) keep(gdst[q], compute(pend[c]));
                }
            }
            if (s + 1 < NST) {
                if (ct_job < nchunk) {
#pragma unroll
                    for (int c = 0; c < CS; ++c) issue(pend[c], ab + Q0 + (s + 1) * CS + c, ct_job, KT);
                }
            } else if (ct_next < nchunk) {
#pragma unroll
                for (int c = 0; c < CSN; ++c) issue(pend[c], ab + Q0N + c, ct_next, KTN);
            }
            // ---- weight contraction over this stage's 32 positions
            const char *wb = wring + (gs & 1) * W_STAGE;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                if (KS == 2 && sub != kh) continue;
                const int slotA = 2 * (2 * s + sub) + lj;
                const int sa = (slotA ^ fswA) * 16;
                const int sb = ((2 * sub + lj) ^ fswB) * 16;
                bf16x8 af[TM][NP], bfr[TN][NP];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int p = 0; p < NP; ++p)
                        af[i][p] = *reinterpret_cast<const bf16x8 *>(atile + p * A_PLANE + aoff[i] + sa);
#pragma unroll
                for (int jn = 0; jn < TN; ++jn)
#pragma unroll
                    for (int p = 0; p < NP; ++p)
                        bfr[jn][p] = *reinterpret_cast<const bf16x8 *>(wb + p * P_BYTES + boff[jn] + sb);
#define EPN_FX_TERM(C_, PA, PB)                                                                              \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int jn = 0; jn < TN; ++jn)         \
        acc[C_][i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PA], bfr[jn][PB], acc[C_][i][jn], 0, 0, 0)
                if constexpr (X3) {
                    EPN_FX_TERM(0, 0, 2); EPN_FX_TERM(NACC - 1, 2, 0); EPN_FX_TERM(0, 1, 1);      // small terms first
                    EPN_FX_TERM(NACC - 1, 0, 1); EPN_FX_TERM(0, 1, 0); EPN_FX_TERM(NACC - 1, 0, 0);
                } else {
                    EPN_FX_TERM(0, 0, 0);
                }
#undef EPN_FX_TERM
            }
            ++gs;
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I4 = std::integral_constant<int, 4>;
    using I8 = std::integral_constant<int, 8>;
    using IH = std::integral_constant<int, CPW / 2>;
    using IC8 = std::integral_constant<int, (CPW / 8 > 0 ? CPW / 8 : 1)>;
    using IC4 = std::integral_constant<int, CPW / 4>;
    static_assert(CPW >= 8, "eight columns per wave at least");

    // ---- prologue: kernel points 0..15 of chunk 0, not overlapped; then the first loads of the pipeline
    {
        int ab = a_base;
        asm volatile("" : "+s"(ab));
#pragma unroll
        for (int q0 = 0; q0 < CPW; q0 += CSMAX) {
#pragma unroll
            for (int c = 0; c < CSMAX; ++c) issue(pend[c], ab + q0 + c, 0, 0);
#pragma unroll
            for (int c = 0; c < CSMAX; ++c)
                if (ab + q0 + c < A.na) keep(gk0[q0 + c], compute(pend[c]));
        }
#pragma unroll
        for (int c = 0; c < IC8::value; ++c) issue(pend[c], ab + c, 0, 1);
    }

    for (int ct = 0; ct < nchunk; ++ct) {
        __syncthreads();                                 // every wave is done reading the A tile
        if constexpr (HALF) {
#pragma unroll
            for (int q = 0; q < CPW; ++q) deposit_pair(q, gk0[q][0], gk0[q][1]);
            phase(I4{}, I1{}, I0{}, IC8{}, I1{}, IH{}, IC8{}, ct, ct, gk1);      // A: produces kt 1, columns 0 .. CPW/2
            __syncthreads();
#pragma unroll
            for (int q = 0; q < CPW; ++q) deposit_pair(q, gk0[q][GR - 2], gk0[q][GR - 1]);
            phase(I4{}, I1{}, IH{}, IC8{}, I0{}, I0{}, IC4{}, ct, ct + 1, gk1);  // B: kt 1, columns CPW/2 .. CPW
        } else {
#pragma unroll
            for (int q = 0; q < CPW; ++q) deposit_quad(q, gk0[q], true);
            phase(I8{}, I1{}, I0{}, IC8{}, I0{}, I0{}, IC4{}, ct, ct + 1, gk1);  // A: 8 stages, produces kt 1
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < CPW; ++q) deposit_quad(q, gk1[q], 16 + 4 * j < A.ks);
        phase(I4{}, I0{}, I0{}, IC4{}, I1{}, I0{}, IC8{}, ct + 1, ct + 1, gk0);  // C: produces kt 0 of the next chunk
    }

    // ---- K split: the second half's partial sums go through LDS (the A tile is free now)
    if constexpr (NACC == 2) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][i][jn][r] += acc[1][i][jn][r];
    }
    if constexpr (KS == 2) {
        __syncthreads();
        float *red = reinterpret_cast<float *>(atile) + (size_t)wmn * (TM * TN * 16 * 64);
        if (kh == 1) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jn = 0; jn < TN; ++jn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[((i * TN + jn) * 16 + r) * 64 + lane] = acc[0][i][jn][r];
        }
        __syncthreads();
        if (kh == 1) return;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][i][jn][r] += red[((i * TN + jn) * 16 + r) * 64 + lane];
    }

    // ---- epilogue: D[row = (r&3) + 8 (r>>2) + 4 lj][n = li]; tile row -> (point, anchor)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lj;
            const int a = row & 63, pt = pt0 + (row >> 6);
            if (a < A.na && pt < F.npts) {
                const size_t o0 = ((size_t)pt * A.na + a) * A.cout;
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) {
                    const int n = (wn * TN + jn) * 32 + li;
                    if (n < A.cout) {
                        if constexpr (X3) static_cast<float *>(F.out)[o0 + n] = acc[0][i][jn][r];
                        else static_cast<__bf16 *>(F.out)[o0 + n] = (__bf16)acc[0][i][jn][r];
                    }
                }
            }
        }
}

}  // namespace

// ---- host side --------------------------------------------------------------------------------------------------
static inline int fx_kq1(const epn_inter_desc *d) { return 16 * (d->ks - 16); }
static inline int fx_kp(const epn_inter_desc *d) { return (d->cin / 16) * (256 + fx_kq1(d)); }

bool inter_fx_ok(const epn_inter_desc *d, int bf16) {
    (void)bf16;
    return inter_mfma_available() && !d->dense_w && d->cin % 16 == 0 && d->cin >= 16 && d->cout % 32 == 0 &&
           d->cout >= 32 && d->cout <= 256 && d->ks == 24 && d->nn <= 64 && d->na > 32 &&
           d->na <= 64 && (long long)d->p1 * d->na * d->cin < (1LL << 31);
}

size_t inter_fx_planes_bytes(const epn_inter_desc *d, int bf16) {
    return (((size_t)(bf16 ? 1 : 3) * d->cout * fx_kp(d) * 2) + 255) & ~(size_t)255;
}

template <typename TF, int NT>
static int launch_fx_fwd_nt(const FxArgs &F, const epn_inter_desc *d, hipStream_t st) {
    constexpr bool X3 = std::is_same<TF, float>::value;
#define EPN_FX_GO(PT_, WGM_, WGN_, TN_, KS_)                                                                         \
    do {                                                                                                             \
        const unsigned grid = (unsigned)((F.npts + (PT_) - 1) / (PT_));                                              \
        EPN_LAUNCH((inter_fx_fwd_kernel<TF, NT, PT_, WGM_, WGN_, TN_, KS_>), dim3(grid), dim3(512), 0, st, F); \
        EPN_CHECK_LAUNCH();                                                                                          \
        return 0;                                                                                                    \
    } while (0)
    if constexpr (X3) {                                // one output point per workgroup: 64 tile rows
        if (d->cout > 128) EPN_FX_GO(1, 2, 4, 2, 1);   // 64 x 256: A tile 48 KB + W ring 96 KB
        if (d->cout > 64) EPN_FX_GO(1, 2, 4, 1, 1);    // 64 x 128
        EPN_FX_GO(1, 2, 2, 1, 2);                      // 64 x 64, the stage's two 16-position halves on two wave groups
    } else {
        if (d->cout > 128) EPN_FX_GO(2, 4, 2, 4, 1);   // 128 x 256: A tile 64 KB + W ring 32 KB
        if (d->cout > 64) EPN_FX_GO(2, 4, 2, 2, 1);
        if (d->cout > 32) EPN_FX_GO(2, 4, 2, 1, 1);
        EPN_FX_GO(2, 4, 1, 1, 2);
    }
#undef EPN_FX_GO
}

// planes: workspace of inter_fx_planes_bytes(d, bf16) bytes
int launch_inter_fx_fwd(const epn_inter_desc *d, const float *rk4, const void *feats, const float *W, void *out,
                        void *planes, int bf16, hipStream_t st) {
    FxArgs F;
    F.I.xyz = d->xyz; F.I.new_xyz = d->new_xyz; F.I.idx = d->ball_idx; F.I.rk4 = rk4;
    F.I.feats = static_cast<const float *>(feats); F.I.W = nullptr; F.I.gout = nullptr; F.I.out = nullptr;
    F.I.sigma_inv = 1.0f / d->sigma;
    F.I.b = d->b; F.I.p1 = d->p1; F.I.p2 = d->p2; F.I.nn = d->nn; F.I.na = d->na; F.I.ks = d->ks; F.I.cin = d->cin;
    F.I.cout = d->cout; F.I.wk = 0;
    F.I.ncol = (long long)d->b * d->p2 * d->na;
    F.I.col_tiles_per_wg = 1;
    F.Wp = planes; F.out = out; F.KP = fx_kp(d); F.kq1 = fx_kq1(d); F.npts = d->b * d->p2;
    const long long pairs = (long long)d->cout * (F.KP / 2);
    EPN_LAUNCH_AUX(fx_prep_w_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, st, W, d->cout, d->cin, d->ks,
                       F.kq1, F.KP, bf16 ? 1 : 3, static_cast<unsigned *>(planes));
    EPN_CHECK_LAUNCH();
    const int nt = (d->nn + 15) / 16;
    if (bf16) {
        if (nt <= 1) return launch_fx_fwd_nt<__bf16, 1>(F, d, st);
        if (nt <= 2) return launch_fx_fwd_nt<__bf16, 2>(F, d, st);
        return launch_fx_fwd_nt<__bf16, 4>(F, d, st);
    }
    if (nt <= 1) return launch_fx_fwd_nt<float, 1>(F, d, st);
    if (nt <= 2) return launch_fx_fwd_nt<float, 2>(F, d, st);
    return launch_fx_fwd_nt<float, 4>(F, d, st);
}

}  // namespace epn
