// Weight contractions of BasicSO3Conv (vgtk/vgtk/so3conv/modules.py:48-55: `W @ feats.view(b, c*ks, p*a)`) and of its
// autograd transposes, as hand-written MFMA GEMMs for gfx950 -- round 1 handed these to the BLAS library.
//
//   NT   C[M][N]   = A[M][K] . Bt[N][K]^T      activation x weight: out = G W^T, dG = dOut (W^T)^T, the spectral blocks
//   TN   C[N1][N2] = X[R][N1]^T . Y[R][N2]     weight gradients: dW = dOut^T G (contraction over the R = b*p*a columns)
//
// Both stream 128-byte row segments through a double-buffered LDS ring with direct-to-LDS loads
// (global_load_lds_dwordx4, one 1 KiB wave instruction = 8 row segments), one barrier per K step.  The NT image is
// XOR-swizzled on the SOURCE address (16-byte slot ^ ((row >> 1) & 7)) so that the ds_read_b128 fragment reads are
// bank-conflict free (cdna_hip_programming.md T2 / rule 21); the TN image is read along its rows and needs none.
// fp32 uses v_mfma_f32_32x32x2_f32 (exact f32, 64 FLOP/clk/SIMD): one ds_read_b128 per operand tile feeds FOUR MFMAs
// because the contraction index may be visited in any order as long as both operands agree (lane group j of a step
// holds k = 8s + 4j .. +3).  bf16 uses v_mfma_f32_32x32x16_bf16 (NT, one b128 = one operand) and
// v_mfma_f32_16x16x32_bf16 fed by ds_read_b64_tr_b16 transposed reads (TN), fp32 accumulation throughout.
#include "conv_internal.h"
#include "gemm.h"

namespace epn {
namespace {
EPN_F2_SENTINEL_DECL

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <typename T> struct ElemOf;
template <> struct ElemOf<float> { static constexpr int PER16 = 4; };
template <> struct ElemOf<__bf16> { static constexpr int PER16 = 8; };

__device__ __forceinline__ void glds16(const void *g, char *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((glb_void *)g, (lds_void *)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ void glds16_nt(const void *g, char *lds_wave_base) {      // non-temporal: a stream read once
    __builtin_amdgcn_global_load_lds((glb_void *)g, (lds_void *)lds_wave_base, 16, 0, 2);
}

__device__ __forceinline__ void store_out(float *p, float v) { *p = v; }
__device__ __forceinline__ void store_out(__bf16 *p, float v) { *p = (__bf16)v; }

// ------------------------------------------------------------------------------------------------ NT
// Block tile BM x BN = (WGM*TM*32) x (WGN*TN*32), WGM*WGN waves, each wave TM x TN MFMA tiles of 32x32.
// K step = KS 16-byte slots of a row: KS = 8 (128 bytes: 32 floats / 64 bf16) or, for contraction lengths that are
// only a multiple of half that (the 32-channel layers in bf16), KS = 4.
template <typename T, typename TO, int WGM, int WGN, int TM, int TN, int KS = 8, int NSTG = 2>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_nt_kernel(GemmNtBatch B) {
    constexpr int NW = WGM * WGN;
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    constexpr int ROWS = BM + BN;                  // LDS rows per stage (A rows, then Bt rows)
    constexpr int ROWB = KS * 16;                  // bytes per LDS row
    constexpr int RPG = 64 / KS;                   // rows per wave-level load instruction (1 KiB)
    constexpr int NG = ROWS / RPG;                 // wave-level load instructions per stage
    constexpr int GPW = (NG + NW - 1) / NW;        // ... per wave
    constexpr int E16 = ElemOf<T>::PER16;          // elements per 16-byte slot
    constexpr int BKE = KS * E16;                  // elements per K step
    constexpr int NS = KS / 2;                     // fragment steps per K step (two slots each: lane groups j = 0, 1)
    __shared__ __attribute__((aligned(1024))) char smem[NSTG * ROWS * ROWB];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // ---- which problem / tile
    // Workgroup -> problem by launch order, then the XCD-aware remap WITHIN the problem (every problem's first
    // workgroup is a multiple of 8, so workgroup % 8 = XCD holds locally too): each XCD gets a contiguous range of
    // every problem's tiles.  One remap over the whole table gave XCD 0 all tiles of the longest-K problem and XCD 7 the
    // short ones -- a grouped launch ran 10 % slower than its problems launched one by one.
    int pi = 0;
#pragma unroll
    for (int i = 1; i < GEMM_MAX_PROB; ++i)
        if (i < B.nprob && blockIdx.x >= B.p[i].tile0) pi = i;
    const GemmNtProb &P = B.p[pi];
    if (blockIdx.x - P.tile0 >= P.ntile) return;
    const unsigned t = epn_xcd_tile(blockIdx.x - P.tile0, P.ntile);
    const long long m0 = (long long)(t / P.tiles_n) * BM;
    const int n0 = (int)(t % P.tiles_n) * BN;
    const T *__restrict__ A = static_cast<const T *>(P.A);
    const T *__restrict__ Bt = static_cast<const T *>(P.Bt);
    const int nk = P.K / BKE;

    // ---- staging pointers: group g covers LDS rows 8g .. 8g+7; lane -> (row 8g + lane/8, physical slot lane%8)
    const T *src[GPW];
#pragma unroll
    for (int i = 0; i < GPW; ++i) {
        const int g = wave + i * NW;
        const int r = RPG * g + lane / KS;                     // LDS row
        // logical 16-byte slot stored at physical slot lane % KS (conflict-free ds_read_b128 of 32 consecutive rows)
        const int slot = (lane % KS) ^ (KS == 8 ? (r >> 1) & 7 : (r >> 2) & 3);
        if (r < BM) {
            long long gr = m0 + r;
            gr = gr < P.M ? gr : P.M - 1;
            src[i] = A + gr * P.lda + slot * E16;
        } else {
            int gn = n0 + (r - BM);
            gn = gn < P.N ? gn : P.N - 1;
            src[i] = Bt + (long long)gn * P.ldb + slot * E16;
        }
    }
    auto stage_one = [&](int buf, int i) {
        const int g = wave + i * NW;
        if (NG % NW == 0 || g < NG) {
#ifdef EPN_NT_NTA
            if (RPG * g < BM) glds16_nt(src[i], smem + buf * (ROWS * ROWB) + g * 1024); else
#endif
            glds16(src[i], smem + buf * (ROWS * ROWB) + g * 1024);
            src[i] += BKE;
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < GPW; ++i) stage_one(buf, i);
    };
    // waves w and w + 4 of an 8-wave workgroup share a SIMD: their loads go to alternating MFMA groups (below)
    const int wpar = NW == 8 ? __builtin_amdgcn_readfirstlane(wave >> 2) : 0;

    const int wm = wave / WGN, wn = wave % WGN;
    const int li = lane & 31, lj = lane >> 5;
    const int fsw = KS == 8 ? (li >> 1) & 7 : (li >> 2) & 3;
    // byte offsets of this lane's fragment rows inside a stage
    int aoff[TM], boff[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) aoff[i] = ((wm * TM + i) * 32 + li) * ROWB;
#pragma unroll
    for (int i = 0; i < TN; ++i) boff[i] = (BM + (wn * TN + i) * 32 + li) * ROWB;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // One barrier per K step.  A direct-to-LDS load costs its wave ~60-180 cycles of issue during which it feeds no MFMA,
    // and the two waves of a SIMD leave the barrier together: issued as one burst (wherever in the step) both waves
    // stall at once and the matrix pipe idles ~7 % (measured, tools/gemm_lab).  So the next stage's loads are spread
    // one per MFMA group over the step, the partner waves on alternating groups (sched_barrier pins the placement; left
    // alone the compiler hoists all of them in front of the first ds_read).  fp32: +3 % over the burst; bf16 is
    // HBM-bound on these shapes and keeps the burst.
    constexpr int NGRP = 4 * NS;               // fp32: MFMA groups per K step (NS fragment steps x 4 contraction pairs)
    constexpr bool BURST = sizeof(T) != 4 || BN <= 64;   // HBM-bound shapes: the whole next stage is requested at once
    static_assert(NSTG == 2 || (NSTG == 3 && BURST && NG % NW == 0), "the 3-stage ring is for the burst (HBM-bound) configurations");
    stage(0);
    if constexpr (NSTG == 3) {
        if (nk > 1) stage(1);
    }
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if constexpr (NSTG == 3) {
            // Two stages in flight (HBM-bound operands: one stage of bytes per CU does not cover the loaded latency):
            // stage kt must have landed, stage kt+1 (this wave's GPW youngest loads) may still be on its way.
            if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        } else {
            __syncthreads();                   // stage kt has landed (vmcnt(0) rides on the barrier); buffer (kt+1)&1 is free
        }
        const char *base = smem + (kt % NSTG) * (ROWS * ROWB);
        if constexpr (BURST) {
            if constexpr (NSTG == 3) {
                if (kt + 2 < nk) stage((kt + 2) % 3);      // buffer of step kt-1: every wave is past it (barrier above)
            } else {
                if (more) stage((kt + 1) & 1);
            }
        }
        if constexpr (sizeof(T) == 4) {
            // fragments of step s+1 are read while the MFMAs of step s issue (two register sets, static indices)
            f32x4 a[2][TM], b[2][TN];
            auto rd = [&](int s, int set) {
                const int so = ((2 * s + lj) ^ fsw) * 16;
#pragma unroll
                for (int i = 0; i < TM; ++i) a[set][i] = *reinterpret_cast<const f32x4 *>(base + aoff[i] + so);
#pragma unroll
                for (int i = 0; i < TN; ++i) b[set][i] = *reinterpret_cast<const f32x4 *>(base + boff[i] + so);
            };
            rd(0, 0);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (s + 1 < NS) rd(s + 1, (s + 1) & 1);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (more && !BURST) {
                        const int grp = 4 * s + e;
#pragma unroll
                        for (int i = 0; i < GPW; ++i) {
                            const int g0 = 1 + (i * (NGRP - 2)) / GPW;             // group of load i, first wave of a SIMD
                            const int g1 = g0 + 1 < NGRP ? g0 + 1 : NGRP - 1;      // ... its partner, one group later
                            if ((g0 == grp && wpar == 0) || (g1 == grp && wpar != 0)) stage_one((kt + 1) & 1, i);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s & 1][i][e], b[s & 1][j][e], acc[i][j], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int so = ((2 * s + lj) ^ fsw) * 16;
                bf16x8 a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const bf16x8 *>(base + aoff[i] + so);
#pragma unroll
                for (int i = 0; i < TN; ++i) b[i] = *reinterpret_cast<const bf16x8 *>(base + boff[i] + so);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: D[row = (r&3) + 8 (r>>2) + 4 lj][col = li]
    TO *__restrict__ C = static_cast<TO *>(P.C);
    if (P.stats) nt_col_stats<TM, TN, TO>(acc, P.stats, P.M, P.N, m0 + wm * TM * 32, n0 + wn * TN * 32, li, lj);
    if (P.c_amax) nt_c_amax<TM, TN, TO>(acc, P.c_amax);
    if (m0 + BM <= P.M && n0 + BN <= P.N && (long long)BM * P.ldc < (1LL << 30)) {
        // interior tile: wave-uniform base + 32-bit lane offset; one scalar multiply and one vector add per output row.
        // (The checked form below costs ~15 VALU instructions per element -- 128 elements per lane: measured as a fixed
        // ~8 us per 256 x 256 tile, a third of the run time of the short-K data-gradient GEMMs.)
        TO *__restrict__ cw = C + (size_t)(m0 + wm * TM * 32) * P.ldc + (n0 + wn * TN * 32);
        const unsigned ldc = (unsigned)P.ldc;
        const unsigned lane_off = 4u * lj * ldc + li;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned o = lane_off + (unsigned)(i * 32 + (r & 3) + 8 * (r >> 2)) * ldc;
#pragma unroll
                for (int j = 0; j < TN; ++j) store_out(cw + o + j * 32, acc[i][j][r]);
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lj;
                if (m < P.M && n < P.N) store_out(C + m * P.ldc + n, acc[i][j][r]);
            }
        }
}

// ------------------------------------------------------------------------------------------------ TN (fp32)
// C[N1][N2] (+ split partials) = sum_r X[r][n1] Y[r][n2].  Block tile BN1 x BN2 = (WGM*TM*32) x (WGN*TN*32); stage =
// BR rows of both operands, row-contiguous in LDS.  MFMA tile tm of a wave covers rows n1 = base + TM*i + tm (i = MFMA
// row) so that ONE ds_read of TM floats at [r][base + TM*i] serves all TM tiles (likewise columns): output rows are a
// permutation the epilogue undoes.
struct TnGeom {
    long long r0, r1;     // row range of this split
};

// Split form (X3, see gemm_x3.hip): fp32 operands split without loss into three bf16 pieces in registers, six
// v_mfma_f32_32x32x16_bf16 per 32x32x16 block instead of eight v_mfma_f32_32x32x2_f32 -- fp32 accuracy at 2.7x the rate.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_rne(float a, float b) {   // v_cvt_pk_bf16_f32
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void split3(const float (&x)[8], bf16x8 &h, bf16x8 &m, bf16x8 &l) {
    u32x4 H, M, L;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const unsigned hp = pack_rne(x[2 * p], x[2 * p + 1]);
        const float r0 = x[2 * p] - __builtin_bit_cast(float, hp << 16);
        const float r1 = x[2 * p + 1] - __builtin_bit_cast(float, hp & 0xffff0000u);
        const unsigned mp = pack_rne(r0, r1);
        H[p] = hp; M[p] = mp;
        L[p] = pack_rne(r0 - __builtin_bit_cast(float, mp << 16), r1 - __builtin_bit_cast(float, mp & 0xffff0000u));
    }
    h = __builtin_bit_cast(bf16x8, H); m = __builtin_bit_cast(bf16x8, M); l = __builtin_bit_cast(bf16x8, L);
}

// X3: both operands split in registers (the narrow / grouped problems, where a separate splitting pass over X would cost
// as much as the GEMM); the wide weight gradients run on gemm_tn_x3_kernel below.
#ifndef EPN_TN_XCD
#define EPN_TN_XCD 1
#endif
#ifndef EPN_TN_NARROW           // exact narrow tiles + ring of stages for the bf16 1x1-convolution weight gradients
#define EPN_TN_NARROW 1
#endif
#ifndef EPN_TN_WIDE_RING        // 1: grouped 128 x 256 launches on the ring kernel; 2: single problems too
#define EPN_TN_WIDE_RING 1
#endif
#ifndef EPN_TN_GROUP_TARGET_BF16
#define EPN_TN_GROUP_TARGET_BF16 768
#endif
#ifndef EPN_TN_GROUP_TARGET_F32
#define EPN_TN_GROUP_TARGET_F32 1024
#endif
#ifndef EPN_TN_SINGLE_TARGET_BF16
#define EPN_TN_SINGLE_TARGET_BF16 512
#endif
// Image of the staged rows of the bf16 weight-gradient kernels (round 6).  A 16-lane group of `ds_read_b64_tr_b16` reads four
// consecutive rows of a 16-column tile (32 bytes each); the instruction is serviced in two 32-lane groups, i.e. eight rows
// ({r .. r+3} of two lane groups 8 rows apart) at once, and the bank of a byte is (a / 4) mod 64.  In the linear image of
// the 128 x 256 tile a row is 768 bytes = 3 x 256: all eight rows start on bank 0 and the 64 chunks of a wave queue on 8
// banks -- SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.75-0.88 in rounds 3-5's PMC files (review, round 5).  1: the 16-byte
// slots of row r are XOR-ed with 2 rho(r), rho = (r & 3) | ((r >> 3) & 1) << 2 -- applied to the SOURCE address of the
// direct-to-LDS loads (their LDS side is lane-linear), so the eight rows of a group land on eight different 32-byte windows
// of the 256-byte bank row.  Needs operand regions that are multiples of 16 slots (128 columns): the 128 x 256 tiles;
// narrower tiles keep the linear image (tools/lds_tr_probe.hip measures the candidates).
#ifndef EPN_TN_SWZ
#define EPN_TN_SWZ 1
#endif
__device__ __forceinline__ int tn_row_swz(int r) { return 2 * ((r & 3) | (((r >> 3) & 1) << 2)); }
// column (bf16 index inside the staged row) where the swizzled image keeps logical column `col` of row r; rows r and r + 4
// share the mask (bit 2 of r is not used), so the second read of a fragment keeps its fixed row offset
template <bool SWZ>
__device__ __forceinline__ int tn_swz_col(int col, int r) {
    if constexpr (!SWZ) return col;
    return (((col >> 3) ^ tn_row_swz(r)) << 3) | (col & 7);
}
#ifndef EPN_TN_WIDE_NSTG
#define EPN_TN_WIDE_NSTG 3
#endif
#ifndef EPN_TN_REDUCE_SP        // shared-quad reduction of the partial slabs for small outputs
#define EPN_TN_REDUCE_SP 1
#endif
#ifndef EPN_TN_NARROW_NSTG
#define EPN_TN_NARROW_NSTG 4
#endif
#ifndef EPN_TN_NARROW_STAGE_KB  // stage size aimed at (a stage = a power-of-two number of 32-row contraction steps)
#define EPN_TN_NARROW_STAGE_KB 32
#endif
constexpr int tn_ring_kr(int wgm, int wgn, int tm, int tn) {
    const int step_b = 32 * 2 * 16 * (wgm * tm + wgn * tn);      // bytes of one 32-row step of [BN1 + BN2] bf16
    int kr = 1;
    while (2 * kr * step_b <= EPN_TN_NARROW_STAGE_KB * 1024 && 2 * kr * step_b * EPN_TN_NARROW_NSTG <= 160 * 1024) kr *= 2;
    return kr;
}
#ifndef EPN_TN_NARROW_WGS       // resident workgroups per CU the split count aims at
#define EPN_TN_NARROW_WGS 1
#endif
// Workgroup -> (tile, split) of a TN problem.  Launch order is split-major (all tiles of one K range, then the next
// range); every tile row streams the whole Y panel and every tile column the whole X panel, so the tiles of ONE K range
// running on ONE XCD read each operand row once from HBM and again from that XCD's L2.  Workgroup b runs on XCD b % 8:
// the bijective remap of epn_common.h hands every XCD a contiguous run of that order (a problem's first workgroup is a
// multiple of 8, tn_plan).  Placement only: the partial slabs and their fixed-order reduction are unchanged.
__device__ __forceinline__ unsigned tn_tile_of(unsigned lb, unsigned ntiles, unsigned nsplit) {
#if EPN_TN_XCD
    return epn_xcd_tile(lb, (ntiles * nsplit + 7u) & ~7u);
#else
    return lb;
#endif
}

template <int WGM, int WGN, int TM, int TN, int BR, int X3 = 0>     // X3: 0 = fp32 MFMA, 3 = three bf16 pieces, 2 = two fp16 pieces
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_tn_f32_kernel(GemmTnBatch B) {
    constexpr int NW = WGM * WGN;
    constexpr int BN1 = WGM * TM * 32, BN2 = WGN * TN * 32;
    constexpr int ROWF = BN1 + BN2;                 // floats per staged row (X part, then Y part)
    constexpr int STAGE_B = BR * ROWF * 4;
    constexpr int NI = STAGE_B / 1024;              // wave-level 1 KiB load instructions per stage
    constexpr int IPW = (NI + NW - 1) / NW;
    static_assert(STAGE_B % 1024 == 0, "stage must be a whole number of 1 KiB pieces");
    __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE_B];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int pi = 0;
#pragma unroll
    for (int i = 1; i < GEMM_MAX_PROB; ++i)
        if (i < B.nprob && blockIdx.x >= B.p[i].block0) pi = i;
    const GemmTnArgs &G = B.p[pi];
    const unsigned tile = tn_tile_of(blockIdx.x - G.block0, G.ntiles, (unsigned)G.nsplit) % G.ntiles;
    const unsigned split = tn_tile_of(blockIdx.x - G.block0, G.ntiles, (unsigned)G.nsplit) / G.ntiles;
    if (split >= (unsigned)G.nsplit) return;           // padding workgroup of a grouped launch (uniform per workgroup)
    const int n1_0 = (int)(tile / G.tiles_n2) * BN1, n2_0 = (int)(tile % G.tiles_n2) * BN2;
    const long long nchunk = G.R / BR;
    const long long c0 = nchunk * split / G.nsplit, c1 = nchunk * (split + 1) / G.nsplit;
    const int nk = (int)(c1 - c0);
    const float *__restrict__ X = static_cast<const float *>(G.X);
    const float *__restrict__ Y = static_cast<const float *>(G.Y);

    // piece q (1 KiB = 256 floats) of a stage: float offset 256 q + 4 lane -> (row, col) of the [BR][ROWF] image
    const float *src[IPW];
    long long sstep[IPW];
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
        const int q = wave + i * NW;
        const int fo = 256 * q + 4 * lane;
        const int r = fo / ROWF, c = fo % ROWF;
        if (c < BN1) {
            int n = n1_0 + c;
            n = n < G.N1 - 4 ? n : G.N1 - 4;
            src[i] = X + (c0 * BR + r) * G.ldx + n;
            sstep[i] = (long long)BR * G.ldx;
        } else {
            int n = n2_0 + (c - BN1);
            n = n < G.N2 - 4 ? n : G.N2 - 4;
            src[i] = Y + (c0 * BR + r) * G.ldy + n;
            sstep[i] = (long long)BR * G.ldy;
        }
    }
    auto stage_one = [&](int buf, int i) {
        const int q = wave + i * NW;
        if (NI % NW == 0 || q < NI) {
            glds16(src[i], smem + buf * STAGE_B + q * 1024);
            src[i] += sstep[i];
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < IPW; ++i) stage_one(buf, i);
    };
    const int wpar = NW == 8 ? __builtin_amdgcn_readfirstlane(wave >> 2) : 0;   // SIMD partner parity (gemm_nt_kernel)

    const int wm = wave / WGN, wn = wave % WGN;
    const int li = lane & 31, lj = lane >> 5;
    const int xo = (wm * TM * 32 + TM * li) * 4;                 // byte offset inside a staged row
    const int yo = (BN1 + wn * TN * 32 + TN * li) * 4;
    float x_scale = 1.0f, y_scale = 1.0f;                        // two-piece form: powers of two from the device maxima
    if constexpr (X3 == 2) { x_scale = f2_scale_of(*G.x_amax); y_scale = f2_scale_of(*G.y_amax); }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (nk > 0) stage(0);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();
        const char *base = smem + (kt & 1) * STAGE_B;
        const bool more = kt + 1 < nk;
        if constexpr (X3 != 0) {
            if (more) stage((kt + 1) & 1);
#pragma unroll
            for (int s = 0; s < BR / 16; ++s) {     // 16 rows per fragment step: lane group lj holds rows 8 lj .. 8 lj + 7
                float xa[TM][8], yb[TN][8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const char *row = base + (16 * s + 8 * lj + k) * (ROWF * 4);
                    if constexpr (TM == 4) {
                        const f32x4 v = *reinterpret_cast<const f32x4 *>(row + xo);
                        xa[0][k] = v[0]; xa[1][k] = v[1]; xa[2][k] = v[2]; xa[3][k] = v[3];
                    } else if constexpr (TM == 2) {
                        const f32x2 v = *reinterpret_cast<const f32x2 *>(row + xo);
                        xa[0][k] = v[0]; xa[1][k] = v[1];
                    } else {
                        xa[0][k] = *reinterpret_cast<const float *>(row + xo);
                    }
                    if constexpr (TN == 4) {
                        const f32x4 v = *reinterpret_cast<const f32x4 *>(row + yo);
                        yb[0][k] = v[0]; yb[1][k] = v[1]; yb[2][k] = v[2]; yb[3][k] = v[3];
                    } else if constexpr (TN == 2) {
                        const f32x2 v = *reinterpret_cast<const f32x2 *>(row + yo);
                        yb[0][k] = v[0]; yb[1][k] = v[1];
                    } else {
                        yb[0][k] = *reinterpret_cast<const float *>(row + yo);
                    }
                }
                if constexpr (X3 == 3) {
                bf16x8 ah[TM], am[TM], al[TM], bh[TN], bm[TN], bl[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) split3(xa[i], ah[i], am[i], al[i]);
#pragma unroll
                for (int j = 0; j < TN; ++j) split3(yb[j], bh[j], bm[j], bl[j]);
#define EPN_X3_TERM(PA, PB)                                                                                 \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)         \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PA[i], PB[j], acc[i][j], 0, 0, 0)
                EPN_X3_TERM(ah, bl);                // small terms first
                EPN_X3_TERM(al, bh);
                EPN_X3_TERM(am, bm);
                EPN_X3_TERM(ah, bm);
                EPN_X3_TERM(am, bh);
                EPN_X3_TERM(ah, bh);
#undef EPN_X3_TERM
                } else {
                gemm_f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) f2_split8(xa[i], x_scale, ah[i], al[i]);
#pragma unroll
                for (int j = 0; j < TN; ++j) f2_split8(yb[j], y_scale, bh[j], bl[j]);
#define EPN_F2_TERM(PA, PB)                                                                                 \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)         \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(PA[i], PB[j], acc[i][j], 0, 0, 0)
                EPN_F2_TERM(ah, bl);
                EPN_F2_TERM(al, bh);
                EPN_F2_TERM(ah, bh);
#undef EPN_F2_TERM
                }
            }
            continue;
        }
        // fragments of k-pair s+1 are read while the MFMAs of pair s issue (two register sets, static indices)
        float a[2][TM], b[2][TN];
        auto rd = [&](int s, int set) {
            const char *row = base + (2 * s + lj) * (ROWF * 4);
            if constexpr (TM == 4) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(row + xo);
                a[set][0] = v[0]; a[set][1] = v[1]; a[set][2] = v[2]; a[set][3] = v[3];
            } else if constexpr (TM == 2) {
                const f32x2 v = *reinterpret_cast<const f32x2 *>(row + xo);
                a[set][0] = v[0]; a[set][1] = v[1];
            } else {
                a[set][0] = *reinterpret_cast<const float *>(row + xo);
            }
            if constexpr (TN == 4) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(row + yo);
                b[set][0] = v[0]; b[set][1] = v[1]; b[set][2] = v[2]; b[set][3] = v[3];
            } else if constexpr (TN == 2) {
                const f32x2 v = *reinterpret_cast<const f32x2 *>(row + yo);
                b[set][0] = v[0]; b[set][1] = v[1];
            } else {
                b[set][0] = *reinterpret_cast<const float *>(row + yo);
            }
        };
        rd(0, 0);
#pragma unroll
        for (int s = 0; s < BR / 2; ++s) {
            if (s + 1 < BR / 2) rd(s + 1, (s + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            if (more) {                         // next stage's loads spread over the step (see gemm_nt_kernel)
                constexpr int NGRP = BR / 2;
#pragma unroll
                for (int i = 0; i < IPW; ++i) {
                    const int g0 = (i * (NGRP - 1)) / IPW;
                    const int g1 = g0 + 1 < NGRP ? g0 + 1 : NGRP - 1;
                    if ((g0 == s && wpar == 0) || (g1 == s && wpar != 0)) stage_one((kt + 1) & 1, i);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s & 1][i], b[s & 1][j], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue: MFMA row ri of tile tm = output row base1 + TM*ri + tm; column li of tile tn = base2 + TN*li + tn
    if constexpr (X3 == 2) {
        const float ux = f2_inverse(x_scale), uy = f2_inverse(y_scale);
        float chk = 0.0f;                           // NaN iff an accumulator of this lane is inf / NaN (gemm.h: EPN_F2_CHECK)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    chk = fmaf(acc[i][j][r], 0.0f, chk);
                    acc[i][j][r] = acc[i][j][r] * ux * uy;
                }
        EPN_F2_CHECK(chk);
    }
    float *__restrict__ C = G.nsplit > 1 ? static_cast<float *>(G.part) + (size_t)split * G.N1 * G.N2
                                         : static_cast<float *>(G.C);
    const long long ldc = G.nsplit > 1 ? G.N2 : G.ldc;
    if (n1_0 + BN1 <= G.N1 && n2_0 + BN2 <= G.N2 && (long long)BN1 * ldc < (1LL << 30)) {
        // interior tile: wave-uniform base + 32-bit lane offset (see gemm_nt_kernel's epilogue)
        float *__restrict__ cw = C + (size_t)(n1_0 + wm * TM * 32) * ldc + (n2_0 + wn * TN * 32);
        const unsigned ld = (unsigned)ldc;
        const unsigned lane_off = (unsigned)(TM * 4 * lj) * ld + TN * li;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned o = lane_off + (unsigned)(TM * ((r & 3) + 8 * (r >> 2)) + i) * ld;
#pragma unroll
                for (int j = 0; j < TN; ++j) cw[o + j] = acc[i][j][r];
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ri = (r & 3) + 8 * (r >> 2) + 4 * lj;
            const int n1 = n1_0 + wm * TM * 32 + TM * ri + i;
            const int n2 = n2_0 + wn * TN * 32 + TN * li;
            if (n1 < G.N1) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    if (n2 + j < G.N2) C[(long long)n1 * ldc + n2 + j] = acc[i][j][r];
            }
        }
}

// ------------------------------------------------------------------------------------------------ TN (fp32, split form)
// Split form with the NARROW operand X (the output gradient: N1 = cout columns, a few % of the bytes of Y) split ahead of
// the GEMM by split_octets_kernel into three bf16 planes laid out [plane][R/8][N1][8]: the eight contraction values a
// lane needs for one output row are one 16-byte chunk, so an X fragment is ONE ds_read_b128 per plane and costs no VALU
// work; only Y (the grouped features, streamed once) is split in registers -- 36 VALU instructions per Y fragment,
// 1.5 per MFMA for a 128 x 64 wave tile instead of 4.5-6 when both operands are split in the kernel (the VALU/issue
// slots beside a 32-cycle MFMA are what bounded that form at 150 TFLOP/s).  MFMA tile i of a wave covers output rows
// base + 32 i + (MFMA row) here (contiguous chunks: conflict-free b128 reads), columns as in gemm_tn_f32_kernel.
__global__ void split_octets_kernel(const float *__restrict__ X, long long ldx, long long R, int N1, u32x4 *__restrict__ planes) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long n = (R >> 3) * N1;
    if (i >= n) return;
    const long long o = i / N1;
    const int c = (int)(i - o * N1);
    float x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = X[(8 * o + k) * ldx + c];
    bf16x8 h, m, l;
    split3(x, h, m, l);
    planes[i] = __builtin_bit_cast(u32x4, h);
    planes[n + i] = __builtin_bit_cast(u32x4, m);
    planes[2 * n + i] = __builtin_bit_cast(u32x4, l);
}

// two-piece fp16 form: planes [2][R/8][N1][8], scaled by the power of two of max|X| (device scalar)
__global__ void split_octets2_kernel(const float *__restrict__ X, long long ldx, long long R, int N1, u32x4 *__restrict__ planes,
                                     const float *__restrict__ amax) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long n = (R >> 3) * N1;
    if (i >= n) return;
    const long long o = i / N1;
    const int c = (int)(i - o * N1);
    float x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = X[(8 * o + k) * ldx + c];
    gemm_f16x8 h, l;
    f2_split8(x, f2_scale_of(*amax), h, l);
    planes[i] = __builtin_bit_cast(u32x4, h);
    planes[n + i] = __builtin_bit_cast(u32x4, l);
}

template <int WGM, int WGN, int TM, int TN, int BR, int NPL = 3>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_tn_x3_kernel(GemmTnBatch B) {
    constexpr int NW = WGM * WGN;
    constexpr int BN1 = WGM * TM * 32, BN2 = WGN * TN * 32;
    constexpr int OCT = BR / 8;                     // row octets per stage
    constexpr int XB = NPL * OCT * BN1 * 16;        // X planes of a stage: [plane][octet][n1] 16-byte chunks
    constexpr int YB = BR * BN2 * 4;                // Y rows (fp32)
    constexpr int STAGE_B = XB + YB;
    constexpr int NIX = XB / 1024, NI = STAGE_B / 1024;
    constexpr int IPW = (NI + NW - 1) / NW;
    static_assert(XB % 1024 == 0 && YB % 1024 == 0 && 2 * STAGE_B <= 160 * 1024, "stage");
    __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE_B];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int pi = 0;
#pragma unroll
    for (int i = 1; i < GEMM_MAX_PROB; ++i)
        if (i < B.nprob && blockIdx.x >= B.p[i].block0) pi = i;
    const GemmTnArgs &G = B.p[pi];
    const unsigned tile = tn_tile_of(blockIdx.x - G.block0, G.ntiles, (unsigned)G.nsplit) % G.ntiles;
    const unsigned split = tn_tile_of(blockIdx.x - G.block0, G.ntiles, (unsigned)G.nsplit) / G.ntiles;
    if (split >= (unsigned)G.nsplit) return;           // padding workgroup of a grouped launch (uniform per workgroup)
    const int n1_0 = (int)(tile / G.tiles_n2) * BN1, n2_0 = (int)(tile % G.tiles_n2) * BN2;
    const long long nchunk = G.R / BR;
    const long long c0 = nchunk * split / G.nsplit, c1 = nchunk * (split + 1) / G.nsplit;
    const int nk = (int)(c1 - c0);

    const char *src[IPW];
    long long sstep[IPW];
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
        const int q = wave + i * NW;
        if (q < NIX) {
            const int ci = 64 * q + lane;            // chunk of the [plane][octet][n1] image
            const int c = ci % BN1, po = ci / BN1;
            const int oct = po % OCT, pl = po / OCT;
            int n = n1_0 + c;
            n = n < G.N1 ? n : G.N1 - 1;
            src[i] = static_cast<const char *>(G.Xp) + (((size_t)pl * (G.R >> 3) + (c0 * OCT + oct)) * G.N1 + n) * 16;
            sstep[i] = (long long)OCT * G.N1 * 16;
        } else {
            const int fo = 256 * (q - NIX) + 4 * lane;
            const int r = fo / BN2, c = fo % BN2;
            int n = n2_0 + c;
            n = n < G.N2 - 4 ? n : G.N2 - 4;
            src[i] = reinterpret_cast<const char *>(static_cast<const float *>(G.Y) + (c0 * BR + r) * G.ldy + n);
            sstep[i] = (long long)BR * G.ldy * 4;
        }
    }
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const int q = wave + i * NW;
            if (NI % NW == 0 || q < NI) {
#ifdef EPN_TN_NTY
                if (q >= NIX) glds16_nt(src[i], smem + buf * STAGE_B + q * 1024); else
#endif
                glds16(src[i], smem + buf * STAGE_B + q * 1024);
                src[i] += sstep[i];
            }
        }
    };

    const int wm = wave / WGN, wn = wave % WGN;
    const int li = lane & 31, lj = lane >> 5;
    const int xo = (wm * TM * 32 + li) * 16;                      // chunk of MFMA tile 0 inside an [octet] row of a plane
    const int yo = XB + (wn * TN * 32 + TN * li) * 4;
    float y_scale = 1.0f;
    if constexpr (NPL == 2) y_scale = f2_scale_of(*G.y_amax);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (nk > 0) stage(0);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();
        const char *base = smem + (kt & 1) * STAGE_B;
        if (kt + 1 < nk) stage((kt + 1) & 1);
#pragma unroll
        for (int s = 0; s < BR / 16; ++s) {         // 16 rows per fragment step: lane group lj holds octet 2 s + lj
            float yb[TN][8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const char *row = base + (16 * s + 8 * lj + k) * (BN2 * 4);
                if constexpr (TN == 4) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(row + yo);
                    yb[0][k] = v[0]; yb[1][k] = v[1]; yb[2][k] = v[2]; yb[3][k] = v[3];
                } else if constexpr (TN == 2) {
                    const f32x2 v = *reinterpret_cast<const f32x2 *>(row + yo);
                    yb[0][k] = v[0]; yb[1][k] = v[1];
                } else {
                    yb[0][k] = *reinterpret_cast<const float *>(row + yo);
                }
            }
            if constexpr (NPL == 3) {
            bf16x8 ah[TM], am[TM], al[TM], bh[TN], bm[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const char *xp = base + (2 * s + lj) * (BN1 * 16) + xo + i * 512;
                ah[i] = *reinterpret_cast<const bf16x8 *>(xp);
                am[i] = *reinterpret_cast<const bf16x8 *>(xp + OCT * BN1 * 16);
                al[i] = *reinterpret_cast<const bf16x8 *>(xp + 2 * OCT * BN1 * 16);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) split3(yb[j], bh[j], bm[j], bl[j]);
#define EPN_X3_TERM(PA, PB)                                                                                 \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)         \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PA[i], PB[j], acc[i][j], 0, 0, 0)
            EPN_X3_TERM(ah, bl);                    // small terms first
            EPN_X3_TERM(al, bh);
            EPN_X3_TERM(am, bm);
            EPN_X3_TERM(ah, bm);
            EPN_X3_TERM(am, bh);
            EPN_X3_TERM(ah, bh);
#undef EPN_X3_TERM
            } else {
            gemm_f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const char *xp = base + (2 * s + lj) * (BN1 * 16) + xo + i * 512;
                ah[i] = *reinterpret_cast<const gemm_f16x8 *>(xp);
                al[i] = *reinterpret_cast<const gemm_f16x8 *>(xp + OCT * BN1 * 16);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) f2_split8(yb[j], y_scale, bh[j], bl[j]);
#define EPN_F2_TERM(PA, PB)                                                                                 \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)         \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(PA[i], PB[j], acc[i][j], 0, 0, 0)
            EPN_F2_TERM(ah, bl);
            EPN_F2_TERM(al, bh);
            EPN_F2_TERM(ah, bh);
#undef EPN_F2_TERM
            }
        }
    }

    if constexpr (NPL == 2) {
        const float ux = f2_inverse(f2_scale_of(*G.x_amax)), uy = f2_inverse(y_scale);
        float chk = 0.0f;                           // NaN iff an accumulator of this lane is inf / NaN (gemm.h: EPN_F2_CHECK)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    chk = fmaf(acc[i][j][r], 0.0f, chk);
                    acc[i][j][r] = acc[i][j][r] * ux * uy;
                }
        EPN_F2_CHECK(chk);
    }
    // ---- epilogue: MFMA row ri of tile i = output row base1 + 32 i + ri; column li of tile j = base2 + TN*li + j
    float *__restrict__ C = G.nsplit > 1 ? static_cast<float *>(G.part) + (size_t)split * G.N1 * G.N2
                                         : static_cast<float *>(G.C);
    const long long ldc = G.nsplit > 1 ? G.N2 : G.ldc;
    if (n1_0 + BN1 <= G.N1 && n2_0 + BN2 <= G.N2 && (long long)BN1 * ldc < (1LL << 30)) {
        float *__restrict__ cw = C + (size_t)(n1_0 + wm * TM * 32) * ldc + (n2_0 + wn * TN * 32);
        const unsigned ld = (unsigned)ldc;
        const unsigned lane_off = (unsigned)(4 * lj) * ld + TN * li;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned o = lane_off + (unsigned)(32 * i + (r & 3) + 8 * (r >> 2)) * ld;
#pragma unroll
                for (int j = 0; j < TN; ++j) cw[o + j] = acc[i][j][r];
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n1 = n1_0 + wm * TM * 32 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lj;
            const int n2 = n2_0 + wn * TN * 32 + TN * li;
            if (n1 < G.N1) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    if (n2 + j < G.N2) C[(long long)n1 * ldc + n2 + j] = acc[i][j][r];
            }
        }
}

// ------------------------------------------------------------------------------------------------ TN (bf16)
// Same problem with bf16 operands, fp32 result: v_mfma_f32_16x16x32_bf16, both fragments by ds_read_b64_tr_b16 (a
// 16-lane group reads a [4 rows][16 columns] block and receives it column-per-lane: lane i gets rows 0..3 of column i).
// Wave tile = (TM*16) x (TN*16); stage = 32 rows (= one MFMA contraction step) of [BN1 + BN2] bf16.
template <int WGM, int WGN, int TM, int TN>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_tn_bf16_kernel(GemmTnBatch B) {
    constexpr int NW = WGM * WGN;
    constexpr int BR = 32;
    constexpr int BN1 = WGM * TM * 16, BN2 = WGN * TN * 16;
    constexpr int ROWE = BN1 + BN2;                 // bf16 per staged row
    constexpr int STAGE_B = BR * ROWE * 2;
    constexpr int NI = STAGE_B / 1024;
    constexpr int IPW = (NI + NW - 1) / NW;
    static_assert(STAGE_B % 1024 == 0, "stage must be a whole number of 1 KiB pieces");
    __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE_B];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int pi = 0;
#pragma unroll
    for (int i = 1; i < GEMM_MAX_PROB; ++i)
        if (i < B.nprob && blockIdx.x >= B.p[i].block0) pi = i;
    const GemmTnArgs &G = B.p[pi];
    const unsigned tile = tn_tile_of(blockIdx.x - G.block0, G.ntiles, (unsigned)G.nsplit) % G.ntiles;
    const unsigned split = tn_tile_of(blockIdx.x - G.block0, G.ntiles, (unsigned)G.nsplit) / G.ntiles;
    if (split >= (unsigned)G.nsplit) return;           // padding workgroup of a grouped launch (uniform per workgroup)
    const int n1_0 = (int)(tile / G.tiles_n2) * BN1, n2_0 = (int)(tile % G.tiles_n2) * BN2;
    const long long nchunk = G.R / BR;
    const long long c0 = nchunk * split / G.nsplit, c1 = nchunk * (split + 1) / G.nsplit;
    const int nk = (int)(c1 - c0);
    const __bf16 *__restrict__ X = static_cast<const __bf16 *>(G.X);
    const __bf16 *__restrict__ Y = static_cast<const __bf16 *>(G.Y);

    constexpr bool SWZ = EPN_TN_SWZ && BN1 % 128 == 0 && BN2 % 128 == 0;   // (see EPN_TN_SWZ)
    const __bf16 *src[IPW];
    long long sstep[IPW];
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
        const int q = wave + i * NW;
        const int eo = 512 * q + 8 * lane;            // bf16 offset inside the [BR][ROWE] image
        const int r = eo / ROWE;
        int c = eo % ROWE;
        if constexpr (SWZ) c = 8 * ((c >> 3) ^ tn_row_swz(r));      // this LDS slot holds the row's slot (s ^ swz): an involution
        if (c < BN1) {
            int n = n1_0 + c;
            n = n < G.N1 - 8 ? n : G.N1 - 8;
            src[i] = X + (c0 * BR + r) * G.ldx + n;
            sstep[i] = (long long)BR * G.ldx;
        } else {
            int n = n2_0 + (c - BN1);
            n = n < G.N2 - 8 ? n : G.N2 - 8;
            src[i] = Y + (c0 * BR + r) * G.ldy + n;
            sstep[i] = (long long)BR * G.ldy;
        }
    }
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const int q = wave + i * NW;
            if (NI % NW == 0 || q < NI) {
                glds16(src[i], smem + buf * STAGE_B + q * 1024);
                src[i] += sstep[i];
            }
        }
    };

    const int wm = wave / WGN, wn = wave % WGN;
    const int li = lane & 15, lg = lane >> 4;
    // transposed read: lane i of a 16-lane group supplies the address of chunk i of the [4][16] block:
    // row i/4, columns 4 (i%4) .. +3; the group g handles contraction rows 8g .. 8g+7 (two reads of 4 rows)
    const int tr_row = 8 * lg + (li >> 2), tr_col = 4 * (li & 3);

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (nk > 0) stage(0);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();
        const char *base = smem + (kt & 1) * STAGE_B;
        bf16x8 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int col = tn_swz_col<SWZ>((wm * TM + i) * 16 + tr_col, tr_row);
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4 *)(base + ((tr_row)*ROWE + col) * 2));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4 *)(base + ((tr_row + 4) * ROWE + col) * 2));
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            a[i] = __builtin_bit_cast(bf16x8, v);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = tn_swz_col<SWZ>(BN1 + (wn * TN + j) * 16 + tr_col, tr_row);
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4 *)(base + ((tr_row)*ROWE + col) * 2));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4 *)(base + ((tr_row + 4) * ROWE + col) * 2));
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            b[j] = __builtin_bit_cast(bf16x8, v);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
                if (i == 0 && j == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (kt + 1 < nk) stage((kt + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
    }

    // D[row = 4 lg + r][col = li]
    float *__restrict__ C = G.nsplit > 1 ? static_cast<float *>(G.part) + (size_t)split * G.N1 * G.N2
                                         : static_cast<float *>(G.C);
    const long long ldc = G.nsplit > 1 ? G.N2 : G.ldc;
    if (n1_0 + BN1 <= G.N1 && n2_0 + BN2 <= G.N2 && (long long)BN1 * ldc < (1LL << 30)) {
        float *__restrict__ cw = C + (size_t)(n1_0 + wm * TM * 16) * ldc + (n2_0 + wn * TN * 16);
        const unsigned ld = (unsigned)ldc;
        const unsigned lane_off = (unsigned)(4 * lg) * ld + li;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned o = lane_off + (unsigned)(i * 16 + r) * ld;
#pragma unroll
                for (int j = 0; j < TN; ++j) cw[o + j * 16] = acc[i][j][r];
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n1 = n1_0 + (wm * TM + i) * 16 + 4 * lg + r;
                const int n2 = n2_0 + (wn * TN + j) * 16 + li;
                if (n1 < G.N1 && n2 < G.N2) C[(long long)n1 * ldc + n2] = acc[i][j][r];
            }
}

// ------------------------------------------------------------------------------------------------ TN (bf16), ring form
// The streaming weight gradients (1x1 convolutions: N1, N2 <= 256 outputs over 10^5..10^6 rows) -- same fragments and
// MFMA as gemm_tn_bf16_kernel, but (a) the tile IS the output (no columns of loads wasted on a 256-wide tile), (b) a
// stage holds KR contraction steps and NSTG stages form a ring with NSTG - 1 requested ahead: the wait in front of the
// barrier counts the younger stages' loads instead of draining them, (c) the transposed LDS reads are inline assembly --
// behind the builtin the compiler puts `s_waitcnt vmcnt(0)` in front of the first read of every step (it cannot tell the
// stage being read from the stages the LDS-direct loads are still writing), which empties the ring -- and are waited for
// by hand (the empty asm statements tie each fragment register to that wait).  Few, long workgroups: measured cold
// (tools/tn_probe.py), 256 workgroups stream faster than 512 or 1024 (fewer concurrent DRAM streams).
template <int WGM, int WGN, int TM, int TN, int NSTG, int KR>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_tn_bf16_ring_kernel(GemmTnBatch B) {
    constexpr int NW = WGM * WGN;
    constexpr int BN1 = WGM * TM * 16, BN2 = WGN * TN * 16;
    constexpr int ROWE = BN1 + BN2;                 // bf16 per staged row
    constexpr int BR = 32 * KR;                     // rows per stage
    constexpr int STAGE_B = BR * ROWE * 2;
    constexpr int NI = STAGE_B / 1024;
    constexpr int IPW = (NI + NW - 1) / NW;
    static_assert(STAGE_B % 1024 == 0, "stage must be a whole number of 1 KiB pieces");
    static_assert(NSTG >= 3 && NSTG <= 4 && NSTG * STAGE_B <= 160 * 1024, "ring of 3 or 4 stages in LDS");
    static_assert(IPW * (NSTG - 2) <= 63, "vmcnt range");
    __shared__ __attribute__((aligned(1024))) char smem[NSTG * STAGE_B];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int pi = 0;
#pragma unroll
    for (int i = 1; i < GEMM_MAX_PROB; ++i)
        if (i < B.nprob && blockIdx.x >= B.p[i].block0) pi = i;
    const GemmTnArgs &G = B.p[pi];
    const unsigned tile = tn_tile_of(blockIdx.x - G.block0, G.ntiles, (unsigned)G.nsplit) % G.ntiles;
    const unsigned split = tn_tile_of(blockIdx.x - G.block0, G.ntiles, (unsigned)G.nsplit) / G.ntiles;
    if (split >= (unsigned)G.nsplit) return;           // padding workgroup (uniform per workgroup)
    const int n1_0 = (int)(tile / G.tiles_n2) * BN1, n2_0 = (int)(tile % G.tiles_n2) * BN2;
    const long long nchunk = G.R / 32;              // 32-row contraction steps
    const long long c0 = nchunk * split / G.nsplit, c1 = nchunk * (split + 1) / G.nsplit;
    const int nk = (int)(c1 - c0);                  // steps of this split
    const int nst = (nk + KR - 1) / KR;             // stages (the last one may be partly used)
    const __bf16 *__restrict__ X = static_cast<const __bf16 *>(G.X);
    const __bf16 *__restrict__ Y = static_cast<const __bf16 *>(G.Y);

    // every wave issues IPW loads per stage so that one vmcnt value holds for all (a wave past the last piece requests
    // piece q - NI again: same bytes to the same place); rows past the end of the operands (last stage of the last
    // split) are clamped to the last row -- loaded, never multiplied
    constexpr bool SWZ = EPN_TN_SWZ && BN1 % 128 == 0 && BN2 % 128 == 0;   // (see EPN_TN_SWZ)
    const __bf16 *src[IPW], *lim[IPW];
    long long sstep[IPW];
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
        const int q = (wave + i * NW) % NI;
        const int eo = 512 * q + 8 * lane;            // bf16 offset inside the [BR][ROWE] image
        const int r = eo / ROWE;
        int c = eo % ROWE;
        if constexpr (SWZ) c = 8 * ((c >> 3) ^ tn_row_swz(r));
        if (c < BN1) {
            int n = n1_0 + c;
            n = n < G.N1 - 8 ? n : G.N1 - 8;
            src[i] = X + (c0 * 32 + r) * G.ldx + n;
            lim[i] = X + (G.R - 1) * G.ldx + n;
            sstep[i] = (long long)BR * G.ldx;
        } else {
            int n = n2_0 + (c - BN1);
            n = n < G.N2 - 8 ? n : G.N2 - 8;
            src[i] = Y + (c0 * 32 + r) * G.ldy + n;
            lim[i] = Y + (G.R - 1) * G.ldy + n;
            sstep[i] = (long long)BR * G.ldy;
        }
    }
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const int q = (wave + i * NW) % NI;
            glds16(KR > 1 && src[i] > lim[i] ? lim[i] : src[i], smem + buf * STAGE_B + q * 1024);
            src[i] += sstep[i];
        }
    };

    const int wm = wave / WGN, wn = wave % WGN;
    const int li = lane & 15, lg = lane >> 4;
    const int tr_row = 8 * lg + (li >> 2), tr_col = 4 * (li & 3);      // see gemm_tn_bf16_kernel

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int s = 0; s < NSTG - 1; ++s)
        if (s < nst) stage(s);
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    for (int kt = 0; kt < nst; ++kt) {
        // stage kt has landed; up to NSTG - 2 younger stages (IPW loads each, this wave's) may still be on their way
        const int ahead = nst - 1 - kt;
        if (NSTG == 4 && ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * IPW) : "memory");
        else if (ahead >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const char *sbase = smem + (kt % NSTG) * STAGE_B;
#pragma unroll
        for (int ks = 0; ks < KR; ++ks) {
            if (KR > 1 && kt * KR + ks >= nk) break;
            const char *base = sbase + ks * (32 * ROWE * 2);
            // Y fragments first, then X's: row i of MFMAs starts when its X fragment (and all of Y) has arrived
            s16x4 lo[TM + TN], hi[TM + TN];
#pragma unroll
            for (int f = 0; f < TM + TN; ++f) {
                const int col = tn_swz_col<SWZ>(f < TN ? BN1 + (wn * TN + f) * 16 + tr_col : (wm * TM + (f - TN)) * 16 + tr_col, tr_row);
                const unsigned ad =
                    (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)(base + (tr_row * ROWE + col) * 2);
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo[f]) : "v"(ad));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi[f]) : "v"(ad), "n"(4 * ROWE * 2));
            }
            bf16x8 b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                // reads return in order: all but the 2 (TM - 1 - i) youngest have landed
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * (TM - 1 - i)) : "memory");
                if (i == 0) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        asm volatile("" : "+v"(lo[j]), "+v"(hi[j]));
                        const s16x8 v = {lo[j][0], lo[j][1], lo[j][2], lo[j][3], hi[j][0], hi[j][1], hi[j][2], hi[j][3]};
                        b[j] = __builtin_bit_cast(bf16x8, v);
                    }
                }
                asm volatile("" : "+v"(lo[TN + i]), "+v"(hi[TN + i]));
                const s16x8 va = {lo[TN + i][0], lo[TN + i][1], lo[TN + i][2], lo[TN + i][3],
                                  hi[TN + i][0], hi[TN + i][1], hi[TN + i][2], hi[TN + i][3]};
                const bf16x8 a = __builtin_bit_cast(bf16x8, va);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[j], acc[i][j], 0, 0, 0);
                    if (ks == 0 && i == 0 && j == 0) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (kt + NSTG - 1 < nst) stage((kt + NSTG - 1) % NSTG);   // the buffer of stage kt - 1 (all waves past it)
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
    }

    // D[row = 4 lg + r][col = li]
    float *__restrict__ C = G.nsplit > 1 ? static_cast<float *>(G.part) + (size_t)split * G.N1 * G.N2
                                         : static_cast<float *>(G.C);
    const long long ldc = G.nsplit > 1 ? G.N2 : G.ldc;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n1 = n1_0 + (wm * TM + i) * 16 + 4 * lg + r;
                const int n2 = n2_0 + (wn * TN + j) * 16 + li;
                if (n1 < G.N1 && n2 < G.N2) C[(long long)n1 * ldc + n2] = acc[i][j][r];
            }
}

// sum the split partials in a fixed order (deterministic): C[i] = sum_s part[s][i].  Streaming: 16-byte loads, eight
// splits in flight per thread (a scalar loop over the splits ran at a third of the HBM rate).
__global__ __launch_bounds__(256) void gemm_tn_reduce_kernel(GemmTnBatch B) {      // blockIdx.y = problem
    const GemmTnArgs &G = B.p[blockIdx.y];
    if (G.nsplit <= 1) return;
    const float *__restrict__ part = static_cast<const float *>(G.part);
    float *__restrict__ C = static_cast<float *>(G.C);
    const size_t n = (size_t)G.N1 * G.N2;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    if ((G.N2 & 3) == 0 && (G.ldc & 3) == 0 && !((uintptr_t)C & 15)) {
        const size_t n4 = n >> 2;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
            const f32x4 *p = reinterpret_cast<const f32x4 *>(part) + i;
            f32x4 a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            int k = 0;
            for (; k + 8 <= G.nsplit; k += 8)
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const f32x4 v = p[(size_t)(k + u) * n4];
                    a[u][0] += v[0]; a[u][1] += v[1]; a[u][2] += v[2]; a[u][3] += v[3];
                }
            for (; k < G.nsplit; ++k) {
                const f32x4 v = p[(size_t)k * n4];
                a[0][0] += v[0]; a[0][1] += v[1]; a[0][2] += v[2]; a[0][3] += v[3];
            }
            f32x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                r[e] = ((a[0][e] + a[1][e]) + (a[2][e] + a[3][e])) + ((a[4][e] + a[5][e]) + (a[6][e] + a[7][e]));
            const size_t e0 = i << 2;
            *reinterpret_cast<f32x4 *>(C + (e0 / G.N2) * G.ldc + (e0 % G.N2)) = r;
        }
        return;
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float s = 0.f;
        for (int k = 0; k < G.nsplit; ++k) s += part[(size_t)k * n + i];
        C[(i / G.N2) * G.ldc + (i % G.N2)] = s;
    }
}

// The same sum for SMALL outputs with MANY partials (the narrow weight gradients: 32 x 32 .. 256 x 128 outputs, 256
// partial slabs): one thread per output quad walks all the slabs in rounds of eight dependent-latency loads -- 32 rounds,
// ~40 us, as long as the GEMM itself.  Here SP threads share a quad: thread g of them sums slabs g, g + SP, g + 2 SP, ...
// (four in flight), the SP partial sums are added in the order g = 0 .. SP-1 through LDS.  Fixed order: deterministic.
template <int SP>
__global__ __launch_bounds__(256) void gemm_tn_reduce_sp_kernel(GemmTnBatch B) {      // blockIdx.y = problem
    constexpr int QB = 256 / SP;                   // output quads per workgroup
    __shared__ f32x4 red[SP][QB];
    const GemmTnArgs &G = B.p[blockIdx.y];
    if (G.nsplit <= 1) return;
    const f32x4 *__restrict__ part = static_cast<const f32x4 *>(G.part);
    float *__restrict__ C = static_cast<float *>(G.C);
    const size_t n4 = ((size_t)G.N1 * G.N2) >> 2;
    const int ql = threadIdx.x % QB, g = threadIdx.x / QB;
    for (size_t q0 = (size_t)blockIdx.x * QB; q0 < n4; q0 += (size_t)gridDim.x * QB) {      // uniform per workgroup
        const size_t q = q0 + ql;
        f32x4 a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (q < n4) {
            int k = g;
            for (; k + 3 * SP < G.nsplit; k += 4 * SP)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const f32x4 v = part[(size_t)(k + u * SP) * n4 + q];
                    a[u][0] += v[0]; a[u][1] += v[1]; a[u][2] += v[2]; a[u][3] += v[3];
                }
            for (; k < G.nsplit; k += SP) {
                const f32x4 v = part[(size_t)k * n4 + q];
                a[0][0] += v[0]; a[0][1] += v[1]; a[0][2] += v[2]; a[0][3] += v[3];
            }
        }
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = (a[0][e] + a[1][e]) + (a[2][e] + a[3][e]);
        red[g][ql] = r;
        __syncthreads();
        if (g == 0 && q < n4) {
            for (int j = 1; j < SP; ++j) {
                const f32x4 v = red[j][ql];
                r[0] += v[0]; r[1] += v[1]; r[2] += v[2]; r[3] += v[3];
            }
            const size_t e0 = q << 2;
            *reinterpret_cast<f32x4 *>(C + (e0 / G.N2) * G.ldc + (e0 % G.N2)) = r;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ small helpers
// dst[c][r] = (TO) src[r][c]   (weights only: a few MB at most)
template <typename TI, typename TO>
__global__ void transpose_cast_kernel(const TI *__restrict__ src, TO *__restrict__ dst, int rows, int cols) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        const int r = by + i, c = bx + threadIdx.x;
        tile[i][threadIdx.x] = (r < rows && c < cols) ? (float)src[(size_t)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        const int c = bx + i, r = by + threadIdx.x;
        if (r < rows && c < cols) dst[(size_t)c * rows + r] = (TO)tile[threadIdx.x][i];
    }
}

template <typename TI, typename TO>
__global__ void cast_scalar_kernel(const TI *__restrict__ src, TO *__restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = (TO)(float)src[i];
}

// four elements per thread and step (16 / 8-byte accesses; both pointers 16-byte aligned: launcher), scalar tail.
// ADD: dst = (TO)(src + (float)add[i]) -- the fp32 scatter target of a bf16 network's data gradient converted and added to the
// gradient that reached the same tensor by the other branch, in one pass
template <typename TI, typename TO, bool ADD = false>
__global__ void cast_kernel(const TI *__restrict__ src, TO *__restrict__ dst, size_t n, const TO *__restrict__ add = nullptr) {
    typedef TI vin __attribute__((ext_vector_type(4)));
    typedef TO vout __attribute__((ext_vector_type(4)));
    const size_t n4 = n >> 2, stride = (size_t)gridDim.x * blockDim.x, t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (size_t i = t0; i < n4; i += stride) {
        const vin v = reinterpret_cast<const vin *>(src)[i];
        float f[4] = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
        if constexpr (ADD) {
            const vout a = reinterpret_cast<const vout *>(add)[i];
#pragma unroll
            for (int q = 0; q < 4; ++q) f[q] += (float)a[q];
        }
        reinterpret_cast<vout *>(dst)[i] = vout{(TO)f[0], (TO)f[1], (TO)f[2], (TO)f[3]};
    }
    for (size_t i = 4 * n4 + t0; i < n; i += stride) {
        float f = (float)src[i];
        if constexpr (ADD) f += (float)add[i];
        dst[i] = (TO)f;
    }
}

// column statistics of the generic path: one thread per (32-row block, column) reads C back
template <typename TO>
__global__ void nt_stats_from_c_kernel(const TO *__restrict__ C, long long M, int N, long long ldc, float *__restrict__ part) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (M >> 5) * N) return;
    const long long rb = i / N;
    const int n = (int)(i % N);
    float s1 = 0.f, s2 = 0.f;
    for (int r = 0; r < 32; ++r) {
        const float v = (float)C[(rb * 32 + r) * ldc + n];
        s1 += v;
        s2 = fmaf(v, v, s2);
    }
    part[i * 2] = s1;
    part[i * 2 + 1] = s2;
}

// generic fallbacks (any shape, VALU): one thread per output element
template <typename T, typename TO>
__global__ void gemm_nt_generic_kernel(const T *__restrict__ A, const T *__restrict__ Bt, TO *__restrict__ C, long long M,
                                       int N, int K, long long lda, long long ldb, long long ldc) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * N) return;
    const long long m = i / N;
    const int n = (int)(i % N);
    float s = 0.f;
    for (int k = 0; k < K; ++k) s = fmaf((float)A[m * lda + k], (float)Bt[(long long)n * ldb + k], s);
    store_out(C + m * ldc + n, s);
}
// generic path: max|C| by a pass over C (the MFMA kernels take it from their accumulators, gemm.h nt_c_amax)
template <typename TO>
__global__ __launch_bounds__(256) void nt_amax_from_c_kernel(const TO *__restrict__ C, long long M, int N, long long ldc, unsigned *__restrict__ out) {
    unsigned m = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < M * N; i += (long long)gridDim.x * 256) {
        const unsigned a = __builtin_bit_cast(unsigned, (float)C[(i / N) * ldc + i % N]) & 0x7fffffffu;
        m = (a > m && a < 0x7f800000u) ? a : m;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned v = (unsigned)__shfl_xor((int)m, o, 64);
        m = v > m ? v : m;
    }
    if ((threadIdx.x & 63) == 0 && m > __atomic_load_n(out, __ATOMIC_RELAXED)) atomicMax(out, m);
}
template <typename T>
__global__ void gemm_tn_generic_kernel(const T *__restrict__ X, const T *__restrict__ Y, float *__restrict__ C,
                                       long long R, int N1, int N2, long long ldx, long long ldy, long long ldc) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)N1 * N2) return;
    const int n1 = (int)(i / N2), n2 = (int)(i % N2);
    float s = 0.f;
    for (long long r = 0; r < R; ++r) s = fmaf((float)X[r * ldx + n1], (float)Y[r * ldy + n2], s);
    C[(long long)n1 * ldc + n2] = s;
}

template <typename T, typename TO, int WGM, int WGN, int TM, int TN, int KS = 8, int NSTG = 2>
int launch_nt_cfg(GemmNtBatch &B, hipStream_t st) {
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    // longest contraction first: the tiles are dispatched in table order, and a tile's run time is ~K, so the short
    // tiles of the other problems fill the last round (the spectral blocks of a layer have K = d*c, d = 1..5)
    for (int i = 1; i < B.nprob; ++i)
        for (int k = i; k > 0 && B.p[k].K > B.p[k - 1].K; --k) {
            const GemmNtProb tmp = B.p[k]; B.p[k] = B.p[k - 1]; B.p[k - 1] = tmp;
        }
    unsigned total = 0;
    for (int i = 0; i < B.nprob; ++i) {
        GemmNtProb &p = B.p[i];
        p.tiles_n = (p.N + BN - 1) / BN;
        p.tile0 = total;
        p.ntile = (unsigned)((p.M + BM - 1) / BM) * p.tiles_n;
        total += i + 1 < B.nprob ? (p.ntile + 7u) & ~7u : p.ntile;
    }
    B.ntiles = total;
    if (total == 0) return 0;
    EPN_LAUNCH((gemm_nt_kernel<T, TO, WGM, WGN, TM, TN, KS, NSTG>), dim3(total), dim3(64 * WGM * WGN), 0, st, B);
    EPN_CHECK_LAUNCH();
    return 0;
}

template <typename T, typename TO>
int launch_nt_typed(GemmNtBatch &B, hipStream_t st) {
    constexpr int E16 = ElemOf<T>::PER16;
    bool fast = true, half_k = false;
    int maxn = 0, minn = 1 << 30;
    for (int i = 0; i < B.nprob; ++i) {
        const GemmNtProb &p = B.p[i];
        if (p.M < 0 || p.N < 1 || p.K < 1 || (p.stats && p.M % 32)) return EPN_EINVAL;
        if (!p.A || !p.Bt || !p.C) return EPN_ENULL;
        if (p.K % (4 * E16) || p.lda % E16 || p.ldb % E16 || ((uintptr_t)p.A & 15) || ((uintptr_t)p.Bt & 15)) fast = false;
        if (p.K % (8 * E16)) half_k = true;      // K a multiple of 4 slots only: the short-K-step kernels
        maxn = p.N > maxn ? p.N : maxn;
        minn = p.N < minn ? p.N : minn;
    }
    if (!fast) {
        for (int i = 0; i < B.nprob; ++i) {
            const GemmNtProb &p = B.p[i];
            if (p.M == 0) continue;
            const long long n = p.M * p.N;
            EPN_LAUNCH((gemm_nt_generic_kernel<T, TO>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                               static_cast<const T *>(p.A), static_cast<const T *>(p.Bt), static_cast<TO *>(p.C), p.M,
                               p.N, p.K, p.lda, p.ldb, p.ldc);
            EPN_CHECK_LAUNCH();
            if (p.c_amax) {
                EPN_LAUNCH_AUX((nt_amax_from_c_kernel<TO>), dim3((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0, st,
                               static_cast<const TO *>(p.C), p.M, p.N, p.ldc, p.c_amax);
                EPN_CHECK_LAUNCH();
            }
            if (p.stats) {
                const long long ns = (p.M >> 5) * p.N;
                EPN_LAUNCH_AUX((nt_stats_from_c_kernel<TO>), dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, st,
                               static_cast<const TO *>(p.C), p.M, p.N, p.ldc, p.stats);
                EPN_CHECK_LAUNCH();
            }
        }
        return 0;
    }
    if (half_k) {
        if (maxn <= 32) return launch_nt_cfg<T, TO, 8, 1, 2, 1, 4>(B, st);
        if (maxn <= 64) return launch_nt_cfg<T, TO, 8, 1, 2, 2, 4>(B, st);
        return launch_nt_cfg<T, TO, 4, 2, 2, 2, 4>(B, st);
    }
    const int pol = kernel_policy();
    if ((pol & ~0xff) == 0x100) {           // tuning override (tools/gemm_bench.py --cfg)
        switch (pol & 0xff) {
            case 1: return launch_nt_cfg<T, TO, 4, 2, 2, 2>(B, st);      // 256 x 128, 8 waves
            case 2: return launch_nt_cfg<T, TO, 4, 2, 2, 4>(B, st);      // 256 x 256, 8 waves
            case 3: return launch_nt_cfg<T, TO, 2, 2, 2, 2>(B, st);      // 128 x 128, 4 waves (2 workgroups / CU)
            case 4: return launch_nt_cfg<T, TO, 2, 4, 2, 2>(B, st);      // 128 x 256, 8 waves
            case 5: return launch_nt_cfg<T, TO, 4, 1, 2, 2>(B, st);      // 256 x 64, 4 waves
            case 6: return launch_nt_cfg<T, TO, 2, 2, 4, 2>(B, st);      // 256 x 128, 4 waves, 128 x 64 per wave
#ifdef EPN_TUNING
            // short-K (data-gradient) shapes: four-wave workgroups with 128 accumulators per wave and a half-width K step, so that TWO
            // fit a CU and one's store epilogue overlaps the other's loads (round 6 sweep, tools/r06_dg_sweep.sh ->
            // profiles/r06_bf16_dg_tile_sweep.txt).  MEASURED 25-30 % SLOWER than the 256 x 256 tile on every bf16 data-gradient
            // shape (245760 x 3072 x 256: 0.76-0.79 vs 0.60 ms; hipBLASLt through torch.mm: 0.60-0.62): kept for the record only
            case 7: return launch_nt_cfg<T, TO, 2, 2, 2, 4, 4, 2>(B, st);   // 128 x 256, 4 waves, K step 32 (bf16) / 16 (fp32), 48 KB
            case 8: return launch_nt_cfg<T, TO, 2, 2, 4, 2, 4, 2>(B, st);   // 256 x 128, 4 waves, same
            case 10: return launch_nt_cfg<T, TO, 2, 2, 2, 4, 8, 2>(B, st);  // 128 x 256, 4 waves, full K step (96 KB: one workgroup per CU)
#endif
            default: break;
        }
    }
    if (sizeof(T) == 4 && B.nprob == 1 && maxn >= 128 && maxn <= 256 && maxn % 128 == 0 &&
        (B.p[0].M / 128) * (maxn / 128) >= 3840)
        // many-tile problems: 128 x 128 tiles, 4 waves, two workgroups per CU (measured on 491520 x 128 and 245760 x 256:
        // 130-140 TFLOP/s against 124-136 for the 256-row tiles; below ~3840 tiles the big tiles win)
        return launch_nt_cfg<T, TO, 2, 2, 2, 2>(B, st);
    if (sizeof(T) == 4 && B.nprob > 1) {
        // grouped spectral blocks (widths d*c, d = 1, 3, 3, 4, 5): narrow tiles waste less of the ragged widths
        // (c = 64: 256 x 64 tiles 0.40 ms vs 0.45; c = 128 / 256: 128 x 128 tiles 0.65 / 1.22 vs 0.69 / 1.25, measured)
        if (maxn <= 320 && minn <= 64) return launch_nt_cfg<T, TO, 4, 1, 2, 2>(B, st);
        return launch_nt_cfg<T, TO, 2, 2, 2, 2>(B, st);
    }
    if (maxn <= 32) return launch_nt_cfg<T, TO, 8, 1, 2, 1>(B, st);      // 512 x 32
    if (maxn <= 64) return launch_nt_cfg<T, TO, 8, 1, 2, 2>(B, st);      // 512 x 64
    // 256 x 256 (fewer LDS fragment reads per MFMA: 6 per 8 instead of 4 per 4).  bf16 too: 11-12 % faster
    // than the 3-stage 256 x 128 instance on the N >= 256 shapes (245760 x 256 x 6144: 0.92 -> 0.82 ms)
    if (minn >= 256) return launch_nt_cfg<T, TO, 4, 2, 2, 4>(B, st);   // (fp32 groups were routed above)
    // 256 x 128.  bf16 is HBM-bound on these shapes (it streams G or dG): a THREE-stage LDS ring keeps two stages of loads
    // in flight (144 KB of LDS; the wait before the barrier is vmcnt(loads of one stage), not 0): 10-16 % faster on the
    // schedule's shapes (245760 x 256 x 6144: 1.07 -> 0.93 ms).  Narrow tiles and fp32 (MFMA-bound) measured no gain.
    if constexpr (sizeof(T) == 2) return launch_nt_cfg<T, TO, 4, 2, 2, 2, 8, 3>(B, st);
    else return launch_nt_cfg<T, TO, 4, 2, 2, 2>(B, st);
}

constexpr int tn_waves(int wgm, int wgn, int, int, int) { return wgm * wgn; }
// split form: single wide weight gradients take the pre-split planes kernel, narrow / grouped ones split in the kernel
// (N2 = the narrowest output; groups of c = 128 blocks measured slower with the extra pass over X: 1.43 vs 1.34 ms,
// c = 256 ones too once timed cold -- X is as large as Y in a spectral block, so the pre-split is a full extra pass)
#ifndef EPN_TN_GROUP_PLANES     // narrowest output from which a GROUP takes the pre-split planes kernel (256 x 256 tiles)
#define EPN_TN_GROUP_PLANES 512  // (three-piece form; the two-piece form: 256, see below) cold, tools/tn_probe.py, three-piece bf16 form: c = 256 groups 1.05 ms in-kernel split vs 1.14 planes; c = 512: 2.40 vs 1.48.
                                 // Two-piece fp16 form (round 5: the planes are 4 bytes per value, the MFMA half is twice as fast, so the in-kernel split of BOTH
                                 // operands weighs more; tools/spectral_dw_probe.py, maxima supplied): c = 256, 4096 points 0.751 -> 0.632 ms, 2048 points 0.368 -> 0.361;
                                 // c = 128 0.420 -> 0.441, c = 64 0.296 -> 0.346 (stay on the in-kernel split)
#endif
inline int tn_group_planes(int x3) { return x3 == 2 ? EPN_TN_GROUP_PLANES / 2 : EPN_TN_GROUP_PLANES; }
inline bool tn_planes_form(int nprob, int N2, int x3) { return N2 >= (nprob == 1 ? 512 : tn_group_planes(x3)); }

template <typename T>
bool tn_fast_ok(const GemmTnArgs &G) {
    constexpr int E16 = ElemOf<T>::PER16;
    return G.R > 0 && G.R % 32 == 0 && G.N1 >= E16 && G.N2 >= E16 && G.ldx % E16 == 0 && G.ldy % E16 == 0 &&
           !((uintptr_t)G.X & 15) && !((uintptr_t)G.Y & 15) && G.N1 % E16 == 0 && G.N2 % E16 == 0;
}

// Plan of a (grouped) TN launch: one block tile for all problems; splits so that every workgroup runs about the same
// number of K steps and the launch has ~2048 workgroups (single problem) / ~1024 (group); partial slabs carved from `ws`.
template <typename T>
size_t tn_plan(GemmTnBatch &B, int *bn1_out, int *bn2_out, void *ws, int x3 = 0, size_t *amax_off = nullptr) {
    // x3: 0 = operands as they are, 3 = fp32 operands in three bf16 pieces, 2 = in two fp16 pieces (same tiles and splits)
    const int bf = sizeof(T) == 2 ? 1 : (x3 ? 2 : 0);
    int max1 = 0, min2 = 1 << 30;
    for (int i = 0; i < B.nprob; ++i) {
        max1 = B.p[i].N1 > max1 ? B.p[i].N1 : max1;
        min2 = B.p[i].N2 < min2 ? B.p[i].N2 : min2;
    }
    int bn1, bn2;
    gemm_tn_tile(B.nprob > 1 && bf == 2 ? 0 : bf, max1, B.nprob > 1 && min2 < 256 ? 256 : min2, &bn1, &bn2);   // groups: the wide tiles
    if (B.nprob > 1 && bf == 2 && min2 >= tn_group_planes(x3)) {       // wide spectral groups (c >= 512), split form: 256 x 256 tiles
        int min1 = 1 << 30;                             // halve the re-reads of X and Y (every tile row / column streams
        for (int i = 0; i < B.nprob; ++i) min1 = B.p[i].N1 < min1 ? B.p[i].N1 : min1;   // the other operand again)
        if (min1 >= 256) { bn1 = 256; bn2 = 256; }
    }
    // narrow spectral groups (c < 256) in the two-piece form: 128 x 128 tiles -- 64 KB of stages, two workgroups per CU, and the
    // ragged widths (c, 3c, 3c, 4c, 5c) pad less: c = 64 0.302 -> 0.279 ms, c = 128 0.422 -> 0.348 (tools/spectral_dw_probe.py)
    if (B.nprob > 1 && x3 == 2 && min2 < 256) { bn1 = 128; bn2 = 128; }
    *bn1_out = bn1; *bn2_out = bn2;
    long long tiles[GEMM_MAX_PROB], chunks[GEMM_MAX_PROB];
    for (int i = 0; i < B.nprob; ++i) {
        GemmTnArgs &G = B.p[i];
        G.tiles_n2 = (G.N2 + bn2 - 1) / bn2;
        G.ntiles = (unsigned)((G.N1 + bn1 - 1) / bn1) * G.tiles_n2;
        tiles[i] = G.ntiles;
        chunks[i] = G.R / 32;
    }
    if (B.nprob == 1) {
        B.p[0].nsplit = gemm_tn_splits(bf, B.p[0].R, B.p[0].N1, B.p[0].N2);
    } else {
        // smallest steps-per-workgroup S (>= 8) whose launch fits `target` workgroups; every problem gets ceil(chunks / S)
        // splits.  The five problems of a group are ragged (d = 1, 3, 3, 4, 5 times the rows AND the width): with ONE round
        // of 256 workgroups the launch ends when its longest ones do (76 % of the CU-time used, SQ_BUSY / GRBM 3.1 against
        // 3.85 for the single-problem kernels); several rounds of shorter workgroups fill in behind each other.  fp32 groups,
        // cold (tools/tn_probe.py), 256 / 384 / 512 / 768 / 1024 / 1536 / 2048 workgroups: c = 64: 0.64 / 0.49 / 0.52 / 0.42 /
        // 0.43 / 0.45 / 0.48 ms, c = 128: 0.84 / 0.67 / 0.71 / 0.60 / 0.61 / 0.64 / 0.67, c = 256: 0.98 / 1.03 / 0.91 / 0.97 /
        // 0.94 / 0.97 / 1.00.  (Round 2 measured the opposite, 6.2 / 6.7 / 7.4 / 7.9 ms per step for 256 / 512 / 1024 / 2048:
        // that was the cost of its slab reduction, one thread per output quad walking every slab -- gemm_tn_reduce_sp_kernel.)
        const long long target = bf == 1 ? EPN_TN_GROUP_TARGET_BF16 : EPN_TN_GROUP_TARGET_F32;
        long long S = 8;
        for (long long cand = 8; cand <= 8192; ++cand) {
            long long blocks = 0;
            for (int i = 0; i < B.nprob; ++i) blocks += tiles[i] * ((chunks[i] + cand - 1) / cand);
            S = cand;
            if (blocks <= target) break;
        }
        for (int i = 0; i < B.nprob; ++i) {
            long long sp = (chunks[i] + S - 1) / S;
            sp = sp < 1 ? 1 : (sp > 512 ? 512 : sp);
            B.p[i].nsplit = (int)sp;
        }
    }
    size_t off = 0;
    unsigned blk = 0;
    for (int i = 0; i < B.nprob; ++i) {
        GemmTnArgs &G = B.p[i];
        G.block0 = blk;
        blk += (G.ntiles * (unsigned)G.nsplit + 7u) & ~7u;      // whole octets: workgroup % 8 (= XCD) holds per problem
        G.part = nullptr; G.part_bytes = 0;
        if (G.nsplit > 1) {
            const size_t nb = (size_t)G.nsplit * G.N1 * G.N2 * sizeof(float);
            if (ws) G.part = static_cast<char *>(ws) + off;
            G.part_bytes = nb;
            off += (nb + 255) & ~(size_t)255;
        }
    }
    B.nblocks = blk;
    for (int i = 0; i < B.nprob; ++i) B.p[i].Xp = nullptr;
    const bool x3_tile = (bn1 == 32 && bn2 == 512) || (bn1 == 64 && bn2 == 512) || (bn1 == 256 && bn2 == 256) ||
                         (bn1 == 128 && (bn2 == 256 || bn2 == 512));      // instances of gemm_tn_x3_kernel
    if (x3 && x3_tile && tn_planes_form(B.nprob, min2, x3))  // bf16 planes of X: 6 bytes per value
        for (int i = 0; i < B.nprob; ++i) {
            GemmTnArgs &G = B.p[i];
            // (a null workspace = size query: a non-null marker keeps the two passes on the same path)
            G.Xp = ws ? static_cast<char *>(ws) + off : reinterpret_cast<const void *>(1);
            off += ((size_t)2 * x3 * G.R * G.N1 + 255) & ~(size_t)255;
        }
    if (x3 == 2) {                              // two-piece form: [max|X_i|, max|Y_i|] per problem when the caller has none
        if (amax_off) *amax_off = off;
        off += 256;
    }
    return off;
}

template <typename T>
int launch_tn_typed(GemmTnBatch &B, void *ws, size_t ws_bytes, hipStream_t st, int x3 = 0) {
    bool fast = true;
    for (int i = 0; i < B.nprob; ++i) {
        const GemmTnArgs &G = B.p[i];
        if (G.R < 0 || G.N1 < 1 || G.N2 < 1) return EPN_EINVAL;
        if (!G.C) return EPN_ENULL;
        if (G.R > 0 && (!G.X || !G.Y)) return EPN_ENULL;
        fast = fast && tn_fast_ok<T>(G);
    }
    if (!fast) {
        for (int i = 0; i < B.nprob; ++i) {
            const GemmTnArgs &G = B.p[i];
            const long long n = (long long)G.N1 * G.N2;
            EPN_LAUNCH((gemm_tn_generic_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                               static_cast<const T *>(G.X), static_cast<const T *>(G.Y), static_cast<float *>(G.C), G.R,
                               G.N1, G.N2, G.ldx, G.ldy, G.ldc);
            EPN_CHECK_LAUNCH();
        }
        return 0;
    }
    int bn1, bn2;
    size_t amax_off = 0;
    const size_t need = tn_plan<T>(B, &bn1, &bn2, ws, x3, &amax_off);
    if (need > 0 && (!ws || ws_bytes < need)) return EPN_EWORKSPACE;
    const dim3 grid(B.nblocks);
    if constexpr (sizeof(T) == 4) {
        if (x3 == 2) {                                  // maxima nobody supplied: one pass over the operand each
            float *slots = reinterpret_cast<float *>(static_cast<char *>(ws) + amax_off);
            for (int i = 0; i < B.nprob; ++i) {
                GemmTnArgs &G = B.p[i];
                if (!G.x_amax) {
                    int rc = launch_absmax(static_cast<const float *>(G.X), G.ldx, G.R, G.N1, slots + 2 * i, st);
                    if (rc) return rc;
                    G.x_amax = slots + 2 * i;
                }
                if (!G.y_amax) {
                    int rc = launch_absmax(static_cast<const float *>(G.Y), G.ldy, G.R, G.N2, slots + 2 * i + 1, st);
                    if (rc) return rc;
                    G.y_amax = slots + 2 * i + 1;
                }
            }
        }
#define EPN_TN(...)                                                                                                  \
    do {                                                                                                             \
        if (x3 == 3) EPN_LAUNCH((gemm_tn_f32_kernel<__VA_ARGS__, 3>), grid, dim3(64 * tn_waves(__VA_ARGS__)), 0, st, B);  \
        else if (x3 == 2) EPN_LAUNCH((gemm_tn_f32_kernel<__VA_ARGS__, 2>), grid, dim3(64 * tn_waves(__VA_ARGS__)), 0, st, B);  \
        else EPN_LAUNCH((gemm_tn_f32_kernel<__VA_ARGS__, 0>), grid, dim3(64 * tn_waves(__VA_ARGS__)), 0, st, B);    \
    } while (0)
#define EPN_TX(...)                                                                                                  \
    do {                                                                                                             \
        if (x3 == 3) EPN_LAUNCH((gemm_tn_x3_kernel<__VA_ARGS__, 3>), grid, dim3(64 * tn_waves(__VA_ARGS__)), 0, st, B);   \
        else EPN_LAUNCH((gemm_tn_x3_kernel<__VA_ARGS__, 2>), grid, dim3(64 * tn_waves(__VA_ARGS__)), 0, st, B);          \
    } while (0)
        if (x3 && B.p[0].Xp) {
            for (int i = 0; i < B.nprob; ++i) {         // X's planes (workspace, after the slabs)
                const GemmTnArgs &G = B.p[i];
                const long long n = (G.R >> 3) * G.N1;
                if (x3 == 3)
                    EPN_LAUNCH_AUX(split_octets_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                                   static_cast<const float *>(G.X), G.ldx, G.R, G.N1, static_cast<u32x4 *>(const_cast<void *>(G.Xp)));
                else
                    EPN_LAUNCH_AUX(split_octets2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                                   static_cast<const float *>(G.X), G.ldx, G.R, G.N1, static_cast<u32x4 *>(const_cast<void *>(G.Xp)), G.x_amax);
                EPN_CHECK_LAUNCH();
            }
            if (bn1 == 32) EPN_TX(1, 8, 1, 2, 32);           // (bn1, bn2) is one of tn_plan's x3_tile pairs
            else if (bn1 == 64) EPN_TX(1, 8, 2, 2, 16);
            else if (bn1 == 256) EPN_TX(2, 4, 4, 2, 16);
            else if (bn2 == 256) EPN_TX(2, 4, 2, 2, 32);     // grouped spectral weight gradients (c >= 128)
            else EPN_TX(1, 8, 4, 2, 16);
        }
        else if (bn1 == 32) EPN_TN(1, 8, 1, 2, 32);
        else if (bn1 == 64 && bn2 == 64) EPN_TN(2, 2, 1, 1, 32);
        else if (bn1 == 64 && bn2 == 128) EPN_TN(2, 2, 1, 2, 32);
        else if (bn1 == 128 && bn2 == 64) EPN_TN(2, 2, 2, 1, 32);
        else if (bn1 == 128 && bn2 == 128) EPN_TN(2, 2, 2, 2, 32);
        else if (bn1 == 64 && bn2 == 512) EPN_TN(1, 4, 2, 4, 16);
        else if (bn1 == 64) EPN_TN(1, 8, 2, 1, 32);
        else if (bn2 == 512) EPN_TN(1, 8, 4, 2, 32);
        else EPN_TN(2, 4, 2, 2, 32);
#undef EPN_TN
#undef EPN_TX
    } else {
#define EPN_TNR(...)                                                                                                     \
    EPN_LAUNCH((gemm_tn_bf16_ring_kernel<__VA_ARGS__, EPN_TN_NARROW_NSTG, tn_ring_kr(__VA_ARGS__)>), grid, dim3(256), 0, st, B)
        if (bn2 <= 128 && B.nprob == 1) {               // narrow single problems (gemm_tn_tile): 2 x 2 waves, ring of stages
            if (bn1 == 32 && bn2 == 32) EPN_TNR(2, 2, 1, 1);
            else if (bn1 == 64 && bn2 == 32) EPN_TNR(2, 2, 2, 1);
            else if (bn1 == 32 && bn2 == 64) EPN_TNR(2, 2, 1, 2);
            else if (bn1 == 64 && bn2 == 64) EPN_TNR(2, 2, 2, 2);
            else if (bn1 == 128 && bn2 == 64) EPN_TNR(2, 2, 4, 2);
            else if (bn1 == 256 && bn2 == 64) EPN_TNR(2, 2, 8, 2);
            else if (bn1 == 32 && bn2 == 128) EPN_TNR(2, 2, 1, 4);
            else if (bn1 == 64 && bn2 == 128) EPN_TNR(2, 2, 2, 4);
            else if (bn1 == 128 && bn2 == 128) EPN_TNR(2, 2, 4, 4);
            else EPN_TNR(2, 2, 8, 4);                   // 256 x 128
        }
#undef EPN_TNR
        else if (bn1 == 32) EPN_LAUNCH((gemm_tn_bf16_kernel<1, 4, 2, 4>), grid, dim3(256), 0, st, B);        // 4 waves: fewer,
        else if (bn1 == 64) EPN_LAUNCH((gemm_tn_bf16_kernel<1, 4, 4, 4>), grid, dim3(256), 0, st, B);   // larger wave tiles
        // 128 x 256 tile on FOUR waves (64 x 128 per wave: 32 MFMAs per 12 transposed LDS reads).  The 8-wave form (16 MFMAs
        // per 16 reads) was LDS-bandwidth bound -- 3 workgroups x 64 KB of fragment reads per 1024 cycles > 128 B/clk:
        // 245760 x 256 x 6144: 1.63 -> 1.23 ms (630 TFLOP/s)
#if EPN_TN_WIDE_RING
        else if (B.nprob > 1 || EPN_TN_WIDE_RING > 1)
            EPN_LAUNCH((gemm_tn_bf16_ring_kernel<2, 2, 4, 8, EPN_TN_WIDE_NSTG, 1>), grid, dim3(256), 0, st, B);
#endif
        else EPN_LAUNCH((gemm_tn_bf16_kernel<2, 2, 4, 8>), grid, dim3(256), 0, st, B);
    }
    EPN_CHECK_LAUNCH();
    bool any_split = false;
    size_t nmax = 0;
    for (int i = 0; i < B.nprob; ++i) {
        any_split = any_split || B.p[i].nsplit > 1;
        const size_t n = (size_t)B.p[i].N1 * B.p[i].N2;
        nmax = n > nmax ? n : nmax;
    }
    if (any_split) {
        const size_t nv = (nmax + 3) / 4;               // one thread per four outputs
        bool quads = true;                              // (the shared-quad form needs the 16-byte path of every problem)
        int smax = 1;
        for (int i = 0; i < B.nprob; ++i) {
            const GemmTnArgs &G = B.p[i];
            quads = quads && (G.N2 & 3) == 0 && (G.ldc & 3) == 0 && !((uintptr_t)G.C & 15);
            smax = G.nsplit > smax ? G.nsplit : smax;
        }
        // many partials: SP threads per quad so that a thread walks its slabs in at most ~4 rounds of four loads (while
        // the launch stays under a million threads)
        int sp = 1;
        while (EPN_TN_REDUCE_SP && quads && sp < 64 && smax > 16 * sp && nv * B.nprob * sp * 4 <= (1u << 20)) sp *= 4;
        if (sp > 1) {
            const size_t qb = 256 / sp;
            const unsigned gx = (unsigned)((nv + qb - 1) / qb < 2048 ? (nv + qb - 1) / qb : 2048);
            if (sp == 4) EPN_LAUNCH_AUX(gemm_tn_reduce_sp_kernel<4>, dim3(gx, B.nprob), dim3(256), 0, st, B);
            else if (sp == 16) EPN_LAUNCH_AUX(gemm_tn_reduce_sp_kernel<16>, dim3(gx, B.nprob), dim3(256), 0, st, B);
            else EPN_LAUNCH_AUX(gemm_tn_reduce_sp_kernel<64>, dim3(gx, B.nprob), dim3(256), 0, st, B);
        } else {
            const unsigned gx = (unsigned)((nv + 255) / 256 < 2048 ? (nv + 255) / 256 : 2048);
            EPN_LAUNCH_AUX(gemm_tn_reduce_kernel, dim3(gx, B.nprob), dim3(256), 0, st, B);
        }
        EPN_CHECK_LAUNCH();
    }
    return 0;
}

}  // namespace

long long f2_nonfinite_take_gemm(bool reset) {
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

// block tile of the TN kernels for an output of N1 x N2 (shared with the workspace query)
void gemm_tn_tile(int dtype, int N1, int N2, int *bn1, int *bn2) {   // dtype: 0 fp32, 1 bf16, 2 fp32 split form
    if (dtype == 1) {
        if (N2 <= 128 && EPN_TN_NARROW) {        // 1x1 convolutions: the tile IS the output, ring form (gemm_tn_bf16_ring_kernel)
            *bn2 = N2 <= 32 ? 32 : (N2 <= 64 ? 64 : 128);
            *bn1 = N1 <= 32 ? 32 : (N1 <= 64 ? 64 : (N1 <= 128 ? 128 : 256));
            if (*bn2 == 32 && *bn1 > 64) *bn1 = 64;
            return;
        }
        if (N1 <= 32) { *bn1 = 32; *bn2 = 256; }
        else if (N1 <= 64) { *bn1 = 64; *bn2 = 256; }
        else { *bn1 = 128; *bn2 = 256; }
    } else {
        // wide outputs (dW of the inter convolutions: N2 = cin*ks): 512-column tiles, 8 MFMAs per pair of LDS reads;
        // narrow ones (spectral blocks, 1x1 convolutions): 256-column tiles
        // narrow single problems (dW of the 1x1 convolutions: N2 = cin <= 128): 64- / 128-column tiles, 4 waves
        if (dtype == 2 && N2 >= 512) {   // split form, wide outputs (gemm_tn_x3_kernel): X planes + Y rows, two stages in LDS
            if (N1 <= 32) { *bn1 = 32; *bn2 = 512; }
            else if (N1 <= 64) { *bn1 = 64; *bn2 = 512; }
            else if (N1 >= 256) { *bn1 = 256; *bn2 = 256; }     // X covered by one tile row: Y is streamed once
            else { *bn1 = 128; *bn2 = 512; }
            return;
        }
        if (N1 <= 32 && N2 > 128) { *bn1 = 32; *bn2 = 512; }      // (narrow N2: the 64-row tiles below, not a 512-column tile)
        else if (N1 <= 64) { *bn1 = 64; *bn2 = N2 >= 512 ? 512 : (N2 > 128 ? 256 : (N2 > 64 ? 128 : 64)); }
        else { *bn1 = 128; *bn2 = N2 >= 512 ? 512 : (N2 > 128 ? 256 : (N2 > 64 ? 128 : 64)); }
    }
}

int gemm_tn_splits(int dtype, long long R, int N1, int N2) {
    int bn1, bn2;
    gemm_tn_tile(dtype, N1, N2, &bn1, &bn2);
    const long long tiles = (long long)((N1 + bn1 - 1) / bn1) * ((N2 + bn2 - 1) / bn2);
    const long long chunks = R / 32;
    const int pol = kernel_policy();
    const long long target = (pol & ~0xff) == 0x200 ? 256LL * (pol & 0xff) : (dtype == 1 ? EPN_TN_SINGLE_TARGET_BF16 : 512);    // 0x200 | v: tuning override
    // two rounds of the 256 CUs.  With the split count rounded down (below) 512 / 1024 / 1536 / 2048 workgroups run the dW
    // shapes of the ModelNet schedule within 1.5 % of each other (18.7 / 18.8 / 18.9 / 19.0 ms summed, fp32 partial slabs and
    // their fixed-order reduction included); fewer workgroups = fewer partial slabs
    // rounded DOWN: the wide tiles take a whole CU's LDS, so 2048 workgroups are exactly 8 rounds of the 256 CUs and one
    // workgroup more is a ninth round that runs 16 workgroups wide (24 tiles x 86 splits = 2064: measured 118 -> 129 TFLOP/s)
    if (dtype == 1 && N2 <= 128 && EPN_TN_NARROW && (pol & ~0xff) != 0x200) {
        // ring form: EPN_TN_NARROW_WGS workgroups per CU, all resident; >= 16 steps per split
        long long sn = 256LL * EPN_TN_NARROW_WGS / tiles;
        const long long cap = chunks / 16 > 1 ? chunks / 16 : 1;
        sn = sn > cap ? cap : sn;
        return (int)(sn < 1 ? 1 : sn);
    }
    long long s = target / tiles;
    // at least 32 K steps per split: a split ends in an N1 x N2 fp32 slab write (+ its share of the reduction), which
    // for the short-and-wide problems (spectral blocks: R = pts*d rows, up to 1280 x 1280 outputs) outweighs 8 steps of loads
    const long long smax = chunks / 32 > 1 ? chunks / 32 : 1;
    if (s > smax) s = smax;
    if (s > 512) s = 512;
    return (int)(s < 1 ? 1 : s);
}

int launch_gemm_nt(GemmNtBatch &B, int dtype, int out_dtype, hipStream_t st) {
    if (B.nprob < 1 || B.nprob > GEMM_MAX_PROB) return EPN_EINVAL;
    if (dtype == 0 && out_dtype == 0) return launch_nt_typed<float, float>(B, st);
    if (dtype == 1 && out_dtype == 1) return launch_nt_typed<__bf16, __bf16>(B, st);
    if (dtype == 1 && out_dtype == 0) return launch_nt_typed<__bf16, float>(B, st);
    return EPN_EINVAL;
}

int launch_gemm_tn_batch(GemmTnBatch &B, int dtype, void *ws, size_t ws_bytes, hipStream_t st) {
    if (B.nprob < 1 || B.nprob > GEMM_MAX_PROB) return EPN_EINVAL;
    if (dtype == 2) return launch_tn_typed<float>(B, ws, ws_bytes, st, 3);    // fp32 operands, three bf16 pieces
    if (dtype == 3) return launch_tn_typed<float>(B, ws, ws_bytes, st, 2);    // fp32 operands, two fp16 pieces
    return dtype == 0 ? launch_tn_typed<float>(B, ws, ws_bytes, st) : launch_tn_typed<__bf16>(B, ws, ws_bytes, st);
}

size_t gemm_tn_batch_workspace(GemmTnBatch &B, int dtype) {
    int bn1, bn2;
    for (int i = 0; i < B.nprob; ++i)
        if (B.p[i].R < 32 || B.p[i].N1 < 1 || B.p[i].N2 < 1) return 0;
    return dtype != 1 ? tn_plan<float>(B, &bn1, &bn2, nullptr, dtype == 2 ? 3 : (dtype == 3 ? 2 : 0))
                      : tn_plan<__bf16>(B, &bn1, &bn2, nullptr);
}

int launch_gemm_tn(GemmTnArgs &G, int dtype, hipStream_t st) {
    GemmTnBatch B;
    B.nprob = 1; B.nblocks = 0; B.p[0] = G;
    return launch_gemm_tn_batch(B, dtype, G.part, G.part_bytes, st);
}

int launch_transpose_cast(const void *src, void *dst, int rows, int cols, int src_bf16, int dst_bf16, hipStream_t st) {
    if (rows < 1 || cols < 1) return EPN_EINVAL;
    if (!src || !dst) return EPN_ENULL;
    const dim3 grid((cols + 31) / 32, (rows + 31) / 32), blk(32, 8);
    if (!src_bf16 && !dst_bf16)
        EPN_LAUNCH((transpose_cast_kernel<float, float>), grid, blk, 0, st, (const float *)src, (float *)dst, rows, cols);
    else if (!src_bf16 && dst_bf16)
        EPN_LAUNCH((transpose_cast_kernel<float, __bf16>), grid, blk, 0, st, (const float *)src, (__bf16 *)dst, rows, cols);
    else if (src_bf16 && !dst_bf16)
        EPN_LAUNCH((transpose_cast_kernel<__bf16, float>), grid, blk, 0, st, (const __bf16 *)src, (float *)dst, rows, cols);
    else
        EPN_LAUNCH((transpose_cast_kernel<__bf16, __bf16>), grid, blk, 0, st, (const __bf16 *)src, (__bf16 *)dst, rows, cols);
    EPN_CHECK_LAUNCH();
    return 0;
}

int launch_cast(const void *src, void *dst, size_t n, int src_bf16, int dst_bf16, hipStream_t st) {
    if (n == 0) return 0;
    if (!src || !dst) return EPN_ENULL;
    // vector accesses need 16-byte aligned pointers; anything else runs the scalar tail loop over everything
    const bool al = !(((uintptr_t)src | (uintptr_t)dst) & 15);
    const size_t nv = al ? n : 0, work = al ? (n + 3) / 4 : n;
    const dim3 grid((unsigned)((work + 255) / 256 < 8192 ? (work + 255) / 256 : 8192)), blk(256);
    if (!src_bf16 && dst_bf16) {
        if (al) EPN_LAUNCH((cast_kernel<float, __bf16>), grid, blk, 0, st, (const float *)src, (__bf16 *)dst, nv, nullptr);
        else EPN_LAUNCH((cast_scalar_kernel<float, __bf16>), grid, blk, 0, st, (const float *)src, (__bf16 *)dst, n);
    } else if (src_bf16 && !dst_bf16) {
        if (al) EPN_LAUNCH((cast_kernel<__bf16, float>), grid, blk, 0, st, (const __bf16 *)src, (float *)dst, nv, nullptr);
        else EPN_LAUNCH((cast_scalar_kernel<__bf16, float>), grid, blk, 0, st, (const __bf16 *)src, (float *)dst, n);
    } else return EPN_EINVAL;
    EPN_CHECK_LAUNCH();
    return 0;
}

int launch_cast_add(const float *src, const void *add_bf16, void *dst_bf16, size_t n, hipStream_t st) {
    if (n == 0) return 0;
    if (!src || !add_bf16 || !dst_bf16) return EPN_ENULL;
    if (((uintptr_t)src | (uintptr_t)add_bf16 | (uintptr_t)dst_bf16) & 15) return EPN_EINVAL;
    const size_t work = (n + 3) / 4;
    const dim3 grid((unsigned)((work + 255) / 256 < 8192 ? (work + 255) / 256 : 8192)), blk(256);
    EPN_LAUNCH((cast_kernel<float, __bf16, true>), grid, blk, 0, st, src, (__bf16 *)dst_bf16, n, (const __bf16 *)add_bf16);
    EPN_CHECK_LAUNCH();
    return 0;
}

}  // namespace epn

using namespace epn;

static int nt_entry(int nprob, const epn_gemm_nt_problem *probs, int dtype, int out_dtype, epn_stream_t stream) {
    if (!probs) return EPN_ENULL;
    if (nprob < 1) return EPN_EINVAL;
    hipStream_t st = epn_stream(stream);
    for (int i0 = 0; i0 < nprob; i0 += GEMM_MAX_PROB) {
        GemmNtBatch B;
        B.nprob = nprob - i0 < GEMM_MAX_PROB ? nprob - i0 : GEMM_MAX_PROB;
        for (int i = 0; i < B.nprob; ++i) {
            const epn_gemm_nt_problem &q = probs[i0 + i];
            GemmNtProb &p = B.p[i];
            p.A = q.A; p.Bt = q.Bt; p.C = q.C; p.M = q.M; p.N = q.N; p.K = q.K; p.lda = q.lda; p.ldb = q.ldb; p.ldc = q.ldc;
            p.tiles_n = 0; p.tile0 = 0; p.stats = q.col_stats; p.c_amax = reinterpret_cast<unsigned *>(q.c_amax); p.a_amax = p.b_amax = nullptr;
        }
        int rc = launch_gemm_nt(B, dtype, out_dtype, st);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int epn_gemm_nt_f32(int nprob, const epn_gemm_nt_problem *probs, epn_stream_t stream) {
    return nt_entry(nprob, probs, 0, 0, stream);
}
extern "C" int epn_gemm_nt_bf16(int nprob, const epn_gemm_nt_problem *probs, int out_f32, epn_stream_t stream) {
    return nt_entry(nprob, probs, 1, out_f32 ? 0 : 1, stream);
}

extern "C" size_t epn_gemm_tn_workspace_bytes(int bf16, long long R, int N1, int N2) {
    if (R < 1 || N1 < 1 || N2 < 1) return 0;
    const int mode = bf16 == 3 ? 2 : bf16;            // 3 = two-piece fp16 form: the tiles and splits of the three-piece form
    const int s = gemm_tn_splits(mode, R, N1, N2);
    const size_t planes = mode == 2 && N2 >= 512 ? (((size_t)(bf16 == 3 ? 4 : 6) * R * N1 + 255) & ~(size_t)255) : 0;   // X's planes
    return (s > 1 ? (((size_t)s * N1 * N2 * sizeof(float) + 255) & ~(size_t)255) : 0) + planes + (bf16 == 3 ? 256 : 0);
}

static int tn_entry(const void *X, long long ldx, const void *Y, long long ldy, float *C, long long ldc, long long R, int N1,
                    int N2, void *ws, size_t ws_bytes, int dtype, epn_stream_t stream) {
    GemmTnArgs G;
    G.X = X; G.Y = Y; G.C = C; G.part = ws; G.part_bytes = ws_bytes; G.R = R; G.N1 = N1; G.N2 = N2;
    G.ldx = ldx; G.ldy = ldy; G.ldc = ldc; G.ntiles = 0; G.tiles_n2 = 0; G.nsplit = 1; G.block0 = 0; G.Xp = nullptr; G.x_amax = G.y_amax = nullptr;
    return launch_gemm_tn(G, dtype, epn_stream(stream));
}
extern "C" int epn_gemm_tn_f32(const float *X, long long ldx, const float *Y, long long ldy, float *C, long long ldc,
                               long long R, int N1, int N2, void *workspace, size_t workspace_bytes, epn_stream_t stream) {
    return tn_entry(X, ldx, Y, ldy, C, ldc, R, N1, N2, workspace, workspace_bytes, 0, stream);
}
extern "C" int epn_gemm_tn_split_f32(const float *X, long long ldx, const float *Y, long long ldy, float *C, long long ldc,
                                     long long R, int N1, int N2, void *workspace, size_t workspace_bytes, epn_stream_t stream) {
    return tn_entry(X, ldx, Y, ldy, C, ldc, R, N1, N2, workspace, workspace_bytes, 2, stream);
}
extern "C" int epn_gemm_tn_bf16(const void *X, long long ldx, const void *Y, long long ldy, float *C, long long ldc,
                                long long R, int N1, int N2, void *workspace, size_t workspace_bytes, epn_stream_t stream) {
    return tn_entry(X, ldx, Y, ldy, C, ldc, R, N1, N2, workspace, workspace_bytes, 1, stream);
}
extern "C" int epn_gemm_tn_f16x2_f32(const float *X, long long ldx, const float *Y, long long ldy, float *C, long long ldc,
                                     long long R, int N1, int N2, const float *x_amax, const float *y_amax, void *workspace,
                                     size_t workspace_bytes, epn_stream_t stream) {
    GemmTnArgs G;
    G.X = X; G.Y = Y; G.C = C; G.part = workspace; G.part_bytes = workspace_bytes; G.R = R; G.N1 = N1; G.N2 = N2;
    G.ldx = ldx; G.ldy = ldy; G.ldc = ldc; G.ntiles = 0; G.tiles_n2 = 0; G.nsplit = 1; G.block0 = 0; G.Xp = nullptr;
    G.x_amax = x_amax; G.y_amax = y_amax;
    return launch_gemm_tn(G, 3, epn_stream(stream));
}

static void tn_fill(GemmTnBatch &B, int nprob, const epn_gemm_tn_problem *probs) {
    B.nprob = nprob; B.nblocks = 0;
    for (int i = 0; i < nprob; ++i) {
        GemmTnArgs &G = B.p[i];
        const epn_gemm_tn_problem &q = probs[i];
        G.X = q.X; G.Y = q.Y; G.C = q.C; G.part = nullptr; G.part_bytes = 0; G.R = q.R; G.N1 = q.N1; G.N2 = q.N2;
        G.ldx = q.ldx; G.ldy = q.ldy; G.ldc = q.ldc; G.ntiles = 0; G.tiles_n2 = 0; G.nsplit = 1; G.block0 = 0; G.Xp = nullptr; G.x_amax = G.y_amax = nullptr;
    }
}
extern "C" size_t epn_gemm_tn_grouped_workspace_bytes(int bf16, int nprob, const epn_gemm_tn_problem *probs) {
    if (!probs || nprob < 1 || nprob > GEMM_MAX_PROB) return 0;
    GemmTnBatch B;
    tn_fill(B, nprob, probs);
    return gemm_tn_batch_workspace(B, bf16);
}
extern "C" int epn_gemm_tn_grouped(int bf16, int nprob, const epn_gemm_tn_problem *probs, void *workspace,
                                   size_t workspace_bytes, epn_stream_t stream) {
    if (!probs) return EPN_ENULL;
    if (nprob < 1 || nprob > GEMM_MAX_PROB) return EPN_EINVAL;
    GemmTnBatch B;
    tn_fill(B, nprob, probs);
    return launch_gemm_tn_batch(B, bf16 == 2 ? 2 : (bf16 ? 1 : 0), workspace, workspace_bytes, epn_stream(stream));
}
extern "C" int epn_gemm_tn_grouped_f16x2(int nprob, const epn_gemm_tn_problem *probs, const float *const *x_amax,
                                         const float *const *y_amax, void *workspace, size_t workspace_bytes, epn_stream_t stream) {
    if (!probs) return EPN_ENULL;
    if (nprob < 1 || nprob > GEMM_MAX_PROB) return EPN_EINVAL;
    GemmTnBatch B;
    tn_fill(B, nprob, probs);
    for (int i = 0; i < nprob; ++i) {
        B.p[i].x_amax = x_amax ? x_amax[i] : nullptr;
        B.p[i].y_amax = y_amax ? y_amax[i] : nullptr;
    }
    return launch_gemm_tn_batch(B, 3, workspace, workspace_bytes, epn_stream(stream));
}

extern "C" int epn_transpose_cast(const void *src, void *dst, int rows, int cols, int src_bf16, int dst_bf16,
                                  epn_stream_t stream) {
    return launch_transpose_cast(src, dst, rows, cols, src_bf16, dst_bf16, epn_stream(stream));
}
extern "C" int epn_cast(const void *src, void *dst, size_t n, int src_bf16, int dst_bf16, epn_stream_t stream) {
    return launch_cast(src, dst, n, src_bf16, dst_bf16, epn_stream(stream));
}
extern "C" int epn_cast_add_bf16(const float *src, const void *add, void *dst, size_t n, epn_stream_t stream) {
    return launch_cast_add(src, add, dst, n, epn_stream(stream));
}
