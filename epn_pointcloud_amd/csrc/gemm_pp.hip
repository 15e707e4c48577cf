// Split-form NT GEMM whose BOTH operands arrive as three bf16 planes ("pre-split"): no fp32 -> 3 x bf16 arithmetic in the
// main loop at all (gemm_nt_x3_kernel splits its A fragments in registers: 36 VALU instructions per fragment, 1.5 per MFMA
// on the 64 x 128 wave tile).  Round 5, review item 3: the producer of A (the grouping kernel / the basis change) would
// emit the planes; this file is the consumer, measured stand-alone first (tools/pp2_probe.py) against gemm_nt_x3_kernel.
//
//   C[M][N] = sum over the six kept piece products of (Ah + Am + Al)[M][K] . (Bh + Bm + Bl)[N][K]^T      (gemm_x3.hip)
//
// Plane layouts (`layout`):
//   0  row-major          [plane][rows][K]                       a K step of KS values is KS*2 bytes of every row
//   1  K-blocked          [plane][K / KS][rows][KS]              a K step of a tile is ONE contiguous block per plane
// LDS image of a stage: [A planes: plane][row][KS] then [B planes: plane][row][KS]; 1 KiB per wave-level direct-to-LDS load.
// KS = 32: 64-byte rows, 16-byte slots XOR-swizzled on the source address (as gemm_nt_x3_kernel's weight planes);
// KS = 16: 32-byte rows, a 32 x 16 fragment IS one contiguous KiB (lane = 2 row + half): conflict-free without a swizzle.
#include "conv_internal.h"
#include "gemm.h"

#ifdef EPN_TUNING
namespace epn {
namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void glds16(const void *g, char *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((glb_void *)g, (lds_void *)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ unsigned pack_rne(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float lo_f(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float hi_f(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

struct PPArgs {
    const __bf16 *Ap, *Bp;      // planes
    float *C;
    long long M, ldc;
    int N, K, layout;
};

// src fp32 [rows][ld] -> three planes in `layout`; two values per thread
__global__ void pp_split_kernel(const float *__restrict__ src, long long ld, long long rows, int K, unsigned *__restrict__ planes,
                                int layout, int KS) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int k2 = K >> 1;
    if (i >= rows * k2) return;
    const long long r = i / k2;
    const int k = 2 * (int)(i % k2);
    const float x0 = src[r * ld + k], x1 = src[r * ld + k + 1];
    const unsigned hp = pack_rne(x0, x1);
    const float r0 = x0 - lo_f(hp), r1 = x1 - hi_f(hp);
    const unsigned mp = pack_rne(r0, r1);
    const size_t plane = (size_t)rows * k2;
    const size_t o = layout == 0 ? (size_t)i : (((size_t)(k / KS) * rows + r) * KS + (k % KS)) >> 1;
    planes[o] = hp;
    planes[plane + o] = mp;
    planes[2 * plane + o] = pack_rne(r0 - lo_f(mp), r1 - hi_f(mp));
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// two-piece fp16 split (round 5 probe): x 2^s = h + l, h = rne_f16(x 2^s), l = rne_f16(x 2^s - h): 22-23 significant bits;
// products hh + hl + lh on v_mfma_f32_32x32x16_f16 -- THREE matrix instructions per 32x32x16 block instead of six
__global__ void pp_split2_kernel(const float *__restrict__ src, long long ld, long long rows, int K, unsigned *__restrict__ planes,
                                 int layout, int KS, float scale) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int k2 = K >> 1;
    if (i >= rows * k2) return;
    const long long r = i / k2;
    const int k = 2 * (int)(i % k2);
    const f32x2 x = {src[r * ld + k] * scale, src[r * ld + k + 1] * scale};
    const f16x2 h = __builtin_convertvector(x, f16x2);
    const f32x2 hf = __builtin_convertvector(h, f32x2);
    const f32x2 rem = {x[0] - hf[0], x[1] - hf[1]};
    const f16x2 l = __builtin_convertvector(rem, f16x2);
    const size_t plane = (size_t)rows * k2;
    const size_t o = layout == 0 ? (size_t)i : (((size_t)(k / KS) * rows + r) * KS + (k % KS)) >> 1;
    planes[o] = __builtin_bit_cast(unsigned, h);
    planes[plane + o] = __builtin_bit_cast(unsigned, l);
}

template <int WGM, int WGN, int TM, int TN, int NSTG, int KS>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_nt_pp2_kernel(PPArgs P, unsigned ntile, int tiles_n, float out_scale) {
    constexpr int NW = WGM * WGN;
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    constexpr int ROWB = KS * 2;
    constexpr int RPI = 1024 / ROWB;
    constexpr int LPR = ROWB / 16;
    constexpr int PA = BM * ROWB, PB = BN * ROWB;
    constexpr int STAGE = 2 * (PA + PB);
    constexpr int NGA = 2 * BM / RPI, NG = 2 * (BM + BN) / RPI;
    constexpr int GPW = (NG + NW - 1) / NW;
    static_assert(NG % NW == 0, "loads divide evenly over the waves");
    static_assert(NSTG * STAGE <= 160 * 1024, "LDS");
    constexpr int SPS = KS / 16;
    __shared__ __attribute__((aligned(1024))) char smem[NSTG * STAGE];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (blockIdx.x >= ntile) return;
    const unsigned t = epn_xcd_tile(blockIdx.x, ntile);
    const long long m0 = (long long)(t / tiles_n) * BM;
    const int n0 = (int)(t % tiles_n) * BN;
    const int nk = P.K / KS;

    const char *src[GPW];
    long long adv[GPW];
#pragma unroll
    for (int i = 0; i < GPW; ++i) {
        const int g = wave + i * NW;
        const bool isA = g < NGA;
        const int rb = RPI * (isA ? g : g - NGA) + lane / LPR;
        const int rows_t = isA ? BM : BN;
        const int plane = rb / rows_t, r = rb % rows_t;
        int part = lane % LPR;
        if (KS == 32) part ^= (r >> 2) & 3;
        const long long rows = isA ? P.M : (long long)P.N;
        long long gr = (isA ? m0 : (long long)n0) + r;
        gr = gr < rows ? gr : rows - 1;
        const __bf16 *base = isA ? P.Ap : P.Bp;
        if (P.layout == 0) {
            src[i] = reinterpret_cast<const char *>(base + ((size_t)plane * rows + gr) * P.K + part * 8);
            adv[i] = ROWB;
        } else {
            src[i] = reinterpret_cast<const char *>(base + (size_t)plane * rows * P.K + (size_t)gr * KS + part * 8);
            adv[i] = rows * ROWB;
        }
    }
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < GPW; ++i) {
            glds16(src[i], smem + buf * STAGE + (wave + i * NW) * 1024);
            src[i] += adv[i];
        }
    };
    const int wm = wave / WGN, wn = wave % WGN;
    const int li = lane & 31, lj = lane >> 5;
    int aoff[TM], boff[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) aoff[i] = ((wm * TM + i) * 32 + li) * ROWB;
#pragma unroll
    for (int i = 0; i < TN; ++i) boff[i] = 2 * PA + ((wn * TN + i) * 32 + li) * ROWB;
    const int fsw = (li >> 2) & 3;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int s = 0; s < NSTG - 1; ++s)
        if (s < nk) stage(s);
    for (int kt = 0; kt < nk; ++kt) {
        const int ahead = nk - 1 - kt < NSTG - 2 ? nk - 1 - kt : NSTG - 2;
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * GPW) : "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + NSTG - 1 < nk) stage((kt + NSTG - 1) % NSTG);
        const char *base = smem + (kt % NSTG) * STAGE;
#pragma unroll
        for (int s = 0; s < SPS; ++s) {
            const int so = KS == 32 ? ((2 * s + lj) ^ fsw) * 16 : lj * 16;
            f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[i] = *reinterpret_cast<const f16x8 *>(base + aoff[i] + so);
                al[i] = *reinterpret_cast<const f16x8 *>(base + aoff[i] + PA + so);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[j] = *reinterpret_cast<const f16x8 *>(base + boff[j] + so);
                bl[j] = *reinterpret_cast<const f16x8 *>(base + boff[j] + PB + so);
            }
#define EPN_PP_TERM(XA, XB)                                                                                 \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)         \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(XA[i], XB[j], acc[i][j], 0, 0, 0)
            EPN_PP_TERM(ah, bl);
            EPN_PP_TERM(al, bh);
            EPN_PP_TERM(ah, bh);
#undef EPN_PP_TERM
        }
    }
    float *__restrict__ C = P.C;
    float *__restrict__ cw = C + (size_t)(m0 + wm * TM * 32) * P.ldc + (n0 + wn * TN * 32);
    const unsigned ldc = (unsigned)P.ldc;
    const unsigned lane_off = 4u * lj * ldc + li;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned o = lane_off + (unsigned)(i * 32 + (r & 3) + 8 * (r >> 2)) * ldc;
#pragma unroll
            for (int j = 0; j < TN; ++j) cw[o + j * 32] = acc[i][j][r] * out_scale;      // (probe: interior tiles only)
        }
}

template <int WGM, int WGN, int TM, int TN, int NSTG, int KS>
int launch_pp2(const PPArgs &P, float out_scale, hipStream_t st) {
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    if (P.K % KS || P.M % BM || P.N % BN) return EPN_EINVAL;
    const int tiles_n = P.N / BN;
    const unsigned ntile = (unsigned)(P.M / BM) * tiles_n;
    EPN_LAUNCH((gemm_nt_pp2_kernel<WGM, WGN, TM, TN, NSTG, KS>), dim3(ntile), dim3(64 * WGM * WGN), 0, st, P, ntile, tiles_n, out_scale);
    EPN_CHECK_LAUNCH();
    return 0;
}

template <int WGM, int WGN, int TM, int TN, int NSTG, int KS>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_nt_pp_kernel(PPArgs P, unsigned ntile, int tiles_n) {
    constexpr int NW = WGM * WGN;
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    constexpr int ROWB = KS * 2;                        // bytes of a staged row
    constexpr int RPI = 1024 / ROWB;                    // rows per 1 KiB load instruction
    constexpr int LPR = ROWB / 16;                      // lanes per row
    constexpr int PA = BM * ROWB, PB = BN * ROWB;       // one plane of the A / B tile
    constexpr int STAGE = 3 * (PA + PB);
    constexpr int NGA = 3 * BM / RPI, NG = 3 * (BM + BN) / RPI;
    constexpr int GPW = (NG + NW - 1) / NW;
    static_assert(NG % NW == 0, "loads divide evenly over the waves");
    static_assert(NSTG * STAGE <= 160 * 1024, "LDS");
    constexpr int SPS = KS / 16;                        // fragment steps per stage
    __shared__ __attribute__((aligned(1024))) char smem[NSTG * STAGE];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (blockIdx.x >= ntile) return;
    const unsigned t = epn_xcd_tile(blockIdx.x, ntile);
    const long long m0 = (long long)(t / tiles_n) * BM;
    const int n0 = (int)(t % tiles_n) * BN;
    const int nk = P.K / KS;

    const char *src[GPW];
    long long adv[GPW];
#pragma unroll
    for (int i = 0; i < GPW; ++i) {
        const int g = wave + i * NW;
        const bool isA = g < NGA;
        const int rb = RPI * (isA ? g : g - NGA) + lane / LPR;      // plane * tile rows + row of the tile
        const int rows_t = isA ? BM : BN;
        const int plane = rb / rows_t, r = rb % rows_t;
        int part = lane % LPR;
        if (KS == 32) part ^= (r >> 2) & 3;
        const long long rows = isA ? P.M : (long long)P.N;
        long long gr = (isA ? m0 : (long long)n0) + r;
        gr = gr < rows ? gr : rows - 1;
        const __bf16 *base = isA ? P.Ap : P.Bp;
        if (P.layout == 0) {
            src[i] = reinterpret_cast<const char *>(base + ((size_t)plane * rows + gr) * P.K + part * 8);
            adv[i] = ROWB;
        } else {
            src[i] = reinterpret_cast<const char *>(base + (size_t)plane * rows * P.K + (size_t)gr * KS + part * 8);
            adv[i] = rows * ROWB;
        }
    }
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < GPW; ++i) {
            glds16(src[i], smem + buf * STAGE + (wave + i * NW) * 1024);
            src[i] += adv[i];
        }
    };

    const int wm = wave / WGN, wn = wave % WGN;
    const int li = lane & 31, lj = lane >> 5;
    int aoff[TM], boff[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) aoff[i] = ((wm * TM + i) * 32 + li) * ROWB;
#pragma unroll
    for (int i = 0; i < TN; ++i) boff[i] = 3 * PA + ((wn * TN + i) * 32 + li) * ROWB;
    const int fsw = (li >> 2) & 3;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
    for (int s = 0; s < NSTG - 1; ++s)
        if (s < nk) stage(s);
    for (int kt = 0; kt < nk; ++kt) {
        // stages kt + 1 .. kt + NSTG - 2 may stay in flight; stage kt must have landed
        const int ahead = nk - 1 - kt < NSTG - 2 ? nk - 1 - kt : NSTG - 2;
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * GPW) : "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + NSTG - 1 < nk) stage((kt + NSTG - 1) % NSTG);
        const char *base = smem + (kt % NSTG) * STAGE;
#pragma unroll
        for (int s = 0; s < SPS; ++s) {
            const int so = KS == 32 ? ((2 * s + lj) ^ fsw) * 16 : lj * 16;
            bf16x8 ah[TM], am[TM], al[TM], bh[TN], bm[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[i] = *reinterpret_cast<const bf16x8 *>(base + aoff[i] + so);
                am[i] = *reinterpret_cast<const bf16x8 *>(base + aoff[i] + PA + so);
                al[i] = *reinterpret_cast<const bf16x8 *>(base + aoff[i] + 2 * PA + so);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[j] = *reinterpret_cast<const bf16x8 *>(base + boff[j] + so);
                bm[j] = *reinterpret_cast<const bf16x8 *>(base + boff[j] + PB + so);
                bl[j] = *reinterpret_cast<const bf16x8 *>(base + boff[j] + 2 * PB + so);
            }
#define EPN_PP_TERM(XA, XB)                                                                                 \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)         \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(XA[i], XB[j], acc[i][j], 0, 0, 0)
            EPN_PP_TERM(ah, bl);                // small terms first
            EPN_PP_TERM(al, bh);
            EPN_PP_TERM(am, bm);
            EPN_PP_TERM(ah, bm);
            EPN_PP_TERM(am, bh);
            EPN_PP_TERM(ah, bh);
#undef EPN_PP_TERM
        }
    }

    float *__restrict__ C = P.C;
    if (m0 + BM <= P.M && n0 + BN <= P.N && (long long)BM * P.ldc < (1LL << 30)) {
        float *__restrict__ cw = C + (size_t)(m0 + wm * TM * 32) * P.ldc + (n0 + wn * TN * 32);
        const unsigned ldc = (unsigned)P.ldc;
        const unsigned lane_off = 4u * lj * ldc + li;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned o = lane_off + (unsigned)(i * 32 + (r & 3) + 8 * (r >> 2)) * ldc;
#pragma unroll
                for (int j = 0; j < TN; ++j) cw[o + j * 32] = acc[i][j][r];
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
                if (m < P.M && n < P.N) C[m * P.ldc + n] = acc[i][j][r];
            }
        }
}

template <int WGM, int WGN, int TM, int TN, int NSTG, int KS>
int launch_pp(const PPArgs &P, hipStream_t st) {
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    if (P.K % KS) return EPN_EINVAL;
    const int tiles_n = (P.N + BN - 1) / BN;
    const unsigned ntile = (unsigned)((P.M + BM - 1) / BM) * tiles_n;
    EPN_LAUNCH((gemm_nt_pp_kernel<WGM, WGN, TM, TN, NSTG, KS>), dim3(ntile), dim3(64 * WGM * WGN), 0, st, P, ntile, tiles_n);
    EPN_CHECK_LAUNCH();
    return 0;
}

}  // namespace
}  // namespace epn

using namespace epn;

// tuning library only (tools/pp2_probe.py)
extern "C" int epn_lab_pp_split(const float *src, long long ld, long long rows, int K, void *planes, int layout, int KS,
                                epn_stream_t stream) {
    hipStream_t st = epn_stream(stream);
    const long long pairs = rows * (K / 2);
    EPN_LAUNCH_AUX(pp_split_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, st, src, ld, rows, K,
                   static_cast<unsigned *>(planes), layout, KS);
    EPN_CHECK_LAUNCH();
    return 0;
}

extern "C" int epn_lab_pp_split2(const float *src, long long ld, long long rows, int K, void *planes, int layout, int KS,
                                 float scale, epn_stream_t stream) {
    hipStream_t st = epn_stream(stream);
    const long long pairs = rows * (K / 2);
    EPN_LAUNCH_AUX(pp_split2_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, st, src, ld, rows, K,
                   static_cast<unsigned *>(planes), layout, KS, scale);
    EPN_CHECK_LAUNCH();
    return 0;
}

extern "C" int epn_lab_gemm_nt_pp2(const void *Ap, const void *Bp, float *C, long long M, int N, int K, long long ldc, int layout,
                                   int cfg, float out_scale, epn_stream_t stream) {
    hipStream_t st = epn_stream(stream);
    PPArgs P{static_cast<const __bf16 *>(Ap), static_cast<const __bf16 *>(Bp), C, M, ldc, N, K, layout};
    switch (cfg) {
        case 0: return launch_pp2<4, 2, 2, 4, 3, 16>(P, out_scale, st);     // 256 x 256, K step 16, three stages (96 KB)
        case 1: return launch_pp2<4, 2, 2, 4, 2, 32>(P, out_scale, st);     // 256 x 256, K step 32, two stages (128 KB)
        case 2: return launch_pp2<4, 2, 2, 4, 4, 16>(P, out_scale, st);     // 256 x 256, K step 16, four stages (128 KB)
        case 3: return launch_pp2<4, 2, 2, 2, 2, 32>(P, out_scale, st);     // 256 x 128, K step 32, two stages (96 KB)
        case 4: return launch_pp2<4, 2, 2, 2, 3, 32>(P, out_scale, st);     // 256 x 128, K step 32, three stages (144 KB)
        default: return EPN_EINVAL;
    }
}

extern "C" int epn_lab_gemm_nt_pp(const void *Ap, const void *Bp, float *C, long long M, int N, int K, long long ldc, int layout,
                                  int cfg, epn_stream_t stream) {
    hipStream_t st = epn_stream(stream);
    PPArgs P{static_cast<const __bf16 *>(Ap), static_cast<const __bf16 *>(Bp), C, M, ldc, N, K, layout};
    switch (cfg) {
        case 0: return launch_pp<4, 2, 2, 4, 3, 16>(P, st);     // 256 x 256, K step 16, three stages (144 KB)
        case 1: return launch_pp<4, 2, 2, 2, 2, 32>(P, st);     // 256 x 128, K step 32, two stages (144 KB)
        case 2: return launch_pp<4, 2, 2, 4, 2, 16>(P, st);     // 256 x 256, K step 16, two stages (96 KB)
        case 3: return launch_pp<2, 4, 2, 2, 2, 32>(P, st);     // 128 x 256, K step 32, two stages
        case 4: return launch_pp<2, 2, 4, 2, 4, 16>(P, st);     // 256 x 128, 4 waves (128 x 64 each), K step 16, four stages (144 KB)
        case 5: return launch_pp<2, 2, 4, 2, 2, 16>(P, st);     // 256 x 128, 4 waves, K step 16, two stages (72 KB: two workgroups per CU)
        case 6: return launch_pp<4, 1, 2, 2, 2, 32>(P, st);     // 256 x 64, K step 32, two stages (4 waves, 120 KB)
        default: return EPN_EINVAL;
    }
}
#endif
