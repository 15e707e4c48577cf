// fp32 weight contractions on the bf16 matrix pipe, at fp32 accuracy ("split" GEMMs).
//
// v_mfma_f32_32x32x2_f32 runs at 64 FLOP/clk/SIMD, v_mfma_f32_32x32x16_bf16 at 1024.  An fp32 value splits WITHOUT LOSS
// into three bf16 pieces x = h + m + l: h = rne_bf16(x), m = rne_bf16(x - h), l = x - h - m (both remainders are exact
// in fp32, and the last one has at most 8 significant bits, i.e. it IS a bf16).  A product a*b is then nine piece
// products, each exact in the fp32 accumulator of the MFMA; the six of relative weight >= 2^-16
//     hh + (hm + mh) + (mm + hl + lh)
// are kept, the three dropped ones (ml, lm, ll) are together below 2^-24 |a||b| -- less than the ONE rounding an fp32
// FMA commits per product.  Six bf16 MFMAs of 32 cycles replace eight fp32 MFMAs of 64 cycles for the same 32x32x16
// block: 2.7x the matrix rate, with the error of an fp32 GEMM (tools/x3_probe.py: rms error vs fp64 0.85x that of the
// native fp32 kernel on the schedule's shapes).  This is NOT a bf16 GEMM: no input bit is discarded.
//
//   NT  C[M][N] = A[M][K] . Bt[N][K]^T   A = activations (fp32 in HBM, split in registers after the LDS read),
//                                        Bt = weights (small): split ONCE per call into three bf16 planes
//                                        [3][N][K] (workspace), which the kernel stages like bf16 operands.
//
// Staging, swizzles and the epilogue follow gemm_nt_kernel (gemm.hip); the K step is 32 values (128-byte fp32 rows of
// A, three 64-byte bf16 rows per weight row).
#include "conv_internal.h"
#include "gemm.h"

namespace epn {
namespace {
EPN_F2_SENTINEL_DECL

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const void *g, char *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((glb_void *)g, (lds_void *)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ void glds16_nt(const void *g, char *lds_wave_base) {      // non-temporal: a stream read once
    __builtin_amdgcn_global_load_lds((glb_void *)g, (lds_void *)lds_wave_base, 16, 0, 2);
}
__device__ __forceinline__ unsigned pack_rne(float a, float b) {   // v_cvt_pk_bf16_f32
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float lo_f(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float hi_f(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// eight fp32 values -> three bf16x8 fragments (4.5 VALU instructions per value)
__device__ __forceinline__ void split3(const f32x4 u, const f32x4 v, bf16x8 &h, bf16x8 &m, bf16x8 &l) {
    const float x[8] = {u[0], u[1], u[2], u[3], v[0], v[1], v[2], v[3]};
    u32x4 H, M, L;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const unsigned hp = pack_rne(x[2 * p], x[2 * p + 1]);
        const float r0 = x[2 * p] - lo_f(hp), r1 = x[2 * p + 1] - hi_f(hp);
        const unsigned mp = pack_rne(r0, r1);
        H[p] = hp; M[p] = mp; L[p] = pack_rne(r0 - lo_f(mp), r1 - hi_f(mp));
    }
    h = __builtin_bit_cast(bf16x8, H); m = __builtin_bit_cast(bf16x8, M); l = __builtin_bit_cast(bf16x8, L);
}

// weights -> planes[3][N][K] (bf16), two values per thread
__global__ void split_planes_kernel(const float *__restrict__ W, long long ldw, int N, int K, unsigned *__restrict__ planes) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // pair index
    const int k2 = K >> 1;
    if (i >= (long long)N * k2) return;
    const int n = (int)(i / k2), k = 2 * (int)(i % k2);
    const float x0 = W[n * ldw + k], x1 = W[n * ldw + k + 1];
    const unsigned hp = pack_rne(x0, x1);
    const float r0 = x0 - lo_f(hp), r1 = x1 - hi_f(hp);
    const unsigned mp = pack_rne(r0, r1);
    const size_t plane = (size_t)N * k2;
    planes[i] = hp;
    planes[plane + i] = mp;
    planes[2 * plane + i] = pack_rne(r0 - lo_f(mp), r1 - hi_f(mp));
}

// the weights of all problems of a grouped launch (the five blocks of a spectral IntraSO3Conv) in ONE launch: blockIdx.y =
// problem (98 -> 42 splitting launches per cls training step)
struct SplitBatch {
    const float *W[GEMM_MAX_PROB];
    unsigned *planes[GEMM_MAX_PROB];
    long long ldw[GEMM_MAX_PROB];
    int N[GEMM_MAX_PROB], K[GEMM_MAX_PROB];
};
__global__ void split_planes_batch_kernel(SplitBatch S) {
    const int q = blockIdx.y;
    const float *__restrict__ W = S.W[q];
    unsigned *__restrict__ planes = S.planes[q];
    const int N = S.N[q], k2 = S.K[q] >> 1;
    const size_t plane = (size_t)N * k2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (long long)N * k2; i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i / k2), k = 2 * (int)(i % k2);
        const float x0 = W[n * S.ldw[q] + k], x1 = W[n * S.ldw[q] + k + 1];
        const unsigned hp = pack_rne(x0, x1);
        const float r0 = x0 - lo_f(hp), r1 = x1 - hi_f(hp);
        const unsigned mp = pack_rne(r0, r1);
        planes[i] = hp;
        planes[plane + i] = mp;
        planes[2 * plane + i] = pack_rne(r0 - lo_f(mp), r1 - hi_f(mp));
    }
}

// two-piece fp16 form (gemm.h): the weights of every problem -> planes[2][N][K] (fp16), every ROW n scaled by the power of two
// that puts max|W[n][:]| at 2^14.  A weight row is one output column of the NT product, so its scale is undone per column in
// the GEMM's epilogue (rowmax[n] is what the kernel reads); the row's maximum is taken by the wave that splits it -- one launch
// per grouped GEMM where a tensor-wide scale needed a memset, a reduction pass and the split (three dependent launches of
// 6-9 us each per weight, 94 weights per classification step), and a row of small weights keeps its full 22 bits.
// Non-finite elements are left out of the maximum (they poison their own products only, as in an fp32 GEMM).
struct Split2Batch {
    const float *W[GEMM_MAX_PROB];
    unsigned *planes[GEMM_MAX_PROB];
    float *rowmax[GEMM_MAX_PROB];
    long long ldw[GEMM_MAX_PROB];
    int N[GEMM_MAX_PROB], K[GEMM_MAX_PROB];
};
__global__ __launch_bounds__(256) void split_rows2_batch_kernel(Split2Batch S) {
    const int q = blockIdx.y;
    const float *__restrict__ W = S.W[q];
    unsigned *__restrict__ planes = S.planes[q];
    const int N = S.N[q], k2 = S.K[q] >> 1;
    const size_t plane = (size_t)N * k2;
    const int lane = threadIdx.x & 63;
    for (int n = blockIdx.x * 4 + (threadIdx.x >> 6); n < N; n += gridDim.x * 4) {      // wave-uniform
        const float *__restrict__ row = W + (size_t)n * S.ldw[q];
        unsigned m = 0;
        for (int i = lane; i < k2; i += 64) {
            const float2 v = make_float2(row[2 * i], row[2 * i + 1]);
            const unsigned a = __builtin_bit_cast(unsigned, v.x) & 0x7fffffffu, b = __builtin_bit_cast(unsigned, v.y) & 0x7fffffffu;
            m = (a > m && a < 0x7f800000u) ? a : m;
            m = (b > m && b < 0x7f800000u) ? b : m;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const unsigned v = (unsigned)__shfl_xor((int)m, o, 64);
            m = v > m ? v : m;
        }
        const float amax = __builtin_bit_cast(float, m);
        const float sc = f2_scale_of(amax);
        if (lane == 0) S.rowmax[q][n] = amax;
        for (int i = lane; i < k2; i += 64) {
            const float2 v = make_float2(row[2 * i], row[2 * i + 1]);
            unsigned h, l;
            f2_split_pair(v.x, v.y, sc, h, l);
            planes[(size_t)n * k2 + i] = h;
            planes[plane + (size_t)n * k2 + i] = l;
        }
    }
}

// max |x| over a strided matrix -> *out (as the bit pattern of a non-negative float: unsigned order = float order).  Non-finite
// values are left out of the maximum: a NaN / inf element then poisons only the rows it belongs to (x 2^s stays NaN / inf in
// the split), as it would in an fp32 GEMM, instead of the scale of the whole tensor.  *out must be zero before the launch
// (launch_absmax).  A streaming read: four independent 16-byte loads per thread and iteration.
__device__ __forceinline__ unsigned absmax4(unsigned m, const u32x4 v) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned a = v[k] & 0x7fffffffu;
        m = (a > m && a < 0x7f800000u) ? a : m;
    }
    return m;
}
// one atomic per WORKGROUP, and only when it would raise the value: 8192 waves hammering one address made the pass over a 126 MB
// operand take 105 us (1.2 TB/s) -- the maximum is monotonic, so a stale read only costs an atomic that changes nothing
__device__ __forceinline__ void absmax_finish(unsigned m, unsigned *out) {      // blockDim.x == 256
    __shared__ unsigned red[4];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned v = (unsigned)__shfl_xor((int)m, o, 64);
        m = v > m ? v : m;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = red[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) m = red[w] > m ? red[w] : m;
        if (m > __atomic_load_n(out, __ATOMIC_RELAXED)) atomicMax(out, m);
    }
}
__global__ __launch_bounds__(256) void absmax_flat_kernel(const u32x4 *__restrict__ src, long long n4, unsigned *__restrict__ out) {
    unsigned m = 0;
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const u32x4 v0 = __builtin_nontemporal_load(src + i), v1 = __builtin_nontemporal_load(src + i + stride);
        const u32x4 v2 = __builtin_nontemporal_load(src + i + 2 * stride), v3 = __builtin_nontemporal_load(src + i + 3 * stride);
        m = absmax4(absmax4(absmax4(absmax4(m, v0), v1), v2), v3);
    }
    for (; i < n4; i += stride) m = absmax4(m, src[i]);
    absmax_finish(m, out);
}
__global__ __launch_bounds__(256) void absmax_kernel(const float *__restrict__ src, long long ld, long long rows, long long cols4,
                                                     unsigned *__restrict__ out) {
    unsigned m = 0;
    const long long n = rows * cols4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / cols4, c = i - r * cols4;
        m = absmax4(m, *reinterpret_cast<const u32x4 *>(src + r * ld + 4 * c));
    }
    absmax_finish(m, out);
}
__global__ void absmax_tail_kernel(const float *__restrict__ src, long long ld, long long rows, long long cols, long long c0,
                                   unsigned *__restrict__ out) {       // columns c0 .. cols of every row (cols % 4 != 0, or unaligned)
    const long long w = cols - c0, n = rows * w;
    unsigned m = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / w, c = c0 + (i - r * w);
        const unsigned a = __builtin_bit_cast(unsigned, src[r * ld + c]) & 0x7fffffffu;
        m = (a > m && a < 0x7f800000u) ? a : m;
    }
    if (m) atomicMax(out, m);
}

template <int WGM, int WGN, int TM, int TN, int NSTG, int NPL = 3>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_nt_x3_kernel(GemmNtBatch B) {
    constexpr int NW = WGM * WGN;
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    constexpr int A_BYTES = BM * 128;              // fp32 rows of the activation tile, 32 values per K step
    constexpr int P_BYTES = BN * 64;               // one bf16 plane of the weight tile
    constexpr int STAGE = A_BYTES + NPL * P_BYTES;  // NPL = 3: bf16 planes (lossless form); 2: fp16 planes (two-piece form)
    constexpr int NGA = BM / 8, NGB = NPL * BN / 16;  // 1 KiB wave-level load instructions per stage (8 / 16 rows each)
    constexpr int NTERM = NPL == 3 ? 6 : 3;         // matrix instructions per (A tile, B tile) and 16 contraction values
    constexpr int NG = NGA + NGB;
    constexpr int GPW = (NG + NW - 1) / NW;
    constexpr int INFLIGHT = NG % NW == 0 ? GPW : GPW - 1;   // loads of the youngest stage a wave may leave pending
    static_assert(NSTG * STAGE <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(1024))) char smem[NSTG * STAGE];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int pi = 0;
#pragma unroll
    for (int i = 1; i < GEMM_MAX_PROB; ++i)
        if (i < B.nprob && blockIdx.x >= B.p[i].tile0) pi = i;
    const GemmNtProb &P = B.p[pi];
    if (blockIdx.x - P.tile0 >= P.ntile) return;
    const unsigned t = epn_xcd_tile(blockIdx.x - P.tile0, P.ntile);
    const long long m0 = (long long)(t / P.tiles_n) * BM;
    const int n0 = (int)(t % P.tiles_n) * BN;
    const int nk = P.K / 32;

    const char *src[GPW];
    int adv[GPW];
#pragma unroll
    for (int i = 0; i < GPW; ++i) {
        const int g = wave + i * NW;
        if (g < NGA) {
            const int r = 8 * g + lane / 8;
            const int slot = (lane % 8) ^ ((r >> 1) & 7);
            long long gr = m0 + r;
            gr = gr < P.M ? gr : P.M - 1;
            src[i] = reinterpret_cast<const char *>(static_cast<const float *>(P.A) + gr * P.lda + slot * 4);
            adv[i] = 128;
        } else {
            const int rb = 16 * (g - NGA) + lane / 4;          // plane * BN + weight row of the tile
            const int plane = rb / BN, n = rb % BN;
            const int slot = (lane % 4) ^ ((rb >> 2) & 3);
            int gn = n0 + n;
            gn = gn < P.N ? gn : P.N - 1;
            src[i] = reinterpret_cast<const char *>(static_cast<const __bf16 *>(P.Bp) + ((size_t)plane * P.N + gn) * P.K + slot * 8);
            adv[i] = 64;
        }
    }
    auto stage_one = [&](int buf, int i) {
        const int g = wave + i * NW;
        if (NG % NW == 0 || g < NG) {
#ifdef EPN_X3_NTA
            if (g < NGA) glds16_nt(src[i], smem + buf * STAGE + g * 1024); else
#endif
            glds16(src[i], smem + buf * STAGE + g * 1024);
            src[i] += adv[i];
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < GPW; ++i) stage_one(buf, i);
    };
    // two-stage ring: the next stage's loads are spread over the twelve MFMA groups of the K step (a direct-to-LDS load
    // costs its wave ~100 cycles of issue; as one burst after the barrier both waves of a SIMD stall together)
    // (the 256 x 256 tile: +3 %; the smaller tiles measured 1-2 % slower spread than as a burst)
    constexpr bool SPREAD = NSTG == 2 && GPW <= 2 * NTERM && TM * TN >= 8;
    const int wpar = NW >= 8 ? __builtin_amdgcn_readfirstlane((wave >> 2) & 1) : 0;

    const int wm = wave / WGN, wn = wave % WGN;
    const int li = lane & 31, lj = lane >> 5;
    const int fswA = (li >> 1) & 7, fswB = (li >> 2) & 3;
    float a_scale = 1.0f;                           // two-piece form: 2^s of the activations (device scalar max|A|)
    if constexpr (NPL == 2) a_scale = f2_scale_of(*P.a_amax);
    int aoff[TM], boff[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) aoff[i] = ((wm * TM + i) * 32 + li) * 128;
#pragma unroll
    for (int i = 0; i < TN; ++i) boff[i] = A_BYTES + ((wn * TN + i) * 32 + li) * 64;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    stage(0);
    if constexpr (NSTG == 3) {
        if (nk > 1) stage(1);
    }
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if constexpr (NSTG == 3) {
            if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + 2 < nk) stage((kt + 2) % 3);
        } else {
            __syncthreads();
            if (more && !SPREAD) stage((kt + 1) & 1);
        }
        const char *base = smem + (kt % NSTG) * STAGE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {           // 16 contraction values per fragment step: lane group lj holds 8 of them
            const int sa0 = ((4 * s + 2 * lj) ^ fswA) * 16, sa1 = ((4 * s + 2 * lj + 1) ^ fswA) * 16;
            const int sb = ((2 * s + lj) ^ fswB) * 16;
#define EPN_X3_LOAD(T_)                                                                                    \
    if constexpr (SPREAD) {                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        if (more) {                                                                                         \
            _Pragma("unroll") for (int i = 0; i < GPW; ++i) {                                               \
                constexpr int LAST = 2 * NTERM - 1;                                                         \
                const int g0 = (i * LAST) / (GPW > 1 ? GPW - 1 : 1);    /* slot of load i, 0 .. LAST */      \
                const int g1 = g0 < LAST ? g0 + 1 : LAST;               /* partner wave: one slot later */   \
                if ((g0 == NTERM * s + (T_) && wpar == 0) || (g1 == NTERM * s + (T_) && wpar != 0))         \
                    stage_one((kt + 1) & 1, i);                                                             \
            }                                                                                               \
        }                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
    }
            if constexpr (NPL == 3) {
            bf16x8 ah[TM], am[TM], al[TM], bh[TN], bm[TN], bl[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[j] = *reinterpret_cast<const bf16x8 *>(base + boff[j] + sb);
                bm[j] = *reinterpret_cast<const bf16x8 *>(base + boff[j] + P_BYTES + sb);
                bl[j] = *reinterpret_cast<const bf16x8 *>(base + boff[j] + 2 * P_BYTES + sb);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
                split3(*reinterpret_cast<const f32x4 *>(base + aoff[i] + sa0),
                       *reinterpret_cast<const f32x4 *>(base + aoff[i] + sa1), ah[i], am[i], al[i]);
#define EPN_X3_TERM(PA, PB)                                                                                 \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)         \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PA[i], PB[j], acc[i][j], 0, 0, 0)
            EPN_X3_LOAD(0) EPN_X3_TERM(ah, bl);                // small terms first
            EPN_X3_LOAD(1) EPN_X3_TERM(al, bh);
            EPN_X3_LOAD(2) EPN_X3_TERM(am, bm);
            EPN_X3_LOAD(3) EPN_X3_TERM(ah, bm);
            EPN_X3_LOAD(4) EPN_X3_TERM(am, bh);
            EPN_X3_LOAD(5) EPN_X3_TERM(ah, bh);
#undef EPN_X3_TERM
            } else {
            gemm_f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[j] = *reinterpret_cast<const gemm_f16x8 *>(base + boff[j] + sb);
                bl[j] = *reinterpret_cast<const gemm_f16x8 *>(base + boff[j] + P_BYTES + sb);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const f32x4 u = *reinterpret_cast<const f32x4 *>(base + aoff[i] + sa0);
                const f32x4 v = *reinterpret_cast<const f32x4 *>(base + aoff[i] + sa1);
                const float x[8] = {u[0], u[1], u[2], u[3], v[0], v[1], v[2], v[3]};
                f2_split8(x, a_scale, ah[i], al[i]);
            }
#define EPN_F2_TERM(PA, PB)                                                                                 \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)         \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(PA[i], PB[j], acc[i][j], 0, 0, 0)
            EPN_X3_LOAD(0) EPN_F2_TERM(ah, bl);                // small terms first
            EPN_X3_LOAD(1) EPN_F2_TERM(al, bh);
            EPN_X3_LOAD(2) EPN_F2_TERM(ah, bh);
#undef EPN_F2_TERM
            }
#undef EPN_X3_LOAD
        }
    }

    // ---- epilogue: D[row = (r&3) + 8 (r>>2) + 4 lj][col = li]   (as gemm_nt_kernel)
    float *__restrict__ C = static_cast<float *>(P.C);
    if constexpr (NPL == 2) {                       // undo the operand scales: one exact power-of-two multiply per value
        const float ua = f2_inverse(a_scale);
        float chk = 0.0f;                           // NaN iff an accumulator of this lane is inf / NaN (gemm.h: EPN_F2_CHECK)
#pragma unroll
        for (int j = 0; j < TN; ++j) {              // the weights are scaled per row = per output column (split_rows2_batch_kernel)
            int n = n0 + (wn * TN + j) * 32 + li;
            n = n < P.N ? n : P.N - 1;
            const float u = ua * f2_inverse(f2_scale_of(P.b_amax[n]));     // 2^-a 2^-b: a, b in [-113, 14] -- exact down to 2^-126, flushed below (as the product would be)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    chk = fmaf(acc[i][j][r], 0.0f, chk);
                    acc[i][j][r] *= u;
                }
        }
        EPN_F2_CHECK(chk);
    }
    if (P.stats) nt_col_stats<TM, TN, float>(acc, P.stats, P.M, P.N, m0 + wm * TM * 32, n0 + wn * TN * 32, li, lj);
    if (P.c_amax) nt_c_amax<TM, TN, float>(acc, P.c_amax);
    // (round 5, measured and dropped: whole tiles leaving through LDS -- a wave's 32 x 32 TN accumulator block transposed in the
    // idle stage buffers and stored as 16-byte pieces of whole rows, 8x fewer store instructions: 2-7 % SLOWER on every
    // output-heavy shape (245760 x 6144 x 256: 2.88 -> 2.94 ms, 983040 x 1536 x 64: 1.25 -> 1.35).  The store tail of these tiles
    // is not bound by store issue; the direct form's 128-byte row segments are already whole cache lines.)
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
                for (int j = 0; j < TN; ++j) {
#ifdef EPN_X3_NTC
                    __builtin_nontemporal_store(acc[i][j][r], cw + o + j * 32);
#else
                    cw[o + j * 32] = acc[i][j][r];
#endif
                }
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

template <int WGM, int WGN, int TM, int TN, int NSTG, int NPL = 3>
int launch_x3_cfg(GemmNtBatch &B, hipStream_t st) {
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    for (int i = 1; i < B.nprob; ++i)           // longest contraction first (see launch_nt_cfg)
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
    EPN_LAUNCH((gemm_nt_x3_kernel<WGM, WGN, TM, TN, NSTG, NPL>), dim3(total), dim3(64 * WGM * WGN), 0, st, B);
    EPN_CHECK_LAUNCH();
    return 0;
}

inline size_t planes_bytes(const GemmNtProb &p) { return ((size_t)6 * p.N * p.K + 255) & ~(size_t)255; }

}  // namespace

long long f2_nonfinite_take_x3(bool reset) {
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

bool gemm_nt_x3_ok(const GemmNtBatch &B) {
    for (int i = 0; i < B.nprob; ++i) {
        const GemmNtProb &p = B.p[i];
        if (p.M < 1 || p.N < 1 || p.K < 32 || p.K % 32 || p.lda % 4 || ((uintptr_t)p.A & 15) || !p.A || !p.Bt || !p.C) return false;
        if (p.stats && p.M % 32) return false;     // (launch_gemm_nt reports the error)
    }
    return B.nprob >= 1 && B.nprob <= GEMM_MAX_PROB;
}

size_t gemm_nt_x3_workspace(const GemmNtBatch &B) {
    size_t n = 0;
    for (int i = 0; i < B.nprob; ++i) n += planes_bytes(B.p[i]);
    return n;
}

// two-piece form: a_amax slot per problem (256 B), then per problem the fp16 planes [2][N][K] and the row maxima [N]
static size_t f2_prob_bytes(long long N, long long K) {
    return (((size_t)4 * N * K + 255) & ~(size_t)255) + (((size_t)4 * N + 255) & ~(size_t)255);
}
size_t gemm_nt_f2_workspace(const GemmNtBatch &B) {
    size_t n = 256;
    for (int i = 0; i < B.nprob; ++i) n += f2_prob_bytes(B.p[i].N, B.p[i].K);
    return n;
}

namespace {
__global__ void scale_scalar_kernel(float *v, float f, unsigned tag) {
    *v *= f;
    if (tag) reinterpret_cast<unsigned *>(v)[1] = tag;
}
// A maximum that travels between two calls in a caller-owned buffer (the tail of `saved`, inter_split.hip) carries a tag word
// behind it.  A buffer without the tag -- written by a 0.2 forward pass, or never written -- must not be trusted as a scale:
// the guard zeroes the slot, and the scan below (which a tagged buffer skips in its first instruction) takes the maximum.
__global__ void amax_guard_kernel(unsigned *slot, unsigned tag) {
    if (slot[1] != tag) slot[0] = 0u;
}
__global__ __launch_bounds__(256) void absmax_untagged_kernel(const u32x4 *__restrict__ src, long long n4, unsigned *__restrict__ slot,
                                                              unsigned tag) {
    if (__atomic_load_n(slot + 1, __ATOMIC_RELAXED) == tag) return;
    unsigned m = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) m = absmax4(m, src[i]);
    absmax_finish(m, slot);
}
}  // namespace
int launch_scale_scalar(float *v, float factor, hipStream_t st, unsigned tag) {   // *v *= factor (a bound derived from a maximum); tag != 0: v[1] = tag
    if (!v) return EPN_ENULL;
    EPN_LAUNCH_AUX(scale_scalar_kernel, dim3(1), dim3(1), 0, st, v, factor, tag);
    EPN_CHECK_LAUNCH();
    return 0;
}
int launch_absmax_unless_tagged(const float *src, long long n, float *slot, unsigned tag, hipStream_t st) {
    if (!slot || !src) return EPN_ENULL;
    if (n % 4 || ((uintptr_t)src & 15)) return EPN_EINVAL;
    EPN_LAUNCH_AUX(amax_guard_kernel, dim3(1), dim3(1), 0, st, reinterpret_cast<unsigned *>(slot), tag);
    EPN_CHECK_LAUNCH();
    const long long n4 = n / 4, want = (n4 + 4 * 256 - 1) / (4 * 256);
    const unsigned g = (unsigned)(want < 1 ? 1 : (want < 2048 ? want : 2048));
    EPN_LAUNCH_AUX(absmax_untagged_kernel, dim3(g), dim3(256), 0, st, reinterpret_cast<const u32x4 *>(src), n4,
                   reinterpret_cast<unsigned *>(slot), tag);
    EPN_CHECK_LAUNCH();
    return 0;
}

int launch_absmax(const float *src, long long ld, long long rows, long long cols, float *out, hipStream_t st) {
    if (!out) return EPN_ENULL;
    EPN_HIP(hipMemsetAsync(out, 0, sizeof(float), st));
    if (rows < 1 || cols < 1) return 0;
    if (!src) return EPN_ENULL;
    const bool vec = !((uintptr_t)src & 15) && ld % 4 == 0;
    if (vec && (ld == cols || rows == 1) && (rows * cols) % 4 == 0) {       // contiguous: one flat stream
        const long long n4 = rows * cols / 4;
        const long long want = (n4 + 4 * 256 - 1) / (4 * 256);
        const unsigned g = (unsigned)(want < 1 ? 1 : (want < 2048 ? want : 2048));
        EPN_LAUNCH_AUX(absmax_flat_kernel, dim3(g), dim3(256), 0, st, reinterpret_cast<const u32x4 *>(src), n4,
                       reinterpret_cast<unsigned *>(out));
        EPN_CHECK_LAUNCH();
        return 0;
    }
    const long long c4 = vec ? cols / 4 : 0;
    if (c4 > 0) {
        const long long n = rows * c4;
        const unsigned g = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
        EPN_LAUNCH_AUX(absmax_kernel, dim3(g), dim3(256), 0, st, src, ld, rows, c4, reinterpret_cast<unsigned *>(out));
        EPN_CHECK_LAUNCH();
    }
    if (4 * c4 < cols) {
        const long long n = rows * (cols - 4 * c4);
        const unsigned g = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
        EPN_LAUNCH_AUX(absmax_tail_kernel, dim3(g), dim3(256), 0, st, src, ld, rows, cols, 4 * c4, reinterpret_cast<unsigned *>(out));
        EPN_CHECK_LAUNCH();
    }
    return 0;
}

template <int NPL>
static int x3_dispatch(GemmNtBatch &B, int maxn, int minn, hipStream_t st) {
    const int pol = kernel_policy();
    if ((pol & ~0xff) == 0x100) {               // tuning override (tools/x3_probe.py)
        switch (pol & 0xff) {
            case 0x21: return launch_x3_cfg<4, 2, 2, 2, 2, NPL>(B, st);     // 256 x 128, 8 waves
            case 0x22: return launch_x3_cfg<4, 2, 2, 4, 2, NPL>(B, st);     // 256 x 256, 8 waves (all of the LDS in the bf16 form)
            case 0x23: return launch_x3_cfg<2, 2, 2, 2, 2, NPL>(B, st);     // 128 x 128, 4 waves
            case 0x24: return launch_x3_cfg<2, 2, 2, 2, 3, NPL>(B, st);
            case 0x25: return launch_x3_cfg<4, 1, 2, 2, 3, NPL>(B, st);     // 256 x 64, 4 waves
            case 0x26: return launch_x3_cfg<4, 1, 2, 2, 2, NPL>(B, st);
            case 0x27: return launch_x3_cfg<2, 2, 4, 2, 2, NPL>(B, st);     // 256 x 128, 4 waves (128 x 64 per wave)
            case 0x28: return launch_x3_cfg<2, 4, 2, 2, 2, NPL>(B, st);     // 128 x 256, 8 waves
            case 0x29: return launch_x3_cfg<2, 2, 2, 4, 2, NPL>(B, st);     // 128 x 256, 4 waves
            case 0x2a: return launch_x3_cfg<8, 1, 2, 1, 2, NPL>(B, st);     // 512 x 32
            case 0x2b: if constexpr (NPL == 2) return launch_x3_cfg<4, 2, 2, 2, 3, NPL>(B, st); else break;   // 256 x 128, three stages (144 KB)
            default: break;
        }
    }
    // 256 x 64 tiles: three stages (120-132 KB: one workgroup per CU) in the three-piece form; the two-piece form's stage is
    // 40 KB, and TWO stages leave room for two workgroups per CU -- what the short contractions of these narrow problems want
    // (c = 64 spectral groups 0.214 -> 0.171 ms, 983040 x 64 x 64 0.120 -> 0.105, 491520 x 64 x 128 0.099 -> 0.089; the long
    // K = 1536 forward GEMM of the cout = 64 layer is unchanged at 1.12 ms: tools/spectral_nt_probe.py)
    constexpr int NSTG64 = NPL == 2 ? 2 : 3;
    if (B.nprob > 1) {                          // grouped spectral blocks: ragged widths, narrow tiles (see launch_nt_typed)
        if (maxn <= 320 && minn <= 64) return launch_x3_cfg<4, 1, 2, 2, NSTG64, NPL>(B, st);
        if (minn >= 256) return launch_x3_cfg<4, 2, 2, 4, 2, NPL>(B, st);   // c = 256 blocks: 0.76 -> 0.68 ms (A is re-read per tile column)
        return launch_x3_cfg<2, 2, 2, 2, 2, NPL>(B, st);
    }
    if (maxn <= 32) return launch_x3_cfg<8, 1, 2, 1, 2, NPL>(B, st);
    if (maxn <= 64) return launch_x3_cfg<4, 1, 2, 2, NSTG64, NPL>(B, st);
    // cout <= 128 in the two-piece form: 128 x 128 tiles of four waves (64 KB of stages: two workgroups per CU) edge out the
    // 256 x 128 tile (491520 x 128 x 1536: 0.857 -> 0.819 ms cold, x 3072: 1.617 -> 1.573, x 128: 0.125 -> 0.116; tools/nt_tile_probe.py)
    if (NPL == 2 && maxn <= 128) return launch_x3_cfg<2, 2, 2, 2, 2, NPL>(B, st);
    if (maxn <= 128 || maxn % 256 > 128 || (maxn % 256 && maxn < 512)) return launch_x3_cfg<4, 2, 2, 2, 2, NPL>(B, st);
    return launch_x3_cfg<4, 2, 2, 4, 2, NPL>(B, st);
}

// npl = 3: lossless three-piece bf16 form (planes of the weights in `ws`); npl = 2: two-piece fp16 form -- `ws` starts with
// the amax slots ([2 i] = max|A_i| when the caller gave none in p.a_amax, [2 i + 1] = max|Bt_i|), then the fp16 planes
int launch_gemm_nt_x3(GemmNtBatch &B, void *ws, size_t ws_bytes, hipStream_t st, int npl) {
    const size_t need = npl == 2 ? gemm_nt_f2_workspace(B) : gemm_nt_x3_workspace(B);
    if (!gemm_nt_x3_ok(B) || !ws || ((uintptr_t)ws & 15) || ws_bytes < need)
        return launch_gemm_nt(B, 0, 0, st);      // the planes are read with 16-byte direct-to-LDS loads
    char *w = static_cast<char *>(ws);
    int maxn = 0, minn = 1 << 30;
    long long maxpairs = 0;
    for (int i = 0; i < B.nprob; ++i) {
        const GemmNtProb &p = B.p[i];
        const long long pairs = (long long)p.N * (p.K / 2);
        maxpairs = pairs > maxpairs ? pairs : maxpairs;
        maxn = p.N > maxn ? p.N : maxn;
        minn = p.N < minn ? p.N : minn;
    }
    const unsigned gx = (unsigned)((maxpairs + 255) / 256 < 4096 ? (maxpairs + 255) / 256 : 4096);
    if (npl == 2) {
        float *slots = reinterpret_cast<float *>(w);
        w += 256;
        Split2Batch S;
        int rows_max = 0;
        for (int i = 0; i < B.nprob; ++i) {
            GemmNtProb &p = B.p[i];
            if (!p.a_amax) {                    // nobody knows max|A|: one pass over it (callers with a producer-side maximum skip this)
                int rc = launch_absmax(static_cast<const float *>(p.A), p.lda, p.M, p.K, slots + i, st);
                if (rc) return rc;
                p.a_amax = slots + i;
            }
            S.W[i] = static_cast<const float *>(p.Bt); S.planes[i] = reinterpret_cast<unsigned *>(w); S.ldw[i] = p.ldb;
            S.N[i] = p.N; S.K[i] = p.K;
            p.Bp = w;
            w += ((size_t)4 * p.N * p.K + 255) & ~(size_t)255;
            S.rowmax[i] = reinterpret_cast<float *>(w);
            p.b_amax = S.rowmax[i];
            w += ((size_t)4 * p.N + 255) & ~(size_t)255;
            rows_max = p.N > rows_max ? p.N : rows_max;
        }
        // a wave per weight row
        const unsigned gr = (unsigned)((rows_max + 3) / 4 < 2048 ? (rows_max + 3) / 4 : 2048);
        EPN_LAUNCH_AUX(split_rows2_batch_kernel, dim3(gr, B.nprob), dim3(256), 0, st, S);
        EPN_CHECK_LAUNCH();
        return x3_dispatch<2>(B, maxn, minn, st);
    }
    SplitBatch S;
    for (int i = 0; i < B.nprob; ++i) {
        GemmNtProb &p = B.p[i];
        S.W[i] = static_cast<const float *>(p.Bt); S.planes[i] = reinterpret_cast<unsigned *>(w); S.ldw[i] = p.ldb;
        S.N[i] = p.N; S.K[i] = p.K;
        p.Bp = w;
        w += planes_bytes(p);
    }
    if (B.nprob == 1) {
        EPN_LAUNCH_AUX(split_planes_kernel, dim3((unsigned)((maxpairs + 255) / 256)), dim3(256), 0, st, S.W[0], S.ldw[0], S.N[0],
                       S.K[0], S.planes[0]);
    } else {
        EPN_LAUNCH_AUX(split_planes_batch_kernel, dim3(gx, B.nprob), dim3(256), 0, st, S);
    }
    EPN_CHECK_LAUNCH();
    return x3_dispatch<3>(B, maxn, minn, st);
}

}  // namespace epn

using namespace epn;

extern "C" size_t epn_gemm_nt_split_workspace_bytes(int nprob, const epn_gemm_nt_problem *probs) {
    if (!probs || nprob < 1) return 0;
    size_t n = 0;
    for (int i = 0; i < nprob; ++i) n += (((size_t)6 * probs[i].N * probs[i].K + 255) & ~(size_t)255);
    return n;
}

extern "C" int epn_gemm_nt_split_f32(int nprob, const epn_gemm_nt_problem *probs, void *workspace, size_t workspace_bytes,
                                     epn_stream_t stream) {
    if (!probs) return EPN_ENULL;
    if (nprob < 1) return EPN_EINVAL;
    hipStream_t st = epn_stream(stream);
    char *w = static_cast<char *>(workspace);
    size_t left = workspace ? workspace_bytes : 0;
    for (int i0 = 0; i0 < nprob; i0 += GEMM_MAX_PROB) {
        GemmNtBatch B;
        B.nprob = nprob - i0 < GEMM_MAX_PROB ? nprob - i0 : GEMM_MAX_PROB;
        for (int i = 0; i < B.nprob; ++i) {
            const epn_gemm_nt_problem &q = probs[i0 + i];
            GemmNtProb &p = B.p[i];
            p.A = q.A; p.Bt = q.Bt; p.C = q.C; p.M = q.M; p.N = q.N; p.K = q.K; p.lda = q.lda; p.ldb = q.ldb; p.ldc = q.ldc;
            p.tiles_n = 0; p.tile0 = 0; p.ntile = 0; p.Bp = nullptr; p.stats = q.col_stats; p.c_amax = reinterpret_cast<unsigned *>(q.c_amax);
            p.a_amax = p.b_amax = nullptr;
        }
        const size_t need = gemm_nt_x3_workspace(B);
        int rc = launch_gemm_nt_x3(B, need <= left ? w : nullptr, need <= left ? need : 0, st);
        if (rc) return rc;
        if (need <= left) { w += need; left -= need; }
    }
    return 0;
}

extern "C" int epn_absmax_f32(const float *src, long long ld, long long rows, long long cols, float *out, epn_stream_t stream) {
    if (rows < 0 || cols < 0 || (rows > 1 && ld < cols)) return EPN_EINVAL;
    return launch_absmax(src, ld, rows, cols, out, epn_stream(stream));
}

extern "C" long long epn_f16x2_overflow_count(int reset) {
    const long long a = f2_nonfinite_take_gemm(reset != 0);
    if (a < 0) return a;
    const long long b = f2_nonfinite_take_x3(reset != 0);
    if (b < 0) return b;
    const long long c = f2_nonfinite_take_bwd(reset != 0);
    return c < 0 ? c : a + b + c;
}

extern "C" size_t epn_gemm_nt_f16x2_workspace_bytes(int nprob, const epn_gemm_nt_problem *probs) {
    if (!probs || nprob < 1) return 0;
    size_t n = 0;
    for (int i0 = 0; i0 < nprob; i0 += GEMM_MAX_PROB) {
        n += 256;
        for (int i = i0; i < nprob && i < i0 + GEMM_MAX_PROB; ++i) n += f2_prob_bytes(probs[i].N, probs[i].K);
    }
    return n;
}

extern "C" int epn_gemm_nt_f16x2_f32(int nprob, const epn_gemm_nt_problem *probs, const float *const *a_amax, void *workspace,
                                     size_t workspace_bytes, epn_stream_t stream) {
    if (!probs) return EPN_ENULL;
    if (nprob < 1) return EPN_EINVAL;
    hipStream_t st = epn_stream(stream);
    char *w = static_cast<char *>(workspace);
    size_t left = workspace ? workspace_bytes : 0;
    for (int i0 = 0; i0 < nprob; i0 += GEMM_MAX_PROB) {
        GemmNtBatch B;
        B.nprob = nprob - i0 < GEMM_MAX_PROB ? nprob - i0 : GEMM_MAX_PROB;
        for (int i = 0; i < B.nprob; ++i) {
            const epn_gemm_nt_problem &q = probs[i0 + i];
            GemmNtProb &p = B.p[i];
            p.A = q.A; p.Bt = q.Bt; p.C = q.C; p.M = q.M; p.N = q.N; p.K = q.K; p.lda = q.lda; p.ldb = q.ldb; p.ldc = q.ldc;
            p.tiles_n = 0; p.tile0 = 0; p.ntile = 0; p.Bp = nullptr; p.stats = q.col_stats; p.c_amax = reinterpret_cast<unsigned *>(q.c_amax);
            p.a_amax = a_amax ? a_amax[i0 + i] : nullptr; p.b_amax = nullptr;
        }
        const size_t need = gemm_nt_f2_workspace(B);
        if (need > left) return EPN_EWORKSPACE;
        int rc = launch_gemm_nt_x3(B, w, need, st, 2);
        if (rc) return rc;
        w += need; left -= need;
    }
    return 0;
}
