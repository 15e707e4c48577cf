// Internal interface of gemm.hip (the BasicSO3Conv weight contractions as MFMA GEMMs).
#pragma once
#include "epn_common.h"

namespace epn {

constexpr int GEMM_MAX_PROB = 6;   // problems per grouped NT launch (the five irreducible blocks of IntraSO3Conv + 1)

struct GemmNtProb {        // C[M][N] = A[M][K] . Bt[N][K]^T
    const void *A, *Bt;
    void *C;
    long long M, lda, ldb, ldc;
    int N, K;
    int tiles_n;           // filled by the launcher
    unsigned tile0;        // first workgroup of this problem inside the grouped launch (a multiple of 8)
    unsigned ntile;        // its tiles; workgroups tile0 + ntile .. next tile0 are padding
    const void *Bp;        // split GEMM (gemm_x3.hip): the three bf16 planes [3][N][K] of Bt, filled by its launcher
    float *stats;          // optional: per-column (sum, sum of squares) of every 32-row block of C, [M/32][N][2] (M % 32 == 0)
    const float *a_amax, *b_amax;   // two-piece fp16 form (gemm_x3.hip, NPL = 2): device scalar max|A|; b_amax[n] = max|Bt[n][:]| per weight row (filled by the launcher) -> power-of-two scales
    unsigned *c_amax;      // optional: device scalar the kernel raises to max|C| (as stored; bit pattern of a non-negative float, atomicMax); the caller zeroes it
};
struct GemmNtBatch {
    int nprob;
    unsigned ntiles;
    GemmNtProb p[GEMM_MAX_PROB];
};

struct GemmTnArgs {        // C[N1][N2] = X[R][N1]^T . Y[R][N2]
    const void *X, *Y;
    void *C, *part;        // part: split partials [nsplit][N1][N2] fp32 (workspace)
    size_t part_bytes;
    long long R, ldx, ldy, ldc;
    int N1, N2;
    unsigned ntiles;
    int tiles_n2, nsplit;
    unsigned block0;       // first workgroup of this problem inside a grouped launch
    const void *Xp;        // split form: the three bf16 planes [3][R/8][N1][8] of X (workspace), set by the plan
    const float *x_amax, *y_amax;   // two-piece fp16 form: device scalars max|X|, max|Y|
};
struct GemmTnBatch {       // up to GEMM_MAX_PROB problems in ONE launch (the irreducible blocks of a spectral IntraSO3Conv)
    int nprob;
    unsigned nblocks;
    GemmTnArgs p[GEMM_MAX_PROB];
};

#ifdef __HIPCC__
// ---- two-piece fp16 split ("f16x2", round 5): x 2^s = h + l with h = rne_f16(x 2^s), l = rne_f16(x 2^s - h) keeps 22-23
// significant bits of every element within 2^-17 of the tensor's largest magnitude (absolute error <= max|x| 2^-39 below
// that); a product is hh + hl + lh on v_mfma_f32_32x32x16_f16 -- THREE matrix instructions per 32 x 32 x 16 block where the
// lossless bf16 form needs six.  The power-of-two scale 2^s puts max|x| into [2^14, 2^15) (fp16 overflows at 65504); it comes
// from a device scalar holding max|x| (the producer's epilogue or epn_absmax_f32), so nothing is synchronised with the host.
// Measured against fp64 (tools/pp2_probe.py, profiles/r05_f16x2_probe.txt): rms error 6.1e-7 at K = 3072 where the native fp32
// MFMA kernel has 9.9e-7 and the lossless 3 x bf16 kernel 8.6e-7 -- the error of an fp32 GEMM is its accumulation, not its inputs.
typedef _Float16 gemm_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 gemm_f16x2 __attribute__((ext_vector_type(2)));
typedef float gemm_f32x2 __attribute__((ext_vector_type(2)));
typedef float gemm_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned gemm_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float f2_scale_of(float amax) {          // power of two s with amax * s in [2^14, 2^15)
    unsigned e = (__builtin_bit_cast(unsigned, amax) >> 23) & 255u;
    e = e < 14u ? 14u : e;                                          // (zero / tiny tensors: the largest finite scale)
    return __builtin_bit_cast(float, (268u - e) << 23);
}
__device__ __forceinline__ float f2_inverse(float pow2) {            // 1 / s for a power of two
    return __builtin_bit_cast(float, (254u - (__builtin_bit_cast(unsigned, pow2) >> 23)) << 23);
}
__device__ __forceinline__ void f2_split_pair(float x0, float x1, float s, unsigned &h, unsigned &l) {
    const gemm_f32x2 x = {x0 * s, x1 * s};
    const gemm_f16x2 hp = __builtin_convertvector(x, gemm_f16x2);
    const gemm_f32x2 hf = __builtin_convertvector(hp, gemm_f32x2);
    const gemm_f32x2 r = {x[0] - hf[0], x[1] - hf[1]};
    h = __builtin_bit_cast(unsigned, hp);
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(r, gemm_f16x2));
}
__device__ __forceinline__ void f2_split8(const float (&x)[8], float s, gemm_f16x8 &h, gemm_f16x8 &l) {
    gemm_u32x4 H, L;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        unsigned hp, lp;
        f2_split_pair(x[2 * p], x[2 * p + 1], s, hp, lp);
        H[p] = hp; L[p] = lp;
    }
    h = __builtin_bit_cast(gemm_f16x8, H);
    l = __builtin_bit_cast(gemm_f16x8, L);
}

// ---- the scale contract made loud (round 6).  The scale leaves a factor 2-4 below fp16's 65504: an operand element above
// 2-4 x the REPORTED maximum becomes inf in the split, the products inf / NaN, and the result is silently non-finite.  Every
// two-piece kernel therefore folds its accumulators into one value per lane in the epilogue (`chk = fma(acc, 0, chk)`: NaN iff
// an accumulator was inf or NaN -- one VALU instruction per accumulator and TILE, nothing per K step) and a wave with a
// non-finite accumulator bumps a sticky device counter (one atomic per wave, only then).  With finite operands a non-zero
// count can only be a violated scale contract; epn_f16x2_overflow_count() reads (and optionally clears) the sum over the
// library's two GEMM translation units.  Each TU owns its counter (no relocatable device code in this build).
#define EPN_F2_SENTINEL_DECL __device__ unsigned g_f2_nonfinite = 0u;
#define EPN_F2_CHECK(chk_)                                                                       \
    do {                                                                                         \
        const float c__ = (chk_);                                                                \
        if (__builtin_amdgcn_ballot_w64(c__ != c__) != 0ull && (threadIdx.x & 63) == 0)          \
            atomicAdd(&g_f2_nonfinite, 1u);                                                      \
    } while (0)

// Column statistics of an NT tile, taken from the accumulators in the epilogue (the per-channel sums a following
// BatchNorm / InstanceNorm needs: SURVEY 8f.1 -- no separate pass over C).  acc[i][j]: 32 x 32 MFMA tile i (rows) x j
// (columns) of the wave, D[row = (r&3) + 8 (r>>2) + 4 lj][col = li]; part[(row / 32)][n][2].  The sums are those of the
// values as STORED (rounded to bf16 first when C is bf16): what a statistics pass over C would read.
typedef float gemm_f32x16 __attribute__((ext_vector_type(16)));
template <int TM, int TN, typename TO>
__device__ __forceinline__ void nt_col_stats(const gemm_f32x16 (&acc)[TM][TN], float *__restrict__ part, long long M, int N,
                                             long long row0, int col0, int li, int lj) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const long long rb = row0 + 32 * i;
        if (rb >= M) continue;                  // wave-uniform; M % 32 == 0: a 32-row block is valid or not as a whole
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[i][j][r];
                if constexpr (sizeof(TO) == 2) v = (float)(__bf16)v;
                s1 += v;
                s2 = fmaf(v, v, s2);
            }
            s1 += __shfl_xor(s1, 32, 64);       // the two 16-row halves of the tile (lane groups lj = 0, 1)
            s2 += __shfl_xor(s2, 32, 64);
            const int n = col0 + 32 * j + li;
            if (lj == 0 && n < N) {
                float2 *o = reinterpret_cast<float2 *>(part + (((size_t)(rb >> 5)) * N + n) * 2);
                *o = make_float2(s1, s2);
            }
        }
    }
}
#endif

// max |C| of an NT tile from its accumulators into the device scalar *out (epilogue by-product for a consumer that needs the
// range of C -- the cloud-resident transpose of the grouping, inter_ungroup_cloud.hip -- without a pass over C).  Values as
// stored (rounded to bf16 first when C is bf16); non-finite ones are left out, as in launch_absmax.  Rows / columns beyond M / N
// hold copies of the last valid ones (the loaders clamp), so no masking is needed.
#ifdef __HIPCC__
template <int TM, int TN, typename TO>
__device__ __forceinline__ void nt_c_amax(const gemm_f32x16 (&acc)[TM][TN], unsigned *__restrict__ out) {
    // one v_max3_f32 per two accumulators (|x| is an operand modifier; maxNum drops NaN): the first version -- integer compares
    // with the non-finite test per element, four instructions per accumulator -- cost the short-K data-gradient GEMMs 6-18 %
    float mf = 0.0f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; r += 2) mf = fmaxf(fmaxf(mf, fabsf(acc[i][j][r])), fabsf(acc[i][j][r + 1]));
    if constexpr (sizeof(TO) == 2) mf = (float)(__bf16)mf;      // rounding is monotonic: max of the rounded values = the rounded max
    unsigned m = __builtin_bit_cast(unsigned, mf);
    m = m < 0x7f800000u ? m : 0u;                               // (an infinite element: left out, as launch_absmax does; its consumer's own check reports it)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned v = (unsigned)__shfl_xor((int)m, o, 64);
        m = v > m ? v : m;
    }
    if ((threadIdx.x & 63) == 0 && m > __atomic_load_n(out, __ATOMIC_RELAXED)) atomicMax(out, m);
}
#endif

// f16x2 sentinel: sticky per-device count of waves that ended a two-piece GEMM tile with a non-finite accumulator, per TU
// (gemm.hip, gemm_x3.hip); take = read (synchronously, on the current device) and optionally clear.  < 0: a hipError_t, negated
long long f2_nonfinite_take_gemm(bool reset);
long long f2_nonfinite_take_x3(bool reset);
long long f2_nonfinite_take_bwd(bool reset);    // inter_bwd_f2.hip

int kernel_policy();   // c_api.hip: epn_set_kernel_policy (0x100 | cfg = NT tile override of the tuning tool)

// dtype / out_dtype: 0 = fp32, 1 = bf16
int launch_gemm_nt(GemmNtBatch &B, int dtype, int out_dtype, hipStream_t st);
// fp32 operands on the bf16 matrix pipe at fp32 accuracy (gemm_x3.hip); falls back to launch_gemm_nt when a problem does
// not qualify (K % 32, alignment) or the workspace is missing
bool gemm_nt_x3_ok(const GemmNtBatch &B);
size_t gemm_nt_x3_workspace(const GemmNtBatch &B);
int launch_gemm_nt_x3(GemmNtBatch &B, void *ws, size_t ws_bytes, hipStream_t st, int npl = 3);   // npl 2: two-piece fp16 form
size_t gemm_nt_f2_workspace(const GemmNtBatch &B);
// max |x| of a strided matrix into a device scalar (memset + one pass)
int launch_absmax(const float *src, long long ld, long long rows, long long cols, float *out, hipStream_t st);
int launch_scale_scalar(float *v, float factor, hipStream_t st, unsigned tag = 0);   // tag != 0: also v[1] = tag (bit pattern)
// slot[0] = max|src[0..n)| UNLESS slot[1] already holds `tag` (then slot[0] is trusted as it is): two launches, the scan
// returns at once for a tagged slot.  n % 4 == 0, src 16-byte aligned.
int launch_absmax_unless_tagged(const float *src, long long n, float *slot, unsigned tag, hipStream_t st);
int launch_gemm_tn(GemmTnArgs &G, int dtype, hipStream_t st);
// grouped: plans tiles / splits for all problems (balanced K steps per workgroup), carves `ws` into the partial slabs
int launch_gemm_tn_batch(GemmTnBatch &B, int dtype, void *ws, size_t ws_bytes, hipStream_t st);
size_t gemm_tn_batch_workspace(GemmTnBatch &B, int dtype);
void gemm_tn_tile(int dtype, int N1, int N2, int *bn1, int *bn2);   // dtype: 0 fp32, 1 bf16, 2 fp32 operands, split form (3 x bf16), 3 = two-piece fp16 form
int gemm_tn_splits(int dtype, long long R, int N1, int N2);
int launch_transpose_cast(const void *src, void *dst, int rows, int cols, int src_bf16, int dst_bf16, hipStream_t st);
int launch_cast(const void *src, void *dst, size_t n, int src_bf16, int dst_bf16, hipStream_t st);
int launch_cast_add(const float *src, const void *add_bf16, void *dst_bf16, size_t n, hipStream_t st);

}  // namespace epn
