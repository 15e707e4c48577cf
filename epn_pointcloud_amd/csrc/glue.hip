// Block glue of SeparableSO3ConvBlock on channels-last tensors: norm + leaky_relu (+ residual),
// forward and backward, as HBM-bound streaming kernels.  x_cl is [groups][rows][c]; every thread owns 4 consecutive
// channels (one 16-byte access per row) and walks rows with a fixed stride, so all global accesses are full-width
// and coalesced and the per-channel reductions stay in registers until one LDS hop + one atomic per block.
#include "conv_internal.h"

namespace epn {
namespace {

constexpr int GT = 256;   // threads per block

// Activation tensors (x, dy, residual, y / dx) are float or __bf16 (template parameter T of the streaming kernels);
// statistics, affine parameters and all arithmetic are fp32.
struct NormArgs {
    const void *x, *dy, *res;
    void *y;
    const float *sums, *dsums, *gamma, *beta;
    float *out_sums, *dgamma, *dbeta;
    long long rows;
    int c, rows_per_block;
    float eps, slope, inv_rows;
    unsigned *amax;       // optional (fp32): max |stored output| of the launch, atomicMax'ed as the bits of a non-negative float
};

// running max |v| of a lane as the bit pattern of a non-negative float (unsigned order = float order); non-finite values are
// left out (gemm_x3.hip absmax4).  The producer-side maxima of the two-piece fp16 GEMMs' operands (gemm.h).
__device__ __forceinline__ unsigned amax_acc4(unsigned m, const f32x4 v) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        // (the element goes through a scalar first: __builtin_bit_cast applied to the vector subscript v[i] itself compiled to
        // element 0 for every i with this toolchain -- the maxima were those of every fourth value, up to 1.7x low on
        // gradients, and an operand more than 2x above its reported maximum overflows fp16 in the two-piece GEMMs: the rare
        // non-finite step tests/test_gpu_bf16.py::test_producer_side_maxima_are_the_maxima now guards against)
        const float f = v[i];
        const unsigned a = __float_as_uint(f) & 0x7fffffffu;
        m = (a > m && a < 0x7f800000u) ? a : m;
    }
    return m;
}
__device__ __forceinline__ void amax_commit(unsigned m, unsigned *out) {      // one atomic per wave, only when it raises the value
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned v = (unsigned)__shfl_xor((int)m, o, 64);
        m = v > m ? v : m;
    }
    if ((threadIdx.x & 63) == 0 && m > __builtin_nontemporal_load(out)) atomicMax(out, m);
}

typedef __bf16 gbf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 ld4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ f32x4 ld4(const __bf16 *p) {
    const gbf16x4 v = *reinterpret_cast<const gbf16x4 *>(p);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
__device__ __forceinline__ void st4(float *p, f32x4 v) { *reinterpret_cast<f32x4 *>(p) = v; }
__device__ __forceinline__ void st4(__bf16 *p, f32x4 v) {
    *reinterpret_cast<gbf16x4 *>(p) = gbf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
}

__device__ __forceinline__ void load_param4(const float *p, int c4, f32x4 &v, float dflt) {
    v = p ? *reinterpret_cast<const f32x4 *>(p + c4) : f32x4{dflt, dflt, dflt, dflt};
}

// mean / rstd of this thread's 4 channels from sums[g][c][2]
__device__ __forceinline__ void stats4(const NormArgs &A, int g, int c4, f32x4 &mean, f32x4 &rstd) {
    const float *s = A.sums + ((size_t)g * A.c + c4) * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float m = s[2 * i] * A.inv_rows;
        const float var = fmaxf(s[2 * i + 1] * A.inv_rows - m * m, 0.0f);
        mean[i] = m;
        rstd[i] = rsqrtf(var + A.eps);
    }
}

// block = (C/4 channel lanes) x (GT / (C/4) row lanes);  grid = (row blocks, groups)
template <typename T>
__global__ __launch_bounds__(GT) void chan_stats_kernel(NormArgs A) {
    __shared__ float red[GT][8];
    const int lanes = A.c >> 2, cl = threadIdx.x % lanes, rl = threadIdx.x / lanes, rstep = GT / lanes;
    const int g = blockIdx.y;
    const long long r0 = (long long)blockIdx.x * A.rows_per_block;
    long long r1 = r0 + A.rows_per_block;
    r1 = r1 < A.rows ? r1 : A.rows;
    const T *x = static_cast<const T *>(A.x) + ((size_t)g * A.rows) * A.c + 4 * cl;
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (long long r = r0 + rl; r < r1; r += rstep) {
        const f32x4 v = ld4(x + r * A.c);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s1[i] += v[i];
            s2[i] += v[i] * v[i];
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        red[threadIdx.x][i] = s1[i];
        red[threadIdx.x][4 + i] = s2[i];
    }
    __syncthreads();
    if (rl == 0) {
        for (int t = 1; t < rstep; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s1[i] += red[t * lanes + cl][i];
                s2[i] += red[t * lanes + cl][4 + i];
            }
        // per-block partial (no atomics: 1024 blocks hammering [groups][c] addresses cost more than the streaming)
        float *o = A.out_sums + (((size_t)g * gridDim.x + blockIdx.x) * A.c + 4 * cl) * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[2 * i] = s1[i];
            o[2 * i + 1] = s2[i];
        }
    }
}

// Finishing kernels: 256 threads = 16 entries x 16 slices of the block-partial axis, LDS tree over the slices.
// sums[g][e] = sum_b part[g][b][e],  e in [0, 2c)
__global__ __launch_bounds__(256) void stats_finish_kernel(const float *__restrict__ part, int nb, int c2,
                                                           float *__restrict__ sums) {
    __shared__ float red[16][17];
    const int el = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int g = blockIdx.y, e = blockIdx.x * 16 + el;
    float acc = 0.f;
    if (e < c2) {
        // 8 independent loads in flight per thread (a dependent chain of nb/16 strided loads cost ~30 us per launch)
        const float *p = part + (size_t)g * nb * c2 + e;
        float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int b = sl;
        for (; b + 112 < nb; b += 128)
#pragma unroll
            for (int u = 0; u < 8; ++u) a8[u] += p[(size_t)(b + 16 * u) * c2];
        for (; b < nb; b += 16) a8[0] += p[(size_t)b * c2];
        acc = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
    }
    red[sl][el] = acc;
    __syncthreads();
    if (sl == 0 && e < c2) {
        for (int t = 1; t < 16; ++t) acc += red[t][el];
        sums[(size_t)g * c2 + e] = acc;
    }
}

// First level for long partial lists: part2[g][z][e] = sum of blocks [256 z, 256 z + 256) of part[g][.][e].
// 256 threads = 64 entries (coalesced) x 4 slices of the block range; fixed order -> deterministic.
__global__ __launch_bounds__(256) void stats_reduce_kernel(const float *__restrict__ part, int nb, int c2,
                                                           float *__restrict__ part2) {
    __shared__ float red[4][64];
    const int el = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + el, z = blockIdx.y, g = blockIdx.z;
    const int b0 = z * 256, b1 = min(b0 + 256, nb);
    float a4[4] = {0.f, 0.f, 0.f, 0.f};
    if (e < c2) {
        const float *p = part + (size_t)g * nb * c2 + e;
        int b = b0 + sl;
        for (; b + 12 < b1; b += 16)
#pragma unroll
            for (int u = 0; u < 4; ++u) a4[u] += p[(size_t)(b + 4 * u) * c2];
        for (; b < b1; b += 4) a4[0] += p[(size_t)b * c2];
    }
    red[sl][el] = (a4[0] + a4[1]) + (a4[2] + a4[3]);
    __syncthreads();
    if (sl == 0 && e < c2)
        part2[((size_t)g * gridDim.y + z) * c2 + e] = (red[0][el] + red[1][el]) + (red[2][el] + red[3][el]);
}

// backward: per (g, ch): (sa, sb) = sum_b part;  dsums = gamma * (sa, sb);  dgamma / dbeta accumulate over groups
// (groups > 1 only for InstanceNorm, which has no affine parameters -> plain stores suffice when groups == 1)
__global__ __launch_bounds__(256) void bwd_finish_kernel(const float *__restrict__ part, int nb, int c,
                                                         const float *__restrict__ gamma, float *__restrict__ dsums,
                                                         float *__restrict__ dgamma, float *__restrict__ dbeta) {
    __shared__ float red[16][2][17];
    const int el = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int g = blockIdx.y, ch = blockIdx.x * 16 + el;
    float sa = 0.f, sb = 0.f;
    if (ch < c) {
        const float *p0 = part + ((size_t)g * nb * c + ch) * 2;
        f32x2 a4[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
        int b = sl;
        for (; b + 48 < nb; b += 64)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const f32x2 v = *reinterpret_cast<const f32x2 *>(p0 + (size_t)(b + 16 * u) * c * 2);
                a4[u][0] += v[0]; a4[u][1] += v[1];
            }
        for (; b < nb; b += 16) {
            const f32x2 v = *reinterpret_cast<const f32x2 *>(p0 + (size_t)b * c * 2);
            a4[0][0] += v[0]; a4[0][1] += v[1];
        }
        sa = (a4[0][0] + a4[1][0]) + (a4[2][0] + a4[3][0]);
        sb = (a4[0][1] + a4[1][1]) + (a4[2][1] + a4[3][1]);
    }
    red[sl][0][el] = sa;
    red[sl][1][el] = sb;
    __syncthreads();
    if (sl == 0 && ch < c) {
        for (int t = 1; t < 16; ++t) {
            sa += red[t][0][el];
            sb += red[t][1][el];
        }
        const float ga = gamma ? gamma[ch] : 1.0f;
        dsums[((size_t)g * c + ch) * 2] = sa * ga;
        dsums[((size_t)g * c + ch) * 2 + 1] = sb * ga;
        // one group (BatchNorm2d, the only norm here with affine parameters): this thread is the channel's only writer
        if (gridDim.y == 1) {
            if (dgamma) dgamma[ch] = sb;
            if (dbeta) dbeta[ch] = sa;
        } else {
            if (dgamma) atomicAdd(dgamma + ch, sb);
            if (dbeta) atomicAdd(dbeta + ch, sa);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(GT) void norm_act_fwd_kernel(NormArgs A) {
    const int lanes = A.c >> 2, cl = threadIdx.x % lanes, rl = threadIdx.x / lanes, rstep = GT / lanes;
    const int g = blockIdx.y, c4 = 4 * cl;
    f32x4 mean, rstd, ga, be;
    stats4(A, g, c4, mean, rstd);
    load_param4(A.gamma, c4, ga, 1.0f);
    load_param4(A.beta, c4, be, 0.0f);
    const long long r0 = (long long)blockIdx.x * A.rows_per_block;
    long long r1 = r0 + A.rows_per_block;
    r1 = r1 < A.rows ? r1 : A.rows;
    const size_t base = ((size_t)g * A.rows) * A.c + c4;
    for (long long r = r0 + rl; r < r1; r += rstep) {
        const size_t off = base + (size_t)r * A.c;
        const f32x4 v = ld4(static_cast<const T *>(A.x) + off);
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float n = (v[i] - mean[i]) * rstd[i] * ga[i] + be[i];
            o[i] = n > 0.0f ? n : n * A.slope;
        }
        if (A.res) {
            const f32x4 rr = ld4(static_cast<const T *>(A.res) + off);
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] += rr[i];
        }
        st4(static_cast<T *>(A.y) + off, o);
    }
}

// dsums[g][c] = (sum dn, sum dn * xhat), dn = dy * leaky'(n) * gamma;  dgamma += sum dy*leaky'*xhat, dbeta += sum dy*leaky'
template <typename T>
__global__ __launch_bounds__(GT) void norm_act_bwd_reduce_kernel(NormArgs A) {
    __shared__ float red[GT][8];
    const int lanes = A.c >> 2, cl = threadIdx.x % lanes, rl = threadIdx.x / lanes, rstep = GT / lanes;
    const int g = blockIdx.y, c4 = 4 * cl;
    f32x4 mean, rstd, ga, be;
    stats4(A, g, c4, mean, rstd);
    load_param4(A.gamma, c4, ga, 1.0f);
    load_param4(A.beta, c4, be, 0.0f);
    const long long r0 = (long long)blockIdx.x * A.rows_per_block;
    long long r1 = r0 + A.rows_per_block;
    r1 = r1 < A.rows ? r1 : A.rows;
    const size_t base = ((size_t)g * A.rows) * A.c + c4;
    f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};   // sum d, sum d*xhat with d = dy*leaky'
#pragma unroll 4
    for (long long r = r0 + rl; r < r1; r += rstep) {
        const size_t off = base + (size_t)r * A.c;
        const f32x4 v = ld4(static_cast<const T *>(A.x) + off);
        const f32x4 d = ld4(static_cast<const T *>(A.dy) + off);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float xh = (v[i] - mean[i]) * rstd[i];
            const float n = xh * ga[i] + be[i];
            const float dd = n > 0.0f ? d[i] : d[i] * A.slope;
            sa[i] += dd;
            sb[i] += dd * xh;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        red[threadIdx.x][i] = sa[i];
        red[threadIdx.x][4 + i] = sb[i];
    }
    __syncthreads();
    if (rl == 0) {
        for (int t = 1; t < rstep; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sa[i] += red[t * lanes + cl][i];
                sb[i] += red[t * lanes + cl][4 + i];
            }
        float *o = A.out_sums + (((size_t)g * gridDim.x + blockIdx.x) * A.c + c4) * 2;   // per-block partial
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[2 * i] = sa[i];
            o[2 * i + 1] = sb[i];
        }
    }
}

template <typename T>
__global__ __launch_bounds__(GT) void norm_act_bwd_apply_kernel(NormArgs A) {
    const int lanes = A.c >> 2, cl = threadIdx.x % lanes, rl = threadIdx.x / lanes, rstep = GT / lanes;
    const int g = blockIdx.y, c4 = 4 * cl;
    f32x4 mean, rstd, ga, be, m1, m2;
    stats4(A, g, c4, mean, rstd);
    load_param4(A.gamma, c4, ga, 1.0f);
    load_param4(A.beta, c4, be, 0.0f);
    const float *ds = A.dsums + ((size_t)g * A.c + c4) * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        m1[i] = ds[2 * i] * A.inv_rows;       // mean(dn)
        m2[i] = ds[2 * i + 1] * A.inv_rows;   // mean(dn * xhat)
    }
    const long long r0 = (long long)blockIdx.x * A.rows_per_block;
    long long r1 = r0 + A.rows_per_block;
    r1 = r1 < A.rows ? r1 : A.rows;
    const size_t base = ((size_t)g * A.rows) * A.c + c4;
    unsigned vmax = 0;
    for (long long r = r0 + rl; r < r1; r += rstep) {
        const size_t off = base + (size_t)r * A.c;
        const f32x4 v = ld4(static_cast<const T *>(A.x) + off);
        const f32x4 d = ld4(static_cast<const T *>(A.dy) + off);
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float xh = (v[i] - mean[i]) * rstd[i];
            const float n = xh * ga[i] + be[i];
            const float dn = (n > 0.0f ? d[i] : d[i] * A.slope) * ga[i];
            o[i] = rstd[i] * (dn - m1[i] - xh * m2[i]);
        }
        st4(static_cast<T *>(A.y) + off, o);
        if constexpr (sizeof(T) == 4) vmax = amax_acc4(vmax, o);
    }
    if constexpr (sizeof(T) == 4) { if (A.amax) amax_commit(vmax, A.amax); }
}

// ---- the tail of a separable block in ONE pass per direction:  y = leaky(norm_a(xa)) + leaky(norm_b(xb)),
// xa = IntraSO3Conv output with its InstanceNorm, xb = the skip branch's 1x1 convolution with the block's norm
// (SPConvNets/utils/base_so3conv.py:204-211).  The skip branch's normalised tensor is never written, and the backward
// pass reads the common output gradient once per pass instead of once per norm.  The tensors are walked as
// [b clouds][rows per cloud][c]; a side is "instance" (statistics per cloud) or "batch" (one set for all clouds).
struct NormSide {
    const void *x;
    void *dx;
    const float *sums, *dsums, *gamma, *beta;
    float *part;          // reduce: per-block partials [b][blocks][c][2]
    float eps, inv_rows;
    int per_cloud;        // 1: statistics indexed by the cloud, 0: shared
};
struct NormArgs2 {
    NormSide a, b;
    const void *dy;
    void *y;
    long long rows;       // per cloud
    int c, rows_per_block;
    float slope;
    unsigned *amax;       // optional (fp32): max |y| (fwd) / max |dx of side b| (bwd_apply): see NormArgs::amax
};

__device__ __forceinline__ void side_stats4(const NormSide &S, int c, int g, int c4, f32x4 &mean, f32x4 &rstd) {
    const float *s = S.sums + ((size_t)(S.per_cloud ? g : 0) * c + c4) * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float m = s[2 * i] * S.inv_rows;
        const float var = fmaxf(s[2 * i + 1] * S.inv_rows - m * m, 0.0f);
        mean[i] = m;
        rstd[i] = rsqrtf(var + S.eps);
    }
}

template <typename T>
__global__ __launch_bounds__(GT) void norm_act2_fwd_kernel(NormArgs2 A) {
    const int lanes = A.c >> 2, cl = threadIdx.x % lanes, rl = threadIdx.x / lanes, rstep = GT / lanes;
    const int g = blockIdx.y, c4 = 4 * cl;
    f32x4 ma, ra, ga, ba, mb, rb, gb, bb;
    side_stats4(A.a, A.c, g, c4, ma, ra);
    side_stats4(A.b, A.c, g, c4, mb, rb);
    load_param4(A.a.gamma, c4, ga, 1.0f); load_param4(A.a.beta, c4, ba, 0.0f);
    load_param4(A.b.gamma, c4, gb, 1.0f); load_param4(A.b.beta, c4, bb, 0.0f);
    const long long r0 = (long long)blockIdx.x * A.rows_per_block;
    long long r1 = r0 + A.rows_per_block;
    r1 = r1 < A.rows ? r1 : A.rows;
    const size_t base = ((size_t)g * A.rows) * A.c + c4;
    unsigned vmax = 0;
    for (long long r = r0 + rl; r < r1; r += rstep) {
        const size_t off = base + (size_t)r * A.c;
        const f32x4 va = ld4(static_cast<const T *>(A.a.x) + off);
        const f32x4 vb = ld4(static_cast<const T *>(A.b.x) + off);
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float na = (va[i] - ma[i]) * ra[i] * ga[i] + ba[i];
            const float nb = (vb[i] - mb[i]) * rb[i] * gb[i] + bb[i];
            o[i] = (na > 0.0f ? na : na * A.slope) + (nb > 0.0f ? nb : nb * A.slope);
        }
        st4(static_cast<T *>(A.y) + off, o);
        if constexpr (sizeof(T) == 4) vmax = amax_acc4(vmax, o);
    }
    if constexpr (sizeof(T) == 4) { if (A.amax) amax_commit(vmax, A.amax); }
}

template <typename T>
__global__ __launch_bounds__(GT) void norm_act2_bwd_reduce_kernel(NormArgs2 A) {
    __shared__ float red[GT][16];
    const int lanes = A.c >> 2, cl = threadIdx.x % lanes, rl = threadIdx.x / lanes, rstep = GT / lanes;
    const int g = blockIdx.y, c4 = 4 * cl;
    f32x4 ma, ra, ga, ba, mb, rb, gb, bb;
    side_stats4(A.a, A.c, g, c4, ma, ra);
    side_stats4(A.b, A.c, g, c4, mb, rb);
    load_param4(A.a.gamma, c4, ga, 1.0f); load_param4(A.a.beta, c4, ba, 0.0f);
    load_param4(A.b.gamma, c4, gb, 1.0f); load_param4(A.b.beta, c4, bb, 0.0f);
    const long long r0 = (long long)blockIdx.x * A.rows_per_block;
    long long r1 = r0 + A.rows_per_block;
    r1 = r1 < A.rows ? r1 : A.rows;
    const size_t base = ((size_t)g * A.rows) * A.c + c4;
    f32x4 sa1 = {0.f, 0.f, 0.f, 0.f}, sa2 = sa1, sb1 = sa1, sb2 = sa1;
#pragma unroll 2
    for (long long r = r0 + rl; r < r1; r += rstep) {
        const size_t off = base + (size_t)r * A.c;
        const f32x4 va = ld4(static_cast<const T *>(A.a.x) + off);
        const f32x4 vb = ld4(static_cast<const T *>(A.b.x) + off);
        const f32x4 d = ld4(static_cast<const T *>(A.dy) + off);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float xa = (va[i] - ma[i]) * ra[i], xb = (vb[i] - mb[i]) * rb[i];
            const float da = (xa * ga[i] + ba[i]) > 0.0f ? d[i] : d[i] * A.slope;
            const float db = (xb * gb[i] + bb[i]) > 0.0f ? d[i] : d[i] * A.slope;
            sa1[i] += da; sa2[i] += da * xa;
            sb1[i] += db; sb2[i] += db * xb;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        red[threadIdx.x][i] = sa1[i]; red[threadIdx.x][4 + i] = sa2[i];
        red[threadIdx.x][8 + i] = sb1[i]; red[threadIdx.x][12 + i] = sb2[i];
    }
    __syncthreads();
    if (rl == 0) {
        for (int t = 1; t < rstep; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sa1[i] += red[t * lanes + cl][i]; sa2[i] += red[t * lanes + cl][4 + i];
                sb1[i] += red[t * lanes + cl][8 + i]; sb2[i] += red[t * lanes + cl][12 + i];
            }
        const size_t po = (((size_t)g * gridDim.x + blockIdx.x) * A.c + c4) * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            A.a.part[po + 2 * i] = sa1[i]; A.a.part[po + 2 * i + 1] = sa2[i];
            A.b.part[po + 2 * i] = sb1[i]; A.b.part[po + 2 * i + 1] = sb2[i];
        }
    }
}

template <typename T>
__global__ __launch_bounds__(GT) void norm_act2_bwd_apply_kernel(NormArgs2 A) {
    const int lanes = A.c >> 2, cl = threadIdx.x % lanes, rl = threadIdx.x / lanes, rstep = GT / lanes;
    const int g = blockIdx.y, c4 = 4 * cl;
    f32x4 ma, ra, ga, ba, mb, rb, gb, bb, a1, a2, b1, b2;
    side_stats4(A.a, A.c, g, c4, ma, ra);
    side_stats4(A.b, A.c, g, c4, mb, rb);
    load_param4(A.a.gamma, c4, ga, 1.0f); load_param4(A.a.beta, c4, ba, 0.0f);
    load_param4(A.b.gamma, c4, gb, 1.0f); load_param4(A.b.beta, c4, bb, 0.0f);
    const float *dsa = A.a.dsums + ((size_t)(A.a.per_cloud ? g : 0) * A.c + c4) * 2;
    const float *dsb = A.b.dsums + ((size_t)(A.b.per_cloud ? g : 0) * A.c + c4) * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a1[i] = dsa[2 * i] * A.a.inv_rows; a2[i] = dsa[2 * i + 1] * A.a.inv_rows;
        b1[i] = dsb[2 * i] * A.b.inv_rows; b2[i] = dsb[2 * i + 1] * A.b.inv_rows;
    }
    const long long r0 = (long long)blockIdx.x * A.rows_per_block;
    long long r1 = r0 + A.rows_per_block;
    r1 = r1 < A.rows ? r1 : A.rows;
    const size_t base = ((size_t)g * A.rows) * A.c + c4;
    unsigned vmax = 0;
    for (long long r = r0 + rl; r < r1; r += rstep) {
        const size_t off = base + (size_t)r * A.c;
        const f32x4 va = ld4(static_cast<const T *>(A.a.x) + off);
        const f32x4 vb = ld4(static_cast<const T *>(A.b.x) + off);
        const f32x4 d = ld4(static_cast<const T *>(A.dy) + off);
        f32x4 oa, ob;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float xa = (va[i] - ma[i]) * ra[i], xb = (vb[i] - mb[i]) * rb[i];
            const float dna = ((xa * ga[i] + ba[i]) > 0.0f ? d[i] : d[i] * A.slope) * ga[i];
            const float dnb = ((xb * gb[i] + bb[i]) > 0.0f ? d[i] : d[i] * A.slope) * gb[i];
            oa[i] = ra[i] * (dna - a1[i] - xa * a2[i]);
            ob[i] = rb[i] * (dnb - b1[i] - xb * b2[i]);
        }
        if (A.a.dx) st4(static_cast<T *>(A.a.dx) + off, oa);
        if (A.b.dx) st4(static_cast<T *>(A.b.dx) + off, ob);
        if constexpr (sizeof(T) == 4) vmax = amax_acc4(vmax, ob);
    }
    if constexpr (sizeof(T) == 4) { if (A.amax && A.b.dx) amax_commit(vmax, A.amax); }
}

int check_norm(int groups, long long rows, int c) {
    if (groups < 0 || rows < 0 || c < 4 || c % 4 != 0 || c > 4 * GT || GT % (c / 4) != 0) return EPN_EINVAL;
    if (groups > 65535) return EPN_EINVAL;
    return 0;
}

NormArgs make_norm(long long rows, int c, float eps, float slope, dim3 &grid, int groups) {
    NormArgs A = {};
    A.rows = rows; A.c = c; A.eps = eps; A.slope = slope;
    A.inv_rows = rows > 0 ? 1.0f / (float)rows : 0.0f;
    // ~1024 blocks in total (4 per CU): enough loads in flight to stream HBM, few enough that the per-block atomics
    // onto the [groups][c] sums do not serialise (2048 blocks cost 0.3 ms of L2 atomic contention per call)
    long long per_group = (1024 + groups - 1) / (groups > 0 ? groups : 1);
    if (per_group < 1) per_group = 1;
    long long rpb = (rows + per_group - 1) / per_group;
    if (rpb < 64) rpb = 64;
    A.rows_per_block = (int)rpb;
    grid = dim3((unsigned)((rows + rpb - 1) / rpb), (unsigned)groups);
    return A;
}

// BatchNorm's running statistics from (sum x, sum x^2) in ONE launch -- what torch.nn.BatchNorm2d does in training mode
// with ~10 elementwise kernels on c-sized vectors (22 norms per step: ~200 launches of 5 us each).
__global__ void bn_running_update_kernel(const float *__restrict__ sums, float n, const float *__restrict__ bias,
                                         float *__restrict__ rmean, float *__restrict__ rvar,
                                         long long *__restrict__ batches, float momentum, int c) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    // every thread reads the counter before thread 0 of block 0 bumps it (one block: c <= 1024, launcher)
    const long long nb = *batches + 1;
    __syncthreads();
    if (i == 0) *batches = nb;
    if (i >= c) return;
    float mean = sums[2 * i] / n;
    const float varb = fmaxf(sums[2 * i + 1] / n - mean * mean, 0.0f);
    const float var = varb * (n / fmaxf(n - 1.0f, 1.0f));           // unbiased, as BatchNorm stores it
    if (bias) mean += bias[i];
    const float m = momentum >= 0.0f ? momentum : 1.0f / (float)nb;  // momentum = None: cumulative moving average
    rmean[i] = fmaf(mean - rmean[i], m, rmean[i]);
    rvar[i] = fmaf(var - rvar[i], m, rvar[i]);
}

}  // namespace
}  // namespace epn

using namespace epn;

extern "C" size_t epn_norm_workspace_bytes(int groups, long long rows, int c) {
    if (check_norm(groups, rows, c) || groups == 0 || rows == 0) return 0;
    dim3 grid;
    make_norm(rows, c, 0.f, 0.f, grid, groups);
    return sizeof(float) * (size_t)groups * grid.x * c * 2;   // one (s1, s2) pair per block and channel
}

static int chan_stats_any(const void *x_cl, int groups, long long rows, int c, float *sums, void *workspace,
                          size_t workspace_bytes, int bf16, epn_stream_t stream) {
    int rc = check_norm(groups, rows, c);
    if (rc) return rc;
    if (!sums) return EPN_ENULL;
    hipStream_t st = epn_stream(stream);
    if (groups == 0 || rows == 0) {
        EPN_HIP(hipMemsetAsync(sums, 0, sizeof(float) * (size_t)groups * c * 2, st));
        return 0;
    }
    if (!x_cl) return EPN_ENULL;
    if (!workspace || workspace_bytes < epn_norm_workspace_bytes(groups, rows, c)) return EPN_EWORKSPACE;
    dim3 grid;
    NormArgs A = make_norm(rows, c, 0.f, 0.f, grid, groups);
    A.x = x_cl; A.out_sums = static_cast<float *>(workspace);
    if (bf16) EPN_LAUNCH(chan_stats_kernel<__bf16>, grid, dim3(GT), 0, st, A);
    else EPN_LAUNCH(chan_stats_kernel<float>, grid, dim3(GT), 0, st, A);
    EPN_CHECK_LAUNCH();
    EPN_LAUNCH_AUX(stats_finish_kernel, dim3(epn_cdiv(2 * c, 16), groups), dim3(256), 0, st, A.out_sums, (int)grid.x,
                       c * 2, sums);
    EPN_CHECK_LAUNCH();
    return 0;
}

static int norm_act_fwd_any(const void *x_cl, int groups, long long rows, int c, const float *sums, const float *gamma,
                            const float *beta, const void *residual_cl, float eps, float slope, void *y_cl, int bf16,
                            epn_stream_t stream) {
    int rc = check_norm(groups, rows, c);
    if (rc) return rc;
    if (groups == 0 || rows == 0) return 0;
    if (!x_cl || !sums || !y_cl) return EPN_ENULL;
    dim3 grid;
    NormArgs A = make_norm(rows, c, eps, slope, grid, groups);
    A.x = x_cl; A.sums = sums; A.gamma = gamma; A.beta = beta; A.res = residual_cl; A.y = y_cl;
    if (bf16) EPN_LAUNCH(norm_act_fwd_kernel<__bf16>, grid, dim3(GT), 0, epn_stream(stream), A);
    else EPN_LAUNCH(norm_act_fwd_kernel<float>, grid, dim3(GT), 0, epn_stream(stream), A);
    EPN_CHECK_LAUNCH();
    return 0;
}

static int norm_act_bwd_reduce_any(const void *x_cl, const void *dy_cl, int groups, long long rows, int c,
                                   const float *sums, const float *gamma, const float *beta, float eps, float slope,
                                   float *dsums, float *dgamma, float *dbeta, void *workspace, size_t workspace_bytes,
                                   int bf16, epn_stream_t stream) {
    int rc = check_norm(groups, rows, c);
    if (rc) return rc;
    if (!dsums) return EPN_ENULL;
    hipStream_t st = epn_stream(stream);
    if (groups == 0 || rows == 0) {
        EPN_HIP(hipMemsetAsync(dsums, 0, sizeof(float) * (size_t)groups * c * 2, st));
        if (dgamma) EPN_HIP(hipMemsetAsync(dgamma, 0, sizeof(float) * c, st));
        if (dbeta) EPN_HIP(hipMemsetAsync(dbeta, 0, sizeof(float) * c, st));
        return 0;
    }
    if (!x_cl || !dy_cl || !sums) return EPN_ENULL;
    if (!workspace || workspace_bytes < epn_norm_workspace_bytes(groups, rows, c)) return EPN_EWORKSPACE;
    dim3 grid;
    NormArgs A = make_norm(rows, c, eps, slope, grid, groups);
    A.x = x_cl; A.dy = dy_cl; A.sums = sums; A.gamma = gamma; A.beta = beta;
    A.out_sums = static_cast<float *>(workspace);
    if (bf16) EPN_LAUNCH(norm_act_bwd_reduce_kernel<__bf16>, grid, dim3(GT), 0, st, A);
    else EPN_LAUNCH(norm_act_bwd_reduce_kernel<float>, grid, dim3(GT), 0, st, A);
    EPN_CHECK_LAUNCH();
    if (groups > 1) {                        // several groups accumulate with atomics; one group stores
        if (dgamma) EPN_HIP(hipMemsetAsync(dgamma, 0, sizeof(float) * c, st));
        if (dbeta) EPN_HIP(hipMemsetAsync(dbeta, 0, sizeof(float) * c, st));
    }
    EPN_LAUNCH_AUX(bwd_finish_kernel, dim3(epn_cdiv(c, 16), groups), dim3(256), 0, st, A.out_sums, (int)grid.x, c, gamma,
                       dsums, dgamma, dbeta);
    EPN_CHECK_LAUNCH();
    return 0;
}

static int amax_prepare(float *amax, int bf16, hipStream_t st) {
    if (!amax) return 0;
    if (bf16) return EPN_EINVAL;             // the maximum is a by-product of the fp32 kernels only
    EPN_HIP(hipMemsetAsync(amax, 0, sizeof(float), st));
    return 0;
}

static int norm_act_bwd_apply_any(const void *x_cl, const void *dy_cl, int groups, long long rows, int c,
                                  const float *sums, const float *dsums, const float *gamma, const float *beta, float eps,
                                  float slope, void *dx_cl, int bf16, epn_stream_t stream, float *amax = nullptr) {
    int rc = check_norm(groups, rows, c);
    if (rc) return rc;
    rc = amax_prepare(amax, bf16, epn_stream(stream));
    if (rc) return rc;
    if (groups == 0 || rows == 0) return 0;
    if (!x_cl || !dy_cl || !sums || !dsums || !dx_cl) return EPN_ENULL;
    dim3 grid;
    NormArgs A = make_norm(rows, c, eps, slope, grid, groups);
    A.x = x_cl; A.dy = dy_cl; A.sums = sums; A.dsums = dsums; A.gamma = gamma; A.beta = beta; A.y = dx_cl;
    A.amax = reinterpret_cast<unsigned *>(amax);
    if (bf16) EPN_LAUNCH(norm_act_bwd_apply_kernel<__bf16>, grid, dim3(GT), 0, epn_stream(stream), A);
    else EPN_LAUNCH(norm_act_bwd_apply_kernel<float>, grid, dim3(GT), 0, epn_stream(stream), A);
    EPN_CHECK_LAUNCH();
    return 0;
}

// ---- pair form: host side.  clouds b, rows per cloud, c; each side: sums / dsums [b or 1][c][2], gamma, beta (or NULL)
static int pair_setup(const epn_norm_pair_side *sa, const epn_norm_pair_side *sb, int b, long long rows, int c, float slope,
                      NormArgs2 &A, dim3 &grid) {
    int rc = check_norm(b, rows, c);
    if (rc) return rc;
    if (!sa || !sb) return EPN_ENULL;
    NormArgs one = make_norm(rows, c, 0.f, slope, grid, b);
    A = NormArgs2{};
    A.rows = rows; A.c = c; A.rows_per_block = one.rows_per_block; A.slope = slope;
    const epn_norm_pair_side *src[2] = {sa, sb};
    NormSide *dst[2] = {&A.a, &A.b};
    for (int i = 0; i < 2; ++i) {
        dst[i]->sums = src[i]->sums; dst[i]->gamma = src[i]->gamma; dst[i]->beta = src[i]->beta;
        dst[i]->eps = src[i]->eps; dst[i]->per_cloud = src[i]->instance ? 1 : 0;
        const double n = src[i]->instance ? (double)rows : (double)rows * b;
        dst[i]->inv_rows = n > 0 ? (float)(1.0 / n) : 0.f;
    }
    return 0;
}

static int norm_act2_fwd_any(const void *xa, const void *xb, int b, long long rows, int c, const epn_norm_pair_side *sa,
                             const epn_norm_pair_side *sb, float slope, void *y, int bf16, epn_stream_t stream,
                             float *amax = nullptr) {
    NormArgs2 A; dim3 grid;
    int rc = pair_setup(sa, sb, b, rows, c, slope, A, grid);
    if (rc) return rc;
    rc = amax_prepare(amax, bf16, epn_stream(stream));
    if (rc) return rc;
    if (b == 0 || rows == 0) return 0;
    if (!xa || !xb || !y || !sa->sums || !sb->sums) return EPN_ENULL;
    A.a.x = xa; A.b.x = xb; A.y = y; A.amax = reinterpret_cast<unsigned *>(amax);
    if (bf16) EPN_LAUNCH(norm_act2_fwd_kernel<__bf16>, grid, dim3(GT), 0, epn_stream(stream), A);
    else EPN_LAUNCH(norm_act2_fwd_kernel<float>, grid, dim3(GT), 0, epn_stream(stream), A);
    EPN_CHECK_LAUNCH();
    return 0;
}

static int norm_act2_bwd_reduce_any(const void *xa, const void *xb, const void *dy, int b, long long rows, int c,
                                    const epn_norm_pair_side *sa, const epn_norm_pair_side *sb, float slope, float *dsums_a,
                                    float *dgamma_a, float *dbeta_a, float *dsums_b, float *dgamma_b, float *dbeta_b,
                                    void *workspace, size_t workspace_bytes, int bf16, epn_stream_t stream) {
    NormArgs2 A; dim3 grid;
    int rc = pair_setup(sa, sb, b, rows, c, slope, A, grid);
    if (rc) return rc;
    if (!dsums_a || !dsums_b) return EPN_ENULL;
    hipStream_t st = epn_stream(stream);
    float *dg[2] = {dgamma_a, dgamma_b}, *db[2] = {dbeta_a, dbeta_b};
    {
        const epn_norm_pair_side *sd[2] = {sa, sb};
        for (int i = 0; i < 2; ++i) {
            if (!(sd[i]->instance && b > 1) && b > 0 && rows > 0) continue;     // one group: the finishing kernel stores
            if (dg[i]) EPN_HIP(hipMemsetAsync(dg[i], 0, sizeof(float) * c, st));
            if (db[i]) EPN_HIP(hipMemsetAsync(db[i], 0, sizeof(float) * c, st));
        }
    }
    if (b == 0 || rows == 0) return 0;
    if (!xa || !xb || !dy || !sa->sums || !sb->sums) return EPN_ENULL;
    const size_t part_bytes = sizeof(float) * (size_t)b * grid.x * c * 2;
    if (!workspace || workspace_bytes < 2 * part_bytes) return EPN_EWORKSPACE;
    A.a.x = xa; A.b.x = xb; A.dy = dy;
    A.a.part = static_cast<float *>(workspace);
    A.b.part = A.a.part + (size_t)b * grid.x * c * 2;
    if (bf16) EPN_LAUNCH(norm_act2_bwd_reduce_kernel<__bf16>, grid, dim3(GT), 0, st, A);
    else EPN_LAUNCH(norm_act2_bwd_reduce_kernel<float>, grid, dim3(GT), 0, st, A);
    EPN_CHECK_LAUNCH();
    // finishing: an "instance" side keeps one row of sums per cloud, a "batch" side adds the partials of all clouds
    const epn_norm_pair_side *src[2] = {sa, sb};
    float *parts[2] = {A.a.part, A.b.part}, *ds[2] = {dsums_a, dsums_b};
    for (int i = 0; i < 2; ++i) {
        const int groups = src[i]->instance ? b : 1, nb = src[i]->instance ? (int)grid.x : (int)grid.x * b;
        EPN_LAUNCH_AUX(bwd_finish_kernel, dim3(epn_cdiv(c, 16), groups), dim3(256), 0, st, parts[i], nb, c, src[i]->gamma,
                       ds[i], dg[i], db[i]);
        EPN_CHECK_LAUNCH();
    }
    return 0;
}

static int norm_act2_bwd_apply_any(const void *xa, const void *xb, const void *dy, int b, long long rows, int c,
                                   const epn_norm_pair_side *sa, const epn_norm_pair_side *sb, float slope,
                                   const float *dsums_a, const float *dsums_b, void *dxa, void *dxb, int bf16,
                                   epn_stream_t stream, float *amax_b = nullptr) {
    NormArgs2 A; dim3 grid;
    int rc = pair_setup(sa, sb, b, rows, c, slope, A, grid);
    if (rc) return rc;
    rc = amax_prepare(amax_b, bf16, epn_stream(stream));
    if (rc) return rc;
    if (b == 0 || rows == 0) return 0;
    if (!xa || !xb || !dy || !sa->sums || !sb->sums || !dsums_a || !dsums_b) return EPN_ENULL;
    A.a.x = xa; A.b.x = xb; A.dy = dy; A.a.dsums = dsums_a; A.b.dsums = dsums_b; A.a.dx = dxa; A.b.dx = dxb;
    A.amax = reinterpret_cast<unsigned *>(amax_b);
    if (bf16) EPN_LAUNCH(norm_act2_bwd_apply_kernel<__bf16>, grid, dim3(GT), 0, epn_stream(stream), A);
    else EPN_LAUNCH(norm_act2_bwd_apply_kernel<float>, grid, dim3(GT), 0, epn_stream(stream), A);
    EPN_CHECK_LAUNCH();
    return 0;
}

extern "C" size_t epn_norm_pair_workspace_bytes(int b, long long rows, int c) {
    return 2 * epn_norm_workspace_bytes(b, rows, c);
}
extern "C" int epn_norm_act_pair_fwd(const void *xa_cl, const void *xb_cl, int b, long long rows, int c,
                                     const epn_norm_pair_side *side_a, const epn_norm_pair_side *side_b, float slope, void *y_cl,
                                     int bf16, epn_stream_t stream) {
    return norm_act2_fwd_any(xa_cl, xb_cl, b, rows, c, side_a, side_b, slope, y_cl, bf16, stream);
}
extern "C" int epn_norm_act_pair_bwd_reduce(const void *xa_cl, const void *xb_cl, const void *dy_cl, int b, long long rows,
                                            int c, const epn_norm_pair_side *side_a, const epn_norm_pair_side *side_b,
                                            float slope, float *dsums_a, float *dgamma_a, float *dbeta_a, float *dsums_b,
                                            float *dgamma_b, float *dbeta_b, void *workspace, size_t workspace_bytes,
                                            int bf16, epn_stream_t stream) {
    return norm_act2_bwd_reduce_any(xa_cl, xb_cl, dy_cl, b, rows, c, side_a, side_b, slope, dsums_a, dgamma_a, dbeta_a,
                                    dsums_b, dgamma_b, dbeta_b, workspace, workspace_bytes, bf16, stream);
}
extern "C" int epn_norm_act_pair_bwd_apply(const void *xa_cl, const void *xb_cl, const void *dy_cl, int b, long long rows,
                                           int c, const epn_norm_pair_side *side_a, const epn_norm_pair_side *side_b,
                                           float slope, const float *dsums_a, const float *dsums_b, void *dxa_cl,
                                           void *dxb_cl, int bf16, epn_stream_t stream) {
    return norm_act2_bwd_apply_any(xa_cl, xb_cl, dy_cl, b, rows, c, side_a, side_b, slope, dsums_a, dsums_b, dxa_cl, dxb_cl,
                                   bf16, stream);
}

// fp32 variants that also leave max |output| in a device scalar (zeroed by the call): the producer-side maxima of the
// two-piece fp16 GEMMs' operands (gemm.h) -- the block output feeds the next block's grouping bound and skip convolution, the two
// gradients are the narrow operands of the data- / weight-gradient GEMMs.  bf16 = 1: EPN_EINVAL.
extern "C" int epn_norm_act_pair_fwd_amax(const void *xa_cl, const void *xb_cl, int b, long long rows, int c,
                                          const epn_norm_pair_side *side_a, const epn_norm_pair_side *side_b, float slope,
                                          void *y_cl, int bf16, float *y_amax, epn_stream_t stream) {
    if (!y_amax) return EPN_ENULL;
    return norm_act2_fwd_any(xa_cl, xb_cl, b, rows, c, side_a, side_b, slope, y_cl, bf16, stream, y_amax);
}
extern "C" int epn_norm_act_pair_bwd_apply_amax(const void *xa_cl, const void *xb_cl, const void *dy_cl, int b, long long rows,
                                                int c, const epn_norm_pair_side *side_a, const epn_norm_pair_side *side_b,
                                                float slope, const float *dsums_a, const float *dsums_b, void *dxa_cl,
                                                void *dxb_cl, int bf16, float *dxb_amax, epn_stream_t stream) {
    if (!dxb_amax) return EPN_ENULL;
    return norm_act2_bwd_apply_any(xa_cl, xb_cl, dy_cl, b, rows, c, side_a, side_b, slope, dsums_a, dsums_b, dxa_cl, dxb_cl,
                                   bf16, stream, dxb_amax);
}
extern "C" int epn_norm_act_bwd_apply_amax_f32(const float *x_cl, const float *dy_cl, int groups, long long rows, int c,
                                               const float *sums, const float *dsums, const float *gamma, const float *beta,
                                               float eps, float slope, float *dx_cl, float *dx_amax, epn_stream_t stream) {
    if (!dx_amax) return EPN_ENULL;
    return norm_act_bwd_apply_any(x_cl, dy_cl, groups, rows, c, sums, dsums, gamma, beta, eps, slope, dx_cl, 0, stream, dx_amax);
}

extern "C" int epn_bn_running_update_f32(const float *sums, double count, const float *conv_bias, float *running_mean,
                                        float *running_var, long long *num_batches_tracked, float momentum, int c,
                                        epn_stream_t stream) {
    if (c < 1 || c > 1024 || count < 1.0) return EPN_EINVAL;
    if (!sums || !running_mean || !running_var || !num_batches_tracked) return EPN_ENULL;
    EPN_LAUNCH(bn_running_update_kernel, dim3(1), dim3(1024), 0, epn_stream(stream), sums, (float)count, conv_bias,
                       running_mean, running_var, num_batches_tracked, momentum, c);
    EPN_CHECK_LAUNCH();
    return 0;
}

// epilogue partials are one row per 32 tensor rows (30720 of them for the first cls layer): reduced 256 rows at a time
// first, then by stats_finish_kernel
constexpr int STATS_ZB = 256;
static long long stats_finish_nz(long long blocks) { return blocks > 2048 ? (blocks + STATS_ZB - 1) / STATS_ZB : 0; }

extern "C" size_t epn_stats_finish_workspace_bytes(int groups, long long blocks_per_group, int c) {
    if (groups < 1 || blocks_per_group < 1 || c < 1) return 0;
    return sizeof(float) * (size_t)groups * stats_finish_nz(blocks_per_group) * c * 2;
}

extern "C" int epn_stats_finish(const float *partials, int groups, long long blocks_per_group, int c, float *sums,
                                void *workspace, size_t workspace_bytes, epn_stream_t stream) {
    if (groups < 0 || blocks_per_group < 0 || c < 1 || blocks_per_group > 0x7fffffffLL) return EPN_EINVAL;
    // the first level launches one grid row (gridDim.y) per 256 blocks and one grid plane (gridDim.z) per group
    if (stats_finish_nz(blocks_per_group) > 65535 || groups > 65535) return EPN_EINVAL;
    if (groups == 0) return 0;
    if (!sums || (blocks_per_group && !partials)) return EPN_ENULL;
    hipStream_t st = epn_stream(stream);
    const long long nz = stats_finish_nz(blocks_per_group);
    if (nz) {
        if (!workspace || workspace_bytes < epn_stats_finish_workspace_bytes(groups, blocks_per_group, c)) return EPN_EWORKSPACE;
        float *part2 = static_cast<float *>(workspace);
        EPN_LAUNCH(stats_reduce_kernel, dim3(epn_cdiv(2 * c, 64), (unsigned)nz, groups), dim3(256), 0, st, partials,
                   (int)blocks_per_group, c * 2, part2);
        EPN_CHECK_LAUNCH();
        partials = part2;
        blocks_per_group = nz;
    }
    EPN_LAUNCH(stats_finish_kernel, dim3(epn_cdiv(2 * c, 16), groups), dim3(256), 0, st, partials, (int)blocks_per_group,
               c * 2, sums);
    EPN_CHECK_LAUNCH();
    return 0;
}

// dsums / dgamma / dbeta of a norm's backward pass from block partials part[g][block][c][2] = (sum d, sum d xhat) that
// another kernel's epilogue produced (epn_so3_basis_dstats_*: one block per point): what norm_act_bwd_reduce's own finishing
// step does, as an entry point.  Workspace: epn_stats_finish_workspace_bytes(groups, blocks_per_group, c).
extern "C" int epn_norm_bwd_finish(const float *partials, int groups, long long blocks_per_group, int c, const float *gamma,
                                   float *dsums, float *dgamma, float *dbeta, void *workspace, size_t workspace_bytes,
                                   epn_stream_t stream) {
    if (groups < 0 || blocks_per_group < 1 || c < 1 || blocks_per_group > 0x7fffffffLL) return EPN_EINVAL;
    if (stats_finish_nz(blocks_per_group) > 65535 || groups > 65535) return EPN_EINVAL;
    if (groups == 0) return 0;
    if (!partials || !dsums) return EPN_ENULL;
    hipStream_t st = epn_stream(stream);
    const long long nz = stats_finish_nz(blocks_per_group);
    if (nz) {
        if (!workspace || workspace_bytes < epn_stats_finish_workspace_bytes(groups, blocks_per_group, c)) return EPN_EWORKSPACE;
        float *part2 = static_cast<float *>(workspace);
        EPN_LAUNCH_AUX(stats_reduce_kernel, dim3(epn_cdiv(2 * c, 64), (unsigned)nz, groups), dim3(256), 0, st, partials,
                       (int)blocks_per_group, c * 2, part2);
        EPN_CHECK_LAUNCH();
        partials = part2;
        blocks_per_group = nz;
    }
    if (groups > 1) {                        // several groups accumulate with atomics; one group stores
        if (dgamma) EPN_HIP(hipMemsetAsync(dgamma, 0, sizeof(float) * c, st));
        if (dbeta) EPN_HIP(hipMemsetAsync(dbeta, 0, sizeof(float) * c, st));
    }
    EPN_LAUNCH(bwd_finish_kernel, dim3(epn_cdiv(c, 16), groups), dim3(256), 0, st, partials, (int)blocks_per_group, c, gamma,
               dsums, dgamma, dbeta);
    EPN_CHECK_LAUNCH();
    return 0;
}

extern "C" int epn_chan_stats_f32(const float *x_cl, int groups, long long rows, int c, float *sums, void *workspace,
                                  size_t workspace_bytes, epn_stream_t stream) {
    return chan_stats_any(x_cl, groups, rows, c, sums, workspace, workspace_bytes, 0, stream);
}
extern "C" int epn_chan_stats_bf16(const void *x_cl, int groups, long long rows, int c, float *sums, void *workspace,
                                   size_t workspace_bytes, epn_stream_t stream) {
    return chan_stats_any(x_cl, groups, rows, c, sums, workspace, workspace_bytes, 1, stream);
}
extern "C" int epn_norm_act_fwd_f32(const float *x_cl, int groups, long long rows, int c, const float *sums,
                                    const float *gamma, const float *beta, const float *residual_cl, float eps,
                                    float slope, float *y_cl, epn_stream_t stream) {
    return norm_act_fwd_any(x_cl, groups, rows, c, sums, gamma, beta, residual_cl, eps, slope, y_cl, 0, stream);
}
extern "C" int epn_norm_act_fwd_bf16(const void *x_cl, int groups, long long rows, int c, const float *sums,
                                     const float *gamma, const float *beta, const void *residual_cl, float eps,
                                     float slope, void *y_cl, epn_stream_t stream) {
    return norm_act_fwd_any(x_cl, groups, rows, c, sums, gamma, beta, residual_cl, eps, slope, y_cl, 1, stream);
}
extern "C" int epn_norm_act_bwd_reduce_f32(const float *x_cl, const float *dy_cl, int groups, long long rows, int c,
                                           const float *sums, const float *gamma, const float *beta, float eps,
                                           float slope, float *dsums, float *dgamma, float *dbeta, void *workspace,
                                           size_t workspace_bytes, epn_stream_t stream) {
    return norm_act_bwd_reduce_any(x_cl, dy_cl, groups, rows, c, sums, gamma, beta, eps, slope, dsums, dgamma, dbeta,
                                   workspace, workspace_bytes, 0, stream);
}
extern "C" int epn_norm_act_bwd_reduce_bf16(const void *x_cl, const void *dy_cl, int groups, long long rows, int c,
                                            const float *sums, const float *gamma, const float *beta, float eps,
                                            float slope, float *dsums, float *dgamma, float *dbeta, void *workspace,
                                            size_t workspace_bytes, epn_stream_t stream) {
    return norm_act_bwd_reduce_any(x_cl, dy_cl, groups, rows, c, sums, gamma, beta, eps, slope, dsums, dgamma, dbeta,
                                   workspace, workspace_bytes, 1, stream);
}
extern "C" int epn_norm_act_bwd_apply_f32(const float *x_cl, const float *dy_cl, int groups, long long rows, int c,
                                          const float *sums, const float *dsums, const float *gamma,
                                          const float *beta, float eps, float slope, float *dx_cl,
                                          epn_stream_t stream) {
    return norm_act_bwd_apply_any(x_cl, dy_cl, groups, rows, c, sums, dsums, gamma, beta, eps, slope, dx_cl, 0, stream);
}
extern "C" int epn_norm_act_bwd_apply_bf16(const void *x_cl, const void *dy_cl, int groups, long long rows, int c,
                                           const float *sums, const float *dsums, const float *gamma, const float *beta,
                                           float eps, float slope, void *dx_cl, epn_stream_t stream) {
    return norm_act_bwd_apply_any(x_cl, dy_cl, groups, rows, c, sums, dsums, gamma, beta, eps, slope, dx_cl, 1, stream);
}

// ---- IntraSO3Conv grouping as a tensor (the "split" form): grouped[col][k*c + ci] = x[(pt*na + idx[a][k])*c + ci].
// Pure HBM copy: one 16-byte element per thread, consecutive threads walk one gathered row, so reads and writes are
// full-width coalesced.  c % 4 == 0 (launcher falls back to the scalar kernel otherwise).
namespace epn {
namespace {

template <typename V>
__global__ __launch_bounds__(256) void intra_group_kernel(const V *__restrict__ x, const int32_t *__restrict__ idx,
                                                          V *__restrict__ g, long long nelem, int na, int kn, int cv) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nelem) return;
    const int c = (int)(i % cv);
    const long long r = i / cv;
    const int k = (int)(r % kn);
    const long long col = r / kn;
    const int a = (int)(col % na);
    const long long pt = col / na;
    g[i] = x[(pt * na + idx[a * kn + k]) * cv + c];
}

}  // namespace
}  // namespace epn

extern "C" int epn_intra_group_bf16(const void *feats_cl, const int32_t *intra_idx, void *grouped, int b, int p, int na,
                                    int kn, int c, epn_stream_t stream) {
    // a pure copy: 8 bf16 channels are one 16-byte element, exactly like 4 floats
    if (c < 8 || c % 8 != 0) return EPN_EINVAL;
    return epn_intra_group_f32(static_cast<const float *>(feats_cl), intra_idx, static_cast<float *>(grouped), b, p, na,
                               kn, c / 2, stream);
}

extern "C" int epn_intra_group_f32(const float *feats_cl, const int32_t *intra_idx, float *grouped, int b, int p,
                                   int na, int kn, int c, epn_stream_t stream) {
    if (b < 0 || p < 0 || na < 1 || kn < 1 || c < 1) return EPN_EINVAL;
    if (b == 0 || p == 0) return 0;
    if (!feats_cl || !intra_idx || !grouped) return EPN_ENULL;
    const long long cols = (long long)b * p * na;
    if (c % 4 == 0) {
        const long long n = cols * kn * (c / 4);
        EPN_LAUNCH((intra_group_kernel<f32x4>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                           epn_stream(stream), reinterpret_cast<const f32x4 *>(feats_cl), intra_idx,
                           reinterpret_cast<f32x4 *>(grouped), n, na, kn, c / 4);
    } else {
        const long long n = cols * kn * c;
        EPN_LAUNCH((intra_group_kernel<float>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                           epn_stream(stream), feats_cl, intra_idx, grouped, n, na, kn, c);
    }
    EPN_CHECK_LAUNCH();
    return 0;
}

// ---- strided skip connection: batched_index_select(skip, 2, sample_idx) (SPConvNets/utils/base_so3conv.py:206-207,
// vgtk/vgtk/spconv/functional.py:361-369) on channels-last data = a gather of whole rows [a][c] (rowlen bytes each);
// its autograd backward = the scatter of those rows into a zeroed tensor (FPS indices are distinct: plain stores).
// One 16-byte element per thread, consecutive threads walk one row: both sides coalesced.
namespace epn {
namespace {
__global__ __launch_bounds__(256) void rows_gather_kernel(const f32x4 *__restrict__ src, const int32_t *__restrict__ idx,
                                                          f32x4 *__restrict__ dst, long long nelem, int p1, int p2,
                                                          int row16, int scatter) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nelem) return;
    const int e = (int)(i % row16);
    const long long r = i / row16;            // b*p2 + p
    const long long b = r / p2;
    const int q = idx[r];
    if (q < 0 || q >= p1) return;
    const long long s = (b * p1 + q) * row16 + e;
    if (scatter) dst[s] = src[i];
    else dst[i] = src[s];
}

// Transpose of the row gather that tolerates REPEATED indices, without atomics: one workgroup per gathered row r.  It scans
// the cloud's index list (p2 int32, staged in LDS) for other occurrences of q = idx[r].  A unique q -- every row of an
// ordinary cloud -- is a plain 16-byte store stream.  Of a repeated q (FPS pads degenerate clouds with index 0) only the
// FIRST occurrence writes: the sum of all its rows, accumulated in fp32 in ascending row order (deterministic; one
// rounding for bf16).  The target is zero-filled by the caller.
constexpr int ROWS_P2_MAX = 8192;
template <bool BF>
__global__ __launch_bounds__(256) void rows_scatter_add_kernel(const f32x4 *__restrict__ src, const int32_t *__restrict__ idx,
                                                               f32x4 *__restrict__ dst, int p1, int p2, int row16) {
    __shared__ int s_row[ROWS_P2_MAX];
    __shared__ int s_before, s_after;
    const long long r = blockIdx.x;           // b*p2 + p
    const long long b = r / p2;
    const int q = idx[r];
    if (q < 0 || q >= p1) return;
    if (threadIdx.x == 0) s_before = s_after = 0;
    const int32_t *row = idx + b * p2;
    const int self = (int)(r - b * p2);
    for (int k = threadIdx.x; k < p2; k += 256) s_row[k] = row[k];
    __syncthreads();
    bool before = false, after = false;
    for (int k = threadIdx.x; k < p2; k += 256) {
        const bool same = s_row[k] == q;
        before = before || (same && k < self);
        after = after || (same && k > self);
    }
    if (before) s_before = 1;
    if (after) s_after = 1;
    __syncthreads();
    if (s_before) return;                     // a later occurrence: the first one sums for all
    const f32x4 *sp = src + r * row16;
    f32x4 *dp = dst + (b * p1 + q) * row16;
    if (!s_after) {
        for (int e = threadIdx.x; e < row16; e += 256) dp[e] = sp[e];
        return;
    }
    for (int e = threadIdx.x; e < row16; e += 256) {
        f32x4 v = sp[e];
        if constexpr (!BF) {
            for (int k = self + 1; k < p2; ++k)
                if (s_row[k] == q) {
                    const f32x4 u = src[(b * p2 + k) * row16 + e];
                    v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3];
                }
        } else {
            float acc[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float vi = v[i];
                const unsigned w = __float_as_uint(vi);
                acc[2 * i] = __uint_as_float(w << 16);
                acc[2 * i + 1] = __uint_as_float(w & 0xffff0000u);
            }
            for (int k = self + 1; k < p2; ++k)
                if (s_row[k] == q) {
                    const f32x4 u = src[(b * p2 + k) * row16 + e];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float ui = u[i];
                        const unsigned w = __float_as_uint(ui);
                        acc[2 * i] += __uint_as_float(w << 16);
                        acc[2 * i + 1] += __uint_as_float(w & 0xffff0000u);
                    }
                }
            typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
            typedef float f2 __attribute__((ext_vector_type(2)));
            float o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bf2 pk = __builtin_convertvector(f2{acc[2 * i], acc[2 * i + 1]}, bf2);
                o[i] = __uint_as_float(__builtin_bit_cast(unsigned, pk));
            }
            v = f32x4{o[0], o[1], o[2], o[3]};
        }
        dp[e] = v;
    }
}
}  // namespace
}  // namespace epn

// src [b][p1][row_bytes], idx i32[b][p2] -> dst [b][p2][row_bytes] (row_bytes % 16 == 0; any element type)
extern "C" int epn_gather_rows(const void *src, const int32_t *idx, void *dst, int b, int p1, int p2, long long row_bytes,
                               epn_stream_t stream) {
    if (b < 0 || p1 < 1 || p2 < 0 || row_bytes < 16 || row_bytes % 16) return EPN_EINVAL;
    const long long n = (long long)b * p2 * (row_bytes / 16);
    if (n == 0) return 0;
    if (!src || !idx || !dst) return EPN_ENULL;
    EPN_LAUNCH(epn::rows_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, epn_stream(stream),
                       static_cast<const f32x4 *>(src), idx, static_cast<f32x4 *>(dst), n, p1, p2, (int)(row_bytes / 16), 0);
    EPN_CHECK_LAUNCH();
    return 0;
}

// transpose: grad_dst [b][p2][row_bytes] -> grad_src [b][p1][row_bytes], zero-filled here; idx must be distinct per cloud
extern "C" int epn_scatter_rows(const void *grad_dst, const int32_t *idx, void *grad_src, int b, int p1, int p2,
                                long long row_bytes, epn_stream_t stream) {
    if (b < 0 || p1 < 1 || p2 < 0 || row_bytes < 16 || row_bytes % 16) return EPN_EINVAL;
    if (!grad_src) return EPN_ENULL;
    hipStream_t st = epn_stream(stream);
    EPN_HIP(hipMemsetAsync(grad_src, 0, (size_t)b * p1 * row_bytes, st));
    const long long n = (long long)b * p2 * (row_bytes / 16);
    if (n == 0) return 0;
    if (!grad_dst || !idx) return EPN_ENULL;
    EPN_LAUNCH(epn::rows_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                       static_cast<const f32x4 *>(grad_dst), idx, static_cast<f32x4 *>(grad_src), n, p1, p2,
                       (int)(row_bytes / 16), 1);
    EPN_CHECK_LAUNCH();
    return 0;
}

// the same transpose, ACCUMULATING (no atomics, deterministic; p2 <= 8192): an index that occurs several times (FPS repeats index 0 when a cloud has fewer live
// points than samples: padded clouds, points inside its 1e-3 dead zone) receives the sum of its rows, as the backward of
// torch.gather / the reference's batched_index_select does.  Elements are fp32 (bf16 = 0) or bf16 (bf16 = 1).
extern "C" int epn_scatter_rows_add(const void *grad_dst, const int32_t *idx, void *grad_src, int b, int p1, int p2,
                                    long long row_bytes, int bf16, epn_stream_t stream) {
    if (b < 0 || p1 < 1 || p2 < 0 || p2 > epn::ROWS_P2_MAX || row_bytes < 16 || row_bytes % 16) return EPN_EINVAL;
    if (!grad_src) return EPN_ENULL;
    hipStream_t st = epn_stream(stream);
    EPN_HIP(hipMemsetAsync(grad_src, 0, (size_t)b * p1 * row_bytes, st));
    const long long n = (long long)b * p2 * (row_bytes / 16);
    if (n == 0) return 0;
    if (!grad_dst || !idx) return EPN_ENULL;
    const unsigned rows = (unsigned)((long long)b * p2);
    if (bf16) EPN_LAUNCH(epn::rows_scatter_add_kernel<true>, dim3(rows), dim3(256), 0, st, static_cast<const f32x4 *>(grad_dst),
                         idx, static_cast<f32x4 *>(grad_src), p1, p2, (int)(row_bytes / 16));
    else EPN_LAUNCH(epn::rows_scatter_add_kernel<false>, dim3(rows), dim3(256), 0, st, static_cast<const f32x4 *>(grad_dst),
                    idx, static_cast<f32x4 *>(grad_src), p1, p2, (int)(row_bytes / 16));
    EPN_CHECK_LAUNCH();
    return 0;
}

// ---- 1x1 convolution of a single input channel (the occupancy feature of the first block): an outer product
//      y[row][c] = x[row] * w[c], and its weight gradient dw[c] = sum_row x[row] * dy[row][c]
namespace epn {
namespace {
__global__ __launch_bounds__(256) void c1_outer_kernel(const float *__restrict__ x, const f32x4 *__restrict__ w,
                                                       f32x4 *__restrict__ y, long long n4, int c4) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float xv = x[i / c4];
    const f32x4 wv = w[i % c4];
    y[i] = f32x4{xv * wv[0], xv * wv[1], xv * wv[2], xv * wv[3]};
}

// block = (c/4 channel lanes) x (256 / (c/4) row lanes); per-channel partial sums, LDS tree, one atomic per channel and block
__global__ __launch_bounds__(256) void c1_outer_bwd_kernel(const float *__restrict__ x, const f32x4 *__restrict__ dy,
                                                           float *__restrict__ dw, long long rows, int c4,
                                                           long long rows_per_block) {
    __shared__ float red[256][4];
    const int cl = threadIdx.x % c4, rl = threadIdx.x / c4, rstep = 256 / c4;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    long long r1 = r0 + rows_per_block;
    r1 = r1 < rows ? r1 : rows;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (rl < rstep)
        for (long long r = r0 + rl; r < r1; r += rstep) {
            const float xv = x[r];
            const f32x4 d = dy[r * c4 + cl];
            s[0] += xv * d[0]; s[1] += xv * d[1]; s[2] += xv * d[2]; s[3] += xv * d[3];
        }
#pragma unroll
    for (int i = 0; i < 4; ++i) red[threadIdx.x][i] = s[i];
    __syncthreads();
    if (rl == 0) {
        for (int t = 1; t < rstep; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) s[i] += red[t * c4 + cl][i];
#pragma unroll
        for (int i = 0; i < 4; ++i) atomicAdd(dw + 4 * cl + i, s[i]);
    }
}
}  // namespace
}  // namespace epn

extern "C" int epn_conv1x1_c1_f32(const float *x, const float *w, float *y, long long rows, int cout, epn_stream_t stream) {
    if (rows < 0 || cout < 4 || cout % 4 || cout > 1024) return EPN_EINVAL;
    if (rows == 0) return 0;
    if (!x || !w || !y) return EPN_ENULL;
    const long long n4 = rows * (cout / 4);
    EPN_LAUNCH(epn::c1_outer_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, epn_stream(stream), x,
                       reinterpret_cast<const f32x4 *>(w), reinterpret_cast<f32x4 *>(y), n4, cout / 4);
    EPN_CHECK_LAUNCH();
    return 0;
}

extern "C" int epn_conv1x1_c1_bwd_weight_f32(const float *x, const float *grad_y, float *grad_w, long long rows, int cout,
                                              epn_stream_t stream) {
    if (rows < 0 || cout < 4 || cout % 4 || cout > 1024 || 256 % (cout / 4)) return EPN_EINVAL;
    if (!grad_w) return EPN_ENULL;
    hipStream_t st = epn_stream(stream);
    EPN_HIP(hipMemsetAsync(grad_w, 0, sizeof(float) * cout, st));
    if (rows == 0) return 0;
    if (!x || !grad_y) return EPN_ENULL;
    const long long blocks = rows < 1024 * 64 ? (rows + 63) / 64 : 1024;
    const long long rpb = (rows + blocks - 1) / blocks;
    EPN_LAUNCH(epn::c1_outer_bwd_kernel, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0, st, x,
                       reinterpret_cast<const f32x4 *>(grad_y), grad_w, rows, cout / 4, rpb);
    EPN_CHECK_LAUNCH();
    return 0;
}
