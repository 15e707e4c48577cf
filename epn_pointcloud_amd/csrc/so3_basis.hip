// Change of anchor basis for IntraSO3Conv in its block-diagonal ("group Fourier") form, epn_pointcloud_amd/so3_fourier.py:
//     Out[pt][r][ch] = sum_s M[r][s] * In[pt][s][ch]          M: na x na (na = 60 anchors), one matrix for every point
// either side being the plain channels-last layout [pt][anchor][c] or the "spectral" layout, where row f of a point lives
// in the buffer of its irreducible block:  ((base_f * pts + pt * d2_f + (f - base_f)) * c + ch), so that every block is
// a dense row-major [pts * d][d * c] GEMM operand for the BLAS library.
// HBM-bound streaming kernel (17 flop/byte): a wave owns one (point, 64-channel block); M sits zero-padded in LDS as
// the MFMA A operand, the input rows are the B operand straight from global memory as one 16-byte load per lane and
// contraction step (lane x holds channels 4x..4x+3 = column x of four N tiles), 240 MFMAs, 16-byte stores.
#include "conv_internal.h"

namespace epn {
namespace {

constexpr int SB_WAVES = 4;
constexpr int SB_LD = 65;     // LDS row pitch of M (floats)
constexpr int SB_TPW = 4;     // (point, channel block) tasks per wave: amortises the 16 KB load of M

struct SbArgs {
    const void *in;           // T = float or __bf16 (feature storage); M and the arithmetic are fp32
    const float *M;
    const int32_t *blk;       // [na][2] = (base row of the irreducible block, d*d) per spectral row
    void *out;
    long long pts;
    int na, c, in_spec, out_spec;
};

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

typedef __bf16 sbf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 sb_ld(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ f32x4 sb_ld(const __bf16 *p) {
    const sbf16x4 v = *reinterpret_cast<const sbf16x4 *>(p);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
__device__ __forceinline__ void sb_st(float *p, f32x4 v) { *reinterpret_cast<f32x4 *>(p) = v; }
__device__ __forceinline__ void sb_st(__bf16 *p, f32x4 v) {
    *reinterpret_cast<sbf16x4 *>(p) = sbf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
}

template <typename T>
__global__ __launch_bounds__(64 * SB_WAVES) void so3_basis_kernel(SbArgs A) {
    __shared__ float Ms[64 * SB_LD];
    __shared__ int bs[64], d2s[64];
    for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x) {
        const int r = i >> 6, s = i & 63;
        Ms[r * SB_LD + s] = (r < A.na && s < A.na) ? A.M[r * A.na + s] : 0.0f;
    }
    if (threadIdx.x < 64) {
        const int f = threadIdx.x < A.na ? threadIdx.x : 0;
        bs[threadIdx.x] = A.blk[2 * f];
        d2s[threadIdx.x] = A.blk[2 * f + 1];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x = lane & 15, j = lane >> 4;
    const int ncb = A.c >> 6;
    const int nst = A.na >> 2;        // contraction steps of 4 rows (na % 4 == 0, launcher)
    for (int it = 0; it < SB_TPW; ++it) {
        const long long task = ((long long)blockIdx.x * SB_WAVES + wave) * SB_TPW + it;
        if (task >= A.pts * ncb) return;
        const long long pt = task / ncb;
        const int cb = (int)(task - pt * ncb);
        const int choff = 64 * cb + 4 * x;

        auto row_addr = [&](int spec, int r) -> size_t {   // float offset of row r of this point
            if (spec) return ((size_t)bs[r] * A.pts + (size_t)pt * d2s[r] + (r - bs[r])) * A.c + choff;
            return ((size_t)pt * A.na + r) * A.c + choff;
        };

        f32x4 acc[4][4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 bv[16];
#pragma unroll
        for (int st = 0; st < 16; ++st)
            if (st < nst) bv[st] = sb_ld(static_cast<const T *>(A.in) + row_addr(A.in_spec, 4 * st + j));
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            if (st < nst) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const float a = Ms[(16 * mt + x) * SB_LD + 4 * st + j];
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma4(a, bv[st][nt], acc[mt][nt]);
                }
            }
        }
        // acc[mt][nt][rr]: output row 16 mt + 4 j + rr, channel 4 x + nt
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int r = 16 * mt + 4 * j + rr;
                if (r < A.na) {
                    const f32x4 v = {acc[mt][0][rr], acc[mt][1][rr], acc[mt][2][rr], acc[mt][3][rr]};
                    sb_st(static_cast<T *>(A.out) + row_addr(A.out_spec, r), v);
                }
            }
    }
}

// bf16 feature storage: the transform on the bf16 matrix pipe.  M is split into two bf16 terms (hi + lo: 2^-17 relative,
// i.e. fp32-grade for an orthogonal 60 x 60 matrix) that sit in LDS as the A operands of v_mfma_f32_16x16x32_bf16; the
// input rows are the B operand without any conversion: a lane loads 8 bytes (channels 4x..4x+3) of 16 rows and
// byte-permutes them into the four N tiles' fragments (contraction slot 8j+e <-> row 32ks + 8j + e).  64 MFMAs per
// (point, 64-channel block) instead of 240 fp32 ones, 3 waves / SIMD instead of 1: 190 -> ~75 us per call (HBM-bound).
constexpr int SBH_LD = 72;    // bf16 per LDS row of M (64 + 8: 144-byte pitch, 16-byte aligned)
typedef __bf16 sbf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned sbu32x2 __attribute__((ext_vector_type(2)));
typedef unsigned sbu32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(64 * SB_WAVES) void so3_basis_bf16_kernel(SbArgs A) {
    __shared__ __attribute__((aligned(16))) __bf16 Mh[64 * SBH_LD];
    __shared__ __attribute__((aligned(16))) __bf16 Ml[64 * SBH_LD];
    __shared__ int bs[64], d2s[64];
    for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x) {
        const int r = i >> 6, q = i & 63;
        const float m = (r < A.na && q < A.na) ? A.M[r * A.na + q] : 0.0f;
        const __bf16 hi = (__bf16)m;
        Mh[r * SBH_LD + q] = hi;
        Ml[r * SBH_LD + q] = (__bf16)(m - (float)hi);
    }
    if (threadIdx.x < 64) {
        const int f = threadIdx.x < A.na ? threadIdx.x : 0;
        bs[threadIdx.x] = A.blk[2 * f];
        d2s[threadIdx.x] = A.blk[2 * f + 1];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x = lane & 15, j = lane >> 4;
    const int ncb = A.c >> 6;
    const __bf16 *in = static_cast<const __bf16 *>(A.in);
    __bf16 *out = static_cast<__bf16 *>(A.out);
    for (int it = 0; it < SB_TPW; ++it) {
        const long long task = ((long long)blockIdx.x * SB_WAVES + wave) * SB_TPW + it;
        if (task >= A.pts * ncb) return;
        const long long pt = task / ncb;
        const int cb = (int)(task - pt * ncb);
        const int choff = 64 * cb + 4 * x;
        auto row_addr = [&](int spec, int r) -> size_t {
            if (spec) return ((size_t)bs[r] * A.pts + (size_t)pt * d2s[r] + (r - bs[r])) * A.c + choff;
            return ((size_t)pt * A.na + r) * A.c + choff;
        };
        sbu32x2 raw[2][8];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int r = 32 * ks + 8 * j + e;
                raw[ks][e] = r < A.na ? *reinterpret_cast<const sbu32x2 *>(in + row_addr(A.in_spec, r)) : sbu32x2{0u, 0u};
            }
        f32x4 acc[4][4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            sbf16x8 b[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                sbu32x4 w;
#pragma unroll
                for (int d = 0; d < 4; ++d)      // slots e = 2d, 2d+1: the nt-th bf16 of two rows' 8-byte loads
                    w[d] = __builtin_amdgcn_perm(raw[ks][2 * d + 1][nt >> 1], raw[ks][2 * d][nt >> 1],
                                                 (nt & 1) ? 0x07060302u : 0x05040100u);
                b[nt] = __builtin_bit_cast(sbf16x8, w);
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int ao = (16 * mt + x) * SBH_LD + 32 * ks + 8 * j;
                const sbf16x8 ah = *reinterpret_cast<const sbf16x8 *>(Mh + ao);
                const sbf16x8 al = *reinterpret_cast<const sbf16x8 *>(Ml + ao);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, b[nt], acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, b[nt], acc[mt][nt], 0, 0, 0);
                }
            }
        }
        // acc[mt][nt][rr]: output row 16 mt + 4 j + rr, channel 4 x + nt
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int r = 16 * mt + 4 * j + rr;
                if (r < A.na) {
                    const f32x4 v = {acc[mt][0][rr], acc[mt][1][rr], acc[mt][2][rr], acc[mt][3][rr]};
                    sb_st(out + row_addr(A.out_spec, r), v);
                }
            }
    }
}

}  // namespace
}  // namespace epn

using namespace epn;

static int so3_basis_any(const void *in, const float *M, const int32_t *blocks, long long pts, int na, int c,
                        int in_spectral, int out_spectral, void *out, int bf16, epn_stream_t stream) {
    if (pts < 0 || na < 4 || na > 64 || (na & 3) || c < 64 || (c & 63)) return EPN_EINVAL;
    if (pts == 0) return 0;
    if (!in || !M || !blocks || !out) return EPN_ENULL;
    SbArgs A;
    A.in = in; A.M = M; A.blk = blocks; A.out = out; A.pts = pts; A.na = na; A.c = c;
    A.in_spec = in_spectral; A.out_spec = out_spectral;
    const long long tasks = pts * (c >> 6);
    const long long per_wg = (long long)SB_WAVES * SB_TPW;
    const dim3 grid((unsigned)((tasks + per_wg - 1) / per_wg));
    if (bf16) hipLaunchKernelGGL(so3_basis_bf16_kernel, grid, dim3(64 * SB_WAVES), 0, epn_stream(stream), A);
    else hipLaunchKernelGGL(so3_basis_kernel<float>, grid, dim3(64 * SB_WAVES), 0, epn_stream(stream), A);
    EPN_CHECK_LAUNCH();
    return 0;
}

extern "C" int epn_so3_basis_f32(const float *in, const float *M, const int32_t *blocks, long long pts, int na, int c,
                                 int in_spectral, int out_spectral, float *out, epn_stream_t stream) {
    return so3_basis_any(in, M, blocks, pts, na, c, in_spectral, out_spectral, out, 0, stream);
}
extern "C" int epn_so3_basis_bf16(const void *in, const float *M, const int32_t *blocks, long long pts, int na, int c,
                                  int in_spectral, int out_spectral, void *out, epn_stream_t stream) {
    return so3_basis_any(in, M, blocks, pts, na, c, in_spectral, out_spectral, out, 1, stream);
}
