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
    int pair;                 // c == 32, 32-bit addressing: two points per task (see the kernels)
    // optional: leaky_relu(norm(.)) applied to the input rows as they are loaded ("norm on load": the block glue's first
    // normalisation folded into the basis change) -- nsums[g][c] = (sum x, sum x^2) as epn_chan_stats writes
    const float *nsums, *ngamma, *nbeta;
    float neps, nslope, ninv_rows;
    int ngroups;
    long long npts_per_group;
    // optional: per-channel (sum, sum of squares) over the na output rows of every point, pstats[pt][c][2] -- the block
    // partials of the statistics a norm after the transform needs (epn_stats_finish sums them), from the accumulators
    float *pstats;
    // optional (inverse transform of a gradient, 32-bit addressing only): dstat_x = the tensor x whose leaky_relu(norm(x)) fed
    // the forward transform (plain layout, as `out`).  The kernel then also writes pstats[pt][c][2] = (sum d, sum d * xhat)
    // over the point's na rows, d = dy * leaky'(norm(x)) -- the block partials of the norm's backward reduction
    // (glue.hip norm_act_bwd_reduce_kernel) taken from the accumulators that hold dy; nsums / ngamma / nbeta / neps /
    // nslope describe that norm and are NOT applied to the input rows in this mode
    const void *dstat_x;
    // optional (fp32 split form): max |output| of the whole launch as the bit pattern of a non-negative float, atomicMax'ed
    // into *amax (zeroed by the entry point) -- the scale source of the two-piece fp16 GEMMs that consume the output (gemm.h)
    unsigned *amax;
};

// per-lane normalisation of its 4 channels: n = (v - mean) * rstd * gamma + beta, leaky (the formula of glue.hip's
// norm_act_fwd_kernel, so folded and unfolded paths agree to the last bit)
struct SbNorm {
    float mean[4], rstd[4], ga[4], be[4];
    float slope;
};
__device__ __forceinline__ void sb_norm_load(const SbArgs &A, long long pt, int ch, SbNorm &N) {
    const int g = A.ngroups == 1 ? 0 : (int)(pt / A.npts_per_group);
    const float *s = A.nsums + ((size_t)g * A.c + ch) * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float m = s[2 * i] * A.ninv_rows;
        const float var = fmaxf(s[2 * i + 1] * A.ninv_rows - m * m, 0.0f);
        N.mean[i] = m;
        N.rstd[i] = rsqrtf(var + A.neps);
        N.ga[i] = A.ngamma ? A.ngamma[ch + i] : 1.0f;
        N.be[i] = A.nbeta ? A.nbeta[ch + i] : 0.0f;
    }
    N.slope = A.nslope;
}
__device__ __forceinline__ float sb_norm1(const SbNorm &N, int i, float v) {
    const float n = (v - N.mean[i]) * N.rstd[i] * N.ga[i] + N.be[i];
    return n > 0.0f ? n : n * N.slope;
}

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

typedef __bf16 sbf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 sb_ld(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ f32x4 sb_ld(const __bf16 *p) {
    const sbf16x4 v = *reinterpret_cast<const sbf16x4 *>(p);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
// acc[mt][nt][rr] = output row 16 mt + 4 j + rr, channel choff0 + nt of one point (rows >= na are exact zeros: M is zero
// padded): per-channel sums over the rows, of the values as stored (BF: rounded to bf16), lane groups j combined by DPP-free
// shuffles, lanes of group j = 0 write their 4 channels
template <bool BF>
__device__ __forceinline__ void sb_point_stats(const f32x4 (&acc)[4][4], float *__restrict__ pstats, long long pt, int c,
                                               int choff0, bool cval, int j) {
    float s1[4], s2[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                float v = acc[mt][nt][rr];
                if constexpr (BF) v = (float)(__bf16)v;
                a += v;
                b = fmaf(v, v, b);
            }
        a += __shfl_xor(a, 16, 64); b += __shfl_xor(b, 16, 64);
        a += __shfl_xor(a, 32, 64); b += __shfl_xor(b, 32, 64);
        s1[nt] = a; s2[nt] = b;
    }
    if (j == 0 && cval) {
        f32x4 *o = reinterpret_cast<f32x4 *>(pstats + ((size_t)pt * c + choff0) * 2);
        o[0] = f32x4{s1[0], s2[0], s1[1], s2[1]};
        o[1] = f32x4{s1[2], s2[2], s1[3], s2[3]};
    }
}

__device__ __forceinline__ void sb_st(float *p, f32x4 v) { *reinterpret_cast<f32x4 *>(p) = v; }
__device__ __forceinline__ void sb_st(__bf16 *p, f32x4 v) {
    *reinterpret_cast<sbf16x4 *>(p) = sbf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
}

// ---- row addressing, 32-bit form (round 4).  For BOTH layouts the byte offset of row r of point pt is
//     off(r, pt) = rb[r] + pt * rs[r]  (+ the lane's channel offset)
// plain:    rb = r c esz,                                   rs = na c esz
// spectral: rb = (base_r pts + (r - base_r)) c esz,         rs = d2_r c esz
// so the two tables (rb, rs) per side are built once per workgroup in LDS and a row costs ONE v_mad_u32_u24 + one add, issued
// as a buffer load / store with that 32-bit offset.  Round 3 evaluated the size_t expression per row -- two 64-bit
// multiplies each (192 v_mul_lo_u32 + 98 v_mad_u64_u32 per task in the ISA), ~1700 VALU instructions per (point, 64-channel
// block) against 64-256 MFMAs: the "HBM-bound streaming kernel" was bound by its address arithmetic (0.33-0.45 of the HBM
// roof).  Rows >= na and lanes without channels carry the offset 0x80000000: loads return zeros, stores are dropped by the
// buffer bounds check -- no branches (the two markers do not wrap when added).  Condition (launcher): the tensor is smaller than
// 2 GiB; larger ones keep the 64-bit path.
struct SbTables { unsigned rbi[64], rsi[64], rbo[64], rso[64]; };
constexpr unsigned SB_OOB = 0x80000000u;       // table entry of a row >= na
constexpr unsigned SB_CHO_OOB = 0x7fffff00u;   // channel offset of a lane without channels: OOB alone, and no 32-bit wrap with SB_OOB

template <int ESZ>
__device__ __forceinline__ void sb_make_tables(const SbArgs &A, SbTables &T) {
    const int r = threadIdx.x;
    if (r < 64) {
        unsigned bi = SB_OOB, si = 0u, bo = SB_OOB, so = 0u;
        if (r < A.na) {
            const unsigned base = (unsigned)A.blk[2 * r], d2 = (unsigned)A.blk[2 * r + 1];
            const unsigned row = (unsigned)A.c * ESZ;
            const unsigned spec_b = (unsigned)(((unsigned long long)base * (unsigned long long)A.pts + (r - base)) * row);
            const unsigned spec_s = d2 * row, plain_b = (unsigned)r * row, plain_s = (unsigned)A.na * row;
            bi = A.in_spec ? spec_b : plain_b; si = A.in_spec ? spec_s : plain_s;
            bo = A.out_spec ? spec_b : plain_b; so = A.out_spec ? spec_s : plain_s;
        }
        T.rbi[r] = bi; T.rsi[r] = si; T.rbo[r] = bo; T.rso[r] = so;
    }
}
__device__ __forceinline__ unsigned sb_off(const unsigned *rb, const unsigned *rs, int r, unsigned pt, unsigned cho) {
    return __umul24(pt, rs[r]) + rb[r] + cho;       // pt < 2^24, rs < 2^24 (launcher)
}
typedef unsigned sb_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned sb_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 sb_bld128(__amdgpu_buffer_rsrc_t rs, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
}
__device__ __forceinline__ sb_u32x2 sb_bld64(__amdgpu_buffer_rsrc_t rs, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0);
}
// (scalar offset literal 0: see store_g in inter_mfma.hip for what an SGPR offset costs on gfx950)
__device__ __forceinline__ void sb_bst(float *, __amdgpu_buffer_rsrc_t rs, unsigned off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(sb_u32x4, v), rs, off, 0, 0);
}
__device__ __forceinline__ void sb_bst(__bf16 *, __amdgpu_buffer_rsrc_t rs, unsigned off, f32x4 v) {
    const sbf16x4 b = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(sb_u32x2, b), rs, off, 0, 0);
}

// backward reduction of the norm in front of the forward transform, from the accumulators that hold dy (see SbArgs::dstat_x):
// x is read at the rows this lane is about to store (same offsets, plain layout), four rows at a time
template <bool BF>
__device__ __forceinline__ void sb_point_dstats(const f32x4 (&acc)[4][4], const SbArgs &A, const SbTables &tab,
                                                __amdgpu_buffer_rsrc_t rx, unsigned upt, unsigned cho, long long pt, int choff0,
                                                int choff, bool cval, int j) {
    SbNorm N;
    sb_norm_load(A, cval ? pt : 0, choff, N);
    float sa[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        f32x4 xv[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const unsigned off = sb_off(tab.rbo, tab.rso, 16 * mt + 4 * j + rr, upt, cho);     // rows >= na: zeros
            if constexpr (BF) {
                const sb_u32x2 w = sb_bld64(rx, off);
                xv[rr] = f32x4{__builtin_bit_cast(float, w[0] << 16), __builtin_bit_cast(float, w[0] & 0xffff0000u),
                               __builtin_bit_cast(float, w[1] << 16), __builtin_bit_cast(float, w[1] & 0xffff0000u)};
            } else {
                xv[rr] = sb_bld128(rx, off);
            }
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                float d = acc[mt][nt][rr];                      // rows >= na: exactly 0 (M is zero padded)
                if constexpr (BF) d = (float)(__bf16)d;        // the value the apply pass will read back
                const float xh = (xv[rr][nt] - N.mean[nt]) * N.rstd[nt];
                const float n = xh * N.ga[nt] + N.be[nt];
                const float dd = n > 0.0f ? d : d * N.slope;
                sa[nt] += dd;
                sb[nt] = fmaf(dd, xh, sb[nt]);
            }
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        sa[nt] += __shfl_xor(sa[nt], 16, 64); sb[nt] += __shfl_xor(sb[nt], 16, 64);
        sa[nt] += __shfl_xor(sa[nt], 32, 64); sb[nt] += __shfl_xor(sb[nt], 32, 64);
    }
    if (j == 0 && cval) {
        f32x4 *o = reinterpret_cast<f32x4 *>(A.pstats + ((size_t)pt * A.c + choff0) * 2);
        o[0] = f32x4{sa[0], sb[0], sa[1], sb[1]};
        o[1] = f32x4{sa[2], sb[2], sa[3], sb[3]};
    }
}

template <typename T, bool SMALL>
__global__ __launch_bounds__(64 * SB_WAVES) void so3_basis_kernel(SbArgs A) {
    __shared__ float Ms[64 * SB_LD];
    __shared__ int bs[64], d2s[64];
    __shared__ SbTables tab;
    for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x) {
        const int r = i >> 6, s = i & 63;
        Ms[r * SB_LD + s] = (r < A.na && s < A.na) ? A.M[r * A.na + s] : 0.0f;
    }
    if (threadIdx.x < 64) {
        const int f = threadIdx.x < A.na ? threadIdx.x : 0;
        bs[threadIdx.x] = A.blk[2 * f];
        d2s[threadIdx.x] = A.blk[2 * f + 1];
    }
    if constexpr (SMALL) sb_make_tables<(int)sizeof(T)>(A, tab);
    __syncthreads();
    const unsigned nbytes = SMALL ? (unsigned)(A.pts * A.na * A.c * (long long)sizeof(T)) : 0u;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(A.in), 0, (int)nbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(A.out, 0, (int)nbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rxs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(A.dstat_x ? A.dstat_x : A.in), 0, (int)nbytes, 0x00020000);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x = lane & 15, j = lane >> 4;
    const int ncb = (A.c + 63) >> 6;      // c % 32 == 0: the last block of a 32 (mod 64) width is half empty (lanes x >= 8)
    const int nst = A.na >> 2;        // contraction steps of 4 rows (na % 4 == 0, launcher)
    for (int it = 0; it < SB_TPW; ++it) {
        const long long task = ((long long)blockIdx.x * SB_WAVES + wave) * SB_TPW + it;
        // c == 32 (the first two blocks of the rotation / 3DMatch networks): a 64-channel block would leave lanes x >= 8 idle
        // (round 3: "half empty"); instead a task is TWO points, lanes x >= 8 carry the second one -- only the per-lane point
        // index of the loads, stores and statistics changes, M is the same for every column
        const bool pair = SMALL && A.pair;
        if (task >= (pair ? (A.pts + 1) >> 1 : A.pts * ncb)) return;
        const long long pt = pair ? 2 * task + (x >> 3) : task / ncb;
        const int cb = pair ? 0 : (int)(task - pt * ncb);
        const int choff0 = pair ? 4 * (x & 7) : 64 * cb + 4 * x;
        const bool cval = pair ? pt < A.pts : choff0 < A.c;        // this lane's 4 channels exist
        const int choff = cval ? choff0 : 0;

        auto row_addr = [&](int spec, int r) -> size_t {   // float offset of row r of this point
            if (spec) return ((size_t)bs[r] * A.pts + (size_t)pt * d2s[r] + (r - bs[r])) * A.c + choff;
            return ((size_t)pt * A.na + r) * A.c + choff;
        };

        f32x4 acc[4][4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const unsigned cho = cval ? (unsigned)choff0 * (unsigned)sizeof(T) : SB_CHO_OOB, upt = (unsigned)pt;
        f32x4 bv[16];
#pragma unroll
        for (int st = 0; st < 16; ++st)
            if (st < nst) {
                if constexpr (SMALL && sizeof(T) == 4) bv[st] = sb_bld128(rin, sb_off(tab.rbi, tab.rsi, 4 * st + j, upt, cho));
                else bv[st] = cval ? sb_ld(static_cast<const T *>(A.in) + row_addr(A.in_spec, 4 * st + j)) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        if (A.nsums && !A.dstat_x) {
            SbNorm N;
            sb_norm_load(A, cval ? pt : 0, choff, N);
#pragma unroll
            for (int st = 0; st < 16; ++st)
                if (st < nst) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float n = sb_norm1(N, i, bv[st][i]);
                        if constexpr (sizeof(T) == 2) n = (float)(__bf16)n;    // what the unfolded path stores and re-reads
                        bv[st][i] = n;
                    }
                }
        }
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
        if constexpr (SMALL) {
            if (A.dstat_x) sb_point_dstats<sizeof(T) == 2>(acc, A, tab, rxs, upt, cho, pt, choff0, choff, cval, j);
            else if (A.pstats) sb_point_stats<sizeof(T) == 2>(acc, A.pstats, pt, A.c, choff0, cval, j);
        } else if (A.pstats) sb_point_stats<sizeof(T) == 2>(acc, A.pstats, pt, A.c, choff0, cval, j);
        // acc[mt][nt][rr]: output row 16 mt + 4 j + rr, channel 4 x + nt
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int r = 16 * mt + 4 * j + rr;
                const f32x4 v = {acc[mt][0][rr], acc[mt][1][rr], acc[mt][2][rr], acc[mt][3][rr]};
                if constexpr (SMALL) sb_bst(static_cast<T *>(nullptr), rout, sb_off(tab.rbo, tab.rso, r, upt, cho), v);
                else if (r < A.na && cval) sb_st(static_cast<T *>(A.out) + row_addr(A.out_spec, r), v);
            }
        if constexpr (SMALL) asm volatile("s_nop 4" ::: "memory");   // store data registers are the next task's accumulators
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

template <bool SMALL>
__global__ __launch_bounds__(64 * SB_WAVES) void so3_basis_bf16_kernel(SbArgs A) {
    __shared__ __attribute__((aligned(16))) __bf16 Mh[64 * SBH_LD];
    __shared__ __attribute__((aligned(16))) __bf16 Ml[64 * SBH_LD];
    __shared__ int bs[64], d2s[64];
    __shared__ SbTables tab;
    if constexpr (SMALL) sb_make_tables<2>(A, tab);
    const unsigned nbytes = SMALL ? (unsigned)(A.pts * A.na * A.c * 2LL) : 0u;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(A.in), 0, (int)nbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(A.out, 0, (int)nbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rxs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(A.dstat_x ? A.dstat_x : A.in), 0, (int)nbytes, 0x00020000);
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
    const int ncb = (A.c + 63) >> 6;      // c % 32 == 0: the last block of a 32 (mod 64) width is half empty (lanes x >= 8)
    const __bf16 *in = static_cast<const __bf16 *>(A.in);
    __bf16 *out = static_cast<__bf16 *>(A.out);
    for (int it = 0; it < SB_TPW; ++it) {
        const long long task = ((long long)blockIdx.x * SB_WAVES + wave) * SB_TPW + it;
        // c == 32 (the first two blocks of the rotation / 3DMatch networks): a 64-channel block would leave lanes x >= 8 idle
        // (round 3: "half empty"); instead a task is TWO points, lanes x >= 8 carry the second one -- only the per-lane point
        // index of the loads, stores and statistics changes, M is the same for every column
        const bool pair = SMALL && A.pair;
        if (task >= (pair ? (A.pts + 1) >> 1 : A.pts * ncb)) return;
        const long long pt = pair ? 2 * task + (x >> 3) : task / ncb;
        const int cb = pair ? 0 : (int)(task - pt * ncb);
        const int choff0 = pair ? 4 * (x & 7) : 64 * cb + 4 * x;
        const bool cval = pair ? pt < A.pts : choff0 < A.c;        // this lane's 4 channels exist
        const int choff = cval ? choff0 : 0;
        auto row_addr = [&](int spec, int r) -> size_t {
            if (spec) return ((size_t)bs[r] * A.pts + (size_t)pt * d2s[r] + (r - bs[r])) * A.c + choff;
            return ((size_t)pt * A.na + r) * A.c + choff;
        };
        const unsigned cho = cval ? (unsigned)choff0 * 2u : SB_CHO_OOB, upt = (unsigned)pt;
        sbu32x2 raw[2][8];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int r = 32 * ks + 8 * j + e;
                if constexpr (SMALL) raw[ks][e] = __builtin_bit_cast(sbu32x2, sb_bld64(rin, sb_off(tab.rbi, tab.rsi, r, upt, cho)));
                else raw[ks][e] = (r < A.na && cval) ? *reinterpret_cast<const sbu32x2 *>(in + row_addr(A.in_spec, r)) : sbu32x2{0u, 0u};
            }
        if (A.nsums && !A.dstat_x) {
            SbNorm N;
            sb_norm_load(A, cval ? pt : 0, choff, N);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (32 * ks + 8 * j + e < A.na) {
                        const sbu32x2 w = raw[ks][e];
                        const float v0 = __builtin_bit_cast(float, w[0] << 16), v1 = __builtin_bit_cast(float, w[0] & 0xffff0000u);
                        const float v2 = __builtin_bit_cast(float, w[1] << 16), v3 = __builtin_bit_cast(float, w[1] & 0xffff0000u);
                        const sbf16x4 o = {(__bf16)sb_norm1(N, 0, v0), (__bf16)sb_norm1(N, 1, v1), (__bf16)sb_norm1(N, 2, v2),
                                           (__bf16)sb_norm1(N, 3, v3)};
                        raw[ks][e] = __builtin_bit_cast(sbu32x2, o);
                    }
                }
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
        if constexpr (SMALL) {
            if (A.dstat_x) sb_point_dstats<true>(acc, A, tab, rxs, upt, cho, pt, choff0, choff, cval, j);
            else if (A.pstats) sb_point_stats<true>(acc, A.pstats, pt, A.c, choff0, cval, j);
        } else if (A.pstats) sb_point_stats<true>(acc, A.pstats, pt, A.c, choff0, cval, j);
        // acc[mt][nt][rr]: output row 16 mt + 4 j + rr, channel 4 x + nt
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int r = 16 * mt + 4 * j + rr;
                const f32x4 v = {acc[mt][0][rr], acc[mt][1][rr], acc[mt][2][rr], acc[mt][3][rr]};
                if constexpr (SMALL) sb_bst(out, rout, sb_off(tab.rbo, tab.rso, r, upt, cho), v);
                else if (r < A.na && cval) sb_st(out + row_addr(A.out_spec, r), v);
            }
        if constexpr (SMALL) asm volatile("s_nop 4" ::: "memory");
    }
}

// fp32 feature storage, split form (see gemm_x3.hip): M in three bf16 planes in LDS, the input rows split without loss
// into three bf16 pieces in registers, six v_mfma_f32_16x16x32_bf16 per (M tile, channel tile, 32 rows) -- 192 MFMAs of
// 16 cycles per task instead of 240 fp32 ones of 32: the transform leaves the matrix pipe as the bound of this
// HBM-streaming kernel.  fp32 in, fp32 out, fp32 accuracy (the dropped piece products are < 2^-24 of |M||x|).
typedef unsigned sbu32x4s __attribute__((ext_vector_type(4)));
typedef __bf16 sbf16x2 __attribute__((ext_vector_type(2)));
typedef float sbf32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned sb_pack(float a, float b) {   // v_cvt_pk_bf16_f32
    const sbf32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, sbf16x2));
}
__device__ __forceinline__ void sb_split3(const float (&x)[8], sbf16x8 &h, sbf16x8 &m, sbf16x8 &l) {
    sbu32x4s H, M, L;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const unsigned hp = sb_pack(x[2 * p], x[2 * p + 1]);
        const float r0 = x[2 * p] - __builtin_bit_cast(float, hp << 16);
        const float r1 = x[2 * p + 1] - __builtin_bit_cast(float, hp & 0xffff0000u);
        const unsigned mp = sb_pack(r0, r1);
        H[p] = hp; M[p] = mp;
        L[p] = sb_pack(r0 - __builtin_bit_cast(float, mp << 16), r1 - __builtin_bit_cast(float, mp & 0xffff0000u));
    }
    h = __builtin_bit_cast(sbf16x8, H); m = __builtin_bit_cast(sbf16x8, M); l = __builtin_bit_cast(sbf16x8, L);
}

template <bool SMALL>
__global__ __launch_bounds__(64 * SB_WAVES) __attribute__((amdgpu_waves_per_eu(2, 2))) void so3_basis_x3_kernel(SbArgs A) {
    __shared__ __attribute__((aligned(16))) __bf16 Mp[3][64 * SBH_LD];
    __shared__ int bs[64], d2s[64];
    __shared__ SbTables tab;
    if constexpr (SMALL) sb_make_tables<4>(A, tab);
    const unsigned nbytes = SMALL ? (unsigned)(A.pts * A.na * A.c * 4LL) : 0u;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(A.in), 0, (int)nbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(A.out, 0, (int)nbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rxs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(A.dstat_x ? A.dstat_x : A.in), 0, (int)nbytes, 0x00020000);
    for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x) {
        const int r = i >> 6, q = i & 63;
        const float v = (r < A.na && q < A.na) ? A.M[r * A.na + q] : 0.0f;
        const __bf16 hi = (__bf16)v;
        const float r1 = v - (float)hi;
        const __bf16 mid = (__bf16)r1;
        Mp[0][r * SBH_LD + q] = hi;
        Mp[1][r * SBH_LD + q] = mid;
        Mp[2][r * SBH_LD + q] = (__bf16)(r1 - (float)mid);
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
    const int ncb = (A.c + 63) >> 6;      // c % 32 == 0: the last block of a 32 (mod 64) width is half empty (lanes x >= 8)
    const float *in = static_cast<const float *>(A.in);
    float *out = static_cast<float *>(A.out);
    float vmax = 0.0f;                    // max |output| of this lane's tasks (A.amax)
    for (int it = 0; it < SB_TPW; ++it) {
        const long long task = ((long long)blockIdx.x * SB_WAVES + wave) * SB_TPW + it;
        // c == 32 (the first two blocks of the rotation / 3DMatch networks): a 64-channel block would leave lanes x >= 8 idle
        // (round 3: "half empty"); instead a task is TWO points, lanes x >= 8 carry the second one -- only the per-lane point
        // index of the loads, stores and statistics changes, M is the same for every column
        const bool pair = SMALL && A.pair;
        if (task >= (pair ? (A.pts + 1) >> 1 : A.pts * ncb)) break;
        const long long pt = pair ? 2 * task + (x >> 3) : task / ncb;
        const int cb = pair ? 0 : (int)(task - pt * ncb);
        const int choff0 = pair ? 4 * (x & 7) : 64 * cb + 4 * x;
        const bool cval = pair ? pt < A.pts : choff0 < A.c;        // this lane's 4 channels exist
        const int choff = cval ? choff0 : 0;
        auto row_addr = [&](int spec, int r) -> size_t {
            if (spec) return ((size_t)bs[r] * A.pts + (size_t)pt * d2s[r] + (r - bs[r])) * A.c + choff;
            return ((size_t)pt * A.na + r) * A.c + choff;
        };
        const unsigned cho = cval ? (unsigned)choff0 * 4u : SB_CHO_OOB, upt = (unsigned)pt;
        f32x4 raw[2][8];          // row 32 ks + 8 j + e, channels 4x .. 4x+3
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int r = 32 * ks + 8 * j + e;
                if constexpr (SMALL) raw[ks][e] = sb_bld128(rin, sb_off(tab.rbi, tab.rsi, r, upt, cho));
                else raw[ks][e] = (r < A.na && cval) ? sb_ld(in + row_addr(A.in_spec, r)) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        if (A.nsums && !A.dstat_x) {
            SbNorm N;
            sb_norm_load(A, cval ? pt : 0, choff, N);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (32 * ks + 8 * j + e < A.na) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) raw[ks][e][i] = sb_norm1(N, i, raw[ks][e][i]);
                    }
        }
        f32x4 acc[4][4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            sbf16x8 bh[4], bm[4], bl[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const float v[8] = {raw[ks][0][nt], raw[ks][1][nt], raw[ks][2][nt], raw[ks][3][nt],
                                    raw[ks][4][nt], raw[ks][5][nt], raw[ks][6][nt], raw[ks][7][nt]};
                sb_split3(v, bh[nt], bm[nt], bl[nt]);
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int ao = (16 * mt + x) * SBH_LD + 32 * ks + 8 * j;
                const sbf16x8 ah = *reinterpret_cast<const sbf16x8 *>(&Mp[0][ao]);
                const sbf16x8 am = *reinterpret_cast<const sbf16x8 *>(&Mp[1][ao]);
                const sbf16x8 al = *reinterpret_cast<const sbf16x8 *>(&Mp[2][ao]);
#define EPN_SB_TERM(PA, PB)                                                                                  \
    _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                        \
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(PA, PB[nt], acc[mt][nt], 0, 0, 0)
                EPN_SB_TERM(ah, bl);
                EPN_SB_TERM(al, bh);
                EPN_SB_TERM(am, bm);
                EPN_SB_TERM(ah, bm);
                EPN_SB_TERM(am, bh);
                EPN_SB_TERM(ah, bh);
#undef EPN_SB_TERM
            }
        }
        if constexpr (SMALL) {
            if (A.dstat_x) sb_point_dstats<false>(acc, A, tab, rxs, upt, cho, pt, choff0, choff, cval, j);
            else if (A.pstats) sb_point_stats<false>(acc, A.pstats, pt, A.c, choff0, cval, j);
        } else if (A.pstats) sb_point_stats<false>(acc, A.pstats, pt, A.c, choff0, cval, j);
        if (A.amax) {                     // rows >= na and lanes without channels hold zeros: no masking needed
            float tmax = 0.0f;            // (fmaxf drops NaNs by itself)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    tmax = __builtin_fmaxf(__builtin_fmaxf(tmax, __builtin_fmaxf(__builtin_fabsf(acc[mt][nt][0]), __builtin_fabsf(acc[mt][nt][1]))),
                                           __builtin_fmaxf(__builtin_fabsf(acc[mt][nt][2]), __builtin_fabsf(acc[mt][nt][3])));
            // An infinite output is left OUT of the maximum, as absmax4 / split_rows2 leave non-finite elements out (advisor
            // finding, round 5: clamped to FLT_MAX it became the scale, 2^-113, and every finite element of the buffer
            // underflowed to zero in the consuming GEMMs; an fp32 GEMM poisons only the rows the element belongs to).  The
            // masked rescan runs only in a wave that saw one.
            if (__builtin_amdgcn_ballot_w64(tmax > 3.4028235e38f) != 0ull) {
                tmax = 0.0f;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) {
                            const float v = __builtin_fabsf(acc[mt][nt][rr]);
                            tmax = __builtin_fmaxf(tmax, v <= 3.4028235e38f ? v : 0.0f);
                        }
            }
            vmax = __builtin_fmaxf(vmax, tmax);
        }
        // acc[mt][nt][rr]: output row 16 mt + 4 j + rr, channel 4 x + nt
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int r = 16 * mt + 4 * j + rr;
                const f32x4 v = {acc[mt][0][rr], acc[mt][1][rr], acc[mt][2][rr], acc[mt][3][rr]};
                if constexpr (SMALL) sb_bst(out, rout, sb_off(tab.rbo, tab.rso, r, upt, cho), v);
                else if (r < A.na && cval) sb_st(out + row_addr(A.out_spec, r), v);
            }
        if constexpr (SMALL) asm volatile("s_nop 4" ::: "memory");
    }
    if (A.amax) {                         // one atomic per wave, and only when it would raise the value seen
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) vmax = __builtin_fmaxf(vmax, __shfl_xor(vmax, o, 64));
        const unsigned m = __builtin_bit_cast(unsigned, vmax);
        if (lane == 0 && m > __builtin_nontemporal_load(A.amax)) atomicMax(A.amax, m);
    }
}

// ---- the convolution weights in the block-diagonal basis:  What^rho[(j, c), (i, o)] = sum_k W[o, c, k] rho(g_k)[i, j]
// (so3_fourier.py).  Every training step re-expresses the weights of every IntraSO3Conv; as torch glue that was a [cout*cin,
// 12] x [60, 12]^T product on the generic kernel plus five slice / permute / clone chains per layer and their autograd
// transposes (~25 small launches per layer and direction).  One thread per (c, o) pair: its twelve weights in registers, the
// table R[f][k] = rho(g_k)[i, j] (f = base + i d + j) in LDS, 60 results.  TR: the same values in the transposed block layout
// What^T[(i, o), (j, c)] (the Bt operand of the forward GEMM), thread index c-fastest so that both layouts are written
// coalesced.  Flat buffers: block rho at offset base_rho * cin * cout.
constexpr int SW_KN_MAX = 16, SW_NA_MAX = 64;
template <bool TR, typename TO = float>      // TO = __bf16: the operand copies of a bf16 network (no cast / transpose-cast launches)
__global__ __launch_bounds__(256) void spectral_weights_kernel(const float *__restrict__ W, const float *__restrict__ R,
                                                               const int32_t *__restrict__ blk, int cout, int cin, int kn,
                                                               int na, TO *__restrict__ out) {
    __shared__ float Rs[SW_NA_MAX * SW_KN_MAX];
    __shared__ int bs[SW_NA_MAX], d2s[SW_NA_MAX];
    for (int i = threadIdx.x; i < na * kn; i += blockDim.x) Rs[(i / kn) * SW_KN_MAX + i % kn] = R[i];
    if ((int)threadIdx.x < na) { bs[threadIdx.x] = blk[2 * threadIdx.x]; d2s[threadIdx.x] = blk[2 * threadIdx.x + 1]; }
    __syncthreads();
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (long long)cin * cout) return;
    const int c = TR ? (int)(id % cin) : (int)(id / cout), o = TR ? (int)(id / cin) : (int)(id % cout);
    float w[SW_KN_MAX];
#pragma unroll
    for (int k = 0; k < SW_KN_MAX; ++k) w[k] = k < kn ? W[((size_t)o * cin + c) * kn + k] : 0.0f;
    // blockIdx.y: a slice of the spectral rows (a 64-channel layer is only 16 workgroups of (c, o) pairs, and one thread's
    // 60 rows in sequence were the whole kernel: 41-64 us per launch)
    const int fper = (na + (int)gridDim.y - 1) / (int)gridDim.y;
    const int f0 = (int)blockIdx.y * fper, f1 = min(f0 + fper, na);
    for (int f = f0; f < f1; ++f) {
        float v = 0.0f;
#pragma unroll
        for (int k = 0; k < SW_KN_MAX; ++k) v = fmaf(w[k], k < kn ? Rs[f * SW_KN_MAX + k] : 0.0f, v);
        const int base = bs[f], d2 = d2s[f];
        const int d = d2 == 1 ? 1 : (d2 == 4 ? 2 : (d2 == 9 ? 3 : (d2 == 16 ? 4 : (d2 == 25 ? 5 : (d2 == 36 ? 6 : 7)))));
        const int e = f - base, i = e / d, j = e - i * d;
        TO *ob = out + (size_t)base * cin * cout;
        if (TR) ob[((size_t)i * cout + o) * ((size_t)d * cin) + (size_t)j * cin + c] = (TO)v;
        else ob[((size_t)j * cin + c) * ((size_t)d * cout) + (size_t)i * cout + o] = (TO)v;
    }
}

// transpose of the above: dW[o, c, k] = sum_f dWhat^rho[(j, c), (i, o)] R[f][k]
__global__ __launch_bounds__(256) void spectral_weights_bwd_kernel(const float *__restrict__ gwhat, const float *__restrict__ R,
                                                                   const int32_t *__restrict__ blk, int cout, int cin, int kn,
                                                                   int na, float *__restrict__ gW) {
    __shared__ float Rs[SW_NA_MAX * SW_KN_MAX];
    __shared__ int bs[SW_NA_MAX], d2s[SW_NA_MAX];
    for (int i = threadIdx.x; i < na * kn; i += blockDim.x) Rs[(i / kn) * SW_KN_MAX + i % kn] = R[i];
    if ((int)threadIdx.x < na) { bs[threadIdx.x] = blk[2 * threadIdx.x]; d2s[threadIdx.x] = blk[2 * threadIdx.x + 1]; }
    __syncthreads();
    // four lanes per (c, o) pair, each a quarter of the spectral rows, combined by two shuffles per weight
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int part = (int)(tid & 3);
    const long long id0 = tid >> 2;
    const bool live = id0 < (long long)cin * cout;
    const long long id = live ? id0 : 0;
    const int c = (int)(id / cout), o = (int)(id % cout);
    float acc[SW_KN_MAX];
#pragma unroll
    for (int k = 0; k < SW_KN_MAX; ++k) acc[k] = 0.0f;
    const int fper = (na + 3) >> 2;
    for (int f = part * fper; f < min((part + 1) * fper, na); ++f) {
        const int base = bs[f], d2 = d2s[f];
        const int d = d2 == 1 ? 1 : (d2 == 4 ? 2 : (d2 == 9 ? 3 : (d2 == 16 ? 4 : (d2 == 25 ? 5 : (d2 == 36 ? 6 : 7)))));
        const int e = f - base, i = e / d, j = e - i * d;
        const float g = gwhat[(size_t)base * cin * cout + ((size_t)j * cin + c) * ((size_t)d * cout) + (size_t)i * cout + o];
#pragma unroll
        for (int k = 0; k < SW_KN_MAX; ++k) acc[k] = fmaf(g, k < kn ? Rs[f * SW_KN_MAX + k] : 0.0f, acc[k]);
    }
#pragma unroll
    for (int k = 0; k < SW_KN_MAX; ++k) {
        acc[k] += __shfl_xor(acc[k], 1, 64);
        acc[k] += __shfl_xor(acc[k], 2, 64);
    }
    if (live && part == 0) {
#pragma unroll
        for (int k = 0; k < SW_KN_MAX; ++k)
            if (k < kn) gW[((size_t)o * cin + c) * kn + k] = acc[k];
    }
}

}  // namespace
}  // namespace epn

using namespace epn;

static int sw_check(const void *a, const void *R, const int32_t *blocks, int cout, int cin, int kn, int na) {
    if (cout < 1 || cin < 1 || kn < 1 || kn > SW_KN_MAX || na < 1 || na > SW_NA_MAX) return EPN_EINVAL;
    if (!a || !R || !blocks) return EPN_ENULL;
    return 0;
}

extern "C" int epn_spectral_weights_f32(const float *W, const float *R, const int32_t *blocks, int cout, int cin, int kn,
                                        int na, float *what, float *what_t, epn_stream_t stream) {
    int rc = sw_check(W, R, blocks, cout, cin, kn, na);
    if (rc) return rc;
    if (!what && !what_t) return EPN_ENULL;
    const unsigned grid = (unsigned)(((long long)cin * cout + 255) / 256);
    const unsigned fy = grid >= 1024 ? 4 : (grid >= 256 ? 6 : 12);          // slices of the spectral rows
    if (what) EPN_LAUNCH(spectral_weights_kernel<false>, dim3(grid, fy), dim3(256), 0, epn_stream(stream), W, R, blocks, cout, cin, kn, na, what);
    if (what_t) EPN_LAUNCH(spectral_weights_kernel<true>, dim3(grid, fy), dim3(256), 0, epn_stream(stream), W, R, blocks, cout, cin, kn, na, what_t);
    EPN_CHECK_LAUNCH();
    return 0;
}

extern "C" int epn_spectral_weights_bf16(const float *W, const float *R, const int32_t *blocks, int cout, int cin, int kn,
                                         int na, void *what, void *what_t, epn_stream_t stream) {
    int rc = sw_check(W, R, blocks, cout, cin, kn, na);
    if (rc) return rc;
    if (!what && !what_t) return EPN_ENULL;
    const unsigned grid = (unsigned)(((long long)cin * cout + 255) / 256);
    const unsigned fy = grid >= 1024 ? 4 : (grid >= 256 ? 6 : 12);
    if (what) EPN_LAUNCH((spectral_weights_kernel<false, __bf16>), dim3(grid, fy), dim3(256), 0, epn_stream(stream), W, R, blocks, cout, cin, kn, na, static_cast<__bf16 *>(what));
    if (what_t) EPN_LAUNCH((spectral_weights_kernel<true, __bf16>), dim3(grid, fy), dim3(256), 0, epn_stream(stream), W, R, blocks, cout, cin, kn, na, static_cast<__bf16 *>(what_t));
    EPN_CHECK_LAUNCH();
    return 0;
}

extern "C" int epn_spectral_weights_bwd_f32(const float *grad_what, const float *R, const int32_t *blocks, int cout, int cin,
                                            int kn, int na, float *grad_W, epn_stream_t stream) {
    int rc = sw_check(grad_what, R, blocks, cout, cin, kn, na);
    if (rc) return rc;
    if (!grad_W) return EPN_ENULL;
    const unsigned grid = (unsigned)((4LL * cin * cout + 255) / 256);
    EPN_LAUNCH(spectral_weights_bwd_kernel, dim3(grid), dim3(256), 0, epn_stream(stream), grad_what, R, blocks, cout, cin, kn, na, grad_W);
    EPN_CHECK_LAUNCH();
    return 0;
}

struct SbNormHost {
    const float *sums, *gamma, *beta;
    int groups;
    long long pts_per_group;
    float eps, slope;
};

static int so3_basis_any(const void *in, const float *M, const int32_t *blocks, long long pts, int na, int c,
                        int in_spectral, int out_spectral, void *out, int bf16, epn_stream_t stream,
                        const SbNormHost *nh = nullptr, float *point_stats = nullptr, const void *dstat_x = nullptr,
                        float *amax_out = nullptr) {
    if (pts < 0 || na < 4 || na > 64 || (na & 3) || c < 32 || (c & 31)) return EPN_EINVAL;
    if (pts == 0) return 0;
    if (!in || !M || !blocks || !out) return EPN_ENULL;
    SbArgs A;
    A.in = in; A.M = M; A.blk = blocks; A.out = out; A.pts = pts; A.na = na; A.c = c; A.pstats = point_stats;
    A.in_spec = in_spectral; A.out_spec = out_spectral; A.dstat_x = dstat_x;
    A.amax = reinterpret_cast<unsigned *>(amax_out);
    if (amax_out) {
        if (bf16 != 2) return EPN_EINVAL;       // the maximum is produced by the fp32 split-form kernel only
        EPN_HIP(hipMemsetAsync(amax_out, 0, sizeof(float), epn_stream(stream)));
    }
    A.nsums = nullptr; A.ngamma = A.nbeta = nullptr; A.neps = 0.f; A.nslope = 0.f; A.ninv_rows = 0.f; A.ngroups = 1;
    A.npts_per_group = pts;
    if (nh) {
        if (!nh->sums) return EPN_ENULL;
        if ((in_spectral && !dstat_x) || nh->groups < 1 || nh->pts_per_group < 1 || (nh->groups > 1 && nh->groups * nh->pts_per_group != pts))
            return EPN_EINVAL;
        A.nsums = nh->sums; A.ngamma = nh->gamma; A.nbeta = nh->beta; A.neps = nh->eps; A.nslope = nh->slope;
        A.ngroups = nh->groups; A.npts_per_group = nh->groups > 1 ? nh->pts_per_group : pts;
        A.ninv_rows = 1.0f / ((float)A.npts_per_group * (float)na);
    }
    const long long bytes_ = pts * na * c * (bf16 == 1 ? 2LL : 4LL);
    A.pair = (c == 32 && bytes_ < 0x7fffff00LL && pts < (1LL << 24)) ? 1 : 0;
    const long long tasks = A.pair ? (pts + 1) / 2 : pts * ((c + 63) >> 6);
    const long long per_wg = (long long)SB_WAVES * SB_TPW;
    const dim3 grid((unsigned)((tasks + per_wg - 1) / per_wg));
    // 32-bit row offsets + buffer instructions when the tensor is below 2 GiB and the per-point strides fit 24 bits
    const long long bytes = pts * na * c * (bf16 == 1 ? 2LL : 4LL);
    if (dstat_x && (!(bytes < 0x7fffff00LL && pts < (1LL << 24)) || !point_stats || !nh || out_spectral)) return EPN_EINVAL;
    const bool small_t = bytes < 0x7fffff00LL && pts < (1LL << 24) && (long long)na * c * 4 < (1LL << 24);
    if (bf16 == 1) {
        if (small_t) EPN_LAUNCH(so3_basis_bf16_kernel<true>, grid, dim3(64 * SB_WAVES), 0, epn_stream(stream), A);
        else EPN_LAUNCH(so3_basis_bf16_kernel<false>, grid, dim3(64 * SB_WAVES), 0, epn_stream(stream), A);
    } else if (bf16 == 2) {                                                  // fp32, split form
        if (small_t) EPN_LAUNCH(so3_basis_x3_kernel<true>, grid, dim3(64 * SB_WAVES), 0, epn_stream(stream), A);
        else EPN_LAUNCH(so3_basis_x3_kernel<false>, grid, dim3(64 * SB_WAVES), 0, epn_stream(stream), A);
    } else {
        if (small_t) EPN_LAUNCH((so3_basis_kernel<float, true>), grid, dim3(64 * SB_WAVES), 0, epn_stream(stream), A);
        else EPN_LAUNCH((so3_basis_kernel<float, false>), grid, dim3(64 * SB_WAVES), 0, epn_stream(stream), A);
    }
    EPN_CHECK_LAUNCH();
    return 0;
}

extern "C" int epn_so3_basis_f32(const float *in, const float *M, const int32_t *blocks, long long pts, int na, int c,
                                 int in_spectral, int out_spectral, float *out, epn_stream_t stream) {
    return so3_basis_any(in, M, blocks, pts, na, c, in_spectral, out_spectral, out, 0, stream);
}
extern "C" int epn_so3_basis_split_f32(const float *in, const float *M, const int32_t *blocks, long long pts, int na, int c,
                                       int in_spectral, int out_spectral, float *out, epn_stream_t stream) {
    return so3_basis_any(in, M, blocks, pts, na, c, in_spectral, out_spectral, out, 2, stream);
}
// ... + max |out| into the device scalar *amax_out (the scale source of the two-piece fp16 GEMMs that read `out`, gemm.h)
extern "C" int epn_so3_basis_amax_split_f32(const float *in, const float *M, const int32_t *blocks, long long pts, int na, int c,
                                            int in_spectral, int out_spectral, float *out, float *amax_out, epn_stream_t stream) {
    if (!amax_out) return EPN_ENULL;
    return so3_basis_any(in, M, blocks, pts, na, c, in_spectral, out_spectral, out, 2, stream, nullptr, nullptr, nullptr, amax_out);
}
extern "C" int epn_so3_basis_norm_amax_split_f32(const float *in, const float *M, const int32_t *blocks, long long pts, int na,
                                                 int c, int out_spectral, float *out, const float *sums, int groups,
                                                 long long pts_per_group, const float *gamma, const float *beta, float eps,
                                                 float slope, float *amax_out, epn_stream_t stream) {
    if (!amax_out) return EPN_ENULL;
    const SbNormHost nh = {sums, gamma, beta, groups, pts_per_group, eps, slope};
    return so3_basis_any(in, M, blocks, pts, na, c, 0, out_spectral, out, 2, stream, &nh, nullptr, nullptr, amax_out);
}
extern "C" int epn_so3_basis_bf16(const void *in, const float *M, const int32_t *blocks, long long pts, int na, int c,
                                  int in_spectral, int out_spectral, void *out, epn_stream_t stream) {
    return so3_basis_any(in, M, blocks, pts, na, c, in_spectral, out_spectral, out, 1, stream);
}

// the change of basis + per-point block partials of the output's per-channel statistics (see SbArgs::pstats)
extern "C" int epn_so3_basis_stats_f32(const float *in, const float *M, const int32_t *blocks, long long pts, int na, int c,
                                       int in_spectral, int out_spectral, float *out, float *point_stats, epn_stream_t stream) {
    if (!point_stats) return EPN_ENULL;
    if (out_spectral) return EPN_EINVAL;      // statistics over the ANCHOR rows of a point: plain output layout only
    return so3_basis_any(in, M, blocks, pts, na, c, in_spectral, out_spectral, out, 0, stream, nullptr, point_stats);
}
extern "C" int epn_so3_basis_stats_split_f32(const float *in, const float *M, const int32_t *blocks, long long pts, int na,
                                             int c, int in_spectral, int out_spectral, float *out, float *point_stats,
                                             epn_stream_t stream) {
    if (!point_stats) return EPN_ENULL;
    if (out_spectral) return EPN_EINVAL;      // statistics over the ANCHOR rows of a point: plain output layout only
    return so3_basis_any(in, M, blocks, pts, na, c, in_spectral, out_spectral, out, 2, stream, nullptr, point_stats);
}
extern "C" int epn_so3_basis_stats_bf16(const void *in, const float *M, const int32_t *blocks, long long pts, int na, int c,
                                        int in_spectral, int out_spectral, void *out, float *point_stats, epn_stream_t stream) {
    if (!point_stats) return EPN_ENULL;
    if (out_spectral) return EPN_EINVAL;      // statistics over the ANCHOR rows of a point: plain output layout only
    return so3_basis_any(in, M, blocks, pts, na, c, in_spectral, out_spectral, out, 1, stream, nullptr, point_stats);
}

// Inverse transform of a spectral GRADIENT + the block partials of the backward reduction of the norm that preceded the
// forward transform (SbArgs::dstat_x): dy = U . grad (plain layout), point_dstats[pt][c][2] = (sum d, sum d xhat) over the
// point's anchors with d = dy * leaky'(norm(x)).  epn_norm_bwd_finish turns the partials into dsums / dgamma / dbeta.
// Tensors of 2 GiB or more: EPN_EINVAL (use epn_so3_basis_* + epn_norm_act_bwd_reduce_*).
static int so3_basis_dstats_any(const void *in, const float *M, const int32_t *blocks, long long pts, int na, int c, void *out,
                                const void *x_cl, const float *sums, int groups, long long pts_per_group, const float *gamma,
                                const float *beta, float eps, float slope, float *point_dstats, int bf16, epn_stream_t stream) {
    if (!x_cl || !point_dstats) return EPN_ENULL;
    const SbNormHost nh = {sums, gamma, beta, groups, pts_per_group, eps, slope};
    return so3_basis_any(in, M, blocks, pts, na, c, 1, 0, out, bf16, stream, &nh, point_dstats, x_cl);
}
extern "C" int epn_so3_basis_dstats_f32(const float *in, const float *M, const int32_t *blocks, long long pts, int na, int c,
                                        float *out, const float *x_cl, const float *sums, int groups, long long pts_per_group,
                                        const float *gamma, const float *beta, float eps, float slope, float *point_dstats,
                                        epn_stream_t stream) {
    return so3_basis_dstats_any(in, M, blocks, pts, na, c, out, x_cl, sums, groups, pts_per_group, gamma, beta, eps, slope,
                                point_dstats, 0, stream);
}
extern "C" int epn_so3_basis_dstats_split_f32(const float *in, const float *M, const int32_t *blocks, long long pts, int na,
                                              int c, float *out, const float *x_cl, const float *sums, int groups,
                                              long long pts_per_group, const float *gamma, const float *beta, float eps,
                                              float slope, float *point_dstats, epn_stream_t stream) {
    return so3_basis_dstats_any(in, M, blocks, pts, na, c, out, x_cl, sums, groups, pts_per_group, gamma, beta, eps, slope,
                                point_dstats, 2, stream);
}
extern "C" int epn_so3_basis_dstats_bf16(const void *in, const float *M, const int32_t *blocks, long long pts, int na, int c,
                                         void *out, const void *x_cl, const float *sums, int groups, long long pts_per_group,
                                         const float *gamma, const float *beta, float eps, float slope, float *point_dstats,
                                         epn_stream_t stream) {
    return so3_basis_dstats_any(in, M, blocks, pts, na, c, out, x_cl, sums, groups, pts_per_group, gamma, beta, eps, slope,
                                point_dstats, 1, stream);
}

// leaky_relu(norm(in)) applied on load, then the change of basis (in plain layout only): sums[g][c] = (sum x, sum x^2)
// over the (points of group g) x anchors rows, groups = 1 (BatchNorm2d) or the number of clouds (InstanceNorm2d)
extern "C" int epn_so3_basis_norm_f32(const float *in, const float *M, const int32_t *blocks, long long pts, int na, int c,
                                      int out_spectral, float *out, const float *sums, int groups, long long pts_per_group,
                                      const float *gamma, const float *beta, float eps, float slope, epn_stream_t stream) {
    const SbNormHost nh = {sums, gamma, beta, groups, pts_per_group, eps, slope};
    return so3_basis_any(in, M, blocks, pts, na, c, 0, out_spectral, out, 0, stream, &nh);
}
extern "C" int epn_so3_basis_norm_split_f32(const float *in, const float *M, const int32_t *blocks, long long pts, int na,
                                            int c, int out_spectral, float *out, const float *sums, int groups,
                                            long long pts_per_group, const float *gamma, const float *beta, float eps,
                                            float slope, epn_stream_t stream) {
    const SbNormHost nh = {sums, gamma, beta, groups, pts_per_group, eps, slope};
    return so3_basis_any(in, M, blocks, pts, na, c, 0, out_spectral, out, 2, stream, &nh);
}
extern "C" int epn_so3_basis_norm_bf16(const void *in, const float *M, const int32_t *blocks, long long pts, int na, int c,
                                       int out_spectral, void *out, const float *sums, int groups, long long pts_per_group,
                                       const float *gamma, const float *beta, float eps, float slope, epn_stream_t stream) {
    const SbNormHost nh = {sums, gamma, beta, groups, pts_per_group, eps, slope};
    return so3_basis_any(in, M, blocks, pts, na, c, 0, out_spectral, out, 1, stream, &nh);
}
