// Fused IntraSO3Conv kernels for gfx950 (f32 MFMA 16x16x4).
//
// out[col][o] = sum_k sum_c W[o][c*kn + k] X[pt*na + idx[a][k]][c],   col = pt*na + a
// (intra_so3conv_grouping + BasicSO3Conv, vgtk/vgtk/so3conv/functional.py:221-233, modules.py:48-55).
// The reference materialises X gathered 12x ([b,c,12,p,a], 3 GB per layer at B=32) before the matmul;
// here the anchor permutation is applied on the fly to the B-operand ADDRESS: a wave owns 16 output
// columns, for each anchor neighbour k every lane points at its own source row and the contraction
// over c runs as 16x16x4 MFMAs with W (re-packed k-major) staged through LDS for the 4 waves.
// The data gradient is the same kernel with (X, idx, W) := (dOut, idx^-1, W^T re-packed): each column
// of intra_idx is a permutation of the anchors, so the transpose is a gather, not a scatter.
#include <cstdlib>

#include "conv_internal.h"

namespace epn {
namespace {

constexpr int NW = 4;

struct IntraArgs {
    const float *X;        // [ncol][ci]
    const int32_t *idx;    // [na][kn]
    const float *Wp;       // [co][kn*ci]  (k-major)
    const float *gout;     // bwd_weight: dOut [ncol][co]
    float *out;            // [ncol][co]   (bwd_weight: dW [co][ci*kn])
    int na, kn, ci, co, wk;
    long long ncol;
    int col_tiles_per_wg;
};

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__global__ __launch_bounds__(64 * NW) void intra_gemm_kernel(IntraArgs A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Ws = smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x = lane & 15, j = lane >> 4;
    const int wss = A.wk + 4;
    const int MT = A.co >> 4;
    const int CK = A.kn * A.ci;
    const long long col0 = ((long long)epn_xcd_tile(blockIdx.x, gridDim.x) * NW + wave) * 16;
    long long colx = col0 + x;
    colx = colx < A.ncol ? colx : A.ncol - 1;
    const int ax = (int)(colx % A.na);
    const long long ptrow = colx - ax;  // pt*na
    const int32_t *irow = A.idx + ax * A.kn;

    f32x4 acc[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int sub = 0; sub < CK / A.wk; ++sub) {
        __syncthreads();
        {
            const int vec_per_row = A.wk >> 2;
            const float *src = A.Wp + (size_t)sub * A.wk;
            for (int i = threadIdx.x; i < A.co * vec_per_row; i += blockDim.x) {
                const int o = i / vec_per_row, v = i - o * vec_per_row;
                *reinterpret_cast<f32x4 *>(Ws + o * wss + 4 * v) =
                    *reinterpret_cast<const f32x4 *>(src + (size_t)o * CK + 4 * v);
            }
        }
        __syncthreads();
        for (int g = 0; g < (A.wk >> 4); ++g) {
            const int ck = sub * A.wk + 16 * g;  // k-major: ck = k*ci + c
            const int k = ck / A.ci, c0 = ck - k * A.ci;
            const f32x4 bf = *reinterpret_cast<const f32x4 *>(A.X + (size_t)(ptrow + irow[k]) * A.ci + c0 + 4 * j);
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
    if (col0 + x < A.ncol) {
#pragma unroll
        for (int m = 0; m < 16; ++m)
            if (m < MT) *reinterpret_cast<f32x4 *>(A.out + (col0 + x) * A.co + 16 * m + 4 * j) = acc[m];
    }
}

// dW[o][c*kn + k] = sum_col dOut[col][o] X[pt*na + idx[a][k]][c].  Wave = one k, one (<=64 x <=64) block of
// (o, c); the 4 waves of a workgroup take 4 consecutive k over the SAME columns (shared dOut lines in L1).
__global__ __launch_bounds__(64 * NW) void intra_bwd_weight_kernel(IntraArgs A) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x = lane & 15, j = lane >> 4;
    const int k = blockIdx.y * NW + wave;
    const int cblocks = (A.ci + 63) / 64;
    const int o0 = (blockIdx.z / cblocks) * 64, c0 = (blockIdx.z % cblocks) * 64;
    const int MO = (A.co - o0 < 64 ? A.co - o0 : 64) >> 4;
    const int NC = (A.ci - c0 < 64 ? A.ci - c0 : 64) >> 4;
    if (k >= A.kn) return;  // no barriers in this kernel

    f32x4 acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    const long long t0 = (long long)blockIdx.x * A.col_tiles_per_wg;
    for (int it = 0; it < A.col_tiles_per_wg; ++it) {
        const long long c_base = (t0 + it) * 16;
        if (c_base >= A.ncol) break;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const long long col = c_base + 4 * s + j;
            const bool ok = col < A.ncol;
            const long long cc = ok ? col : A.ncol - 1;
            const int a = (int)(cc % A.na);
            const long long src = cc - a + A.idx[a * A.kn + k];
            float af[4], bf[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) af[m] = (m < MO && ok) ? A.gout[cc * A.co + o0 + 16 * m + x] : 0.0f;
#pragma unroll
            for (int n = 0; n < 4; ++n) bf[n] = (n < NC && ok) ? A.X[src * A.ci + c0 + 16 * n + x] : 0.0f;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    if (m < MO && n < NC) acc[m][n] = mfma4(af[m], bf[n], acc[m][n]);
        }
    }
    // acc[m][n]: lane (x = c within tile, j), register r -> o = o0 + 16m + 4j + r
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
            if (m < MO && n < NC)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    atomicAdd(A.out + ((size_t)(o0 + 16 * m + 4 * j + r) * A.ci + c0 + 16 * n + x) * A.kn + k,
                              acc[m][n][r]);
}

// Same contraction with 16-byte operand loads (co, ci multiples of 64): lane (x, j) reads dOut[col][o0+4x..+3] and
// X[src][c0+4x..+3]; component t of those vectors feeds row/column tile t, i.e. tile t covers the channels
// {4*i + t}.  One dwordx4 per operand per contraction step instead of four dwords.
__global__ __launch_bounds__(64 * NW) void intra_bwd_weight_v4_kernel(IntraArgs A) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x = lane & 15, j = lane >> 4;
    const int cblocks = A.ci / 64;
    const int o0 = (blockIdx.z / cblocks) * 64, c0 = (blockIdx.z % cblocks) * 64;
    // one wave per anchor neighbour; with a single neighbour (the 1x1 skip convolution) the four waves split the
    // workgroup's column tiles instead of three of them idling
    const bool single = A.kn == 1;
    const int k = single ? 0 : blockIdx.y * NW + wave;
    if (k >= A.kn) return;  // no barriers in this kernel

    f32x4 acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    const long long t0 = (long long)blockIdx.x * A.col_tiles_per_wg;
    for (int it = single ? wave : 0; it < A.col_tiles_per_wg; it += single ? NW : 1) {
        const long long c_base = (t0 + it) * 16;
        if (c_base >= A.ncol) break;
        f32x4 af[4], bf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {      // all eight 16-byte loads of the tile in flight before the first MFMA
            const long long col = c_base + 4 * s + j;
            const bool ok = col < A.ncol;
            const long long cc = ok ? col : A.ncol - 1;
            const int a = (int)(cc % A.na);
            const long long src = cc - a + A.idx[a * A.kn + k];
            af[s] = *reinterpret_cast<const f32x4 *>(A.gout + cc * A.co + o0 + 4 * x);
            bf[s] = *reinterpret_cast<const f32x4 *>(A.X + src * A.ci + c0 + 4 * x);
            if (!ok) af[s] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = mfma4(af[s][m], bf[s][n], acc[m][n]);
    }
    // acc[m][n]: lane (x, j), register r -> o = o0 + 4*(4j + r) + m,  c = c0 + 4x + n
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                atomicAdd(A.out + ((size_t)(o0 + 4 * (4 * j + r) + m) * A.ci + c0 + 4 * x + n) * A.kn + k,
                          acc[m][n][r]);
}

// Weight gradient with point tiles in LDS (co, ci multiples of 64; kn <= 16; na <= 64).
// For one point, dOut_pt [na x co] and X_pt [na x ci] are all any anchor-neighbour k needs:
//   dW_k[o][c] += sum_a dOut_pt[a][o] * X_pt[idx[a][k]][c].
// A workgroup = kn waves (one per k) sharing a 64 x 64 block of (o, c): the two point tiles are staged ONCE per
// point (double-buffered, register-prefetched) and read kn times from LDS; the contraction runs over the anchors
// (na = 60 -> 15 MFMA steps, no padding).  12x less global traffic than one-wave-per-k with private loads.
constexpr int PT_LD = 68;   // LDS row stride (floats) of a 64-wide point tile

__global__ __launch_bounds__(768) void intra_bwd_weight_pt_kernel(IntraArgs A, long long npts, int pts_per_wg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int k = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave = anchor neighbour
    const int x = lane & 15, j = lane >> 4;
    const int cblocks = A.ci / 64;
    const int o0 = (blockIdx.y / cblocks) * 64, c0 = (blockIdx.y % cblocks) * 64;
    const int na4 = (A.na + 3) & ~3;                 // anchor rows padded to a multiple of 4 (pad rows are zero)
    const int tile = na4 * PT_LD;                    // floats per tile
    float *Ds = smem;                                // [2][na4][PT_LD]  dOut_pt[:, o0:o0+64]
    float *Fs = smem + 2 * tile;                     // [2][na4][PT_LD]  X_pt[:, c0:c0+64]
    const int nthreads = blockDim.x;
    const int nsteps = na4 >> 2;

    // source row (within the point) of this lane's contraction slot, per step: idx[4s + j][k]
    int srow[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int a = 4 * s + j;
        srow[s] = (s < nsteps && a < A.na) ? A.idx[a * A.kn + k] : 0;
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging: 2 tiles x na x 16 float4 per point
    const int nvec = 2 * A.na * 16;
    constexpr int SP = 4;   // nvec <= 1024-thread-independent bound: 2*64*16 = 2048 <= nthreads * SP for >= 512 threads
    f32x4 pre[SP];
    auto fetch = [&](long long pt) {
#pragma unroll
        for (int u = 0; u < SP; ++u) {
            const int i = threadIdx.x + nthreads * u;
            if (i < nvec) {
                const int which = i >= A.na * 16;
                const int r = (which ? i - A.na * 16 : i) >> 4, v = i & 15;
                const float *src = which ? A.X + ((size_t)pt * A.na + r) * A.ci + c0 + 4 * v
                                         : A.gout + ((size_t)pt * A.na + r) * A.co + o0 + 4 * v;
                pre[u] = *reinterpret_cast<const f32x4 *>(src);
            }
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int u = 0; u < SP; ++u) {
            const int i = threadIdx.x + nthreads * u;
            if (i < nvec) {
                const int which = i >= A.na * 16;
                const int r = (which ? i - A.na * 16 : i) >> 4, v = i & 15;
                float *dst = (which ? Fs : Ds) + buf * tile + r * PT_LD + 4 * v;
                *reinterpret_cast<f32x4 *>(dst) = pre[u];
            }
        }
    };
    // zero the pad rows of both buffers once
    for (int i = threadIdx.x; i < 2 * 2 * (na4 - A.na) * PT_LD; i += nthreads) {
        const int per = (na4 - A.na) * PT_LD;
        const int t = i / per, off = i - t * per;          // t: 0..3 -> (Ds,Fs) x (buf 0,1)
        ((t & 1) ? Fs : Ds)[(t >> 1) * tile + A.na * PT_LD + off] = 0.0f;
    }

    const long long pt0 = (long long)blockIdx.x * pts_per_wg;
    long long pt1 = pt0 + pts_per_wg;
    pt1 = pt1 < npts ? pt1 : npts;
    if (pt0 < pt1) fetch(pt0);
    int buf = 0;
    for (long long pt = pt0; pt < pt1; ++pt) {
        store(buf);
        if (pt + 1 < pt1) fetch(pt + 1);   // lands while this point's 16 * nsteps MFMAs run
        __syncthreads();
        if (k < A.kn) {
            const float *D = Ds + buf * tile + 4 * x;
            const float *F = Fs + buf * tile + 4 * x;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                if (s < nsteps) {
                    const f32x4 af = *reinterpret_cast<const f32x4 *>(D + (4 * s + j) * PT_LD);
                    const f32x4 bf = *reinterpret_cast<const f32x4 *>(F + srow[s] * PT_LD);
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int n = 0; n < 4; ++n) acc[m][n] = mfma4(af[m], bf[n], acc[m][n]);
                }
            }
        }
        buf ^= 1;   // the barrier of the next iteration orders these reads before buffer `buf` is rewritten
    }
    if (k < A.kn) {
        // acc[m][n]: lane (x, j), register r -> o = o0 + 4*(4j + r) + m,  c = c0 + 4x + n
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    atomicAdd(A.out + ((size_t)(o0 + 4 * (4 * j + r) + m) * A.ci + c0 + 4 * x + n) * A.kn + k,
                              acc[m][n][r]);
    }
}

// Wp[o][k*ci + c] = W[o][c*kn + k]                     (forward pack)
// Wq[c][k*co + o] = W[o][c*kn + k]                     (data-gradient pack: roles of o and c swapped)
__global__ void pack_w_kernel(const float *__restrict__ W, int co, int ci, int kn, int transpose,
                              float *__restrict__ dst) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)co * ci * kn) return;
    if (!transpose) {
        const int c = i % ci;
        const int k = (i / ci) % kn;
        const int o = i / ((size_t)ci * kn);
        dst[i] = W[((size_t)o * ci + c) * kn + k];
    } else {
        const int o = i % co;
        const int k = (i / co) % kn;
        const int c = i / ((size_t)co * kn);
        dst[i] = W[((size_t)o * ci + c) * kn + k];
    }
}

int pick_wk(int ck, int co) {
    int wk = 16;
    for (int cand = 16; cand <= ck; cand += 16)
        if (ck % cand == 0 && (size_t)co * (cand + 4) * sizeof(float) <= 32 * 1024) wk = cand;
    return wk;
}

int run_gemm(const float *X, const int32_t *idx, const float *Wp, long long ncol, int na, int kn, int ci, int co,
             float *out, hipStream_t st) {
    IntraArgs A;
    A.X = X; A.idx = idx; A.Wp = Wp; A.gout = nullptr; A.out = out;
    A.na = na; A.kn = kn; A.ci = ci; A.co = co; A.ncol = ncol; A.col_tiles_per_wg = 1;
    A.wk = pick_wk(kn * ci, co);
    const size_t lds = (size_t)co * (A.wk + 4) * sizeof(float);
    const unsigned grid = (unsigned)((ncol + 16 * NW - 1) / (16 * NW));
    EPN_LAUNCH(intra_gemm_kernel, dim3(grid), dim3(64 * NW), lds, st, A);
    EPN_CHECK_LAUNCH();
    return 0;
}

}  // namespace

bool intra_uses_mfma(int na, int kn, int cin, int cout) {
    (void)na; (void)kn;
    return cin % 16 == 0 && cout % 16 == 0 && cin >= 16 && cout >= 16 && cin <= 256 && cout <= 256;
}

size_t intra_workspace_floats(int kn, int cin, int cout) { return rnd64((size_t)cout * cin * kn); }

int launch_intra_fwd_mfma(const float *feats, const int32_t *iidx, const float *W, int b, int p, int na, int kn,
                          int cin, int cout, float *out, float *ws, hipStream_t st) {
    const size_t n = (size_t)cout * cin * kn;
    EPN_LAUNCH_AUX(pack_w_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, W, cout, cin, kn, 0, ws);
    EPN_CHECK_LAUNCH();
    return run_gemm(feats, iidx, ws, (long long)b * p * na, na, kn, cin, cout, out, st);
}

int launch_intra_bwd_data_mfma(const float *dOut, const int32_t *inv_idx, const float *W, int b, int p, int na,
                               int kn, int cin, int cout, float *dF, float *ws, hipStream_t st) {
    const size_t n = (size_t)cout * cin * kn;
    EPN_LAUNCH_AUX(pack_w_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, W, cout, cin, kn, 1, ws);
    EPN_CHECK_LAUNCH();
    return run_gemm(dOut, inv_idx, ws, (long long)b * p * na, na, kn, cout, cin, dF, st);
}

int launch_intra_bwd_weight_mfma(const float *feats, const float *dOut, const int32_t *iidx, int b, int p, int na,
                                 int kn, int cin, int cout, float *dW, hipStream_t st) {
    IntraArgs A;
    A.X = feats; A.idx = iidx; A.Wp = nullptr; A.gout = dOut; A.out = dW;
    A.na = na; A.kn = kn; A.ci = cin; A.co = cout; A.wk = 0;
    A.ncol = (long long)b * p * na;
    const long long tiles = (A.ncol + 15) / 16;
    const int kblocks = (kn + NW - 1) / NW;
    const int tblocks = ((cout + 63) / 64) * ((cin + 63) / 64);
    long long splits = (256 * 4 + kblocks * tblocks - 1) / (kblocks * tblocks);
    if (kn == 1) splits = (splits + NW - 1) / NW;   // all four waves of a workgroup work on the single neighbour: same
                                                    // number of partial results (final atomics) as one wave per workgroup
    if (splits > tiles) splits = tiles;
    if (splits < 1) splits = 1;
    A.col_tiles_per_wg = (int)((tiles + splits - 1) / splits);
    const unsigned gx = (unsigned)((tiles + A.col_tiles_per_wg - 1) / A.col_tiles_per_wg);
    if (cout % 64 == 0 && cin % 64 == 0 && kn >= 4 && kn <= 12 && na <= 64) {   // one wave per k
        const long long npts = (long long)b * p;
        const int blocks = (cout / 64) * (cin / 64);
        long long wgs = (256 * 2 + blocks - 1) / blocks;            // ~2 workgroups per CU in total
        if (wgs > npts) wgs = npts;
        const int per = (int)((npts + wgs - 1) / wgs);
        const unsigned gxp = (unsigned)((npts + per - 1) / per);
        const int na4 = (na + 3) & ~3;
        const size_t lds = (size_t)4 * na4 * PT_LD * sizeof(float);
        const int threads = 64 * kn < 512 ? 512 : (64 * kn > 768 ? 768 : 64 * kn);   // staging assumes >= 512 threads
        EPN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&intra_bwd_weight_pt_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        EPN_LAUNCH(intra_bwd_weight_pt_kernel, dim3(gxp, blocks), dim3(threads), lds, st, A, npts, per);
    } else if (cout % 64 == 0 && cin % 64 == 0)
        EPN_LAUNCH(intra_bwd_weight_v4_kernel, dim3(gx, kblocks, tblocks), dim3(64 * NW), 0, st, A);
    else
        EPN_LAUNCH(intra_bwd_weight_kernel, dim3(gx, kblocks, tblocks), dim3(64 * NW), 0, st, A);
    EPN_CHECK_LAUNCH();
    return 0;
}

}  // namespace epn
