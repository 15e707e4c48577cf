// InterSO3Conv for a single input channel (cin = 1): the first layer of every shipped model, whose input is the
// occupancy feature of get_occupancy_features (vgtk/vgtk/so3conv/functional.py:25-44).  With one channel the layer
// is not matrix work (24 grouped values and a [cout x 24] weight per column): it is bound by the weight generation
// on the VALU and by the output write.  One lane owns one column (b, p, a):
//   G[k]   = sum_n F[idx[n], a] * relu(1 - |g_n - R_a kappa_k|^2 / sigma)        (functional.py:190-200, expanded)
//   out[o] = sum_k W[o][k] * G[k]                                                  (modules.py:52)
// The weight gradient reduces dOut (x) G over all columns with MFMAs (M = kernel point, N = output channel,
// contraction = columns), the grouped values passing through LDS once.
#include "conv_internal.h"

namespace epn {
namespace {

struct C1Args {
    const float *xyz, *new_xyz;
    const int32_t *idx;
    const float *rk;      // [na][ks][3]
    const float *feats;   // [b][p1][na]  (cin = 1)
    const float *W;       // fwd: [cout][ks]
    const float *gout;    // bwd: [ncol][cout]
    float *out;           // fwd: [ncol][cout];  bwd: dW [cout][ks]
    float *gsave;         // fwd, optional: the grouped values G[ncol][ks] kept for the weight gradient
    const float *gload;   // bwd, optional: G from the forward pass (no weight generation in the backward pass)
    float sigma_inv;
    int p1, p2, nn, na, ks, cout;
    long long ncol;
    int groups_per_wg;
};

// grouped values of one column into g[0..ks)
__device__ __forceinline__ void group_column(const C1Args &A, long long col, float (&g)[EPN_KS_MAX]) {
    const int a = (int)(col % A.na);
    const long long pt = col / A.na;
    const int bb = (int)(pt / A.p2), pp = (int)(pt - (long long)bb * A.p2);
    // expanded form of relu(1 - |g_n - R_a kappa_k|^2 / sigma) (the one the MFMA kernels use, inter_mfma.hip):
    //   alpha_n + beta_k + (2/sigma) g_n . (R_a kappa_k),  alpha_n = 1 - |g_n|^2/sigma,  beta_k = -|kappa_k|^2/sigma
    // 5 VALU operations per (neighbour, kernel point) instead of 9; differs from the literal form by fp32 rounding
    // (<= 5e-7 on w)
    // kernel points in PAIRS: the per-(neighbour, kernel point) arithmetic below is written on two-float vectors so that
    // hipcc emits v_pk_fma_f32 / v_pk_add_f32 (two fp32 operations per lane and instruction on this part): 3.5 instead of 6
    // VALU instructions per weight -- this kernel is nothing but that loop (15k instructions per column at K = 128)
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    constexpr int KP = EPN_KS_MAX / 2;
    f32x2_t rx[KP], ry[KP], rz[KP], beta[KP], g2[KP];
    const float two_si = 2.0f * A.sigma_inv;
#pragma unroll
    for (int k = 0; k < EPN_KS_MAX; ++k) {
        const float *e = A.rk + ((size_t)a * A.ks + (k < A.ks ? k : 0)) * 3;
        const float x = e[0], y = e[1], z = e[2];
        beta[k >> 1][k & 1] = -(x * x + y * y + z * z) * A.sigma_inv;
        rx[k >> 1][k & 1] = two_si * x; ry[k >> 1][k & 1] = two_si * y; rz[k >> 1][k & 1] = two_si * z;
        g2[k >> 1][k & 1] = 0.f;
    }
    const int kpairs = (A.ks + 1) >> 1;          // an odd ks: the pad slot repeats kernel point 0 and is dropped below
    const int32_t *row = A.idx + ((size_t)bb * A.p2 + pp) * A.nn;
    const float *s = A.xyz + (size_t)bb * 3 * A.p1;
    const float *c = A.new_xyz + (size_t)bb * 3 * A.p2;
    const float cx = c[pp], cy = c[A.p2 + pp], cz = c[2 * A.p2 + pp];
    const float *f = A.feats + ((size_t)bb * A.p1) * A.na + a;
    // Neighbours in batches of four (round 4): index -> coordinates -> feature is a chain of dependent loads, and with one
    // neighbour per iteration (and a `continue` the compiler cannot hoist loads across) every neighbour cost a full L2 / HBM
    // round trip per wave -- at two or three waves per SIMD the K = 64 / 128 first layers of the rotation / 3DMatch networks
    // were bound by that latency (1.0 / 1.75 ms), not by the 3.5 VALU instructions per weight.  A shadow index reads row 0
    // with feature value 0 instead of being skipped: same sum.
    constexpr int NB4 = 4;
    for (int n0 = 0; n0 < A.nn; n0 += NB4) {
        int qv[NB4];
        bool okv[NB4];
#pragma unroll
        for (int u = 0; u < NB4; ++u) {
            const int q = n0 + u < A.nn ? row[n0 + u] : -1;
            okv[u] = q >= 0 && q < A.p1;       // shadow index: zero feature row (spconv/functional.py:91-95)
            qv[u] = okv[u] ? q : 0;
        }
        float gxv[NB4], gyv[NB4], gzv[NB4], fvv[NB4];
#pragma unroll
        for (int u = 0; u < NB4; ++u) {
            gxv[u] = s[qv[u]]; gyv[u] = s[A.p1 + qv[u]]; gzv[u] = s[2 * A.p1 + qv[u]];
            fvv[u] = f[(size_t)qv[u] * A.na];
        }
#pragma unroll
        for (int u = 0; u < NB4; ++u) {
        const float gx = gxv[u] - cx, gy = gyv[u] - cy, gz = gzv[u] - cz;
        const float alpha = 1.0f - (gx * gx + gy * gy + gz * gz) * A.sigma_inv;
        const float fv = okv[u] ? fvv[u] : 0.0f;
        const f32x2_t a2 = {alpha, alpha}, gx2 = {gx, gx}, gy2 = {gy, gy}, gz2 = {gz, gz}, fv2 = {fv, fv};
#pragma unroll
        for (int kp = 0; kp < KP; ++kp) {
            if (kp < kpairs) {
                // same operation order per element as the scalar form: ((alpha + beta) + gx rx) + gy ry) + gz rz
                f32x2_t sv = (a2 + beta[kp]) + gx2 * rx[kp];
                sv = sv + gy2 * ry[kp];
                sv = sv + gz2 * rz[kp];
                // (scalar temporaries: __builtin_bit_cast applied to a vector ELEMENT reads element 0 for both -- hipcc)
                const float e0 = sv[0], e1 = sv[1];
                const int s0 = __float_as_int(e0), s1 = __float_as_int(e1);
                const f32x2_t w = {__int_as_float(s0 > 0 ? s0 : 0), __int_as_float(s1 > 0 ? s1 : 0)};
                g2[kp] = g2[kp] + fv2 * w;
            }
        }
        }
    }
#pragma unroll
    for (int k = 0; k < EPN_KS_MAX; ++k) g[k] = k < A.ks ? g2[k >> 1][k & 1] : 0.f;
}

__global__ __launch_bounds__(256) void inter_c1_fwd_kernel(C1Args A) {
    extern __shared__ __attribute__((aligned(16))) float Ws[];   // [cout][ks]
    for (int i = threadIdx.x; i < A.cout * A.ks; i += blockDim.x) Ws[i] = A.W[i];
    __syncthreads();
    const long long col = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= A.ncol) return;
    float g[EPN_KS_MAX];
    group_column(A, col, g);
    if (A.gsave) {   // 96 bytes per lane, consecutive lanes consecutive rows: the weight gradient re-reads this instead of
                     // regenerating 24 x K weights per column (the backward kernel WAS that loop a second time)
        float *gs = A.gsave + col * A.ks;
        if ((A.ks & 3) == 0) {
#pragma unroll
            for (int k = 0; k < EPN_KS_MAX; k += 4)
                if (k < A.ks) *reinterpret_cast<f32x4 *>(gs + k) = f32x4{g[k], g[k + 1], g[k + 2], g[k + 3]};
        } else {
#pragma unroll
            for (int k = 0; k < EPN_KS_MAX; ++k)
                if (k < A.ks) gs[k] = g[k];
        }
    }
    float *o = A.out + col * A.cout;
    for (int o4 = 0; o4 < A.cout; o4 += 4) {   // cout % 4 == 0 (launcher)
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < EPN_KS_MAX; ++k) {
            if (k < A.ks) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] += Ws[(o4 + r) * A.ks + k] * g[k];   // LDS broadcast reads
            }
        }
        *reinterpret_cast<f32x4 *>(o + o4) = acc;
    }
}

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// dW[o][k] = sum_col dOut[col][o] * G[col][k];  4 waves, each wave contracts its own 64 columns per group.
__global__ __launch_bounds__(256) void inter_c1_bwd_weight_kernel(C1Args A) {
    __shared__ float Gs[256][EPN_KS_MAX + 1];   // [column of the group][kernel point], zero padded to 32
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int x = lane & 15, j = lane >> 4;
    const int NTO = A.cout >> 4;   // <= 4 (launcher)
    f32x4 acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int it = 0; it < A.groups_per_wg; ++it) {
        const long long base = ((long long)blockIdx.x * A.groups_per_wg + it) * 256;
        if (base >= A.ncol) break;
        const long long col = base + threadIdx.x;
        if (A.gload) {
            // the wave's 64 rows of saved G are one contiguous run of 64 ks floats: coalesced loads, scattered to the tile
            const long long w0 = base + wave * 64;
            const long long wn = A.ncol - w0 < 64 ? A.ncol - w0 : 64;       // rows of this wave inside the tensor (may be <= 0)
            const float *src = A.gload + w0 * A.ks;
            const int total = wn > 0 ? (int)wn * A.ks : 0;
            if ((A.ks & 3) == 0 && ((reinterpret_cast<size_t>(src) & 15) == 0)) {
                for (int e = 4 * lane; e < 64 * A.ks; e += 256) {
                    const f32x4 v = e < total ? *reinterpret_cast<const f32x4 *>(src + e) : f32x4{0.f, 0.f, 0.f, 0.f};
                    const int r = e / A.ks, k0 = e - r * A.ks;
#pragma unroll
                    for (int q = 0; q < 4; ++q) Gs[wave * 64 + r][k0 + q] = v[q];
                }
            } else {
                for (int e = lane; e < 64 * A.ks; e += 64) {
                    const int r = e / A.ks;
                    Gs[wave * 64 + r][e - r * A.ks] = e < total ? src[e] : 0.f;
                }
            }
            for (int e = lane; e < 64 * (EPN_KS_MAX - A.ks); e += 64) {       // zero padding up to 32 kernel points
                const int r = e / (EPN_KS_MAX - A.ks);
                Gs[wave * 64 + r][A.ks + e - r * (EPN_KS_MAX - A.ks)] = 0.f;
            }
        } else {
            float g[EPN_KS_MAX];
            if (col < A.ncol) {
                group_column(A, col, g);
            } else {
#pragma unroll
                for (int k = 0; k < EPN_KS_MAX; ++k) g[k] = 0.f;
            }
            // only this wave reads the rows it writes: wave-level ordering is enough
#pragma unroll
            for (int k = 0; k < EPN_KS_MAX; ++k) Gs[threadIdx.x][k] = k < A.ks ? g[k] : 0.f;
        }
        __builtin_amdgcn_wave_barrier();
        const long long wbase = base + wave * 64;
#pragma unroll 4
        for (int s = 0; s < 16; ++s) {   // contraction step: columns wbase + 4s + j
            const long long cc = wbase + 4 * s + j;
            const bool ok = cc < A.ncol;
            float bf[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) bf[n] = (n < NTO && ok) ? A.gout[cc * A.cout + 16 * n + x] : 0.0f;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const float af = Gs[wave * 64 + 4 * s + j][16 * m + x];
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    if (n < NTO) acc[m][n] = mfma4(af, bf[n], acc[m][n]);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    // acc[m][n]: lane (x = o within tile n, j), register r -> k = 16m + 4j + r
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
            if (n < NTO)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = 16 * m + 4 * j + r;
                    if (k < A.ks) atomicAdd(A.out + (size_t)(16 * n + x) * A.ks + k, acc[m][n][r]);
                }
}

C1Args make_c1(const epn_inter_desc *d, const float *rk) {
    C1Args A;
    A.xyz = d->xyz; A.new_xyz = d->new_xyz; A.idx = d->ball_idx; A.rk = rk;
    A.feats = nullptr; A.W = nullptr; A.gout = nullptr; A.out = nullptr; A.gsave = nullptr; A.gload = nullptr;
    A.sigma_inv = 1.0f / d->sigma;
    A.p1 = d->p1; A.p2 = d->p2; A.nn = d->nn; A.na = d->na; A.ks = d->ks; A.cout = d->cout;
    A.ncol = (long long)d->b * d->p2 * d->na;
    A.groups_per_wg = 1;
    return A;
}

}  // namespace

bool inter_c1_fwd_ok(const epn_inter_desc *d) {
    return d->cin == 1 && !d->dense_w && d->ks <= EPN_KS_MAX && d->cout % 4 == 0 &&
           (size_t)d->cout * d->ks * sizeof(float) <= 48 * 1024;
}

bool inter_c1_bwd_weight_ok(const epn_inter_desc *d) {
    return d->cin == 1 && !d->dense_w && d->ks <= EPN_KS_MAX && d->cout % 16 == 0 && d->cout <= 64;
}

int launch_inter_c1_fwd(const epn_inter_desc *d, const float *rk, const float *feats, const float *W, float *out,
                        hipStream_t st, float *grouped_save) {
    C1Args A = make_c1(d, rk);
    A.feats = feats; A.W = W; A.out = out; A.gsave = grouped_save;
    const unsigned grid = (unsigned)((A.ncol + 255) / 256);
    EPN_LAUNCH(inter_c1_fwd_kernel, dim3(grid), dim3(256), (size_t)d->cout * d->ks * sizeof(float), st, A);
    EPN_CHECK_LAUNCH();
    return 0;
}

int launch_inter_c1_bwd_weight(const epn_inter_desc *d, const float *rk, const float *feats, const float *dOut,
                               float *dW, hipStream_t st, const float *grouped_saved) {
    C1Args A = make_c1(d, rk);
    A.feats = feats; A.gout = dOut; A.out = dW; A.gload = grouped_saved;
    const long long groups = (A.ncol + 255) / 256;
    // two rounds of workgroups: each ends in cout*ks atomics onto the same 1536 addresses (2048 workgroups: 1.48 ms,
    // of which ~0.9 ms contention; 512: see DESIGN 5)
    long long wgs = groups < 512 ? groups : 512;
    A.groups_per_wg = (int)((groups + wgs - 1) / wgs);
    const unsigned grid = (unsigned)((groups + A.groups_per_wg - 1) / A.groups_per_wg);
    EPN_LAUNCH(inter_c1_bwd_weight_kernel, dim3(grid), dim3(256), 0, st, A);
    EPN_CHECK_LAUNCH();
    return 0;
}

}  // namespace epn
