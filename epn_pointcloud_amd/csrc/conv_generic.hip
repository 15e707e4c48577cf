// Generic (any-shape) HIP kernels of the SO(3) separable convolution: one thread per output element,
// grouped features materialised in a caller-provided workspace.  They serve
//   * layers the MFMA kernels do not cover (cin or cout not a multiple of 16, e.g. the first
//     layer's cin = 1) and
//   * as an independent on-device cross-check of the fused MFMA kernels in tests.
// Math restated from vgtk/vgtk/so3conv/functional.py:180-233, vgtk/vgtk/spconv/functional.py:372-390
// and vgtk/vgtk/so3conv/modules.py:48-55; feature tensors are channels-last ([b][p][a][c]).
#include "epn_common.h"
#include "conv_internal.h"

namespace epn {

// rk[a][k][d] = sum_j anchors[a][d][j] * kernels[k][j]      (functional.py:190)
__global__ void rk_table_kernel(const float *__restrict__ anchors, const float *__restrict__ kernels,
                                int na, int ks, float *__restrict__ rk) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= na * ks * 3) return;
    const int d = i % 3, k = (i / 3) % ks, a = i / (3 * ks);
    const float *R = anchors + a * 9 + d * 3;
    const float *kp = kernels + k * 3;
    rk[i] = R[0] * kp[0] + R[1] * kp[1] + R[2] * kp[2];
}

__device__ __forceinline__ float influence(float gx, float gy, float gz, const float *rk3, float sigma) {
    const float dx = gx - rk3[0], dy = gy - rk3[1], dz = gz - rk3[2];
    const float d2 = dx * dx + dy * dy + dz * dz;
    return fmaxf(1.0f - d2 / sigma, 0.0f);  // functional.py:198-200
}

struct Geo {  // relative coordinate of neighbour n of output point (b,p); shadow index -> 1e4
    const float *xyz, *new_xyz;
    const int32_t *idx;
    int p1, p2, nn;
    __device__ __forceinline__ int q(int b, int p, int n) const { return idx[((size_t)b * p2 + p) * nn + n]; }
    __device__ __forceinline__ void g(int b, int p, int qq, float &gx, float &gy, float &gz) const {
        const float *s = xyz + (size_t)b * 3 * p1;
        const float *c = new_xyz + (size_t)b * 3 * p2;
        const bool sh = qq < 0 || qq >= p1;
        gx = (sh ? 1e4f : s[qq]) - c[p];
        gy = (sh ? 1e4f : s[p1 + qq]) - c[p2 + p];
        gz = (sh ? 1e4f : s[2 * p1 + qq]) - c[2 * p2 + p];
    }
};

// w[b,p,a,k,n] dense (API compatibility: InterSO3Conv returns inter_w)
__global__ void inter_weights_kernel(Geo geo, const float *__restrict__ anchors,
                                     const float *__restrict__ kernels, float sigma, int b, int na, int ks,
                                     float *__restrict__ w) {
    const size_t total = (size_t)b * geo.p2 * na * ks * geo.nn;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int n = i % geo.nn;
        size_t r = i / geo.nn;
        const int k = r % ks; r /= ks;
        const int a = r % na; r /= na;
        const int p = r % geo.p2;
        const int bi = r / geo.p2;
        float gx, gy, gz;
        geo.g(bi, p, geo.q(bi, p, n), gx, gy, gz);
        float rk3[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {  // same expression as rk_table_kernel
            const float *R = anchors + a * 9 + d * 3;
            const float *kp = kernels + k * 3;
            rk3[d] = R[0] * kp[0] + R[1] * kp[1] + R[2] * kp[2];
        }
        w[i] = influence(gx, gy, gz, rk3, sigma);
    }
}

// G[col][c][k] = sum_n F[b, idx[b,p,n], a, c] * w[b,p,a,k,n];  thread = (col, c), c fastest
template <int KS_MAX>
__global__ void inter_group_kernel(Geo geo, const float *__restrict__ rk, const float *__restrict__ dense_w,
                                   float sigma, const float *__restrict__ feats, int b, int na, int ks,
                                   int cin, float *__restrict__ G) {
    const size_t total = (size_t)b * geo.p2 * na * cin;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = i % cin;
    const size_t col = i / cin;
    const int a = col % na;
    const int p = (col / na) % geo.p2;
    const int bi = col / ((size_t)na * geo.p2);
    float acc[KS_MAX];
#pragma unroll
    for (int k = 0; k < KS_MAX; ++k) acc[k] = 0.f;
    for (int n = 0; n < geo.nn; ++n) {
        const int qq = geo.q(bi, p, n);
        const bool sh = qq < 0 || qq >= geo.p1;
        const float f = sh ? 0.f : feats[(((size_t)bi * geo.p1 + qq) * na + a) * cin + c];
        float gx = 0.f, gy = 0.f, gz = 0.f;
        if (!dense_w) geo.g(bi, p, qq, gx, gy, gz);
#pragma unroll
        for (int k = 0; k < KS_MAX; ++k) {
            if (k < ks) {
                const float w = dense_w ? dense_w[((col * ks) + k) * geo.nn + n]
                                        : influence(gx, gy, gz, rk + ((size_t)a * ks + k) * 3, sigma);
                acc[k] += f * w;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < KS_MAX; ++k)
        if (k < ks) G[(col * cin + c) * ks + k] = acc[k];
}

// out[col][o] = sum_ck X[col][ck] * W[o][ck]      (BasicSO3Conv, modules.py:48-55); thread = (col,o)
__global__ void rowgemm_nt_kernel(const float *__restrict__ X, const float *__restrict__ W, size_t ncol,
                                  int ck, int cout, float *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncol * cout) return;
    const int o = i % cout;
    const size_t col = i / cout;
    const float *x = X + col * ck;
    const float *w = W + (size_t)o * ck;
    float acc = 0.f;
    for (int j = 0; j < ck; ++j) acc += x[j] * w[j];
    out[i] = acc;
}

// dX[col][ck] = sum_o dOut[col][o] * W[o][ck];  thread = (col, ck)
__global__ void rowgemm_nn_kernel(const float *__restrict__ dOut, const float *__restrict__ W, size_t ncol,
                                  int ck, int cout, float *__restrict__ dX) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncol * ck) return;
    const int j = i % ck;
    const size_t col = i / ck;
    const float *g = dOut + col * cout;
    float acc = 0.f;
    for (int o = 0; o < cout; ++o) acc += g[o] * W[(size_t)o * ck + j];
    dX[i] = acc;
}

// dW[o][ck] += sum_{col in chunk} dOut[col][o] * X[col][ck];  thread = (o, ck), blockIdx.y = chunk
__global__ void colreduce_dw_kernel(const float *__restrict__ dOut, const float *__restrict__ X, size_t ncol,
                                    size_t chunk, int ck, int cout, float *__restrict__ dW) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)cout * ck) return;
    const int j = i % ck;
    const int o = i / ck;
    const size_t c0 = (size_t)blockIdx.y * chunk;
    const size_t c1 = c0 + chunk < ncol ? c0 + chunk : ncol;
    float acc = 0.f;
    for (size_t col = c0; col < c1; ++col) acc += dOut[col * cout + o] * X[col * ck + j];
    atomicAdd(dW + i, acc);
}

// dF[b, idx[b,p,n], a, c] += sum_k dG[col][c][k] * w[b,p,a,k,n];  thread = (col, c)
template <int KS_MAX>
__global__ void inter_scatter_kernel(Geo geo, const float *__restrict__ rk, const float *__restrict__ dense_w,
                                     float sigma, const float *__restrict__ dG, int b, int na, int ks, int cin,
                                     float *__restrict__ dF) {
    const size_t total = (size_t)b * geo.p2 * na * cin;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = i % cin;
    const size_t col = i / cin;
    const int a = col % na;
    const int p = (col / na) % geo.p2;
    const int bi = col / ((size_t)na * geo.p2);
    float dg[KS_MAX];
#pragma unroll
    for (int k = 0; k < KS_MAX; ++k) dg[k] = k < ks ? dG[(col * cin + c) * ks + k] : 0.f;
    for (int n = 0; n < geo.nn; ++n) {
        const int qq = geo.q(bi, p, n);
        if (qq < 0 || qq >= geo.p1) continue;
        float gx = 0.f, gy = 0.f, gz = 0.f;
        if (!dense_w) geo.g(bi, p, qq, gx, gy, gz);
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < KS_MAX; ++k) {
            if (k < ks) {
                const float w = dense_w ? dense_w[((col * ks) + k) * geo.nn + n]
                                        : influence(gx, gy, gz, rk + ((size_t)a * ks + k) * 3, sigma);
                t += dg[k] * w;
            }
        }
        atomicAdd(dF + (((size_t)bi * geo.p1 + qq) * na + a) * cin + c, t);
    }
}

// ---------------------------------------------------------------------------------- intra, generic
// out[col][o] = sum_{c,k} W[o][c*kn+k] * F[b,p,intra_idx[a,k],c];  thread = (col, o)
__global__ void intra_fwd_kernel(const float *__restrict__ feats, const int32_t *__restrict__ iidx,
                                 const float *__restrict__ W, size_t npts, int na, int kn, int cin, int cout,
                                 float *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npts * na * cout) return;
    const int o = i % cout;
    const size_t col = i / cout;
    const int a = col % na;
    const size_t pt = col / na;
    const float *w = W + (size_t)o * cin * kn;
    float acc = 0.f;
    for (int k = 0; k < kn; ++k) {
        const float *f = feats + (pt * na + iidx[a * kn + k]) * cin;
        for (int c = 0; c < cin; ++c) acc += w[c * kn + k] * f[c];
    }
    out[i] = acc;
}

// dF[b,p,intra_idx[a,k],c] += sum_o W[o][c*kn+k] * dOut[col][o];  thread = (col, c)
__global__ void intra_bwd_data_kernel(const float *__restrict__ dOut, const int32_t *__restrict__ iidx,
                                      const float *__restrict__ W, size_t npts, int na, int kn, int cin,
                                      int cout, float *__restrict__ dF) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npts * na * cin) return;
    const int c = i % cin;
    const size_t col = i / cin;
    const int a = col % na;
    const size_t pt = col / na;
    const float *g = dOut + col * cout;
    for (int k = 0; k < kn; ++k) {
        float acc = 0.f;
        for (int o = 0; o < cout; ++o) acc += W[((size_t)o * cin + c) * kn + k] * g[o];
        atomicAdd(dF + (pt * na + iidx[a * kn + k]) * cin + c, acc);
    }
}

// dW[o][c*kn+k] += sum_{col in chunk} dOut[col][o] * F[b,p,intra_idx[a,k],c];  thread = (o, c*kn+k)
__global__ void intra_bwd_weight_kernel(const float *__restrict__ feats, const float *__restrict__ dOut,
                                        const int32_t *__restrict__ iidx, size_t ncol, size_t chunk, int na,
                                        int kn, int cin, int cout, float *__restrict__ dW) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)cout * cin * kn) return;
    const int k = i % kn;
    const int c = (i / kn) % cin;
    const int o = i / ((size_t)kn * cin);
    const size_t c0 = (size_t)blockIdx.y * chunk;
    const size_t c1 = c0 + chunk < ncol ? c0 + chunk : ncol;
    float acc = 0.f;
    for (size_t col = c0; col < c1; ++col) {
        const int a = col % na;
        const size_t pt = col / na;
        acc += dOut[col * cout + o] * feats[(pt * na + iidx[a * kn + k]) * cin + c];
    }
    atomicAdd(dW + i, acc);
}

// ---------------------------------------------------------------------------------- host launchers
static Geo make_geo(const epn_inter_desc *d) {
    Geo g;
    g.xyz = d->xyz; g.new_xyz = d->new_xyz; g.idx = d->ball_idx;
    g.p1 = d->p1; g.p2 = d->p2; g.nn = d->nn;
    return g;
}

int launch_rk_table(const epn_inter_desc *d, float *rk, hipStream_t st) {
    const int n = d->na * d->ks * 3;
    EPN_LAUNCH_AUX(rk_table_kernel, dim3(epn_cdiv(n, 256)), dim3(256), 0, st, d->anchors, d->kernels, d->na,
                       d->ks, rk);
    EPN_CHECK_LAUNCH();
    return 0;
}

int launch_inter_weights(const epn_inter_desc *d, float *w, hipStream_t st) {
    const size_t total = (size_t)d->b * d->p2 * d->na * d->ks * d->nn;
    const int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    EPN_LAUNCH(inter_weights_kernel, dim3(blocks), dim3(256), 0, st, make_geo(d), d->anchors, d->kernels, d->sigma,
                       d->b, d->na, d->ks, w);
    EPN_CHECK_LAUNCH();
    return 0;
}

int launch_inter_group(const epn_inter_desc *d, const float *rk, const float *feats, float *G, hipStream_t st) {
    const size_t total = (size_t)d->b * d->p2 * d->na * d->cin;
    if (d->ks <= EPN_KS_MAX)
        EPN_LAUNCH(inter_group_kernel<EPN_KS_MAX>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                           make_geo(d), rk, d->dense_w, d->sigma, feats, d->b, d->na, d->ks, d->cin, G);
    else   // kpsphere66 (kernel_size = 3, vgtk/vgtk/so3conv/functional.py:86-96)
        EPN_LAUNCH(inter_group_kernel<EPN_KS_GENERIC_MAX>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                           make_geo(d), rk, d->dense_w, d->sigma, feats, d->b, d->na, d->ks, d->cin, G);
    EPN_CHECK_LAUNCH();
    return 0;
}

int launch_inter_scatter(const epn_inter_desc *d, const float *rk, const float *dG, float *dF, hipStream_t st) {
    const size_t total = (size_t)d->b * d->p2 * d->na * d->cin;
    if (d->ks <= EPN_KS_MAX)
        EPN_LAUNCH(inter_scatter_kernel<EPN_KS_MAX>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                           make_geo(d), rk, d->dense_w, d->sigma, dG, d->b, d->na, d->ks, d->cin, dF);
    else
        EPN_LAUNCH(inter_scatter_kernel<EPN_KS_GENERIC_MAX>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                           make_geo(d), rk, d->dense_w, d->sigma, dG, d->b, d->na, d->ks, d->cin, dF);
    EPN_CHECK_LAUNCH();
    return 0;
}

int launch_rowgemm_nt(const float *X, const float *W, size_t ncol, int ck, int cout, float *out, hipStream_t st) {
    const size_t total = ncol * cout;
    EPN_LAUNCH(rowgemm_nt_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, X, W, ncol, ck,
                       cout, out);
    EPN_CHECK_LAUNCH();
    return 0;
}

int launch_rowgemm_nn(const float *dOut, const float *W, size_t ncol, int ck, int cout, float *dX, hipStream_t st) {
    const size_t total = ncol * ck;
    EPN_LAUNCH(rowgemm_nn_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dOut, W, ncol, ck,
                       cout, dX);
    EPN_CHECK_LAUNCH();
    return 0;
}

int launch_colreduce_dw(const float *dOut, const float *X, size_t ncol, int ck, int cout, float *dW,
                        hipStream_t st) {
    const size_t chunk = 512;
    dim3 grid(epn_cdiv((long long)cout * ck, 256), (unsigned)((ncol + chunk - 1) / chunk));
    EPN_LAUNCH(colreduce_dw_kernel, grid, dim3(256), 0, st, dOut, X, ncol, chunk, ck, cout, dW);
    EPN_CHECK_LAUNCH();
    return 0;
}

int launch_intra_fwd_generic(const float *feats, const int32_t *iidx, const float *W, size_t npts, int na, int kn,
                             int cin, int cout, float *out, hipStream_t st) {
    const size_t total = npts * na * cout;
    EPN_LAUNCH(intra_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, feats, iidx, W, npts,
                       na, kn, cin, cout, out);
    EPN_CHECK_LAUNCH();
    return 0;
}

int launch_intra_bwd_data_generic(const float *dOut, const int32_t *iidx, const float *W, size_t npts, int na, int kn,
                                  int cin, int cout, float *dF, hipStream_t st) {
    const size_t total = npts * na * cin;
    EPN_LAUNCH(intra_bwd_data_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dOut, iidx, W,
                       npts, na, kn, cin, cout, dF);
    EPN_CHECK_LAUNCH();
    return 0;
}

int launch_intra_bwd_weight_generic(const float *feats, const float *dOut, const int32_t *iidx, size_t npts, int na,
                                    int kn, int cin, int cout, float *dW, hipStream_t st) {
    const size_t ncol = npts * na, chunk = 480;
    dim3 grid(epn_cdiv((long long)cout * cin * kn, 256), (unsigned)((ncol + chunk - 1) / chunk));
    EPN_LAUNCH(intra_bwd_weight_kernel, grid, dim3(256), 0, st, feats, dOut, iidx, ncol, chunk, na, kn, cin,
                       cout, dW);
    EPN_CHECK_LAUNCH();
    return 0;
}

}  // namespace epn
