// vgtk.cuda.zpconv: the four grouping functions of the legacy ZPConv path (vgtk/vgtk/cuda/zpconv_cuda.cpp:41-112,
// kernels zpconv_cuda_kernel.cu:33-195).  No shipped model reaches them (their Python callers are commented out); they
// are provided for API completeness, in the reference's own channel-major layouts.
//   inter forward : out[b,c,k,p,a]  = sum_ni feats[b,c,nbr[b,p,a,k,ni],a] * w[b,p,a,k,ni]
//   intra forward : out[b,c,k,p,ao] = sum_ni feats[b,c,p,nbr[ao,ni]]      * w[ao,k,ni]
// The reference scatters every term with an atomicAdd into a zero-filled output; the forward passes here are plain
// gathers (one thread per output element, the anchor index fastest so reads and writes are coalesced), atomic-free and
// deterministic.  The backward passes are true scatters and keep the fp32 atomics.
#include "epn_common.h"

namespace {

__global__ __launch_bounds__(256) void zp_inter_fwd_kernel(const int32_t *__restrict__ nbr, const float *__restrict__ w,
                                                           const float *__restrict__ feats, float *__restrict__ out,
                                                           int b, int c, int np, int nq, int na, int ks, int ann) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;     // ((((b*c + ci)*ks + k)*np + p)*na + a)
    const long long total = (long long)b * c * ks * np * na;
    if (i >= total) return;
    const int a = (int)(i % na);
    const int p = (int)((i / na) % np);
    const int k = (int)((i / ((long long)na * np)) % ks);
    const int ci = (int)((i / ((long long)na * np * ks)) % c);
    const int bn = (int)(i / ((long long)na * np * ks * c));
    const size_t q0 = ((((size_t)bn * np + p) * na + a) * ks + k) * ann;
    const float *f = feats + ((size_t)bn * c + ci) * nq * na + a;
    float acc = 0.f;
    for (int ni = 0; ni < ann; ++ni) {
        const int qn = nbr[q0 + ni];
        if (qn >= 0 && qn < nq) acc += f[(size_t)qn * na] * w[q0 + ni];
    }
    out[i] = acc;
}

__global__ __launch_bounds__(256) void zp_inter_bwd_kernel(const int32_t *__restrict__ nbr, const float *__restrict__ w,
                                                           const float *__restrict__ gout, float *__restrict__ gfeats,
                                                           int b, int c, int np, int nq, int na, int ks, int ann) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;     // same indexing as the forward output
    const long long total = (long long)b * c * ks * np * na;
    if (i >= total) return;
    const int a = (int)(i % na);
    const int p = (int)((i / na) % np);
    const int k = (int)((i / ((long long)na * np)) % ks);
    const int ci = (int)((i / ((long long)na * np * ks)) % c);
    const int bn = (int)(i / ((long long)na * np * ks * c));
    const size_t q0 = ((((size_t)bn * np + p) * na + a) * ks + k) * ann;
    float *g = gfeats + ((size_t)bn * c + ci) * nq * na + a;
    const float go = gout[i];
    for (int ni = 0; ni < ann; ++ni) {
        const int qn = nbr[q0 + ni];
        if (qn >= 0 && qn < nq) atomicAdd(g + (size_t)qn * na, go * w[q0 + ni]);
    }
}

__global__ __launch_bounds__(256) void zp_intra_fwd_kernel(const int32_t *__restrict__ nbr, const float *__restrict__ w,
                                                           const float *__restrict__ feats, float *__restrict__ out,
                                                           int b, int c, int np, int na_in, int na_out, int ks, int ann) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;     // ((((b*c + ci)*ks + k)*np + p)*na_out + ao)
    const long long total = (long long)b * c * ks * np * na_out;
    if (i >= total) return;
    const int ao = (int)(i % na_out);
    const int p = (int)((i / na_out) % np);
    const int k = (int)((i / ((long long)na_out * np)) % ks);
    const long long bc = i / ((long long)na_out * np * ks);
    const float *f = feats + ((size_t)bc * np + p) * na_in;
    float acc = 0.f;
    for (int ni = 0; ni < ann; ++ni) {
        const int qa = nbr[ao * ann + ni];
        if (qa >= 0 && qa < na_in) acc += f[qa] * w[((size_t)ao * ks + k) * ann + ni];
    }
    out[i] = acc;
}

__global__ __launch_bounds__(256) void zp_intra_bwd_kernel(const int32_t *__restrict__ nbr, const float *__restrict__ w,
                                                           const float *__restrict__ gout, float *__restrict__ gfeats,
                                                           int b, int c, int np, int na_in, int na_out, int ks, int ann) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)b * c * ks * np * na_out;
    if (i >= total) return;
    const int ao = (int)(i % na_out);
    const int p = (int)((i / na_out) % np);
    const int k = (int)((i / ((long long)na_out * np)) % ks);
    const long long bc = i / ((long long)na_out * np * ks);
    float *g = gfeats + ((size_t)bc * np + p) * na_in;
    const float go = gout[i];
    for (int ni = 0; ni < ann; ++ni) {
        const int qa = nbr[ao * ann + ni];
        if (qa >= 0 && qa < na_in) atomicAdd(g + qa, go * w[((size_t)ao * ks + k) * ann + ni]);
    }
}

int check_dims(int b, int c, int np, int x, int na, int ks, int ann) {
    if (b < 0 || c < 1 || np < 0 || x < 1 || na < 1 || ks < 1 || ann < 1) return EPN_EINVAL;
    return 0;
}

unsigned blocks_of(long long total) { return (unsigned)((total + 255) / 256); }

}  // namespace

extern "C" int epn_zp_inter_fwd_f32(const int32_t *anchor_neighbors, const float *anchor_weights, const float *feats,
                                    int b, int c, int np, int nq, int na, int ks, int ann, float *anchor_feats,
                                    epn_stream_t stream) {
    int rc = check_dims(b, c, np, nq, na, ks, ann);
    if (rc) return rc;
    const long long total = (long long)b * c * ks * np * na;
    if (total == 0) return 0;
    if (!anchor_neighbors || !anchor_weights || !feats || !anchor_feats) return EPN_ENULL;
    EPN_LAUNCH(zp_inter_fwd_kernel, dim3(blocks_of(total)), dim3(256), 0, epn_stream(stream), anchor_neighbors,
                       anchor_weights, feats, anchor_feats, b, c, np, nq, na, ks, ann);
    EPN_CHECK_LAUNCH();
    return 0;
}

extern "C" int epn_zp_inter_bwd_f32(const int32_t *anchor_neighbors, const float *anchor_weights,
                                    const float *grad_anchor_feats, int b, int c, int np, int nq, int na, int ks, int ann,
                                    float *grad_feats, epn_stream_t stream) {
    int rc = check_dims(b, c, np, nq, na, ks, ann);
    if (rc) return rc;
    if (!grad_feats) return EPN_ENULL;
    EPN_HIP(hipMemsetAsync(grad_feats, 0, sizeof(float) * (size_t)b * c * nq * na, epn_stream(stream)));
    const long long total = (long long)b * c * ks * np * na;
    if (total == 0) return 0;
    if (!anchor_neighbors || !anchor_weights || !grad_anchor_feats) return EPN_ENULL;
    EPN_LAUNCH(zp_inter_bwd_kernel, dim3(blocks_of(total)), dim3(256), 0, epn_stream(stream), anchor_neighbors,
                       anchor_weights, grad_anchor_feats, grad_feats, b, c, np, nq, na, ks, ann);
    EPN_CHECK_LAUNCH();
    return 0;
}

extern "C" int epn_zp_intra_fwd_f32(const int32_t *anchor_neighbors, const float *anchor_weights, const float *feats,
                                    int b, int c, int np, int na_in, int na_out, int ks, int ann, float *anchor_feats,
                                    epn_stream_t stream) {
    int rc = check_dims(b, c, np, na_in, na_out, ks, ann);
    if (rc) return rc;
    const long long total = (long long)b * c * ks * np * na_out;
    if (total == 0) return 0;
    if (!anchor_neighbors || !anchor_weights || !feats || !anchor_feats) return EPN_ENULL;
    EPN_LAUNCH(zp_intra_fwd_kernel, dim3(blocks_of(total)), dim3(256), 0, epn_stream(stream), anchor_neighbors,
                       anchor_weights, feats, anchor_feats, b, c, np, na_in, na_out, ks, ann);
    EPN_CHECK_LAUNCH();
    return 0;
}

extern "C" int epn_zp_intra_bwd_f32(const int32_t *anchor_neighbors, const float *anchor_weights,
                                    const float *grad_anchor_feats, int b, int c, int np, int na_in, int na_out, int ks,
                                    int ann, float *grad_feats, epn_stream_t stream) {
    int rc = check_dims(b, c, np, na_in, na_out, ks, ann);
    if (rc) return rc;
    if (!grad_feats) return EPN_ENULL;
    EPN_HIP(hipMemsetAsync(grad_feats, 0, sizeof(float) * (size_t)b * c * np * na_in, epn_stream(stream)));
    const long long total = (long long)b * c * ks * np * na_out;
    if (total == 0) return 0;
    if (!anchor_neighbors || !anchor_weights || !grad_anchor_feats) return EPN_ENULL;
    EPN_LAUNCH(zp_intra_bwd_kernel, dim3(blocks_of(total)), dim3(256), 0, epn_stream(stream), anchor_neighbors,
                       anchor_weights, grad_anchor_feats, grad_feats, b, c, np, na_in, na_out, ks, ann);
    EPN_CHECK_LAUNCH();
    return 0;
}

// ---- vgtk.cuda.grouping.anchor_query (grouping_cuda.cpp:88-108, kernel grouping_cuda_kernel.cu:180-247): for every
// grouped neighbour offset g = grouped_xyz[b,:,p,n], anchor direction a and 2-D kernel point (kw, kh):
//     norm = |g| + 1e-6,  theta = acos(g . anchor_a / norm),  w[b,p,a,k,n] = (kw - norm)^2 + ((kh - theta) * norm)^2
// One thread per (b, p, n): the 3 coordinates are read once, the na*ks results are written with n fastest (the
// reference's layout), consecutive threads = consecutive n -> coalesced stores.  sample_idx / grouped_indices / nq are
// part of the reference signature but unused by its kernel, and so here.
namespace epn {
namespace {
__device__ __forceinline__ float aq_sqrt(float v) { return sqrtf(v); }
__device__ __forceinline__ double aq_sqrt(double v) { return sqrt(v); }
__device__ __forceinline__ float aq_acos(float v) { return acosf(v); }
__device__ __forceinline__ double aq_acos(double v) { return acos(v); }

template <typename T>   // float / double (AT_DISPATCH_FLOATING_TYPES, grouping_cuda_kernel.cu:505-510)
__global__ void anchor_query_kernel(const T *__restrict__ gxyz, const T *__restrict__ anchors,
                                    const T *__restrict__ kp, T *__restrict__ w, int b, int np, int nn, int na,
                                    int ks) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long per_b = (long long)np * nn;
    if (i >= (long long)b * per_b) return;
    const int bi = (int)(i / per_b);
    const long long r = i - (long long)bi * per_b;       // p*nn + n
    const long long pi = r / nn;
    const int ni = (int)(r - pi * nn);
    const T *g = gxyz + (size_t)bi * 3 * per_b;
    const T x = g[r], y = g[per_b + r], z = g[2 * per_b + r];
    // `scalar_t norm = sqrt(x*x + y*y + z*z) + 1e-6;` (:221): the literal is a DOUBLE -- with scalar_t = float the sum is
    // formed in double and rounded once on assignment (not a float add of 1e-6f: the two differ in the last bit at times)
    const T norm = (T)((double)aq_sqrt(x * x + y * y + z * z) + 1e-6);
    T *o = w + (((size_t)bi * np + pi) * na) * ks * nn + ni;
    for (int a = 0; a < na; ++a) {
        const T theta = aq_acos((x * anchors[3 * a] + y * anchors[3 * a + 1] + z * anchors[3 * a + 2]) / norm);
        for (int k = 0; k < ks; ++k) {
            const T d0 = kp[2 * k] - norm, d1 = (kp[2 * k + 1] - theta) * norm;
            o[((size_t)a * ks + k) * nn] = d0 * d0 + d1 * d1;
        }
    }
}
}  // namespace
}  // namespace epn

extern "C" int epn_anchor_query_f32(const float *grouped_xyz, const float *anchors, const float *kernel_points, int b,
                                    int np, int nn, int na, int ks, float *anchor_weights, epn_stream_t stream) {
    if (b < 0 || np < 0 || nn < 1 || na < 1 || ks < 1) return EPN_EINVAL;
    const long long total = (long long)b * np * nn;
    if (total == 0) return 0;
    if (!grouped_xyz || !anchors || !kernel_points || !anchor_weights) return EPN_ENULL;
    EPN_LAUNCH(epn::anchor_query_kernel<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, epn_stream(stream),
                       grouped_xyz, anchors, kernel_points, anchor_weights, b, np, nn, na, ks);
    EPN_CHECK_LAUNCH();
    return 0;
}

extern "C" int epn_anchor_query_f64(const double *grouped_xyz, const double *anchors, const double *kernel_points, int b,
                                    int np, int nn, int na, int ks, double *anchor_weights, epn_stream_t stream) {
    if (b < 0 || np < 0 || nn < 1 || na < 1 || ks < 1) return EPN_EINVAL;
    const long long total = (long long)b * np * nn;
    if (total == 0) return 0;
    if (!grouped_xyz || !anchors || !kernel_points || !anchor_weights) return EPN_ENULL;
    EPN_LAUNCH(epn::anchor_query_kernel<double>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, epn_stream(stream),
                       grouped_xyz, anchors, kernel_points, anchor_weights, b, np, nn, na, ks);
    EPN_CHECK_LAUNCH();
    return 0;
}
