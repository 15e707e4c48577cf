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
    hipLaunchKernelGGL(zp_inter_fwd_kernel, dim3(blocks_of(total)), dim3(256), 0, epn_stream(stream), anchor_neighbors,
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
    hipLaunchKernelGGL(zp_inter_bwd_kernel, dim3(blocks_of(total)), dim3(256), 0, epn_stream(stream), anchor_neighbors,
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
    hipLaunchKernelGGL(zp_intra_fwd_kernel, dim3(blocks_of(total)), dim3(256), 0, epn_stream(stream), anchor_neighbors,
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
    hipLaunchKernelGGL(zp_intra_bwd_kernel, dim3(blocks_of(total)), dim3(256), 0, epn_stream(stream), anchor_neighbors,
                       anchor_weights, grad_anchor_feats, grad_feats, b, c, np, na_in, na_out, ks, ann);
    EPN_CHECK_LAUNCH();
    return 0;
}
