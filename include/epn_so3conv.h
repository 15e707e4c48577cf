/*
 * include/epn_so3conv.h -- C ABI of libepn_so3conv.so (MI355X / gfx950).
 *
 * Drop-in boundary for the EPN SE(3) separable point-convolution hot path.  Every entry point is
 * `extern "C"`, takes plain DEVICE pointers + sizes + a HIP stream (passed as void*), launches
 * asynchronously on that stream, owns no memory and keeps no global state other than epn_set_kernel_policy (thread-safe) and the thread-local diagnostic
 * string behind epn_last_kernel.  Return
 * value: 0 on success, otherwise a hipError_t code (launch errors) or a negative EPN_E* code
 * (argument errors).  `epn_strerror` turns either into text.
 *
 * Each function names the reference interface it replaces (paths relative to the reference repo
 * nintendops/EPN_PointCloud).  The reference's pybind functions allocate their outputs; here the
 * CALLER allocates (the Python mirror in epn_pointcloud_amd/vgtk/cuda does it with torch).
 *
 * Layout notes
 *   xyz tensors are channel-major  [b,3,n]  exactly as the reference passes them.
 *   Feature tensors at THIS boundary are channels-last: feats_cl[b][p][a][c]  (the memory image of a
 *   torch tensor of logical shape [b,c,p,a] in torch.channels_last format).  The public nn.Module
 *   API keeps the reference's logical [b,c,p,a] shape.
 */
#ifndef EPN_SO3CONV_H
#define EPN_SO3CONV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EPN_EINVAL (-1)      /* bad size / unsupported shape                      */
#define EPN_EWORKSPACE (-2)  /* workspace pointer NULL or too small               */
#define EPN_ENULL (-3)       /* required pointer is NULL                          */

typedef void *epn_stream_t; /* hipStream_t; NULL = the null stream */

/* Integer revision of this binary interface: bumped whenever an EXISTING signature or struct layout changes (new entry
 * points alone do not bump it).  A host compares epn_abi_version() of the library it loaded with the EPN_ABI_VERSION of
 * the header it was compiled against and refuses to run on a mismatch -- revision 2 changed `epn_ball_query_f64`'s radius
 * from double to float and grew `struct epn_gemm_nt_problem`, which a caller built against revision 1 would not notice;
 * revision 3 (round 6) grew that struct again (`c_amax`)
 * (INTEGRATION.md "ABI revisions").  epn_pointcloud_amd/_lib.py performs exactly this check at load time. */
#define EPN_ABI_VERSION 3
int epn_abi_version(void);
const char *epn_version(void);
const char *epn_strerror(int code);
/* Diagnostic: the device kernel (exact template instance, "(anonymous namespace)::" removed, e.g.
 * "epn::gemm_nt_x3_kernel<4, 2, 2, 4, 2>") that the calling thread's most recent library call launched as its main
 * kernel -- what rocprofv3 will call it -- so a benchmark can attribute time without re-deriving the launchers' tile
 * choices.  Thread-local, cleared by the read; "" when nothing was launched since the last read.  The returned
 * pointer is valid until the thread's next call of this function. */
const char *epn_last_kernel(void);

/* Cross-check switch, the library's only process-wide state (a relaxed atomic; default 0): 0 = every entry point picks
 * its best kernel, 1 = the any-shape generic kernels everywhere (an independent on-device implementation used by the
 * parity tests).  Not for production use; set it before launching from other threads.  Any other value: EPN_EINVAL.
 * (Libraries built with -DEPN_TUNING -- `python -m epn_pointcloud_amd.build --tuning`, used by tools/ only -- also accept
 * 0x100|v .. 0x400|v, the A/B switches of the tuning tools: GEMM tile overrides, 0x401 = per-slot atomic scatter.) */
int epn_set_kernel_policy(int policy);

/* ------------------------------------------------------------------ index kernels ---------- */

/* Replaces vgtk.cuda.grouping.ball_query  (vgtk/vgtk/cuda/grouping_cuda.cpp:71-86, kernel
 * grouping_cuda_kernel.cu:67-113).  new_xyz f32[b,3,m], xyz f32[b,3,n] -> idx i32[b,m,nsample].
 * The kernel writes every slot (the reference zero-fills first, grouping_cuda.cpp:80-82). */
int epn_ball_query_f32(const float *new_xyz, const float *xyz, int b, int n, int m, float radius,
                       int nsample, int32_t *idx, epn_stream_t stream);

/* Replaces vgtk.cuda.grouping.furthest_point_sampling (grouping_cuda.cpp:160-174, kernel
 * grouping_cuda_kernel.cu:351-466).  xyz f32[b,3,n] -> idx i32[b,m].  The 1e10 `temp` buffer of
 * the reference lives in registers.  Requires 1 <= m, 1 <= n <= 32768 (EPN_EINVAL beyond: use
 * epn_fps_temp_f32, which keeps the running minima in a caller-provided `temp` f32[b,n] like the
 * reference kernel -- its strided loop has no size limit, grouping_cuda_kernel.cu:380-396 -- and
 * gives the same indices for every n). */
int epn_fps_f32(const float *xyz, int b, int n, int m, int32_t *idx, epn_stream_t stream);
int epn_fps_temp_f32(const float *xyz, int b, int n, int m, float *temp, int32_t *idx, epn_stream_t stream);

/* Replace vgtk.cuda.gathering.gather_points_forward / _backward (gathering_cuda.cpp:29-60,
 * kernels gathering_cuda_kernel.cu:43-98).  points f32[b,c,n], idx i32[b,m] -> out f32[b,c,m];
 * grad_out f32[b,c,m] -> grad_points f32[b,c,n] (zero-filled here, then scatter-added). */
int epn_gather_fwd_f32(const float *points, const int32_t *idx, int b, int c, int n, int m,
                       float *out, epn_stream_t stream);
int epn_gather_bwd_f32(const float *grad_out, const int32_t *idx, int b, int c, int n, int m,
                       float *grad_points, epn_stream_t stream);

/* fp64 dispatch of the same four extensions: the reference instantiates them for float AND double
 * (AT_DISPATCH_FLOATING_TYPES: grouping_cuda_kernel.cu:477 ball query, :638-726 FPS; gathering_cuda_kernel.cu:117,151).
 * Same semantics with double coordinates / features; indices stay int32.  epn_fps_f64 takes the reference's `temp`
 * buffer (f64[b,n], grouping_cuda.cpp:167-168; initialised to 1e10 here) because it keeps the running minima there, as the
 * reference kernel does -- fp64 sampling is a compatibility path, not a tuned one.  `radius` stays a FLOAT and is squared in
 * float before it is widened, as the reference's templated kernel does (`float radius`, `scalar_t radius2 = radius *
 * radius;`, grouping_cuda_kernel.cu:67,80; host side grouping_cuda.cpp:74). */
int epn_ball_query_f64(const double *new_xyz, const double *xyz, int b, int n, int m, float radius, int nsample,
                       int32_t *idx, epn_stream_t stream);
int epn_fps_f64(const double *xyz, int b, int n, int m, double *temp, int32_t *idx, epn_stream_t stream);
int epn_gather_fwd_f64(const double *points, const int32_t *idx, int b, int c, int n, int m, double *out,
                       epn_stream_t stream);
int epn_gather_bwd_f64(const double *grad_out, const int32_t *idx, int b, int c, int n, int m, double *grad_points,
                       epn_stream_t stream);

/* ------------------------------------------------------------------ InterSO3Conv ------------ */

/* Geometry + shapes of one inter convolution (vgtk/vgtk/so3conv/functional.py:118-178).
 * The fused kernels regenerate the kernel-influence weights
 *     w[b,p,a,k,n] = relu(1 - |xyz[:,idx[b,p,n]] - new_xyz[:,p] - R_a kappa_k|^2 / sigma)
 * (functional.py:180-218) on the fly; they are never written to HBM. */
typedef struct epn_inter_desc {
    const float *xyz;        /* [b,3,p1] support coordinates                                   */
    const float *new_xyz;    /* [b,3,p2] query (output) coordinates                            */
    const int32_t *ball_idx; /* [b,p2,nn] neighbour indices into p1 (from epn_ball_query_f32)  */
    const float *anchors;    /* [na,3,3] rotation matrices                                     */
    const float *kernels;    /* [ks,3]   kernel points                                         */
    const float *dense_w;    /* optional [b,p2,na,ks,nn]: use THESE weights instead of geometry */
    float sigma;
    int b, p1, p2, nn, na, ks, cin, cout;
} epn_inter_desc;

/* Bytes of scratch the inter kernels need for this descriptor (tables; plus the materialised
 * grouped features on the generic path used when cin or cout is not a multiple of 16). */
size_t epn_inter_workspace_bytes(const epn_inter_desc *d);

/* Replaces InterSO3Conv.forward's compute (vgtk/vgtk/so3conv/modules.py:157-174 =
 * inter_so3conv_grouping -> inter_zpconv_grouping_naive (vgtk/vgtk/spconv/functional.py:372-390)
 * -> BasicSO3Conv (modules.py:48-55)):
 *   out_cl[b,p,a,o] = sum_{c,k} W[o, c*ks+k] * sum_n feats_cl[b, idx[b,p,n], a, c] * w[b,p,a,k,n]
 * feats_cl f32[b,p1,na,cin], W f32[cout, cin*ks] -> out_cl f32[b,p2,na,cout]. */
int epn_inter_so3conv_fwd_f32(const epn_inter_desc *d, const float *feats_cl, const float *W,
                              float *out_cl, void *workspace, size_t workspace_bytes,
                              epn_stream_t stream);

/* Autograd transposes (reference: torch autograd through gather/einsum/matmul).
 * grad_out_cl f32[b,p2,na,cout].
 *   bwd_data  : grad_feats_cl f32[b,p1,na,cin]  (zero-filled here, then accumulated)
 *   bwd_weight: grad_W f32[cout, cin*ks]        (zero-filled here, then accumulated) */
int epn_inter_so3conv_bwd_data_f32(const epn_inter_desc *d, const float *grad_out_cl,
                                   const float *W, float *grad_feats_cl, void *workspace,
                                   size_t workspace_bytes, epn_stream_t stream);
int epn_inter_so3conv_bwd_weight_f32(const epn_inter_desc *d, const float *feats_cl,
                                     const float *grad_out_cl, float *grad_W, void *workspace,
                                     size_t workspace_bytes, epn_stream_t stream);

/* The first layer of every model (cin = 1: the occupancy feature of get_occupancy_features, so3conv/functional.py:25-44;
 * InterSO3Conv(1 -> cout), cls_so3net_pn.py:101) is not matrix work: its cost is generating the ks x nn kernel-influence
 * weights per column on the VALU -- and round 3's weight gradient paid it a second time.  These two entry points keep the
 * 24 grouped values of a column (`grouped` f32[b*p2*na][ks], 96 bytes per column: 94 MB at B=32 against the 3 GB inter_w
 * the reference keeps for the same purpose) from the forward pass, and the weight gradient contracts grad_out with them:
 *   epn_inter_so3conv_fwd_c1_f32        : as epn_inter_so3conv_fwd_f32 for cin = 1; `grouped` may be NULL (inference)
 *   epn_inter_so3conv_bwd_weight_c1_f32 : grad_W f32[cout][ks] = grad_out^T . grouped (zero-filled here, then accumulated)
 * Shapes: epn_inter_c1_ok (cin = 1, no dense inter_w, cout in {16, 32, 48, 64}, ks <= 32); EPN_EINVAL otherwise.
 * Forward, ks = 24, even na, cout in {16, 32, 64}, nn <= 128: when the features do not depend on the anchor (checked on
 * the device, no host synchronisation; true for the occupancy feature) the relu arguments come off the matrix pipe
 * (inter_c1_fwd_mfma_kernel: rank-5 product, operands split without loss into 3 x bf16), otherwise -- or with
 * EPN_C1_MFMA=0 -- from the VALU kernel.  Outputs of the two agree to ~2e-6 of the tensor maximum.  The weight gradient
 * equals epn_gemm_tn_split_f32(grad_out rows, grouped), which is what the Python host calls (3-5x faster; EPN_C1_DW). */
int epn_inter_c1_ok(const epn_inter_desc *d);
int epn_inter_so3conv_fwd_c1_f32(const epn_inter_desc *d, const float *feats_cl, const float *W, float *out_cl,
                                 float *grouped, void *workspace, size_t workspace_bytes, epn_stream_t stream);
int epn_inter_so3conv_bwd_weight_c1_f32(const epn_inter_desc *d, const float *grouped, const float *grad_out_cl,
                                        float *grad_W, epn_stream_t stream);

/* 1 when the fused MFMA kernels serve this descriptor, 0 when the generic kernels do. */
int epn_inter_is_fused(const epn_inter_desc *d);

/* Materialise the weights for API compatibility (InterSO3Conv returns inter_w):
 * w f32[b,p2,na,ks,nn], formula of vgtk/vgtk/so3conv/functional.py:190-200. */
int epn_inter_weights_f32(const epn_inter_desc *d, float *w, epn_stream_t stream);

/* ------------------------------------------------------------------ IntraSO3Conv ------------ */

/* Replaces IntraSO3Conv.forward's compute (vgtk/vgtk/so3conv/modules.py:197-200 =
 * intra_so3conv_grouping (functional.py:221-233) -> BasicSO3Conv):
 *   out_cl[b,p,a,o] = sum_{c,k} W[o, c*kn+k] * feats_cl[b, p, intra_idx[a,k], c]
 * feats_cl f32[b,p,na,cin], intra_idx i32[na,kn], W f32[cout, cin*kn] -> out_cl f32[b,p,na,cout]. */
size_t epn_intra_workspace_bytes(int na, int kn, int cin, int cout);
int epn_intra_is_fused(int na, int kn, int cin, int cout);

int epn_intra_so3conv_fwd_f32(const float *feats_cl, const int32_t *intra_idx, const float *W,
                              int b, int p, int na, int kn, int cin, int cout, float *out_cl,
                              void *workspace, size_t workspace_bytes, epn_stream_t stream);

/* Autograd transposes.  `inv_idx` i32[na,kn] (optional) is the column-wise inverse of intra_idx,
 * inv_idx[intra_idx[a,k], k] = a; it exists when every column of intra_idx is a permutation of the
 * anchors (true for the icosahedral table) and makes the data gradient an atomic-free gather.  With
 * inv_idx == NULL a scatter-add kernel is used.  grad_feats_cl / grad_W are fully overwritten. */
int epn_intra_so3conv_bwd_data_f32(const float *grad_out_cl, const int32_t *intra_idx,
                                   const int32_t *inv_idx, const float *W, int b, int p, int na,
                                   int kn, int cin, int cout, float *grad_feats_cl, void *workspace,
                                   size_t workspace_bytes, epn_stream_t stream);
int epn_intra_so3conv_bwd_weight_f32(const float *feats_cl, const float *grad_out_cl,
                                     const int32_t *intra_idx, int b, int p, int na, int kn,
                                     int cin, int cout, float *grad_W, epn_stream_t stream);

/* ------------------------------------------------------------------ block glue ---- */

/* Normalisation + leaky_relu of SeparableSO3ConvBlock (SPConvNets/utils/base_so3conv.py:116-126, 52-62, 208-211:
 * `relu(norm(x))` with norm = BatchNorm2d or InstanceNorm2d(affine=False), relu = F.leaky_relu) on channels-last
 * tensors, as two memory-bound passes instead of torch's 4-6 (+ layout copies).  x_cl f32[groups][rows][c]:
 *   BatchNorm2d   : groups = 1, rows = b*p*a  (statistics per channel over every other axis)
 *   InstanceNorm2d: groups = b, rows = p*a    (statistics per (sample, channel))
 * sums f32[groups][c][2] = (sum x, sum x^2) is produced by epn_chan_stats_f32 (block partials in the workspace, then a
 * small finishing kernel: no atomics, deterministic) and consumed by the
 * apply passes: mean = s1/rows, var = s2/rows - mean^2 (biased, as both torch norms use for normalisation).
 *   y = leaky(((x - mean) * rsqrt(var + eps)) * gamma + beta, slope) (+ residual)      gamma/beta/residual optional
 * Backward: dsums f32[groups][c][2] = (sum dn, sum dn*xhat) with dn = dy * leaky'(.) * gamma, from
 * epn_norm_act_bwd_reduce_f32 (also writes dgamma / dbeta when given), then
 *   dx = rstd * (dn - mean(dn) - xhat * mean(dn * xhat))                               epn_norm_act_bwd_apply_f32 */
size_t epn_norm_workspace_bytes(int groups, long long rows, int c);   /* scratch of the two reduction entry points */
/* BatchNorm2d's running statistics (training mode) from sums[c][2] of `count` values per channel, in one launch:
 * mean (+ conv_bias, the bias of the producing convolution when it was not added to x), unbiased variance,
 * num_batches_tracked += 1, running = running + momentum * (batch - running); momentum < 0 = None (cumulative average,
 * 1 / num_batches_tracked).  Reference: torch.nn.BatchNorm2d inside SeparableSO3ConvBlock (base_so3conv.py:168-212). */
int epn_bn_running_update_f32(const float *sums, double count, const float *conv_bias, float *running_mean,
                              float *running_var, long long *num_batches_tracked, float momentum, int c,
                              epn_stream_t stream);
int epn_chan_stats_f32(const float *x_cl, int groups, long long rows, int c, float *sums, void *workspace,
                       size_t workspace_bytes, epn_stream_t stream);
int epn_norm_act_fwd_f32(const float *x_cl, int groups, long long rows, int c, const float *sums,
                         const float *gamma, const float *beta, const float *residual_cl, float eps, float slope,
                         float *y_cl, epn_stream_t stream);
int epn_norm_act_bwd_reduce_f32(const float *x_cl, const float *dy_cl, int groups, long long rows, int c,
                                const float *sums, const float *gamma, const float *beta, float eps, float slope,
                                float *dsums, float *dgamma, float *dbeta, void *workspace,
                                size_t workspace_bytes, epn_stream_t stream);
int epn_norm_act_bwd_apply_f32(const float *x_cl, const float *dy_cl, int groups, long long rows, int c,
                               const float *sums, const float *dsums, const float *gamma, const float *beta,
                               float eps, float slope, float *dx_cl, epn_stream_t stream);

/* The tail of a SeparableSO3ConvBlock in one pass per direction (SPConvNets/utils/base_so3conv.py:204-211):
 *   y = leaky_relu(norm_a(xa)) + leaky_relu(norm_b(xb))
 * xa = the IntraSO3Conv output with its InstanceNorm2d, xb = the skip branch's 1x1 convolution with the block's norm.  The
 * skip branch's normalised tensor is never written; the backward passes read the common output gradient once.  Tensors are
 * [b clouds][rows per cloud][c] (float, or __bf16 with bf16 = 1; c % 4 == 0, 256 % (c/4) == 0).  A side is "instance"
 * (sums / dsums [b][c][2], statistics per cloud) or batch (sums / dsums [1][c][2] over all clouds); sums come from
 * epn_chan_stats_*.  bwd_reduce also writes dgamma / dbeta of a side when given; dx of a side may be NULL in bwd_apply. */
typedef struct epn_norm_pair_side {
    const float *sums;      /* [b or 1][c][2]: (sum x, sum x^2) */
    const float *gamma;     /* [c] or NULL */
    const float *beta;      /* [c] or NULL */
    float eps;
    int instance;           /* 1: statistics per cloud (InstanceNorm2d), 0: one set (BatchNorm2d) */
} epn_norm_pair_side;
size_t epn_norm_pair_workspace_bytes(int b, long long rows, int c);
int epn_norm_act_pair_fwd(const void *xa_cl, const void *xb_cl, int b, long long rows, int c,
                          const epn_norm_pair_side *side_a, const epn_norm_pair_side *side_b, float slope, void *y_cl,
                          int bf16, epn_stream_t stream);
int epn_norm_act_pair_bwd_reduce(const void *xa_cl, const void *xb_cl, const void *dy_cl, int b, long long rows, int c,
                                 const epn_norm_pair_side *side_a, const epn_norm_pair_side *side_b, float slope,
                                 float *dsums_a, float *dgamma_a, float *dbeta_a, float *dsums_b, float *dgamma_b,
                                 float *dbeta_b, void *workspace, size_t workspace_bytes, int bf16, epn_stream_t stream);
int epn_norm_act_pair_bwd_apply(const void *xa_cl, const void *xb_cl, const void *dy_cl, int b, long long rows, int c,
                                const epn_norm_pair_side *side_a, const epn_norm_pair_side *side_b, float slope,
                                const float *dsums_a, const float *dsums_b, void *dxa_cl, void *dxb_cl, int bf16,
                                epn_stream_t stream);

/* fp32 variants of three of the passes above that also write max |output| (of y / of dx / of side b's dx) into a device
 * scalar, zeroed by the call: the producer-side maxima of the two-piece fp16 GEMMs that consume those tensors (see
 * epn_gemm_nt_f16x2_f32).  bf16 = 1: EPN_EINVAL. */
int epn_norm_act_pair_fwd_amax(const void *xa_cl, const void *xb_cl, int b, long long rows, int c,
                               const epn_norm_pair_side *side_a, const epn_norm_pair_side *side_b, float slope, void *y_cl,
                               int bf16, float *y_amax, epn_stream_t stream);
int epn_norm_act_pair_bwd_apply_amax(const void *xa_cl, const void *xb_cl, const void *dy_cl, int b, long long rows, int c,
                                     const epn_norm_pair_side *side_a, const epn_norm_pair_side *side_b, float slope,
                                     const float *dsums_a, const float *dsums_b, void *dxa_cl, void *dxb_cl, int bf16,
                                     float *dxb_amax, epn_stream_t stream);
int epn_norm_act_bwd_apply_amax_f32(const float *x_cl, const float *dy_cl, int groups, long long rows, int c,
                                    const float *sums, const float *dsums, const float *gamma, const float *beta, float eps,
                                    float slope, float *dx_cl, float *dx_amax, epn_stream_t stream);

/* replaces vgtk.cuda.grouping.initial_anchor_query (vgtk/vgtk/cuda/grouping_cuda.cpp:138-158, kernel
 * grouping_cuda_kernel.cu:116-167; only consumer: KernelPropagation, vgtk/vgtk/so3conv/modules.py:57-119).
 *   centers f32[b][3][nc]   xyz f32[m][3] (fragment points, shared by the batch)   kernel_points f32[ks][na][3]
 *   anchor_weights f32[b][ks][nc][na] = sum over fragment points within `radius` of the centre (sqrt distance, <=) of
 *                                       max(1 - |centre + kernel_point - x|^2 / sigma, 0)
 *   anchor_ctn     f32[b][ks][nc][na] = number of such points (same value for every ks, na)
 * Both outputs are fully written (the reference zero-fills and accumulates with atomics, and reads xyz[3*pm] past the
 * end for its padding threads; neither is reproduced).  ks*na <= 2048. */
int epn_initial_anchor_query_f32(const float *centers, const float *xyz, const float *kernel_points, int b, int nc,
                                 int m, int na, int ks, float radius, float sigma, float *anchor_weights,
                                 float *anchor_ctn, epn_stream_t stream);
/* the scalar_t = double instantiation of the same kernel (AT_DISPATCH_FLOATING_TYPES on xyz.type(),
 * grouping_cuda_kernel.cu:558-563; outputs take the inputs' dtype, grouping_cuda.cpp:149-154).  radius and sigma stay
 * FLOAT parameters as in the reference (:127-128) and are widened where they meet a double. */
int epn_initial_anchor_query_f64(const double *centers, const double *xyz, const double *kernel_points, int b, int nc,
                                 int m, int na, int ks, float radius, float sigma, double *anchor_weights,
                                 double *anchor_ctn, epn_stream_t stream);

/* InterSO3Conv with the grouped features kept ON CHIP (north_star: "the [B,N,K,A,C] tile staged through LDS"): the same
 * result as epn_inter_so3conv_fwd_f32 -- vgtk/vgtk/so3conv/modules.py:157-174 = inter_so3conv_grouping
 * (vgtk/vgtk/spconv/functional.py:372-390) -> BasicSO3Conv (modules.py:48-55) -- with the neighbour contraction as the
 * A-tile producer of the weight contraction; no [cols, cin*ks] tensor is written.
 *   _onchip_f32: fp32 features / output; the weight contraction runs on the bf16 matrix pipe with both operands split
 *                losslessly into three bf16 pieces (as epn_gemm_nt_split_f32: fp32 accuracy; non-finite inputs give NaN).
 *   _bf16:       bf16 features / output ("bf16 features, fp32 accumulate"), W fp32 master weights (rounded per call).
 * Shapes served: epn_inter_onchip_ok(d, bf16) != 0 (cin % 16 == 0, cout % 32 == 0, cout <= 256, ks in {16, 24},
 * nn <= 64, 32 < na <= 64, no dense_w); EPN_EINVAL otherwise.  Workspace: epn_inter_onchip_workspace_bytes. */
int epn_inter_onchip_ok(const epn_inter_desc *d, int bf16);
size_t epn_inter_onchip_workspace_bytes(const epn_inter_desc *d, int bf16);
int epn_inter_so3conv_fwd_onchip_f32(const epn_inter_desc *d, const float *feats_cl, const float *W, float *out_cl,
                                     void *workspace, size_t workspace_bytes, epn_stream_t stream);
int epn_inter_so3conv_fwd_bf16(const epn_inter_desc *d, const void *feats_cl, const float *W, void *out_cl,
                               void *workspace, size_t workspace_bytes, epn_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------------
 * Grouping only ("split" convolution): the grouped features as a tensor, the weight contraction left to the caller's
 * BLAS.  replaces vgtk/vgtk/so3conv/functional.py:118-140 (inter_so3conv_grouping: ball grouping + anchor weights +
 * inter_zpconv_grouping_naive, vgtk/vgtk/spconv/functional.py:372-421) and its autograd backward; what follows it in the
 * reference, BasicSO3Conv.forward (vgtk/vgtk/so3conv/modules.py:38-52: W @ feats.view(b, c*ks, p*a)), stays a matmul.
 *   grouped f32[b*p2*na][cin*ks]   row = column (b, p, a), element c*ks + k  -- the reference's [b, c, ks, p2, na] tensor
 *                                  with the (c, ks) axes last, so  out_cl[col][o] = grouped[col][:] . W[o][:]
 * epn_inter_ungroup_f32 is the transpose: grad_feats_cl[b][idx][a][c] += sum_k w * grad_grouped (zero-fills first).
 * Its workgroups take 8-16 output points that are neighbours in space (Morton order of new_xyz, computed into the
 * workspace), sum the contributions of the slots that name the same input point in LDS and issue one fp32 atomic per
 * distinct (input point, anchor, channel): ~3x fewer atomics than one per slot.
 * cout / dense_w of the descriptor are ignored (dense_w must be NULL).  Neither inter_w nor the gathered neighbour
 * features are materialised; the fused entry points above additionally avoid `grouped` itself (inference, or when
 * HBM is short: `grouped` is cin*ks*4 bytes per column, 6 GB for a 64-channel layer at B=32). */
size_t epn_inter_group_workspace_bytes(const epn_inter_desc *d);
int epn_inter_group_f32(const epn_inter_desc *d, const float *feats_cl, float *grouped, void *workspace,
                        size_t workspace_bytes, epn_stream_t stream);
int epn_inter_ungroup_f32(const epn_inter_desc *d, const float *grad_grouped, float *grad_feats_cl, void *workspace,
                          size_t workspace_bytes, epn_stream_t stream);
/* epn_inter_ungroup_acc_{f32,bf16}: the same transpose ADDED to the contents of grad_feats_cl (fp32) instead of written
 * over zeros -- for a tensor whose gradient also arrives by another path (the input of a SeparableSO3ConvBlock feeds the
 * inter convolution AND the skip branch, base_so3conv.py:186-207): hand in the other path's gradient and neither a
 * zero-fill nor a separate addition pass is needed. */
int epn_inter_ungroup_acc_f32(const epn_inter_desc *d, const float *grad_grouped, float *grad_feats_cl, void *workspace,
                              size_t workspace_bytes, epn_stream_t stream);
int epn_inter_ungroup_acc_bf16(const epn_inter_desc *d, const void *grad_grouped, float *grad_feats_cl, void *workspace,
                               size_t workspace_bytes, epn_stream_t stream);

/* Packed column order of `grouped` (what the split convolution of this package uses between its own kernels).  In the
 * element order c*ks + k above, one store instruction of the grouping kernel -- four kernel points of 16 channels -- is 16
 * pieces of 64 (or 32) bytes, ks*4 bytes or more apart; the kernel is bound by those stores (6 GB per 64-channel layer at
 * B=32).  In the packed order every store instruction covers one contiguous 1 KiB / 512 B range of the row: 8-20 % less
 * kernel time.  A matrix product does not care about the order of its contraction index as long as both operands agree, so
 * the WEIGHTS are permuted instead (a few MB): same result as the plain order bit for bit up to summation order.
 *   epn_inter_packed_position(cin, ks, position[cin*ks]) : host table, position[c*ks + k] = column of element (c, k):
 *       CG = 4 if cin % 64 == 0 else 2;  slot(c) = c - c % (16 CG) + 16 (c % CG) + (c % (16 CG)) / CG;
 *       k < 16:  slot * min(ks, 16) + k;   k >= 16:  min(ks, 16) * cin + slot * (ks - 16) + (k - 16)
 *   epn_inter_group_packed_{f32,bf16} : as epn_inter_group_*, columns in packed order; shapes: epn_inter_group_packed_ok
 *       (the MFMA grouping kernel's shapes with na >= 16, cin % 32 == 0, nn <= 64), EPN_EINVAL otherwise
 *   epn_inter_pack_weights_{f32,bf16} : packed[o][position[q]] = W[o][q]   (fp32 master weights -> fp32 / bf16 operand)
 *   epn_inter_unpack_weight_grad_f32  : grad_W[o][q] = grad_packed[o][position[q]]   (gradient computed against packed G)
 * The gradient of `grouped` stays in the plain order (epn_inter_ungroup_*: its reads are not what bounds that kernel). */
int epn_inter_group_packed_ok(const epn_inter_desc *d);
int epn_inter_packed_position(int cin, int ks, int32_t *position);
int epn_inter_group_packed_f32(const epn_inter_desc *d, const float *feats_cl, float *grouped, void *workspace,
                               size_t workspace_bytes, epn_stream_t stream);
int epn_inter_group_packed_bf16(const epn_inter_desc *d, const void *feats_cl, void *grouped, void *workspace,
                                size_t workspace_bytes, epn_stream_t stream);
int epn_inter_pack_weights_f32(const float *W, int cout, int cin, int ks, float *packed, epn_stream_t stream);
int epn_inter_pack_weights_bf16(const float *W, int cout, int cin, int ks, void *packed, epn_stream_t stream);
int epn_inter_unpack_weight_grad_f32(const float *grad_packed, int cout, int cin, int ks, float *grad_W,
                                     epn_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------------
 * Composed split form: InterSO3Conv forward / backward as ONE call per direction on the kernel chain the benchmark times
 * replaces InterSO3Conv.forward (vgtk/vgtk/so3conv/modules.py:157-174: L.inter_so3conv_grouping, so3conv/functional.py:118-178,
 * then BasicSO3Conv's matmul, modules.py:48-55) and the autograd transposes of both -- the same boundary as
 * epn_inter_so3conv_{fwd,bwd_data,bwd_weight}_f32 above (the fused fp32-MFMA kernels), but running
 *   forward : epn_inter_group_packed_* (plain order where the packed one does not apply) -> epn_inter_pack_weights_* ->
 *             epn_gemm_nt_split_f32 (fp32: lossless 3 x bf16 split, fp32 accuracy) / epn_gemm_nt_bf16
 *   backward: epn_gemm_tn_split_f32 / _bf16 (+ epn_inter_unpack_weight_grad_f32) for grad_W;
 *             epn_transpose_cast -> epn_gemm_nt_* -> epn_inter_ungroup[_acc]_* for grad_feats
 * i.e. exactly what epn_pointcloud_amd/ops.py's InterSO3ConvSplitFn issues; a binding of the reference's four-call surface
 * (grouping_cuda.cpp:176-181 + the two matmul gradients) gets the benchmarked path.  Buffers are the caller's:
 *   saved     : the grouped features G[b*p2*na][cin*ks] (feature dtype) written by forward and read by backward --
 *               epn_inter_split_saved_bytes(d, bf16); column order is an implementation detail of the pair
 *   workspace : scratch of one call -- epn_inter_split_workspace_bytes(d, bf16, backward_pass); the backward scratch holds
 *               the gradient of the grouped features (same size as `saved`)
 *   out_col_stats (optional, NULL: off): per-32-row-block column statistics of out_cl, see epn_gemm_nt_problem::col_stats
 *   backward: grad_feats_cl (fp32 scatter target for either feature dtype; NULL: not wanted; accumulate != 0: ADD to its
 *             contents, see epn_inter_ungroup_acc_*), grad_W f32[cout][cin*ks] (NULL: not wanted).
 * Shapes: the MFMA grouping kernels' (no dense inter_w, cin % 16 == 0, ks % 4 == 0, ks <= 32, nn <= 128) -- epn_inter_split_ok;
 * EPN_EINVAL otherwise (use the fused entry points).  feats_cl / out_cl / grad_out_cl are float (…_f32) or bf16 (…_bf16);
 * W and grad_W are always fp32 (master weights). */
int epn_inter_split_ok(const epn_inter_desc *d);
size_t epn_inter_split_saved_bytes(const epn_inter_desc *d, int bf16);
size_t epn_inter_split_workspace_bytes(const epn_inter_desc *d, int bf16, int backward_pass);
int epn_inter_so3conv_fwd_split_f32(const epn_inter_desc *d, const float *feats_cl, const float *W, float *out_cl,
                                    float *out_col_stats, void *saved, size_t saved_bytes, void *workspace,
                                    size_t workspace_bytes, epn_stream_t stream);
int epn_inter_so3conv_fwd_split_bf16(const epn_inter_desc *d, const void *feats_cl, const float *W, void *out_cl,
                                     float *out_col_stats, void *saved, size_t saved_bytes, void *workspace,
                                     size_t workspace_bytes, epn_stream_t stream);
int epn_inter_so3conv_bwd_split_f32(const epn_inter_desc *d, const float *grad_out_cl, const float *W, const void *saved,
                                    size_t saved_bytes, float *grad_feats_cl, int accumulate, float *grad_W,
                                    void *workspace, size_t workspace_bytes, epn_stream_t stream);
int epn_inter_so3conv_bwd_split_bf16(const epn_inter_desc *d, const void *grad_out_cl, const float *W, const void *saved,
                                     size_t saved_bytes, float *grad_feats_cl, int accumulate, float *grad_W,
                                     void *workspace, size_t workspace_bytes, epn_stream_t stream);

/* IntraSO3Conv grouping as a tensor: replaces L.intra_so3conv_grouping (vgtk/vgtk/so3conv/functional.py:255-268,
 * feats[..., intra_idx]); IntraSO3Conv's BasicSO3Conv matmul (modules.py:197-200) then runs on the caller's BLAS.
 *   grouped f32[b*p*na][kn*c]   element k*c + ci = feats_cl[b][p][intra_idx[a][k]][ci]   (anchor-neighbour major, so
 *   the matching weight is W[o][ci*kn + k] re-ordered to [o][k*c + ci]; the reference tensor is [b, c, kn, p, na]).
 * Its transpose is the same gather through the inverse permutation (epn_intra_so3conv_bwd_data_f32 covers it fused). */
int epn_intra_group_f32(const float *feats_cl, const int32_t *intra_idx, float *grouped, int b, int p, int na, int kn,
                        int c, epn_stream_t stream);

/* Change of anchor basis for the block-diagonal form of IntraSO3Conv (epn_pointcloud_amd/so3_fourier.py): the twelve
 * anchor permutations of L.intra_so3conv_grouping (vgtk/vgtk/so3conv/functional.py:255-268) are simultaneously
 * block-diagonalised by one orthogonal matrix U, so IntraSO3Conv.forward (modules.py:197-200) becomes: U^T, one GEMM per
 * irreducible block, U -- 2.95x fewer flops than the 12-neighbour contraction and no grouped tensor.
 *   out[pt][r][ch] = sum_s M[r][s] * in[pt][s][ch]      M f32[na][na] row-major, pts points, c channels (c % 32 == 0)
 * `*_spectral` = 0: plain channels-last rows ((pt*na + r)*c);  1: row f of a point lives in its block's buffer at
 * ((blocks[f][0]*pts + pt*blocks[f][1] + (f - blocks[f][0]))*c), blocks i32[na][2] = (first row of the block, d*d), so
 * each block is a dense row-major [pts*d][d*c] matrix.  Forward transform: M = U^T, out_spectral = 1; inverse: M = U,
 * in_spectral = 1 (each is the other's transpose, i.e. its backward). */
int epn_so3_basis_f32(const float *in, const float *M, const int32_t *blocks, long long pts, int na, int c,
                      int in_spectral, int out_spectral, float *out, epn_stream_t stream);
/* The same transform (out_spectral must be 0) that also writes point_stats[pt][c][2] = (sum, sum of squares) over the na
 * output rows of every point, from the accumulators: the block partials (one block per point) of the per-channel statistics
 * of `out` for epn_stats_finish -- the InstanceNorm that follows IntraSO3Conv (base_so3conv.py:204-211) without a
 * statistics pass over `out`.  _split_f32: fp32 on the bf16 matrix pipe; _bf16: bf16 features (sums of the
 * rounded values). */
/* IntraSO3Conv's weights in the block-diagonal basis (so3_fourier.py): for every irreducible block rho (dimension d, first
 * spectral row `base`, blocks[f] = (base, d*d) as above) What^rho[(j, c), (i, o)] = sum_k W[o, c, k] R[base + i d + j][k],
 * R f32[na][kn] = the representation matrices rho(g_k)[i, j] of the kn anchor neighbours.  what: flat, block rho at offset
 * base * cin * cout as a row-major [d*cin][d*cout] matrix; what_t: the same blocks transposed ([d*cout][d*cin], the Bt
 * operand of epn_gemm_nt for the forward product); either may be NULL.  epn_spectral_weights_bwd_f32 is the transpose
 * (grad_W[o][c][k] from grad_what in the `what` layout).  Replaces nothing in the reference (the spectral form is this
 * library's); it replaces ~25 small torch launches per layer and direction. */
int epn_spectral_weights_f32(const float *W, const float *R, const int32_t *blocks, int cout, int cin, int kn, int na,
                             float *what, float *what_t, epn_stream_t stream);
/* the same blocks rounded to bf16 (fp32 master weights -> the operands of a bf16 network's GEMMs) */
int epn_spectral_weights_bf16(const float *W, const float *R, const int32_t *blocks, int cout, int cin, int kn, int na,
                              void *what, void *what_t, epn_stream_t stream);
int epn_spectral_weights_bwd_f32(const float *grad_what, const float *R, const int32_t *blocks, int cout, int cin, int kn,
                                 int na, float *grad_W, epn_stream_t stream);
int epn_so3_basis_stats_f32(const float *in, const float *M, const int32_t *blocks, long long pts, int na, int c,
                            int in_spectral, int out_spectral, float *out, float *point_stats, epn_stream_t stream);
int epn_so3_basis_stats_split_f32(const float *in, const float *M, const int32_t *blocks, long long pts, int na, int c,
                                  int in_spectral, int out_spectral, float *out, float *point_stats, epn_stream_t stream);
int epn_so3_basis_stats_bf16(const void *in, const float *M, const int32_t *blocks, long long pts, int na, int c,
                             int in_spectral, int out_spectral, void *out, float *point_stats, epn_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------------
 * PointnetSO3Conv: the aggregation tail of every shipped model
 * replaces vgtk/vgtk/so3conv/modules.py:203-235 (PointnetSO3Conv.forward: centre xyz, rotate it into every anchor frame
 * with einsum 'aji,bjn->bina', concatenate to the features, 1x1 Conv2d "embed", torch.max over the point axis) -- one
 * fused pass, neither the concatenated tensor nor the per-point embedding is written to HBM.
 *   feats_cl f32[b][p][a][c]   channels-last features          xyz   f32[b][3][p]
 *   anchors  f32[a][3][3] or NULL (a == 1: no rotation, modules.py:227-228)
 *   W        f32[co][c+3]      embed.weight (feature channels first, then the 3 coordinates)     bias f32[co] or NULL
 *   out      f32[b][a][co]     = logical [b, co, a] viewed channels-last
 *   argmax   i32[b][a][co]     point index of the maximum (first maximum wins), consumed by the backward entry points
 *   centre   f32[b][3]         per-cloud mean of xyz, consumed by epn_pointnet_so3conv_bwd_weight_f32
 * Backward = torch.max backward composed with the 1x1 convolution: gradients flow through the arg-max point only.
 * xyz receives no gradient (the reference never asks for one: coordinates are inputs). */
int epn_pointnet_so3conv_fwd_f32(const float *feats_cl, const float *xyz, const float *anchors, const float *W,
                                 const float *bias, float *out, int32_t *argmax, float *centre, int b, int p, int a,
                                 int c, int co, epn_stream_t stream);
int epn_pointnet_so3conv_bwd_data_f32(const float *grad_out, const int32_t *argmax, const float *W,
                                      float *grad_feats_cl, int b, int p, int a, int c, int co, epn_stream_t stream);
int epn_pointnet_so3conv_bwd_weight_f32(const float *grad_out, const int32_t *argmax, const float *feats_cl,
                                        const float *xyz, const float *anchors, const float *centre, float *grad_W,
                                        float *grad_bias, int b, int p, int a, int c, int co, epn_stream_t stream);

/* PointnetSO3Conv composed with the library's GEMMs (same module, vgtk/vgtk/so3conv/modules.py:219-235; the form the
 * benchmarked networks run for c % 16 == 0): the caller forms the embedding's feature part with epn_gemm_nt_*
 *     Z[b][p][a][o] = sum_c W[o][c] F[b][p][a][c]          (rows = the channels-last feature rows; fp32 result)
 * and these entries do the rest.
 *   epn_pointnet_max_f32:  out[b][a][o] = max_p (Z + sum_j W[o][c+j] ext_j[b][p][a] + bias[o]), argmax = first maximum,
 *                          centre[b][3] = mean_p xyz (ext = R_a^T (xyz - centre); anchors NULL: identity).  W is the
 *                          full [co][c+3] weight, of which only the three coordinate columns are read.
 *   epn_pointnet_dz_*:     dZ[b][p][a][o] = argmax[b][a][o] == p ? grad_out[b][a][o] : 0 (every element written;
 *                          co % 8 == 0, 16-byte aligned pointers) -- then dF = dZ W (epn_gemm_nt_*), dW[:, :c] = dZ^T F
 *                          (epn_gemm_tn_*).  _bf16: dZ in bf16 for bf16 features.
 *   epn_pointnet_bwd_coord_f32: grad_W[o][c + j] = sum_{b,a} grad_out ext_j[b][argmax][a] (the other columns of the
 *                          [co][c+3] buffer are not touched), grad_bias[o] = sum_{b,a} grad_out (may be NULL); fixed
 *                          summation order. */
int epn_pointnet_max_f32(const float *Z, const float *xyz, const float *anchors, const float *W, const float *bias,
                         float *out, int32_t *argmax, float *centre, int b, int p, int a, int c, int co,
                         epn_stream_t stream);
int epn_pointnet_dz_f32(const float *grad_out, const int32_t *argmax, float *dZ, int b, int p, int a, int co,
                        epn_stream_t stream);
int epn_pointnet_dz_bf16(const float *grad_out, const int32_t *argmax, void *dZ, int b, int p, int a, int co,
                         epn_stream_t stream);
int epn_pointnet_bwd_coord_f32(const float *grad_out, const int32_t *argmax, const float *xyz, const float *anchors,
                               const float *centre, float *grad_W, float *grad_bias, int b, int p, int a, int c, int co,
                               epn_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------------
 * vgtk.cuda.zpconv: grouping functions of the legacy ZPConv path (unreachable from the shipped models,
 * provided for API completeness).  Layouts are the reference's (channel-major, contiguous).
 * replaces inter_zpconv_forward / inter_zpconv_backward / intra_zpconv_forward / intra_zpconv_backward
 * (vgtk/vgtk/cuda/zpconv_cuda.cpp:41-112; kernels zpconv_cuda_kernel.cu:33-195).
 *   inter: anchor_neighbors i32[b][np][na][ks][ann], anchor_weights f32 (same shape), feats f32[b][c][nq][na]
 *          anchor_feats f32[b][c][ks][np][na] = sum_ni feats[b][c][nbr][a] * w      (forward: gather, no atomics)
 *          grad_feats   f32[b][c][nq][na]  (zero-filled here, fp32 atomic scatter)
 *   intra: anchor_neighbors i32[na_out][ann], anchor_weights f32[na_out][ks][ann], feats f32[b][c][np][na_in]
 *          anchor_feats f32[b][c][ks][np][na_out];  grad_feats f32[b][c][np][na_in]
 * Indices outside [0, nq) / [0, na_in) contribute nothing (the reference reads out of bounds). */
/* replaces vgtk.cuda.grouping.anchor_query (vgtk/vgtk/cuda/grouping_cuda.cpp:88-108, kernel
 * grouping_cuda_kernel.cu:180-247; legacy ZPConv, every call site in the reference is commented out):
 *   grouped_xyz f32[b][3][np][nn] (local coordinates), anchors f32[na][3] (unit directions), kernel_points f32[ks][2]
 *   anchor_weights f32[b][np][na][ks][nn] = (kw - norm)^2 + ((kh - theta) * norm)^2,
 *       norm = |g| + 1e-6, theta = acos(g . anchor / norm)      (fully written; the reference pre-fills 1e6)
 * The reference's sample_idx / grouped_indices / nq arguments are unused by its kernel and not part of this entry. */
int epn_anchor_query_f32(const float *grouped_xyz, const float *anchors, const float *kernel_points, int b, int np,
                         int nn, int na, int ks, float *anchor_weights, epn_stream_t stream);
/* scalar_t = double (grouping_cuda_kernel.cu:505-510; the output takes grouped_xyz's dtype, grouping_cuda.cpp:103-104).
 * In BOTH instantiations `+ 1e-6` adds a double literal (:221): the f32 entry forms |g| + 1e-6 in double and rounds
 * once. */
int epn_anchor_query_f64(const double *grouped_xyz, const double *anchors, const double *kernel_points, int b, int np,
                         int nn, int na, int ks, double *anchor_weights, epn_stream_t stream);
int epn_zp_inter_fwd_f32(const int32_t *anchor_neighbors, const float *anchor_weights, const float *feats, int b, int c,
                         int np, int nq, int na, int ks, int ann, float *anchor_feats, epn_stream_t stream);
int epn_zp_inter_bwd_f32(const int32_t *anchor_neighbors, const float *anchor_weights, const float *grad_anchor_feats,
                         int b, int c, int np, int nq, int na, int ks, int ann, float *grad_feats, epn_stream_t stream);
int epn_zp_intra_fwd_f32(const int32_t *anchor_neighbors, const float *anchor_weights, const float *feats, int b, int c,
                         int np, int na_in, int na_out, int ks, int ann, float *anchor_feats, epn_stream_t stream);
int epn_zp_intra_bwd_f32(const int32_t *anchor_neighbors, const float *anchor_weights, const float *grad_anchor_feats,
                         int b, int c, int np, int na_in, int na_out, int ks, int ann, float *grad_feats,
                         epn_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------------
 * BasicSO3Conv's weight contraction and its autograd transposes as hand-written MFMA GEMMs
 * replaces BasicSO3Conv.forward (vgtk/vgtk/so3conv/modules.py:48-55: `torch.matmul(self.W, x.view(b, c*ks, p*a))`) and
 * the two matmul gradients torch autograd derives from it.  Operands are row-major with explicit leading dimensions.
 *   NT: C[M][N]   = A[M][K] . Bt[N][K]^T   (activations x weights: out = grouped W^T, grad_grouped = grad_out (W^T)^T,
 *                                           the irreducible blocks of the spectral IntraSO3Conv); up to any number of
 *                                           problems per call (one grouped launch per 6).
 *   TN: C[N1][N2] = X[R][N1]^T . Y[R][N2]  (weight gradients: contraction over the R = b*p*a columns; split over R into
 *                                           fp32 partials in `workspace`, summed in a fixed order: deterministic).
 * fp32: v_mfma_f32_32x32x2_f32, exact f32 (the split entry points below: the bf16 pipe at fp32 accuracy).  bf16: bf16 operands, fp32 accumulation; NT writes bf16 (or fp32 when
 * out_f32 != 0), TN always writes fp32.  Fast path: K % 32 == 0 (fp32) / 64 (bf16), 16-byte aligned rows; anything
 * else runs on a generic kernel.  C is fully overwritten. */
typedef struct epn_gemm_nt_problem {
    const void *A, *Bt;
    void *C;
    long long M, lda, ldb, ldc;
    int N, K;
    /* Optional (NULL: off): per-column statistics of C taken from the accumulators in the kernel's epilogue --
     * col_stats[(m / 32)][n][2] = (sum, sum of squares) of rows 32 (m / 32) .. + 31 of column n, as stored (bf16 outputs:
     * of the rounded values); M % 32 == 0.  epn_stats_finish turns them into the sums[groups][N][2] the norm entry points
     * take: the per-channel statistics pass of a following BatchNorm / InstanceNorm (base_so3conv.py:196-204) without
     * reading C again. */
    float *col_stats;
    /* Optional (NULL: off; library 0.4): device scalar the kernel RAISES to max|C| (values as stored; non-finite ones left
     * out) from its accumulators -- zero it before the call.  What epn_inter_ungroup_cloud_* takes as dg_amax when C is the
     * gradient of the grouped features. */
    float *c_amax;
} epn_gemm_nt_problem;
int epn_gemm_nt_f32(int nprob, const epn_gemm_nt_problem *probs, epn_stream_t stream);
/* sums[g][c][2] = sum over the blocks_per_group consecutive 32-row blocks of group g of partials[block][c][2] (fixed
 * order: deterministic).  groups = 1 for BatchNorm2d, the number of clouds for InstanceNorm2d (rows per cloud % 32 == 0). */
size_t epn_stats_finish_workspace_bytes(int groups, long long blocks_per_group, int c);
int epn_stats_finish(const float *partials, int groups, long long blocks_per_group, int c, float *sums, void *workspace,
                     size_t workspace_bytes, epn_stream_t stream);
int epn_gemm_nt_bf16(int nprob, const epn_gemm_nt_problem *probs, int out_f32, epn_stream_t stream);
/* Split form of the fp32 NT contraction: the same fp32 operands and fp32 result, computed on the bf16 matrix pipe.
 * Every fp32 value is split WITHOUT LOSS into three bf16 pieces (round-to-nearest remainders); the six piece products
 * of weight >= 2^-16 are accumulated in fp32, the three dropped ones are below 2^-24 |a||b| (less than the rounding
 * of one fp32 FMA): fp32 accuracy at 2.7x the matrix rate of v_mfma_f32_32x32x2_f32.  Bt (the weights) is split once
 * per call into `workspace` (epn_gemm_nt_split_workspace_bytes); problems that do not qualify (K % 32, row alignment)
 * or a missing workspace run on epn_gemm_nt_f32's kernels. */
size_t epn_gemm_nt_split_workspace_bytes(int nprob, const epn_gemm_nt_problem *probs);
int epn_gemm_nt_split_f32(int nprob, const epn_gemm_nt_problem *probs, void *workspace, size_t workspace_bytes,
                          epn_stream_t stream);
/* `bf16`: 0 = fp32 operands (epn_gemm_tn_f32), 1 = bf16 (epn_gemm_tn_bf16), 2 = fp32 operands in the split form
 * (epn_gemm_tn_split_f32: the partial slabs plus, for N2 >= 512, the three bf16 planes of X = 6 bytes per value) */
size_t epn_gemm_tn_workspace_bytes(int bf16, long long R, int N1, int N2);
int epn_gemm_tn_f32(const float *X, long long ldx, const float *Y, long long ldy, float *C, long long ldc, long long R,
                    int N1, int N2, void *workspace, size_t workspace_bytes, epn_stream_t stream);
int epn_gemm_tn_bf16(const void *X, long long ldx, const void *Y, long long ldy, float *C, long long ldc, long long R,
                     int N1, int N2, void *workspace, size_t workspace_bytes, epn_stream_t stream);
/* split form of epn_gemm_tn_f32 (see epn_gemm_nt_split_f32).  Wide outputs (N2 >= 512): X, the narrow operand, is split
 * ahead of the GEMM into bf16 planes laid out [plane][R/8][N1][8] in the workspace, Y in registers; narrow ones: both
 * operands in registers.  Workspace size: epn_gemm_tn_workspace_bytes(2, ...); too small -> EPN_EWORKSPACE. */
int epn_gemm_tn_split_f32(const float *X, long long ldx, const float *Y, long long ldy, float *C, long long ldc,
                          long long R, int N1, int N2, void *workspace, size_t workspace_bytes, epn_stream_t stream);
/* Grouped TN (`bf16`: 0 = fp32, 1 = bf16 operands, 2 = fp32 operands in the split form): up to 6 problems in ONE launch (the five weight-gradient GEMMs of a spectral IntraSO3Conv layer, each too
 * small to fill the chip alone); splits are planned so that every workgroup runs about the same number of K steps. */
typedef struct epn_gemm_tn_problem {
    const void *X, *Y;
    float *C;
    long long R, ldx, ldy, ldc;
    int N1, N2;
} epn_gemm_tn_problem;
size_t epn_gemm_tn_grouped_workspace_bytes(int bf16, int nprob, const epn_gemm_tn_problem *probs);
int epn_gemm_tn_grouped(int bf16, int nprob, const epn_gemm_tn_problem *probs, void *workspace, size_t workspace_bytes,
                        epn_stream_t stream);

/* ---- fp32 contractions as THREE fp16 matrix products ("f16x2", round 5; csrc/gemm.h) -------------------------------------
 * Same math as BasicSO3Conv.forward and its autograd transposes (vgtk/vgtk/so3conv/modules.py:48-55), fp32 operands, fp32
 * accumulation, fp32 result.  Each operand is scaled by the power of two that puts its largest magnitude at 2^14 and split
 * into two fp16 pieces x 2^s = h + l (22-23 significant bits for every element within 2^-17 of the tensor's maximum,
 * absolute error <= max|x| 2^-39 below); a product is hh + hl + lh on v_mfma_f32_32x32x16_f16 -- half the matrix
 * instructions of the lossless three-piece bf16 form (epn_gemm_*_split_f32), with a measured error against fp64 at or below
 * that of the fp32 matrix instruction.  The scale comes from a DEVICE scalar holding max|x|: pass the
 * pointer when a producer already knows it (epn_absmax_f32 computes one: memset + one pass), or NULL and the entry point
 * makes that pass itself into its workspace.  Nothing is synchronised with the host; capturable into a HIP graph.
 * `a_amax` / `x_amax` / `y_amax`: arrays of nprob device pointers (entries may be NULL), or NULL.
 * Workspace: epn_gemm_nt_f16x2_workspace_bytes; epn_gemm_tn_workspace_bytes(3, ...) / epn_gemm_tn_grouped_workspace_bytes(3, ...). */
/* The anchor basis change in the fp32 split form + max |out| into the device scalar *amax_out (zeroed by the call): the
 * producer-side maximum of the spectral buffers, so that the two-piece GEMMs that read them need no pass of their own.
 * Arguments as epn_so3_basis_split_f32 / epn_so3_basis_norm_split_f32. */
int epn_so3_basis_amax_split_f32(const float *in, const float *M, const int32_t *blocks, long long pts, int na, int c,
                                 int in_spectral, int out_spectral, float *out, float *amax_out, epn_stream_t stream);
int epn_so3_basis_norm_amax_split_f32(const float *in, const float *M, const int32_t *blocks, long long pts, int na, int c,
                                      int out_spectral, float *out, const float *sums, int groups, long long pts_per_group,
                                      const float *gamma, const float *beta, float eps, float slope, float *amax_out,
                                      epn_stream_t stream);
int epn_absmax_f32(const float *src, long long ld, long long rows, long long cols, float *out, epn_stream_t stream);
size_t epn_gemm_nt_f16x2_workspace_bytes(int nprob, const epn_gemm_nt_problem *probs);
int epn_gemm_nt_f16x2_f32(int nprob, const epn_gemm_nt_problem *probs, const float *const *a_amax, void *workspace,
                          size_t workspace_bytes, epn_stream_t stream);
int epn_gemm_tn_f16x2_f32(const float *X, long long ldx, const float *Y, long long ldy, float *C, long long ldc, long long R,
                          int N1, int N2, const float *x_amax, const float *y_amax, void *workspace, size_t workspace_bytes,
                          epn_stream_t stream);
int epn_gemm_tn_grouped_f16x2(int nprob, const epn_gemm_tn_problem *probs, const float *const *x_amax, const float *const *y_amax,
                              void *workspace, size_t workspace_bytes, epn_stream_t stream);
/* Data gradient of InterSO3Conv with the gradient of the grouped features kept ON CHIP (round 6; csrc/inter_bwd_f2.hip):
 *     grad_feats[b, idx[b,p,n], a, c] (+)= sum_k w[b,p,a,k,n] * sum_o grad_out[col][o] W[o][c*ks + k]
 * -- autograd's transpose of BasicSO3Conv's matmul (vgtk/vgtk/so3conv/modules.py:48-55) chained with the scatter-add that is
 * the backward of inter_zpconv_grouping_naive's gather (vgtk/vgtk/spconv/functional.py:372-390), in ONE kernel: the
 * [cols, cin*ks] gradient of the grouped features (which epn_gemm_nt_f16x2_f32 + epn_inter_ungroup_f32 write to and read back
 * from HBM: 27 GB per classification step) lives in registers.  The contraction over the output channels runs in the two-piece
 * fp16 form (three v_mfma_f32_16x16x32_f16 per block, fp32 accumulate) inside the workgroup of the LDS-pre-reduced scatter.
 * go_amax: device scalar max|grad_out| (required: epn_absmax_f32 computes one).  accumulate != 0: added to what grad_feats_cl
 * holds.  Shapes: ks == 24, 16 <= na <= 64, nn <= 32, cout in {64, 128, 256}, cin % 16 == 0, p2 % 8 == 0 (epn_..._ok). */
int epn_inter_bwd_data_f16x2_ok(const epn_inter_desc *d);
size_t epn_inter_bwd_data_f16x2_workspace_bytes(const epn_inter_desc *d);
int epn_inter_bwd_data_f16x2_f32(const epn_inter_desc *d, const float *grad_out_cl, const float *W, const float *go_amax,
                                 float *grad_feats_cl, int accumulate, void *workspace, size_t workspace_bytes,
                                 epn_stream_t stream);
/* Transpose of the grouping with a whole cloud's gradient rows resident in LDS (round 6; csrc/inter_ungroup_cloud.hip):
 *     grad_feats[b, idx[b,p,n], a, c] = (add[...] +) sum over (p, n) of sum_k w[b,p,a,k,n] * grad_grouped[(b,p,a)][c*ks + k]
 * -- the backward of inter_zpconv_grouping_naive's gather (vgtk/vgtk/spconv/functional.py:372-390), as epn_inter_ungroup_*, with
 * NO global atomics: a workgroup owns every output point of one (cloud, anchor) and with them the destination rows outright,
 * accumulates them in LDS as 64-bit fixed-point integers (integer LDS atomics run at the LDS store rate on gfx950, the
 * floating-point ones 31 x slower) and writes them once.  Consequences: no zero fill of the target, the result is written in the
 * gradient's own type (bf16 entry: bf16 in, bf16 out -- no fp32 scatter target + conversion pass; out_f32 != 0: fp32 out and
 * add, the contract of epn_inter_ungroup_bf16), `add` (same shape and type as
 * grad_feats_cl; may BE grad_feats_cl; or NULL) is folded into the write-out, and the sum does not depend on the order in which
 * waves arrive: bitwise repeatable.
 * Fixed point: unit = max|grad_grouped| * 2^-(43 .. 50) (from the device scalar *dg_amax and the largest multiplicity of the
 * index table; every contribution is rounded ONCE to the unit, the sum is exact, one rounding to the output type).  dg_amax
 * NULL: the entry takes the maximum itself (one more pass over grad_grouped).  An UNDERSTATED maximum makes contributions leave
 * the representable range: such a workgroup writes NaN to its rows and bumps a sticky device counter
 * (epn_inter_ungroup_cloud_range_count; reset != 0 clears it), as does a non-finite grad_grouped.
 * Shapes (epn_..._ok): the MFMA grouping's (cin % 16 == 0, ks % 4 == 0, ks <= 32, nn <= 64), p2 <= 4096, and p1 small enough
 * for (p1 + 4) * 16 * 8 bytes of accumulators (p1 <= 1148). */
int epn_inter_ungroup_cloud_ok(const epn_inter_desc *d);
size_t epn_inter_ungroup_cloud_workspace_bytes(const epn_inter_desc *d);
int epn_inter_ungroup_cloud_f32(const epn_inter_desc *d, const float *grad_grouped, const float *dg_amax, float *grad_feats_cl,
                                const float *add, void *workspace, size_t workspace_bytes, epn_stream_t stream);
int epn_inter_ungroup_cloud_bf16(const epn_inter_desc *d, const void *grad_grouped, const float *dg_amax, void *grad_feats_cl,
                                 const void *add, int out_f32, void *workspace, size_t workspace_bytes, epn_stream_t stream);
long long epn_inter_ungroup_cloud_range_count(int reset);
/* The scale contract of the two-piece form made loud (round 6).  The power-of-two scale leaves a factor 2-4 below fp16's
 * 65504, so an operand element above 2-4 x the maximum the caller REPORTED becomes inf in the split and the product silently
 * non-finite -- where the fp32 matmul these entry points replace (torch.matmul in vgtk/vgtk/so3conv/modules.py:48-55) would
 * have returned a finite number.  Every two-piece kernel checks its accumulators once per tile in the epilogue; a wave that
 * ends with a non-finite accumulator bumps a sticky per-device counter.  epn_f16x2_overflow_count returns that count for the
 * CURRENT device (reset != 0 also clears it); with finite operands any non-zero value is a violated maximum (a stale or
 * under-reported `*_amax`).  Non-finite operands raise it too (their rows are legitimately non-finite) -- clear the counter
 * after feeding such inputs on purpose.  Synchronous (a 4-byte device read): call it outside timed regions and never during
 * stream capture.  A negative return is a negated hipError_t.  bench.py reports it in its line; tests/conftest.py fails any
 * GPU test that leaves it non-zero. */
long long epn_f16x2_overflow_count(int reset);
/* dst[cols][rows] = src[rows][cols]^T with an optional fp32 <-> bf16 conversion (weights: W^T for the data gradient,
 * bf16 copies of the fp32 master weights); epn_cast converts a flat array. */
int epn_transpose_cast(const void *src, void *dst, int rows, int cols, int src_bf16, int dst_bf16, epn_stream_t stream);
int epn_cast(const void *src, void *dst, size_t n, int src_bf16, int dst_bf16, epn_stream_t stream);
/* dst bf16[n] = bf16(src f32[n] + float(add bf16[n])) in one pass (16-byte aligned pointers): the fp32 scatter target of a
 * bf16 network's InterSO3Conv data gradient converted AND added to the gradient that reached the same tensor through the
 * block's skip branch. */
int epn_cast_add_bf16(const float *src, const void *add, void *dst, size_t n, epn_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------------
 * bf16 feature path (BASELINE configs 3-4: "bf16 features / fp32 accumulate"; the reference dispatches float/double only,
 * grouping_cuda_kernel.cu:477,638, so these follow the same interfaces as their _f32 twins with feature tensors stored
 * as bfloat16).  Coordinates, indices, kernel-influence weights w, statistics, affine parameters, weight gradients
 * and every accumulation stay fp32; products inside the GEMMs are bf16 x bf16 -> fp32.
 *   epn_inter_group_bf16   : feats_cl bf16 [b][p1][na][cin] -> grouped bf16 [b*p2*na][cin*ks]   (cin % 16 == 0)
 *   epn_inter_ungroup_bf16 : grad_grouped bf16 -> grad_feats_cl **fp32** (atomic scatter target; convert with epn_cast)
 *   epn_intra_group_bf16   : pure gather of bf16 rows (c % 8 == 0)
 *   epn_so3_basis_bf16, epn_chan_stats_bf16, epn_norm_act_*_bf16: as the _f32 entry points, x / y / dy / dx / residual
 *   in bf16.  The first layer (cin = 1) and PointnetSO3Conv run their fp32 kernels on tensors converted by epn_cast. */
int epn_inter_group_bf16(const epn_inter_desc *d, const void *feats_cl, void *grouped, void *workspace,
                         size_t workspace_bytes, epn_stream_t stream);
int epn_inter_ungroup_bf16(const epn_inter_desc *d, const void *grad_grouped, float *grad_feats_cl, void *workspace,
                           size_t workspace_bytes, epn_stream_t stream);
int epn_intra_group_bf16(const void *feats_cl, const int32_t *intra_idx, void *grouped, int b, int p, int na, int kn,
                         int c, epn_stream_t stream);
int epn_so3_basis_bf16(const void *in, const float *M, const int32_t *blocks, long long pts, int na, int c,
                       int in_spectral, int out_spectral, void *out, epn_stream_t stream);

/* Block glue folded into the basis change ("norm on load", SPConvNets/utils/base_so3conv.py:196-204: InterSO3ConvBlock's
 * norm + leaky_relu feeding IntraSO3Conv): out = basis_change(leaky_relu(norm(in))) with `in` in the plain channels-last
 * layout and sums[g][c] = (sum x, sum x^2) from epn_chan_stats_* (groups = 1: BatchNorm2d, or the number of clouds with
 * pts_per_group points each: InstanceNorm2d; gamma / beta may be NULL).  The normalised tensor is never written. */
int epn_so3_basis_norm_f32(const float *in, const float *M, const int32_t *blocks, long long pts, int na, int c,
                           int out_spectral, float *out, const float *sums, int groups, long long pts_per_group,
                           const float *gamma, const float *beta, float eps, float slope, epn_stream_t stream);
int epn_so3_basis_norm_bf16(const void *in, const float *M, const int32_t *blocks, long long pts, int na, int c,
                            int out_spectral, void *out, const float *sums, int groups, long long pts_per_group,
                            const float *gamma, const float *beta, float eps, float slope, epn_stream_t stream);
/* fp32 in / fp32 out on the bf16 matrix pipe (split form, see epn_gemm_nt_split_f32): M and the input rows are split
 * without loss into three bf16 pieces, six piece products per multiply, fp32 accumulation */
int epn_so3_basis_split_f32(const float *in, const float *M, const int32_t *blocks, long long pts, int na, int c,
                            int in_spectral, int out_spectral, float *out, epn_stream_t stream);
int epn_so3_basis_norm_split_f32(const float *in, const float *M, const int32_t *blocks, long long pts, int na, int c,
                                 int out_spectral, float *out, const float *sums, int groups, long long pts_per_group,
                                 const float *gamma, const float *beta, float eps, float slope, epn_stream_t stream);
/* The backward reduction of the block's FIRST norm (InterSO3ConvBlock's norm + leaky_relu in front of IntraSO3Conv,
 * SPConvNets/utils/base_so3conv.py:196-204) taken from the kernel that produces its output gradient: the inverse basis change
 * of the spectral gradient writes dy (plain layout) AND, from its accumulators, per point the partial sums
 *   point_dstats[pt][c][2] = (sum_a d, sum_a d * xhat),   d = dy * leaky'(norm(x)),  xhat = (x - mean) * rstd
 * of the tensor x_cl the forward transform normalised (sums / groups / pts_per_group / gamma / beta / eps / slope as for
 * epn_so3_basis_norm_*).  epn_norm_bwd_finish reduces block partials part[g][block][c][2] to the dsums[g][c][2], dgamma[c],
 * dbeta[c] that epn_norm_act_bwd_apply_* takes (workspace: epn_stats_finish_workspace_bytes(groups, blocks_per_group, c));
 * together they replace epn_norm_act_bwd_reduce_* (one full read of x and dy) for that norm.  Tensors of 2 GiB or more:
 * EPN_EINVAL (keep the separate reduction). */
int epn_so3_basis_dstats_f32(const float *in, const float *M, const int32_t *blocks, long long pts, int na, int c, float *out,
                             const float *x_cl, const float *sums, int groups, long long pts_per_group, const float *gamma,
                             const float *beta, float eps, float slope, float *point_dstats, epn_stream_t stream);
int epn_so3_basis_dstats_split_f32(const float *in, const float *M, const int32_t *blocks, long long pts, int na, int c,
                                   float *out, const float *x_cl, const float *sums, int groups, long long pts_per_group,
                                   const float *gamma, const float *beta, float eps, float slope, float *point_dstats,
                                   epn_stream_t stream);
int epn_so3_basis_dstats_bf16(const void *in, const float *M, const int32_t *blocks, long long pts, int na, int c, void *out,
                              const void *x_cl, const float *sums, int groups, long long pts_per_group, const float *gamma,
                              const float *beta, float eps, float slope, float *point_dstats, epn_stream_t stream);
int epn_norm_bwd_finish(const float *partials, int groups, long long blocks_per_group, int c, const float *gamma,
                        float *dsums, float *dgamma, float *dbeta, void *workspace, size_t workspace_bytes,
                        epn_stream_t stream);
int epn_chan_stats_bf16(const void *x_cl, int groups, long long rows, int c, float *sums, void *workspace,
                        size_t workspace_bytes, epn_stream_t stream);
int epn_norm_act_fwd_bf16(const void *x_cl, int groups, long long rows, int c, const float *sums, const float *gamma,
                          const float *beta, const void *residual_cl, float eps, float slope, void *y_cl,
                          epn_stream_t stream);
int epn_norm_act_bwd_reduce_bf16(const void *x_cl, const void *dy_cl, int groups, long long rows, int c,
                                 const float *sums, const float *gamma, const float *beta, float eps, float slope,
                                 float *dsums, float *dgamma, float *dbeta, void *workspace, size_t workspace_bytes,
                                 epn_stream_t stream);
int epn_norm_act_bwd_apply_bf16(const void *x_cl, const void *dy_cl, int groups, long long rows, int c,
                                const float *sums, const float *dsums, const float *gamma, const float *beta, float eps,
                                float slope, void *dx_cl, epn_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------------
 * Deterministic (atomic-free) data gradient of the grouping -- the transpose of epn_inter_group_* without the fp32 atomic
 * scatter of epn_inter_ungroup_* (reference: torch autograd through gather + einsum, whose index_add is itself
 * non-deterministic on CUDA; SURVEY a17 asked for an atomic-free form):
 *   1. epn_inter_inverse_list: ball_idx i32[b][p2][nn] -> CSR per cloud, offsets i32[b][p1+1], entries i32[b][p2*nn]
 *      (entry = p*nn + n with ball_idx[b][p][n] == q, in increasing order) -- once per geometry;
 *   2. epn_inter_ungroup_det_*: per-slot contributions slab[b][p][n][a][c] = sum_k w * grad_grouped (plain stores;
 *      slab has the element type of grad_grouped, b*p2*nn*na*cin elements), then grad_feats_cl[b][q][a][c] = the sum
 *      over q's list in list order (fp32 accumulation; fully overwritten; f32 -> float, bf16 -> bf16 output).
 *      When slab_bytes leaves room for b*p2*nn more bytes behind the (256-byte rounded) slab and the output points
 *      divide into the scatter's workgroups, the contributions are pre-reduced: a workgroup of 8-16 spatially adjacent
 *      output points sums the slots of each distinct destination in LDS (ascending slot order) and stores ONE slab row
 *      per destination, marked in those bytes; step 2 adds only the marked rows (a third of the slab traffic).
 * Bitwise repeatable.  Requires cin % 16 == 0 and na >= 16 (the MFMA grouping kernels). */
int epn_inter_inverse_list(const int32_t *ball_idx, int b, int p1, int p2, int nn, int32_t *offsets, int32_t *entries,
                           epn_stream_t stream);
int epn_inter_ungroup_det_f32(const epn_inter_desc *d, const float *grad_grouped, float *grad_feats_cl,
                              const int32_t *offsets, const int32_t *entries, void *slab, size_t slab_bytes,
                              void *workspace, size_t workspace_bytes, epn_stream_t stream);
int epn_inter_ungroup_det_bf16(const epn_inter_desc *d, const void *grad_grouped, void *grad_feats_cl,
                               const int32_t *offsets, const int32_t *entries, void *slab, size_t slab_bytes,
                               void *workspace, size_t workspace_bytes, epn_stream_t stream);

/* Strided skip connection of SeparableSO3ConvBlock: replaces zptk.functional.batched_index_select(skip_feature, 2,
 * sample_idx) (SPConvNets/utils/base_so3conv.py:206-207; vgtk/vgtk/spconv/functional.py:361-369: torch.gather with a
 * broadcast index) on channels-last data, where it is a gather of whole [a][c] rows, and its autograd backward.
 *   src [b][p1][row_bytes], idx i32[b][p2] -> dst [b][p2][row_bytes]            (row_bytes % 16 == 0, any element type)
 *   epn_scatter_rows: grad_src [b][p1][row_bytes] = 0, then row idx[b][p] <- grad_dst[b][p] (indices distinct per cloud,
 *   as FPS indices are: plain stores, deterministic) */
int epn_gather_rows(const void *src, const int32_t *idx, void *dst, int b, int p1, int p2, long long row_bytes,
                    epn_stream_t stream);
int epn_scatter_rows(const void *grad_dst, const int32_t *idx, void *grad_src, int b, int p1, int p2, long long row_bytes,
                     epn_stream_t stream);
/* epn_scatter_rows that ACCUMULATES (atomic-free and deterministic: the first occurrence of an index sums all of its
 * rows in fp32, ascending row order; p2 <= 8192): repeated indices -- FPS repeats index 0 for clouds with fewer live
 * points than samples -- receive the sum of their rows, like the backward of torch.gather / batched_index_select
 * (vgtk/vgtk/spconv/functional.py:28-40).  Elements fp32 (bf16 = 0) or bf16 (bf16 = 1). */
int epn_scatter_rows_add(const void *grad_dst, const int32_t *idx, void *grad_src, int b, int p1, int p2,
                         long long row_bytes, int bf16, epn_stream_t stream);

/* 1x1 convolution of a SINGLE input channel (the skip branch of every model's first block, whose input is the
 * occupancy feature: nn.Conv2d(1, cout, 1), SPConvNets/utils/base_so3conv.py:186): y[row][c] = x[row] * w[c], rows =
 * b*p*a of the channels-last tensor, cout % 4 == 0; and its weight gradient grad_w[c] = sum_row x[row] * grad_y[row][c]
 * (zero-fills grad_w first; 256 % (cout/4) == 0). */
int epn_conv1x1_c1_f32(const float *x, const float *w, float *y, long long rows, int cout, epn_stream_t stream);
int epn_conv1x1_c1_bwd_weight_f32(const float *x, const float *grad_y, float *grad_w, long long rows, int cout,
                                  epn_stream_t stream);

/* Attention pooling over the anchors in the 3DMatch head: replaces `attn = F.softmax(attn, dim=3)` and
 * `x_out = (x.feats * attn).sum(-1, keepdim=True)` of InvOutBlockMVD.forward (SPConvNets/utils/base_so3conv.py:603-606)
 * and their autograd backward, on channels-last rows [row = (b, p)][a][c] (na <= 64):
 *   fwd: attn[row][a][c] = softmax over a of logits[row][a][c];  pooled[row][c] = sum_a feats[row][a][c] * attn[row][a][c]
 *   bwd: g = grad_pooled * feats (+ grad_attn);  grad_logits = attn * (g - sum_a attn g);  grad_feats = grad_pooled * attn
 *        grad_pooled or grad_attn_cl may be NULL (not both: EPN_ENULL), grad_feats_cl may be NULL (not wanted). */
int epn_anchor_softmax_pool_fwd_f32(const float *feats_cl, const float *logits_cl, float *attn_cl, float *pooled,
                                    long long rows, int na, int c, epn_stream_t stream);
int epn_anchor_softmax_pool_bwd_f32(const float *feats_cl, const float *attn_cl, const float *grad_pooled,
                                    const float *grad_attn_cl, float *grad_feats_cl, float *grad_logits_cl,
                                    long long rows, int na, int c, epn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* EPN_SO3CONV_H */
