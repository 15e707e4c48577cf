// Internal declarations shared by the convolution translation units of libepn_so3conv.so.
#pragma once
#include "epn_common.h"

#define EPN_KS_MAX 32   // kernel points per anchor are padded to 32 slots (reference uses 24)
#define EPN_KS_GENERIC_MAX 96   // the any-shape kernels take the reference's larger kernel-point sets (kernel_size 2 / 3: 30 / 66 points)
#define EPN_NN_MAX 128  // neighbours per output point supported by the fused kernels

namespace epn {

// workspace carve-up (float offsets) for one inter descriptor
struct InterWs {
    size_t rk_off, rk4_off, beta_off, order_off, big_off, total_floats;
};
static inline size_t rnd64(size_t x) { return (x + 63) & ~(size_t)63; }
static inline bool inter_mfma_shape_ok(const epn_inter_desc *d) {
    return d->cin % 16 == 0 && d->cout % 16 == 0 && d->cout <= 256 && d->ks <= EPN_KS_MAX && d->ks % 4 == 0 &&
           d->nn <= EPN_NN_MAX && d->cin >= 16 && d->cout >= 16;
}
bool inter_mfma_available();  // false while only the stand-in TU is linked
static inline bool inter_uses_mfma(const epn_inter_desc *d) { return inter_mfma_available() && inter_mfma_shape_ok(d); }
static inline InterWs inter_ws(const epn_inter_desc *d) {
    InterWs w;
    w.rk_off = 0;
    w.rk4_off = rnd64((size_t)d->na * d->ks * 3);
    w.beta_off = w.rk4_off + rnd64((size_t)d->na * EPN_KS_MAX * 4);
    w.order_off = w.beta_off + rnd64((size_t)d->na * EPN_KS_MAX);   // b*p2 int32: spatial order of the output points
    w.big_off = w.order_off + rnd64((size_t)d->b * d->p2);
    // generic path: materialised grouped features; MFMA path: transposed weight for bwd_data
    const size_t big = inter_uses_mfma(d) ? (size_t)d->cout * d->cin * d->ks
                                          : (size_t)d->b * d->p2 * d->na * d->cin * d->ks;
    w.total_floats = w.big_off + rnd64(big);
    return w;
}

// conv_generic.hip
int launch_rk_table(const epn_inter_desc *d, float *rk, hipStream_t st);
int launch_inter_weights(const epn_inter_desc *d, float *w, hipStream_t st);
int launch_inter_group(const epn_inter_desc *d, const float *rk, const float *feats, float *G, hipStream_t st);
int launch_inter_scatter(const epn_inter_desc *d, const float *rk, const float *dG, float *dF, hipStream_t st);
int launch_rowgemm_nt(const float *X, const float *W, size_t ncol, int ck, int cout, float *out, hipStream_t st);
int launch_rowgemm_nn(const float *dOut, const float *W, size_t ncol, int ck, int cout, float *dX, hipStream_t st);
int launch_colreduce_dw(const float *dOut, const float *X, size_t ncol, int ck, int cout, float *dW, hipStream_t st);
int launch_intra_fwd_generic(const float *feats, const int32_t *iidx, const float *W, size_t npts, int na, int kn,
                             int cin, int cout, float *out, hipStream_t st);
int launch_intra_bwd_data_generic(const float *dOut, const int32_t *iidx, const float *W, size_t npts, int na, int kn,
                                  int cin, int cout, float *dF, hipStream_t st);
int launch_intra_bwd_weight_generic(const float *feats, const float *dOut, const int32_t *iidx, size_t npts, int na,
                                    int kn, int cin, int cout, float *dW, hipStream_t st);

// inter_c1.hip (single input channel: the first layer of every shipped model)
bool inter_c1_fwd_ok(const epn_inter_desc *d);
bool inter_c1_bwd_weight_ok(const epn_inter_desc *d);
int launch_inter_c1_fwd(const epn_inter_desc *d, const float *rk, const float *feats, const float *W, float *out,
                        hipStream_t st, float *grouped_save = nullptr, unsigned *flag = nullptr);   // flag: 4 B of workspace
int launch_inter_c1_bwd_weight(const epn_inter_desc *d, const float *rk, const float *feats, const float *dOut,
                               float *dW, hipStream_t st, const float *grouped_saved = nullptr);

// inter_mfma.hip / intra_mfma.hip (fused MFMA kernels; cin, cout multiples of 16)
int launch_inter_tables_mfma(const epn_inter_desc *d, const float *rk, float *rk4, float *beta, hipStream_t st);
int launch_inter_fwd_mfma(const epn_inter_desc *d, const float *rk4, const float *beta, const float *feats,
                          const float *W, float *out, hipStream_t st);
int launch_inter_bwd_data_mfma(const epn_inter_desc *d, const float *rk4, const float *wt_scratch, const float *dOut,
                               const float *W, float *dF, hipStream_t st);
int launch_inter_bwd_weight_mfma(const epn_inter_desc *d, const float *rk4, const float *beta, const float *feats,
                                 const float *dOut, float *dW, hipStream_t st);
// grouping only ("split" path: the weight contraction is left to the BLAS library); any cout
static inline bool inter_group_mfma_ok(const epn_inter_desc *d) {
    return inter_mfma_available() && !d->dense_w && d->cin % 16 == 0 && d->ks <= EPN_KS_MAX && d->ks % 4 == 0 &&
           d->nn <= EPN_NN_MAX && (long long)d->p1 * d->na * d->cin < (1LL << 31);
}
// bf16 != 0: feats / G (group) and dG (ungroup) are bf16; the scatter target dF is fp32 either way
// Packed column order of the grouped features (epn_inter_group_packed_*).  The plain order G[col][c*ks + k] makes a wave's
// store instruction -- 16 lanes of one kernel-point quad = 16 channels -- 16 separate 64-byte (k < 16) or 32-byte pieces,
// ks*4 bytes apart (4*ks*4 in the wide kernel, whose lanes own every CG-th channel).  Packed: all k < 16 columns first, in
// the order the wide kernel's lanes hold the channels (group g of 16 CG channels, then the lane's channel e, then lane x),
// then the k >= 16 columns the same way: every store instruction covers ONE contiguous 1 KiB / 512 B range of the row.
// The weight contraction permutes W's columns instead (epn_inter_pack_weights_*): a GEMM does not care.
__host__ __device__ inline int inter_packed_cg(int cin) { return cin % 64 == 0 ? 4 : 2; }
__host__ __device__ inline int inter_packed_slot(int c, int cg) {
    const int gw = 16 * cg, cl = c % gw;
    return c - cl + 16 * (cl % cg) + cl / cg;
}
__host__ __device__ inline int inter_packed_position(int c, int k, int cin, int ks) {
    const int slot = inter_packed_slot(c, inter_packed_cg(cin));
    const int w0 = ks < 16 ? ks : 16;
    return k < 16 ? slot * w0 + k : w0 * cin + slot * (ks - 16) + (k - 16);
}
// packed != 0: G in the packed column order (needs inter_group_packed_ok)
static inline bool inter_group_packed_ok(const epn_inter_desc *d) {
    return inter_group_mfma_ok(d) && d->na >= 16 && d->cin % 32 == 0 && d->nn <= 64;
}
int launch_inter_group_mfma(const epn_inter_desc *d, const float *rk4, const void *feats, void *G, int bf16,
                            hipStream_t st, int packed = 0);
// W[cout][cin*ks] (fp32) -> the same matrix with its columns in the packed order (fp32 or bf16), and the inverse for a
// weight gradient computed against packed grouped features
int launch_inter_pack_weights(const float *W, int cout, int cin, int ks, void *Wp, int bf16, hipStream_t st);
int launch_inter_unpack_weight_grad(const float *gWp, int cout, int cin, int ks, float *gW, hipStream_t st);
// order: b*p2 int32 of scratch for the Morton order of the output points (nullptr: per-slot atomic scatter)
int launch_inter_ungroup_mfma(const epn_inter_desc *d, const float *rk4, const void *dG, float *dF, int32_t *order,
                              int bf16, hipStream_t st);
int launch_morton_order(const float *new_xyz, int b, int p2, int32_t *order, hipStream_t st);
// inter_ungroup_cloud.hip: the transpose of the grouping with a cloud's gradient rows resident in LDS (64-bit fixed-point
// accumulators, no global atomics).  extra = workspace behind the rotated-kernel table, inter_ungroup_cloud_extra_bytes; dg_amax:
// device scalar max|dG|; add: optional tensor of dF's shape and type added to the result (may be dF itself); bf16: dG is bf16;
// out_bf16: dF (and add) are bf16
bool inter_ungroup_cloud_ok(const epn_inter_desc *d);
size_t inter_ungroup_cloud_extra_bytes(const epn_inter_desc *d);
int launch_inter_ungroup_cloud(const epn_inter_desc *d, const float *rk4, const void *dG, const float *dg_amax, void *dF,
                               const void *add, int bf16, int out_bf16, void *extra, hipStream_t st);
long long ungroup_cloud_range_take(bool reset);
// inter_bwd_f2.hip: the data gradient of InterSO3Conv with dG kept on chip (two-piece fp16 contraction inside the LDS-reduced
// scatter's workgroup).  rk4p: na*32*4 floats of scratch (a kernel-point order of its own), order: b*p2 int32, extra:
// inter_bwd_f2_extra_bytes; go_amax: device scalar max|dOut|.  dF is accumulated into (zero it for a plain gradient).
bool inter_bwd_f2_ok(const epn_inter_desc *d);
size_t inter_bwd_f2_extra_bytes(const epn_inter_desc *d);
int launch_inter_bwd_f2(const epn_inter_desc *d, float *rk4p, int32_t *order, const float *dOut, const float *W,
                        const float *go_amax, float *dF, void *extra, hipStream_t st);
// deterministic (atomic-free) data gradient of the grouping: inverse neighbour list + per-slot slab + ordered reduction
int launch_inverse_list(const int32_t *idx, int b, int p1, int p2, int nn, int32_t *off, int32_t *ent, hipStream_t st);
int launch_inter_ungroup_det_mfma(const epn_inter_desc *d, const float *rk4, const void *dG, void *dF, void *slab,
                                  const int32_t *off, const int32_t *ent, int bf16, hipStream_t st,
                                  int32_t *order = nullptr, unsigned char *canon = nullptr);
// inter_fx.hip: InterSO3Conv with the grouped features kept on chip (grouping = A-tile producer of the weight contraction);
// fp32 features: lossless 3 x bf16 split on the bf16 MFMAs; bf16 features: bf16 MFMAs.  planes: inter_fx_planes_bytes bytes
bool inter_fx_ok(const epn_inter_desc *d, int bf16);
size_t inter_fx_planes_bytes(const epn_inter_desc *d, int bf16);
int launch_inter_fx_fwd(const epn_inter_desc *d, const float *rk4, const void *feats, const float *W, void *out,
                        void *planes, int bf16, hipStream_t st);
bool intra_uses_mfma(int na, int kn, int cin, int cout);
size_t intra_workspace_floats(int kn, int cin, int cout);
int launch_intra_fwd_mfma(const float *feats, const int32_t *iidx, const float *W, int b, int p, int na, int kn,
                          int cin, int cout, float *out, float *ws, hipStream_t st);
int launch_intra_bwd_data_mfma(const float *dOut, const int32_t *inv_idx, const float *W, int b, int p, int na,
                               int kn, int cin, int cout, float *dF, float *ws, hipStream_t st);
int launch_intra_bwd_weight_mfma(const float *feats, const float *dOut, const int32_t *iidx, int b, int p, int na,
                                 int kn, int cin, int cout, float *dW, hipStream_t st);

}  // namespace epn
