// extern "C" entry points of libepn_so3conv.so for the convolution path (see include/epn_so3conv.h).
// Dispatch: fused MFMA kernels when cin and cout are multiples of 16, generic kernels otherwise
// (epn_set_kernel_policy(1) forces the generic path; used by the cross-check tests).
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <cxxabi.h>

#include "conv_internal.h"

using namespace epn;

static std::atomic<int> g_policy{0};   // the library's only process-wide state: 0 = best kernel, 1 = generic kernels,
                                       // 2 = the first layer's VALU kernel instead of its matrix-pipe form (cross-check)
static bool force_generic() { return g_policy.load(std::memory_order_relaxed) == 1; }

namespace epn {
bool first_layer_on_valu() { return g_policy.load(std::memory_order_relaxed) == 2; }
#ifdef EPN_TUNING
int kernel_policy() { return g_policy.load(std::memory_order_relaxed); }
#else
int kernel_policy() { return 0; }    // tile / scatter overrides exist only in -DEPN_TUNING builds
#endif
}

// ---- which kernel did the last call launch?  (thread-local; read and cleared by epn_last_kernel)
namespace {
thread_local const void *t_stub = nullptr;
thread_local bool t_stub_aux = true;
}  // namespace
namespace epn {
void note_kernel(const void *host_stub, bool aux) {
    if (aux && t_stub) return;                    // a helper never replaces what is already recorded
    t_stub = host_stub;                           // the last main kernel of a call wins
    t_stub_aux = aux;
}
}  // namespace epn

extern "C" const char *epn_last_kernel(void) {
    static thread_local char out[320];
    out[0] = 0;
    const void *stub = t_stub;
    t_stub = nullptr;
    t_stub_aux = true;
    if (!stub) return out;
    const char *mangled = hipKernelNameRefByPtr(stub, nullptr);
    if (!mangled) return out;
    int status = 0;
    char *dem = abi::__cxa_demangle(mangled, nullptr, nullptr, &status);
    const char *b = (status == 0 && dem) ? dem : mangled;
    if (!std::strncmp(b, "void ", 5)) b += 5;
    // drop "(anonymous namespace)::" and the argument list (the first '(' outside template brackets)
    static const char anon[] = "(anonymous namespace)::";
    size_t o = 0;
    int depth = 0;
    for (size_t i = 0; b[i] && o + 1 < sizeof(out);) {
        if (!std::strncmp(b + i, anon, sizeof(anon) - 1)) { i += sizeof(anon) - 1; continue; }
        if (b[i] == '<') ++depth;
        else if (b[i] == '>') --depth;
        else if (b[i] == '(' && depth == 0) break;
        out[o++] = b[i++];
    }
    out[o] = 0;
    std::free(dem);
    return out;
}

extern "C" int epn_set_kernel_policy(int policy) {
#ifdef EPN_TUNING   // tools/ builds (python -m epn_pointcloud_amd.build --tuning): 0x100 | cfg .. 0x400 | cfg = A/B switches
    if (policy != 0 && policy != 1 && policy != 2 && (policy & ~0xff) != 0x100 && (policy & ~0xff) != 0x200 && (policy & ~0xff) != 0x400 && (policy & ~0xff) != 0x800 && (policy & ~0xff) != 0x900) return EPN_EINVAL;
#else
    if (policy != 0 && policy != 1 && policy != 2) return EPN_EINVAL;
#endif
    g_policy.store(policy, std::memory_order_relaxed);
    return 0;
}

static int check_desc(const epn_inter_desc *d) {
    if (!d) return EPN_ENULL;
    if (d->b < 0 || d->p1 < 1 || d->p2 < 0 || d->nn < 1 || d->na < 1 || d->ks < 1 || d->cin < 1 || d->cout < 1)
        return EPN_EINVAL;
    if (d->ks > EPN_KS_GENERIC_MAX) return EPN_EINVAL;   // fused kernels: ks <= EPN_KS_MAX (inter_mfma_shape_ok); generic beyond
    if (!(d->sigma > 0.f) && !d->dense_w) return EPN_EINVAL;
    if (!d->ball_idx) return EPN_ENULL;
    if (!d->dense_w && (!d->xyz || !d->new_xyz || !d->anchors || !d->kernels)) return EPN_ENULL;
    return 0;
}

static bool use_mfma(const epn_inter_desc *d) { return inter_uses_mfma(d) && !force_generic() && !d->dense_w; }

// 0.2: epn_gemm_nt_problem gained the trailing `col_stats` member (round 3) and epn_ball_query_f64 takes `float radius`
// (round 4) -- callers compiled against the 0.1 header must be rebuilt; INTEGRATION.md "ABI revisions"
// 0.3 (round 5): epn_abi_version() added; new entry point epn_fps_temp_f32; nothing existing changed
// 0.4 (round 6): epn_f16x2_overflow_count (the two-piece kernels' overflow sentinel); the tail of the composed split form's
// `saved` buffer carries a tag word and an untagged tail is re-derived, not trusted; no signature changed.  (0.3 DID change
// behaviour without changing a signature -- fp32 `saved` grew by 256 bytes and the composed entries moved to the two-piece
// GEMMs: INTEGRATION.md "ABI revisions" says so; the note above was too short.)
// 0.5 (round 6, second half; EPN_ABI_VERSION 3): epn_gemm_nt_problem gained the trailing `c_amax` member (max|C| from the
// kernels' epilogue) -- callers that build the struct must be recompiled; new entry points epn_inter_ungroup_cloud_* (the transpose
// of the grouping with a cloud's gradient rows resident in LDS); the composed bf16 split backward runs on it
extern "C" const char *epn_version(void) { return "epn_so3conv 0.5 (gfx950)"; }
extern "C" int epn_abi_version(void) { return EPN_ABI_VERSION; }

extern "C" const char *epn_strerror(int code) {
    switch (code) {
        case 0: return "success";
        case EPN_EINVAL: return "epn: invalid size or unsupported shape";
        case EPN_EWORKSPACE: return "epn: workspace missing or too small";
        case EPN_ENULL: return "epn: required pointer is NULL";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "epn: unknown error";
    }
}

extern "C" int epn_inter_is_fused(const epn_inter_desc *d) {
    return d && (use_mfma(d) || (inter_c1_fwd_ok(d) && !force_generic())) ? 1 : 0;
}

extern "C" int epn_intra_is_fused(int na, int kn, int cin, int cout) {
    return intra_uses_mfma(na, kn, cin, cout) && !force_generic() ? 1 : 0;
}

extern "C" size_t epn_inter_workspace_bytes(const epn_inter_desc *d) {
    if (!d) return 0;
    InterWs w = inter_ws(d);
    // the generic path may be forced at run time, so always report the larger requirement when asked to
    if (force_generic() || d->dense_w) {
        epn_inter_desc g = *d;
        g.cin = d->cin; g.cout = d->cout;
        const size_t big = (size_t)d->b * d->p2 * d->na * d->cin * d->ks;
        return (w.big_off + rnd64(big)) * sizeof(float);
    }
    return w.total_floats * sizeof(float);
}

// need_rk: the [na][ks][3] rotated-kernel table of the generic / cin = 1 kernels; the MFMA kernels read the [na][32][4] table
// only, which launch_inter_tables_mfma derives from the anchors itself (one table launch per call instead of two)
static int prep(const epn_inter_desc *d, void *workspace, size_t bytes, bool need_big, InterWs &ws, float *&base,
                hipStream_t st, bool need_rk = true) {
    int rc = check_desc(d);
    if (rc) return rc;
    ws = inter_ws(d);
    const size_t big = need_big ? rnd64((size_t)d->b * d->p2 * d->na * d->cin * d->ks)
                                : ws.total_floats - ws.big_off;
    if (!workspace || bytes < (ws.big_off + big) * sizeof(float)) return EPN_EWORKSPACE;
    base = static_cast<float *>(workspace);
    if (!d->dense_w && need_rk) {
        rc = launch_rk_table(d, base + ws.rk_off, st);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int epn_inter_weights_f32(const epn_inter_desc *d, float *w, epn_stream_t stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!w || !d->xyz || !d->new_xyz || !d->anchors || !d->kernels) return EPN_ENULL;
    if (d->b == 0 || d->p2 == 0) return 0;
    return launch_inter_weights(d, w, epn_stream(stream));
}

extern "C" int epn_inter_so3conv_fwd_f32(const epn_inter_desc *d, const float *feats_cl, const float *W,
                                         float *out_cl, void *workspace, size_t workspace_bytes,
                                         epn_stream_t stream) {
    hipStream_t st = epn_stream(stream);
    InterWs ws;
    float *base = nullptr;
    const bool mf = d && use_mfma(d);
    int rc = prep(d, workspace, workspace_bytes, !mf, ws, base, st, !mf);
    if (rc) return rc;
    if (!feats_cl || !W || !out_cl) return EPN_ENULL;
    if (d->b == 0 || d->p2 == 0) return 0;
    if (mf) {
        rc = launch_inter_tables_mfma(d, base + ws.rk_off, base + ws.rk4_off, base + ws.beta_off, st);
        if (rc) return rc;
        return launch_inter_fwd_mfma(d, base + ws.rk4_off, base + ws.beta_off, feats_cl, W, out_cl, st);
    }
    if (inter_c1_fwd_ok(d) && !force_generic())
        return launch_inter_c1_fwd(d, base + ws.rk_off, feats_cl, W, out_cl, st, nullptr,
                                   reinterpret_cast<unsigned *>(base + ws.beta_off));   // (beta table: unused by cin = 1)
    float *G = base + ws.big_off;
    rc = launch_inter_group(d, base + ws.rk_off, feats_cl, G, st);
    if (rc) return rc;
    return launch_rowgemm_nt(G, W, (size_t)d->b * d->p2 * d->na, d->cin * d->ks, d->cout, out_cl, st);
}

// ---- cin = 1 (first layer of every model) with the grouped values kept for the weight gradient
extern "C" int epn_inter_c1_ok(const epn_inter_desc *d) {
    return d && !check_desc(d) && inter_c1_fwd_ok(d) && inter_c1_bwd_weight_ok(d) && !force_generic() ? 1 : 0;
}
extern "C" int epn_inter_so3conv_fwd_c1_f32(const epn_inter_desc *d, const float *feats_cl, const float *W, float *out_cl,
                                            float *grouped, void *workspace, size_t workspace_bytes, epn_stream_t stream) {
    if (!epn_inter_c1_ok(d)) return EPN_EINVAL;
    hipStream_t st = epn_stream(stream);
    InterWs ws;
    float *base = nullptr;
    int rc = prep(d, workspace, workspace_bytes, false, ws, base, st);
    if (rc) return rc;
    if (!feats_cl || !W || !out_cl) return EPN_ENULL;
    if (d->b == 0 || d->p2 == 0) return 0;
    return launch_inter_c1_fwd(d, base + ws.rk_off, feats_cl, W, out_cl, st, grouped,
                               reinterpret_cast<unsigned *>(base + ws.beta_off));       // (beta table: unused by cin = 1)
}
extern "C" int epn_inter_so3conv_bwd_weight_c1_f32(const epn_inter_desc *d, const float *grouped, const float *grad_out_cl,
                                                   float *grad_W, epn_stream_t stream) {
    if (!epn_inter_c1_ok(d)) return EPN_EINVAL;
    if (!grad_W) return EPN_ENULL;
    hipStream_t st = epn_stream(stream);
    EPN_HIP(hipMemsetAsync(grad_W, 0, sizeof(float) * (size_t)d->cout * d->ks, st));
    if (d->b == 0 || d->p2 == 0) return 0;
    if (!grouped || !grad_out_cl) return EPN_ENULL;
    return launch_inter_c1_bwd_weight(d, nullptr, nullptr, grad_out_cl, grad_W, st, grouped);
}

extern "C" int epn_inter_so3conv_bwd_data_f32(const epn_inter_desc *d, const float *grad_out_cl, const float *W,
                                              float *grad_feats_cl, void *workspace, size_t workspace_bytes,
                                              epn_stream_t stream) {
    hipStream_t st = epn_stream(stream);
    InterWs ws;
    float *base = nullptr;
    const bool mf = d && use_mfma(d);
    int rc = prep(d, workspace, workspace_bytes, !mf, ws, base, st, !mf);
    if (rc) return rc;
    if (!grad_feats_cl) return EPN_ENULL;
    EPN_HIP(hipMemsetAsync(grad_feats_cl, 0, sizeof(float) * (size_t)d->b * d->p1 * d->na * d->cin, st));
    if (d->b == 0 || d->p2 == 0) return 0;
    if (!grad_out_cl || !W) return EPN_ENULL;
    if (mf) {
        rc = launch_inter_tables_mfma(d, base + ws.rk_off, base + ws.rk4_off, base + ws.beta_off, st);
        if (rc) return rc;
        return launch_inter_bwd_data_mfma(d, base + ws.rk4_off, base + ws.big_off, grad_out_cl, W, grad_feats_cl, st);
    }
    float *dG = base + ws.big_off;
    rc = launch_rowgemm_nn(grad_out_cl, W, (size_t)d->b * d->p2 * d->na, d->cin * d->ks, d->cout, dG, st);
    if (rc) return rc;
    return launch_inter_scatter(d, base + ws.rk_off, dG, grad_feats_cl, st);
}

extern "C" int epn_inter_so3conv_bwd_weight_f32(const epn_inter_desc *d, const float *feats_cl,
                                                const float *grad_out_cl, float *grad_W, void *workspace,
                                                size_t workspace_bytes, epn_stream_t stream) {
    hipStream_t st = epn_stream(stream);
    InterWs ws;
    float *base = nullptr;
    const bool mf = d && use_mfma(d);
    int rc = prep(d, workspace, workspace_bytes, !mf, ws, base, st, !mf);
    if (rc) return rc;
    if (!grad_W) return EPN_ENULL;
    EPN_HIP(hipMemsetAsync(grad_W, 0, sizeof(float) * (size_t)d->cout * d->cin * d->ks, st));
    if (d->b == 0 || d->p2 == 0) return 0;
    if (!feats_cl || !grad_out_cl) return EPN_ENULL;
    if (mf) {
        rc = launch_inter_tables_mfma(d, base + ws.rk_off, base + ws.rk4_off, base + ws.beta_off, st);
        if (rc) return rc;
        return launch_inter_bwd_weight_mfma(d, base + ws.rk4_off, base + ws.beta_off, feats_cl, grad_out_cl, grad_W,
                                            st);
    }
    if (inter_c1_bwd_weight_ok(d) && !force_generic())
        return launch_inter_c1_bwd_weight(d, base + ws.rk_off, feats_cl, grad_out_cl, grad_W, st);
    float *G = base + ws.big_off;
    rc = launch_inter_group(d, base + ws.rk_off, feats_cl, G, st);
    if (rc) return rc;
    return launch_colreduce_dw(grad_out_cl, G, (size_t)d->b * d->p2 * d->na, d->cin * d->ks, d->cout, grad_W, st);
}

// ---- grouping only: G[col][c*ks + k] and its transpose; the caller runs the weight contraction as a plain GEMM
extern "C" size_t epn_inter_group_workspace_bytes(const epn_inter_desc *d) {
    if (!d) return 0;
    InterWs w = inter_ws(d);
    return w.big_off * sizeof(float);   // the rotated-kernel tables only
}

static int prep_tables(const epn_inter_desc *d, void *workspace, size_t bytes, InterWs &ws, float *&base,
                       hipStream_t st, bool need_rk = true) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (d->dense_w) return EPN_EINVAL;
    ws = inter_ws(d);
    if (!workspace || bytes < ws.big_off * sizeof(float)) return EPN_EWORKSPACE;
    base = static_cast<float *>(workspace);
    return need_rk ? launch_rk_table(d, base + ws.rk_off, st) : 0;
}

// ---- on-chip form: grouping fused into the weight contraction as its A-tile producer (inter_fx.hip)
extern "C" int epn_inter_onchip_ok(const epn_inter_desc *d, int bf16) {
    return d && !check_desc(d) && inter_fx_ok(d, bf16) && !force_generic() ? 1 : 0;
}

extern "C" size_t epn_inter_onchip_workspace_bytes(const epn_inter_desc *d, int bf16) {
    if (!d) return 0;
    InterWs w = inter_ws(d);
    return w.big_off * sizeof(float) + inter_fx_planes_bytes(d, bf16);   // tables + the permuted bf16 planes of W
}

static int inter_onchip_fwd(const epn_inter_desc *d, const void *feats_cl, const float *W, void *out_cl, void *workspace,
                            size_t workspace_bytes, int bf16, epn_stream_t stream) {
    hipStream_t st = epn_stream(stream);
    int rc = check_desc(d);
    if (rc) return rc;
    if (!inter_fx_ok(d, bf16)) return EPN_EINVAL;
    if (workspace_bytes < epn_inter_onchip_workspace_bytes(d, bf16)) return EPN_EWORKSPACE;
    InterWs ws;
    float *base = nullptr;
    rc = prep_tables(d, workspace, workspace_bytes, ws, base, st, false);
    if (rc) return rc;
    if (d->b == 0 || d->p2 == 0) return 0;
    if (!feats_cl || !W || !out_cl) return EPN_ENULL;
    rc = launch_inter_tables_mfma(d, base + ws.rk_off, base + ws.rk4_off, base + ws.beta_off, st);
    if (rc) return rc;
    return launch_inter_fx_fwd(d, base + ws.rk4_off, feats_cl, W, out_cl, base + ws.big_off, bf16, st);
}

extern "C" int epn_inter_so3conv_fwd_onchip_f32(const epn_inter_desc *d, const float *feats_cl, const float *W,
                                                float *out_cl, void *workspace, size_t workspace_bytes,
                                                epn_stream_t stream) {
    return inter_onchip_fwd(d, feats_cl, W, out_cl, workspace, workspace_bytes, 0, stream);
}
extern "C" int epn_inter_so3conv_fwd_bf16(const epn_inter_desc *d, const void *feats_cl, const float *W, void *out_cl,
                                          void *workspace, size_t workspace_bytes, epn_stream_t stream) {
    return inter_onchip_fwd(d, feats_cl, W, out_cl, workspace, workspace_bytes, 1, stream);
}

static int inter_group_any(const epn_inter_desc *d, const void *feats_cl, void *grouped, void *workspace,
                           size_t workspace_bytes, int bf16, epn_stream_t stream, int packed = 0) {
    hipStream_t st = epn_stream(stream);
    InterWs ws;
    float *base = nullptr;
    int rc = prep_tables(d, workspace, workspace_bytes, ws, base, st, !(d && inter_group_mfma_ok(d) && !force_generic()));
    if (rc) return rc;
    if (d->b == 0 || d->p2 == 0) return 0;
    if (!feats_cl || !grouped) return EPN_ENULL;
    if (inter_group_mfma_ok(d) && !force_generic()) {
        rc = launch_inter_tables_mfma(d, base + ws.rk_off, base + ws.rk4_off, base + ws.beta_off, st);
        if (rc) return rc;
        return launch_inter_group_mfma(d, base + ws.rk4_off, feats_cl, grouped, bf16, st, packed);
    }
    if (bf16 || packed) return EPN_EINVAL;   // the bf16 feature path needs cin % 16 == 0 (every layer but the first, which is c1)
    return launch_inter_group(d, base + ws.rk_off, static_cast<const float *>(feats_cl), static_cast<float *>(grouped), st);
}

static int inter_ungroup_any(const epn_inter_desc *d, const void *grad_grouped, float *grad_feats_cl, void *workspace,
                             size_t workspace_bytes, int bf16, epn_stream_t stream, bool accumulate = false) {
    hipStream_t st = epn_stream(stream);
    InterWs ws;
    float *base = nullptr;
    int rc = prep_tables(d, workspace, workspace_bytes, ws, base, st, !(d && inter_group_mfma_ok(d) && !force_generic()));
    if (rc) return rc;
    if (!grad_feats_cl) return EPN_ENULL;
    if (!accumulate)
        EPN_HIP(hipMemsetAsync(grad_feats_cl, 0, sizeof(float) * (size_t)d->b * d->p1 * d->na * d->cin, st));
    if (d->b == 0 || d->p2 == 0) return 0;
    if (!grad_grouped) return EPN_ENULL;
    if (inter_group_mfma_ok(d) && !force_generic()) {
        rc = launch_inter_tables_mfma(d, base + ws.rk_off, base + ws.rk4_off, base + ws.beta_off, st);
        if (rc) return rc;
        return launch_inter_ungroup_mfma(d, base + ws.rk4_off, grad_grouped, grad_feats_cl,
                                         reinterpret_cast<int32_t *>(base + ws.order_off), bf16, st);
    }
    if (bf16) return EPN_EINVAL;
    return launch_inter_scatter(d, base + ws.rk_off, static_cast<const float *>(grad_grouped), grad_feats_cl, st);
}

extern "C" int epn_inter_group_f32(const epn_inter_desc *d, const float *feats_cl, float *grouped, void *workspace,
                                   size_t workspace_bytes, epn_stream_t stream) {
    return inter_group_any(d, feats_cl, grouped, workspace, workspace_bytes, 0, stream);
}
extern "C" int epn_inter_group_bf16(const epn_inter_desc *d, const void *feats_cl, void *grouped, void *workspace,
                                    size_t workspace_bytes, epn_stream_t stream) {
    return inter_group_any(d, feats_cl, grouped, workspace, workspace_bytes, 1, stream);
}
// ---- packed column order of the grouped features (contiguous stores in the grouping kernel); W's columns follow
extern "C" int epn_inter_group_packed_ok(const epn_inter_desc *d) {
    return d && !check_desc(d) && inter_group_packed_ok(d) && !force_generic() ? 1 : 0;
}
extern "C" int epn_inter_packed_position(int cin, int ks, int32_t *position) {
    if (cin < 32 || cin % 32 || ks < 4 || ks > EPN_KS_MAX || ks % 4) return EPN_EINVAL;
    if (!position) return EPN_ENULL;
    for (int c = 0; c < cin; ++c)
        for (int k = 0; k < ks; ++k) position[c * ks + k] = inter_packed_position(c, k, cin, ks);
    return 0;
}
extern "C" int epn_inter_group_packed_f32(const epn_inter_desc *d, const float *feats_cl, float *grouped, void *workspace,
                                          size_t workspace_bytes, epn_stream_t stream) {
    if (!epn_inter_group_packed_ok(d)) return EPN_EINVAL;
    return inter_group_any(d, feats_cl, grouped, workspace, workspace_bytes, 0, stream, 1);
}
extern "C" int epn_inter_group_packed_bf16(const epn_inter_desc *d, const void *feats_cl, void *grouped, void *workspace,
                                           size_t workspace_bytes, epn_stream_t stream) {
    if (!epn_inter_group_packed_ok(d)) return EPN_EINVAL;
    return inter_group_any(d, feats_cl, grouped, workspace, workspace_bytes, 1, stream, 1);
}
static int pack_args_ok(int cout, int cin, int ks) {
    return cout >= 0 && cin >= 32 && cin % 32 == 0 && ks >= 4 && ks <= EPN_KS_MAX && ks % 4 == 0;
}
extern "C" int epn_inter_pack_weights_f32(const float *W, int cout, int cin, int ks, float *packed, epn_stream_t stream) {
    if (!pack_args_ok(cout, cin, ks)) return EPN_EINVAL;
    if (cout && (!W || !packed)) return EPN_ENULL;
    return launch_inter_pack_weights(W, cout, cin, ks, packed, 0, epn_stream(stream));
}
extern "C" int epn_inter_pack_weights_bf16(const float *W, int cout, int cin, int ks, void *packed, epn_stream_t stream) {
    if (!pack_args_ok(cout, cin, ks)) return EPN_EINVAL;
    if (cout && (!W || !packed)) return EPN_ENULL;
    return launch_inter_pack_weights(W, cout, cin, ks, packed, 1, epn_stream(stream));
}
extern "C" int epn_inter_unpack_weight_grad_f32(const float *grad_packed, int cout, int cin, int ks, float *grad_W,
                                                epn_stream_t stream) {
    if (!pack_args_ok(cout, cin, ks)) return EPN_EINVAL;
    if (cout && (!grad_packed || !grad_W)) return EPN_ENULL;
    return launch_inter_unpack_weight_grad(grad_packed, cout, cin, ks, grad_W, epn_stream(stream));
}

extern "C" int epn_inter_ungroup_f32(const epn_inter_desc *d, const float *grad_grouped, float *grad_feats_cl,
                                     void *workspace, size_t workspace_bytes, epn_stream_t stream) {
    return inter_ungroup_any(d, grad_grouped, grad_feats_cl, workspace, workspace_bytes, 0, stream);
}
extern "C" int epn_inter_ungroup_bf16(const epn_inter_desc *d, const void *grad_grouped, float *grad_feats_cl,
                                      void *workspace, size_t workspace_bytes, epn_stream_t stream) {
    return inter_ungroup_any(d, grad_grouped, grad_feats_cl, workspace, workspace_bytes, 1, stream);
}

// the same, ADDED to what grad_feats_cl holds (a gradient that reached the same tensor by another path)
extern "C" int epn_inter_ungroup_acc_f32(const epn_inter_desc *d, const float *grad_grouped, float *grad_feats_cl,
                                         void *workspace, size_t workspace_bytes, epn_stream_t stream) {
    return inter_ungroup_any(d, grad_grouped, grad_feats_cl, workspace, workspace_bytes, 0, stream, true);
}
extern "C" int epn_inter_ungroup_acc_bf16(const epn_inter_desc *d, const void *grad_grouped, float *grad_feats_cl,
                                          void *workspace, size_t workspace_bytes, epn_stream_t stream) {
    return inter_ungroup_any(d, grad_grouped, grad_feats_cl, workspace, workspace_bytes, 1, stream, true);
}

// ---- transpose of the grouping with a cloud's gradient rows resident in LDS (inter_ungroup_cloud.hip)
extern "C" int epn_inter_ungroup_cloud_ok(const epn_inter_desc *d) {
    return d && !check_desc(d) && !d->dense_w && inter_ungroup_cloud_ok(d) && !force_generic() ? 1 : 0;
}
extern "C" size_t epn_inter_ungroup_cloud_workspace_bytes(const epn_inter_desc *d) {
    if (!d || check_desc(d) || !inter_ungroup_cloud_ok(d)) return 0;
    InterWs w = inter_ws(d);
    return w.big_off * sizeof(float) + inter_ungroup_cloud_extra_bytes(d);
}
static int inter_ungroup_cloud_any(const epn_inter_desc *d, const void *grad_grouped, const float *dg_amax, void *grad_feats_cl,
                                   const void *add, void *workspace, size_t workspace_bytes, int bf16, int out_bf16, epn_stream_t stream) {
    hipStream_t st = epn_stream(stream);
    int rc = check_desc(d);
    if (rc) return rc;
    if (d->dense_w || !inter_ungroup_cloud_ok(d) || force_generic()) return EPN_EINVAL;
    InterWs ws = inter_ws(d);
    if (!workspace || workspace_bytes < ws.big_off * sizeof(float) + inter_ungroup_cloud_extra_bytes(d)) return EPN_EWORKSPACE;
    if (!grad_feats_cl) return EPN_ENULL;
    if (d->b == 0) return 0;
    if (d->p2 == 0) {                      // no output points: the gradient is what `add` holds, or zero
        const size_t bytes = (size_t)d->b * d->p1 * d->na * d->cin * (out_bf16 ? 2 : 4);
        if (add && add != grad_feats_cl) EPN_HIP(hipMemcpyAsync(grad_feats_cl, add, bytes, hipMemcpyDeviceToDevice, st));
        else if (!add) EPN_HIP(hipMemsetAsync(grad_feats_cl, 0, bytes, st));
        return 0;
    }
    if (!grad_grouped) return EPN_ENULL;
    float *base = static_cast<float *>(workspace);
    rc = launch_inter_tables_mfma(d, base + ws.rk_off, base + ws.rk4_off, base + ws.beta_off, st);
    if (rc) return rc;
    return launch_inter_ungroup_cloud(d, base + ws.rk4_off, grad_grouped, dg_amax, grad_feats_cl, add, bf16, out_bf16, base + ws.big_off, st);
}
extern "C" int epn_inter_ungroup_cloud_f32(const epn_inter_desc *d, const float *grad_grouped, const float *dg_amax,
                                           float *grad_feats_cl, const float *add, void *workspace, size_t workspace_bytes,
                                           epn_stream_t stream) {
    return inter_ungroup_cloud_any(d, grad_grouped, dg_amax, grad_feats_cl, add, workspace, workspace_bytes, 0, 0, stream);
}
extern "C" int epn_inter_ungroup_cloud_bf16(const epn_inter_desc *d, const void *grad_grouped, const float *dg_amax,
                                            void *grad_feats_cl, const void *add, int out_f32, void *workspace,
                                            size_t workspace_bytes, epn_stream_t stream) {
    return inter_ungroup_cloud_any(d, grad_grouped, dg_amax, grad_feats_cl, add, workspace, workspace_bytes, 1, out_f32 ? 0 : 1, stream);
}
extern "C" long long epn_inter_ungroup_cloud_range_count(int reset) { return ungroup_cloud_range_take(reset != 0); }

// ---- data gradient with the grouped-feature gradient kept on chip (inter_bwd_f2.hip)
extern "C" int epn_inter_bwd_data_f16x2_ok(const epn_inter_desc *d) {
    return d && !check_desc(d) && !d->dense_w && inter_bwd_f2_ok(d) && !force_generic() ? 1 : 0;
}
extern "C" size_t epn_inter_bwd_data_f16x2_workspace_bytes(const epn_inter_desc *d) {
    if (!d || check_desc(d) || !inter_bwd_f2_ok(d)) return 0;
    InterWs w = inter_ws(d);
    return w.big_off * sizeof(float) + inter_bwd_f2_extra_bytes(d);
}
extern "C" int epn_inter_bwd_data_f16x2_f32(const epn_inter_desc *d, const float *grad_out_cl, const float *W, const float *go_amax,
                                            float *grad_feats_cl, int accumulate, void *workspace, size_t workspace_bytes,
                                            epn_stream_t stream) {
    hipStream_t st = epn_stream(stream);
    int rc = check_desc(d);
    if (rc) return rc;
    if (d->dense_w || !inter_bwd_f2_ok(d) || force_generic()) return EPN_EINVAL;
    InterWs ws = inter_ws(d);
    if (!workspace || workspace_bytes < ws.big_off * sizeof(float) + inter_bwd_f2_extra_bytes(d)) return EPN_EWORKSPACE;
    if (!grad_feats_cl) return EPN_ENULL;
    if (!accumulate)
        EPN_HIP(hipMemsetAsync(grad_feats_cl, 0, sizeof(float) * (size_t)d->b * d->p1 * d->na * d->cin, st));
    if (d->b == 0 || d->p2 == 0) return 0;
    if (!grad_out_cl || !W || !go_amax) return EPN_ENULL;
    float *base = static_cast<float *>(workspace);
    return launch_inter_bwd_f2(d, base + ws.rk4_off, reinterpret_cast<int32_t *>(base + ws.order_off), grad_out_cl, W, go_amax,
                               grad_feats_cl, base + ws.big_off, st);
}

extern "C" int epn_inter_inverse_list(const int32_t *ball_idx, int b, int p1, int p2, int nn, int32_t *offsets,
                                      int32_t *entries, epn_stream_t stream) {
    if (b < 0 || p1 < 1 || p2 < 0 || nn < 1) return EPN_EINVAL;
    if (b == 0) return 0;
    if (!ball_idx || !offsets || !entries) return EPN_ENULL;
    return launch_inverse_list(ball_idx, b, p1, p2, nn, offsets, entries, epn_stream(stream));
}

static int inter_ungroup_det_any(const epn_inter_desc *d, const void *grad_grouped, void *grad_feats_cl,
                                 const int32_t *offsets, const int32_t *entries, void *slab, size_t slab_bytes,
                                 void *workspace, size_t workspace_bytes, int bf16, epn_stream_t stream) {
    hipStream_t st = epn_stream(stream);
    InterWs ws;
    float *base = nullptr;
    int rc = prep_tables(d, workspace, workspace_bytes, ws, base, st, false);
    if (rc) return rc;
    if (!inter_group_mfma_ok(d) || d->na < 16 || (d->na * d->cin) % 4) return EPN_EINVAL;
    if (!grad_feats_cl) return EPN_ENULL;
    if (d->b == 0) return 0;
    if (!grad_grouped || !offsets || !entries || !slab) return EPN_ENULL;
    const size_t need = (size_t)d->b * d->p2 * d->nn * d->na * d->cin * (bf16 ? 2 : 4);
    if (slab_bytes < need) return EPN_EWORKSPACE;
    rc = launch_inter_tables_mfma(d, base + ws.rk_off, base + ws.rk4_off, base + ws.beta_off, st);
    if (rc) return rc;
    // room behind the slab for one byte per (point, neighbour slot): the pre-reduced form (a third of the slab traffic)
    const size_t flags = (size_t)d->b * d->p2 * d->nn;
    unsigned char *canon = slab_bytes >= ((need + 255) & ~(size_t)255) + flags ? static_cast<unsigned char *>(slab) + ((need + 255) & ~(size_t)255) : nullptr;
    return launch_inter_ungroup_det_mfma(d, base + ws.rk4_off, grad_grouped, grad_feats_cl, slab, offsets, entries, bf16, st,
                                         reinterpret_cast<int32_t *>(base + ws.order_off), canon);
}
extern "C" int epn_inter_ungroup_det_f32(const epn_inter_desc *d, const float *grad_grouped, float *grad_feats_cl,
                                         const int32_t *offsets, const int32_t *entries, void *slab, size_t slab_bytes,
                                         void *workspace, size_t workspace_bytes, epn_stream_t stream) {
    return inter_ungroup_det_any(d, grad_grouped, grad_feats_cl, offsets, entries, slab, slab_bytes, workspace,
                                 workspace_bytes, 0, stream);
}
extern "C" int epn_inter_ungroup_det_bf16(const epn_inter_desc *d, const void *grad_grouped, void *grad_feats_cl,
                                          const int32_t *offsets, const int32_t *entries, void *slab, size_t slab_bytes,
                                          void *workspace, size_t workspace_bytes, epn_stream_t stream) {
    return inter_ungroup_det_any(d, grad_grouped, grad_feats_cl, offsets, entries, slab, slab_bytes, workspace,
                                 workspace_bytes, 1, stream);
}

static int check_intra(int b, int p, int na, int kn, int cin, int cout) {
    if (b < 0 || p < 0 || na < 1 || kn < 1 || cin < 1 || cout < 1) return EPN_EINVAL;
    return 0;
}

extern "C" size_t epn_intra_workspace_bytes(int na, int kn, int cin, int cout) {
    (void)na;
    if (kn < 1 || cin < 1 || cout < 1) return 0;
    return intra_workspace_floats(kn, cin, cout) * sizeof(float);
}

extern "C" int epn_intra_so3conv_fwd_f32(const float *feats_cl, const int32_t *intra_idx, const float *W, int b,
                                         int p, int na, int kn, int cin, int cout, float *out_cl,
                                         void *workspace, size_t workspace_bytes, epn_stream_t stream) {
    int rc = check_intra(b, p, na, kn, cin, cout);
    if (rc) return rc;
    if (b == 0 || p == 0) return 0;
    if (!feats_cl || !intra_idx || !W || !out_cl) return EPN_ENULL;
    hipStream_t st = epn_stream(stream);
    if (intra_uses_mfma(na, kn, cin, cout) && !force_generic()) {
        if (!workspace || workspace_bytes < intra_workspace_floats(kn, cin, cout) * sizeof(float))
            return EPN_EWORKSPACE;
        return launch_intra_fwd_mfma(feats_cl, intra_idx, W, b, p, na, kn, cin, cout, out_cl,
                                     static_cast<float *>(workspace), st);
    }
    return launch_intra_fwd_generic(feats_cl, intra_idx, W, (size_t)b * p, na, kn, cin, cout, out_cl, st);
}

extern "C" int epn_intra_so3conv_bwd_data_f32(const float *grad_out_cl, const int32_t *intra_idx,
                                              const int32_t *inv_idx, const float *W, int b, int p, int na, int kn,
                                              int cin, int cout, float *grad_feats_cl, void *workspace,
                                              size_t workspace_bytes, epn_stream_t stream) {
    int rc = check_intra(b, p, na, kn, cin, cout);
    if (rc) return rc;
    if (b == 0 || p == 0) return 0;
    if (!grad_out_cl || !intra_idx || !W || !grad_feats_cl) return EPN_ENULL;
    hipStream_t st = epn_stream(stream);
    if (inv_idx && intra_uses_mfma(na, kn, cin, cout) && !force_generic()) {
        if (!workspace || workspace_bytes < intra_workspace_floats(kn, cin, cout) * sizeof(float))
            return EPN_EWORKSPACE;
        return launch_intra_bwd_data_mfma(grad_out_cl, inv_idx, W, b, p, na, kn, cin, cout, grad_feats_cl,
                                          static_cast<float *>(workspace), st);
    }
    EPN_HIP(hipMemsetAsync(grad_feats_cl, 0, sizeof(float) * (size_t)b * p * na * cin, st));
    return launch_intra_bwd_data_generic(grad_out_cl, intra_idx, W, (size_t)b * p, na, kn, cin, cout, grad_feats_cl,
                                         st);
}

extern "C" int epn_intra_so3conv_bwd_weight_f32(const float *feats_cl, const float *grad_out_cl,
                                                const int32_t *intra_idx, int b, int p, int na, int kn, int cin,
                                                int cout, float *grad_W, epn_stream_t stream) {
    int rc = check_intra(b, p, na, kn, cin, cout);
    if (rc) return rc;
    if (!grad_W) return EPN_ENULL;
    hipStream_t st = epn_stream(stream);
    EPN_HIP(hipMemsetAsync(grad_W, 0, sizeof(float) * (size_t)cout * cin * kn, st));
    if (b == 0 || p == 0) return 0;
    if (!feats_cl || !grad_out_cl || !intra_idx) return EPN_ENULL;
    if (intra_uses_mfma(na, kn, cin, cout) && !force_generic())
        return launch_intra_bwd_weight_mfma(feats_cl, grad_out_cl, intra_idx, b, p, na, kn, cin, cout, grad_W, st);
    return launch_intra_bwd_weight_generic(feats_cl, grad_out_cl, intra_idx, (size_t)b * p, na, kn, cin, cout, grad_W,
                                           st);
}
