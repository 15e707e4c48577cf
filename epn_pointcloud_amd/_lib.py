"""ctypes binding of libepn_so3conv.so (include/epn_so3conv.h) on torch device tensors.

PyTorch is plumbing here: it owns device memory and the current HIP stream; every compute call goes
through the C ABI.  There is NO CPU or eager fallback: a missing library or a non-device tensor
raises (the reference's extensions do the same through CHECK_CUDA, grouping_cuda.cpp:66-68).
"""
import ctypes
import os

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EPN_LIB", os.path.join(_PKG, "libepn_so3conv.so"))   # EPN_LIB: A/B builds (tools/)

ABI_VERSION = 3          # EPN_ABI_VERSION of the include/epn_so3conv.h this binding was written against

EXPORTS = [
    "epn_version", "epn_abi_version", "epn_strerror", "epn_set_kernel_policy",
    "epn_ball_query_f32", "epn_fps_f32", "epn_fps_temp_f32", "epn_gather_fwd_f32", "epn_gather_bwd_f32", "epn_initial_anchor_query_f32",
    "epn_initial_anchor_query_f64", "epn_anchor_query_f64",
    "epn_inter_workspace_bytes", "epn_inter_is_fused", "epn_inter_so3conv_fwd_f32",
    "epn_inter_so3conv_bwd_data_f32", "epn_inter_so3conv_bwd_weight_f32", "epn_inter_weights_f32",
    "epn_intra_workspace_bytes", "epn_intra_is_fused", "epn_intra_so3conv_fwd_f32",
    "epn_intra_so3conv_bwd_data_f32", "epn_intra_so3conv_bwd_weight_f32",
    "epn_norm_workspace_bytes", "epn_bn_running_update_f32", "epn_chan_stats_f32", "epn_norm_act_fwd_f32", "epn_norm_act_bwd_reduce_f32", "epn_norm_act_bwd_apply_f32",
    "epn_inter_group_workspace_bytes", "epn_inter_group_f32", "epn_inter_ungroup_f32", "epn_intra_group_f32", "epn_so3_basis_f32",
    "epn_anchor_query_f32", "epn_zp_inter_fwd_f32", "epn_zp_inter_bwd_f32", "epn_zp_intra_fwd_f32", "epn_zp_intra_bwd_f32",
    "epn_pointnet_so3conv_fwd_f32", "epn_pointnet_so3conv_bwd_data_f32", "epn_pointnet_max_f32", "epn_pointnet_dz_f32", "epn_pointnet_dz_bf16", "epn_pointnet_bwd_coord_f32",
    "epn_pointnet_so3conv_bwd_weight_f32",
    "epn_gemm_nt_f32", "epn_gemm_nt_bf16", "epn_gemm_nt_split_workspace_bytes", "epn_gemm_nt_split_f32", "epn_gemm_tn_workspace_bytes", "epn_gemm_tn_f32", "epn_gemm_tn_split_f32", "epn_gemm_tn_bf16",
    "epn_transpose_cast", "epn_cast", "epn_gemm_tn_grouped_workspace_bytes", "epn_gemm_tn_grouped",
    "epn_so3_basis_amax_split_f32", "epn_so3_basis_norm_amax_split_f32",
    "epn_norm_act_pair_fwd_amax", "epn_norm_act_pair_bwd_apply_amax", "epn_norm_act_bwd_apply_amax_f32",
    "epn_absmax_f32", "epn_f16x2_overflow_count", "epn_inter_bwd_data_f16x2_ok", "epn_inter_bwd_data_f16x2_workspace_bytes",
    "epn_inter_bwd_data_f16x2_f32", "epn_inter_ungroup_cloud_ok", "epn_inter_ungroup_cloud_workspace_bytes", "epn_inter_ungroup_cloud_f32",
    "epn_inter_ungroup_cloud_bf16", "epn_inter_ungroup_cloud_range_count", "epn_gemm_nt_f16x2_workspace_bytes", "epn_gemm_nt_f16x2_f32", "epn_gemm_tn_f16x2_f32", "epn_gemm_tn_grouped_f16x2",
    "epn_inter_group_bf16", "epn_inter_ungroup_bf16", "epn_intra_group_bf16", "epn_so3_basis_bf16",
    "epn_gather_rows", "epn_scatter_rows", "epn_conv1x1_c1_f32", "epn_conv1x1_c1_bwd_weight_f32",
    "epn_anchor_softmax_pool_fwd_f32", "epn_anchor_softmax_pool_bwd_f32",
    "epn_ball_query_f64", "epn_fps_f64", "epn_gather_fwd_f64", "epn_gather_bwd_f64", "epn_so3_basis_norm_f32", "epn_so3_basis_norm_bf16", "epn_so3_basis_split_f32", "epn_so3_basis_norm_split_f32", "epn_inter_inverse_list", "epn_inter_ungroup_det_f32", "epn_inter_ungroup_det_bf16",
    "epn_chan_stats_bf16", "epn_norm_act_fwd_bf16", "epn_norm_act_bwd_reduce_bf16", "epn_norm_act_bwd_apply_bf16",
    "epn_last_kernel", "epn_scatter_rows_add", "epn_norm_pair_workspace_bytes", "epn_norm_act_pair_fwd",
    "epn_norm_act_pair_bwd_reduce", "epn_norm_act_pair_bwd_apply", "epn_inter_onchip_ok", "epn_inter_onchip_workspace_bytes", "epn_inter_so3conv_fwd_onchip_f32", "epn_inter_so3conv_fwd_bf16",
    "epn_inter_group_packed_ok", "epn_inter_packed_position", "epn_inter_group_packed_f32", "epn_inter_group_packed_bf16",
    "epn_inter_pack_weights_f32", "epn_inter_pack_weights_bf16", "epn_inter_unpack_weight_grad_f32",
    "epn_inter_ungroup_acc_f32", "epn_inter_ungroup_acc_bf16", "epn_stats_finish", "epn_stats_finish_workspace_bytes",
    "epn_so3_basis_stats_f32", "epn_so3_basis_stats_split_f32", "epn_so3_basis_stats_bf16",
    "epn_spectral_weights_f32", "epn_spectral_weights_bwd_f32", "epn_spectral_weights_bf16", "epn_cast_add_bf16",
    "epn_so3_basis_dstats_f32", "epn_so3_basis_dstats_split_f32", "epn_so3_basis_dstats_bf16", "epn_norm_bwd_finish",
    "epn_inter_c1_ok", "epn_inter_so3conv_fwd_c1_f32", "epn_inter_so3conv_bwd_weight_c1_f32",
    "epn_inter_split_ok", "epn_inter_split_saved_bytes", "epn_inter_split_workspace_bytes",
    "epn_inter_so3conv_fwd_split_f32", "epn_inter_so3conv_fwd_split_bf16", "epn_inter_so3conv_bwd_split_f32",
    "epn_inter_so3conv_bwd_split_bf16",
]

_vp, _ci, _cf, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t


class InterDesc(ctypes.Structure):
    """struct epn_inter_desc (include/epn_so3conv.h)."""
    _fields_ = [("xyz", _vp), ("new_xyz", _vp), ("ball_idx", _vp), ("anchors", _vp), ("kernels", _vp),
                ("dense_w", _vp), ("sigma", _cf), ("b", _ci), ("p1", _ci), ("p2", _ci), ("nn", _ci),
                ("na", _ci), ("ks", _ci), ("cin", _ci), ("cout", _ci)]


class NormPairSide(ctypes.Structure):
    """struct epn_norm_pair_side (include/epn_so3conv.h)."""
    _fields_ = [("sums", _vp), ("gamma", _vp), ("beta", _vp), ("eps", _cf), ("instance", _ci)]


class GemmNtProblem(ctypes.Structure):
    """struct epn_gemm_nt_problem (include/epn_so3conv.h)."""
    _fields_ = [("A", _vp), ("Bt", _vp), ("C", _vp), ("M", ctypes.c_longlong), ("lda", ctypes.c_longlong),
                ("ldb", ctypes.c_longlong), ("ldc", ctypes.c_longlong), ("N", _ci), ("K", _ci), ("col_stats", _vp), ("c_amax", _vp)]


class GemmTnProblem(ctypes.Structure):
    """struct epn_gemm_tn_problem (include/epn_so3conv.h)."""
    _fields_ = [("X", _vp), ("Y", _vp), ("C", _vp), ("R", ctypes.c_longlong), ("ldx", ctypes.c_longlong),
                ("ldy", ctypes.c_longlong), ("ldc", ctypes.c_longlong), ("N1", _ci), ("N2", _ci)]


_lib = None


def get_lib():
    """Load the HIP library or fail loudly (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension was not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
            "epn_pointcloud_amd has no CPU/eager fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    try:
        got = int(lib.epn_abi_version())
    except AttributeError:
        got = None
    if got != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} implements ABI revision {got}, this binding expects {ABI_VERSION} "
                           "(include/epn_so3conv.h EPN_ABI_VERSION): rebuild the library -- signatures differ between revisions")
    lib.epn_version.restype = ctypes.c_char_p
    lib.epn_strerror.restype = ctypes.c_char_p
    lib.epn_last_kernel.restype = ctypes.c_char_p
    lib.epn_strerror.argtypes = [_ci]
    lib.epn_set_kernel_policy.argtypes = [_ci]
    lib.epn_set_kernel_policy.restype = _ci
    lib.epn_ball_query_f32.argtypes = [_vp, _vp, _ci, _ci, _ci, _cf, _ci, _vp, _vp]
    lib.epn_fps_f32.argtypes = [_vp, _ci, _ci, _ci, _vp, _vp]
    lib.epn_initial_anchor_query_f32.argtypes = [_vp, _vp, _vp, _ci, _ci, _ci, _ci, _ci, _cf, _cf, _vp, _vp, _vp]
    lib.epn_initial_anchor_query_f64.argtypes = lib.epn_initial_anchor_query_f32.argtypes
    lib.epn_gather_fwd_f32.argtypes = [_vp, _vp, _ci, _ci, _ci, _ci, _vp, _vp]
    lib.epn_gather_bwd_f32.argtypes = [_vp, _vp, _ci, _ci, _ci, _ci, _vp, _vp]
    lib.epn_ball_query_f64.argtypes = [_vp, _vp, _ci, _ci, _ci, ctypes.c_float, _ci, _vp, _vp]
    lib.epn_fps_f64.argtypes = [_vp, _ci, _ci, _ci, _vp, _vp, _vp]
    lib.epn_fps_temp_f32.argtypes = lib.epn_fps_f64.argtypes
    lib.epn_gather_fwd_f64.argtypes = lib.epn_gather_fwd_f32.argtypes
    lib.epn_gather_bwd_f64.argtypes = lib.epn_gather_bwd_f32.argtypes
    dp = ctypes.POINTER(InterDesc)
    lib.epn_inter_workspace_bytes.argtypes = [dp]
    lib.epn_inter_workspace_bytes.restype = _sz
    lib.epn_inter_so3conv_fwd_f32.argtypes = [dp, _vp, _vp, _vp, _vp, _sz, _vp]
    lib.epn_inter_so3conv_bwd_data_f32.argtypes = [dp, _vp, _vp, _vp, _vp, _sz, _vp]
    lib.epn_inter_so3conv_bwd_weight_f32.argtypes = [dp, _vp, _vp, _vp, _vp, _sz, _vp]
    lib.epn_inter_weights_f32.argtypes = [dp, _vp, _vp]
    lib.epn_inter_onchip_ok.argtypes = [dp, _ci]
    lib.epn_inter_onchip_ok.restype = _ci
    lib.epn_inter_onchip_workspace_bytes.argtypes = [dp, _ci]
    lib.epn_inter_onchip_workspace_bytes.restype = _sz
    lib.epn_inter_so3conv_fwd_onchip_f32.argtypes = [dp, _vp, _vp, _vp, _vp, _sz, _vp]
    lib.epn_inter_so3conv_fwd_bf16.argtypes = [dp, _vp, _vp, _vp, _vp, _sz, _vp]
    for _n in ("epn_so3_basis_dstats_f32", "epn_so3_basis_dstats_split_f32", "epn_so3_basis_dstats_bf16"):
        getattr(lib, _n).argtypes = [_vp, _vp, _vp, ctypes.c_longlong, _ci, _ci, _vp, _vp, _vp, _ci, ctypes.c_longlong, _vp, _vp,
                                     _cf, _cf, _vp, _vp]
    lib.epn_norm_bwd_finish.argtypes = [_vp, _ci, ctypes.c_longlong, _ci, _vp, _vp, _vp, _vp, _vp, _sz, _vp]
    lib.epn_inter_c1_ok.argtypes = [dp]
    lib.epn_inter_c1_ok.restype = _ci
    lib.epn_inter_so3conv_fwd_c1_f32.argtypes = [dp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]
    lib.epn_inter_so3conv_bwd_weight_c1_f32.argtypes = [dp, _vp, _vp, _vp, _vp]
    lib.epn_inter_split_ok.argtypes = [dp]
    lib.epn_inter_split_ok.restype = _ci
    lib.epn_inter_split_saved_bytes.argtypes = [dp, _ci]
    lib.epn_inter_split_saved_bytes.restype = _sz
    lib.epn_inter_split_workspace_bytes.argtypes = [dp, _ci, _ci]
    lib.epn_inter_split_workspace_bytes.restype = _sz
    for _n in ("epn_inter_so3conv_fwd_split_f32", "epn_inter_so3conv_fwd_split_bf16"):
        getattr(lib, _n).argtypes = [dp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _sz, _vp]
    for _n in ("epn_inter_so3conv_bwd_split_f32", "epn_inter_so3conv_bwd_split_bf16"):
        getattr(lib, _n).argtypes = [dp, _vp, _vp, _vp, _sz, _vp, _ci, _vp, _vp, _sz, _vp]
    lib.epn_inter_group_workspace_bytes.argtypes = [dp]
    lib.epn_inter_group_workspace_bytes.restype = _sz
    lib.epn_inter_group_f32.argtypes = [dp, _vp, _vp, _vp, _sz, _vp]
    lib.epn_inter_ungroup_f32.argtypes = [dp, _vp, _vp, _vp, _sz, _vp]
    for _n in ("epn_so3_basis_stats_f32", "epn_so3_basis_stats_split_f32", "epn_so3_basis_stats_bf16"):
        getattr(lib, _n).argtypes = [_vp, _vp, _vp, ctypes.c_longlong, _ci, _ci, _ci, _ci, _vp, _vp, _vp]
    lib.epn_spectral_weights_f32.argtypes = [_vp, _vp, _vp, _ci, _ci, _ci, _ci, _vp, _vp, _vp]
    lib.epn_cast_add_bf16.argtypes = [_vp, _vp, _vp, _sz, _vp]
    lib.epn_spectral_weights_bf16.argtypes = [_vp, _vp, _vp, _ci, _ci, _ci, _ci, _vp, _vp, _vp]
    lib.epn_spectral_weights_bwd_f32.argtypes = [_vp, _vp, _vp, _ci, _ci, _ci, _ci, _vp, _vp]
    lib.epn_stats_finish.argtypes = [_vp, _ci, ctypes.c_longlong, _ci, _vp, _vp, _sz, _vp]
    lib.epn_stats_finish_workspace_bytes.argtypes = [_ci, ctypes.c_longlong, _ci]
    lib.epn_stats_finish_workspace_bytes.restype = _sz
    lib.epn_inter_ungroup_acc_f32.argtypes = [dp, _vp, _vp, _vp, _sz, _vp]
    lib.epn_inter_ungroup_acc_bf16.argtypes = [dp, _vp, _vp, _vp, _sz, _vp]
    lib.epn_inter_group_packed_ok.argtypes = [dp]
    lib.epn_inter_group_packed_ok.restype = _ci
    lib.epn_inter_packed_position.argtypes = [_ci, _ci, _vp]
    lib.epn_inter_group_packed_f32.argtypes = [dp, _vp, _vp, _vp, _sz, _vp]
    lib.epn_inter_group_packed_bf16.argtypes = [dp, _vp, _vp, _vp, _sz, _vp]
    lib.epn_inter_pack_weights_f32.argtypes = [_vp, _ci, _ci, _ci, _vp, _vp]
    lib.epn_inter_pack_weights_bf16.argtypes = [_vp, _ci, _ci, _ci, _vp, _vp]
    lib.epn_inter_unpack_weight_grad_f32.argtypes = [_vp, _ci, _ci, _ci, _vp, _vp]
    lib.epn_intra_group_f32.argtypes = [_vp, _vp, _vp, _ci, _ci, _ci, _ci, _ci, _vp]
    for _n in ("epn_zp_inter_fwd_f32", "epn_zp_inter_bwd_f32", "epn_zp_intra_fwd_f32", "epn_zp_intra_bwd_f32"):
        getattr(lib, _n).argtypes = [_vp, _vp, _vp, _ci, _ci, _ci, _ci, _ci, _ci, _ci, _vp, _vp]
    lib.epn_anchor_query_f32.argtypes = [_vp, _vp, _vp, _ci, _ci, _ci, _ci, _ci, _vp, _vp]
    lib.epn_anchor_query_f64.argtypes = lib.epn_anchor_query_f32.argtypes
    lib.epn_so3_basis_f32.argtypes = [_vp, _vp, _vp, ctypes.c_longlong, _ci, _ci, _ci, _ci, _vp, _vp]
    lib.epn_so3_basis_norm_f32.argtypes = [_vp, _vp, _vp, ctypes.c_longlong, _ci, _ci, _ci, _vp, _vp, _ci, ctypes.c_longlong,
                                           _vp, _vp, _cf, _cf, _vp]
    lib.epn_so3_basis_norm_bf16.argtypes = lib.epn_so3_basis_norm_f32.argtypes
    lib.epn_so3_basis_norm_split_f32.argtypes = lib.epn_so3_basis_norm_f32.argtypes
    lib.epn_so3_basis_split_f32.argtypes = lib.epn_so3_basis_f32.argtypes
    lib.epn_so3_basis_amax_split_f32.argtypes = [_vp, _vp, _vp, ctypes.c_longlong, _ci, _ci, _ci, _ci, _vp, _vp, _vp]
    lib.epn_so3_basis_norm_amax_split_f32.argtypes = [_vp, _vp, _vp, ctypes.c_longlong, _ci, _ci, _ci, _vp, _vp, _ci,
                                                      ctypes.c_longlong, _vp, _vp, _cf, _cf, _vp, _vp]
    lib.epn_inter_is_fused.argtypes = [dp]
    lib.epn_inter_is_fused.restype = _ci
    lib.epn_intra_is_fused.argtypes = [_ci, _ci, _ci, _ci]
    lib.epn_intra_is_fused.restype = _ci
    lib.epn_intra_workspace_bytes.argtypes = [_ci, _ci, _ci, _ci]
    lib.epn_intra_workspace_bytes.restype = _sz
    lib.epn_intra_so3conv_fwd_f32.argtypes = [_vp, _vp, _vp, _ci, _ci, _ci, _ci, _ci, _ci, _vp, _vp, _sz, _vp]
    lib.epn_intra_so3conv_bwd_data_f32.argtypes = [_vp, _vp, _vp, _vp, _ci, _ci, _ci, _ci, _ci, _ci, _vp, _vp, _sz,
                                                   _vp]
    lib.epn_intra_so3conv_bwd_weight_f32.argtypes = [_vp, _vp, _vp, _ci, _ci, _ci, _ci, _ci, _ci, _vp, _vp]
    _ll = ctypes.c_longlong
    lib.epn_norm_workspace_bytes.argtypes = [_ci, _ll, _ci]
    lib.epn_norm_workspace_bytes.restype = _sz
    lib.epn_chan_stats_f32.argtypes = [_vp, _ci, _ll, _ci, _vp, _vp, _sz, _vp]
    lib.epn_bn_running_update_f32.argtypes = [_vp, ctypes.c_double, _vp, _vp, _vp, _vp, _cf, _ci, _vp]
    lib.epn_norm_act_fwd_f32.argtypes = [_vp, _ci, _ll, _ci, _vp, _vp, _vp, _vp, _cf, _cf, _vp, _vp]
    lib.epn_norm_act_bwd_reduce_f32.argtypes = [_vp, _vp, _ci, _ll, _ci, _vp, _vp, _vp, _cf, _cf, _vp, _vp, _vp, _vp,
                                                _sz, _vp]
    lib.epn_norm_act_bwd_apply_f32.argtypes = [_vp, _vp, _ci, _ll, _ci, _vp, _vp, _vp, _vp, _cf, _cf, _vp, _vp]
    gp = ctypes.POINTER(GemmNtProblem)
    lib.epn_gemm_nt_f32.argtypes = [_ci, gp, _vp]
    lib.epn_gemm_nt_bf16.argtypes = [_ci, gp, _ci, _vp]
    lib.epn_gemm_nt_split_workspace_bytes.argtypes = [_ci, gp]
    lib.epn_gemm_nt_split_workspace_bytes.restype = ctypes.c_size_t
    lib.epn_gemm_nt_split_f32.argtypes = [_ci, gp, _vp, ctypes.c_size_t, _vp]
    lib.epn_gemm_tn_workspace_bytes.argtypes = [_ci, _ll, _ci, _ci]
    lib.epn_gemm_tn_f32.argtypes = [_vp, _ll, _vp, _ll, _vp, _ll, _ll, _ci, _ci, _vp, _sz, _vp]
    lib.epn_gemm_tn_bf16.argtypes = [_vp, _ll, _vp, _ll, _vp, _ll, _ll, _ci, _ci, _vp, _sz, _vp]
    lib.epn_gemm_tn_split_f32.argtypes = [_vp, _ll, _vp, _ll, _vp, _ll, _ll, _ci, _ci, _vp, _sz, _vp]
    tp = ctypes.POINTER(GemmTnProblem)
    lib.epn_gemm_tn_grouped_workspace_bytes.argtypes = [_ci, _ci, tp]
    lib.epn_gemm_tn_grouped_workspace_bytes.restype = _sz
    lib.epn_gemm_tn_grouped.argtypes = [_ci, _ci, tp, _vp, _sz, _vp]
    pp = ctypes.POINTER(_vp)                 # const float *const * (arrays of device pointers, entries may be NULL)
    lib.epn_absmax_f32.argtypes = [_vp, _ll, _ll, _ll, _vp, _vp]
    lib.epn_f16x2_overflow_count.argtypes = [_ci]
    lib.epn_inter_bwd_data_f16x2_ok.argtypes = [dp]
    lib.epn_inter_bwd_data_f16x2_ok.restype = _ci
    lib.epn_inter_bwd_data_f16x2_workspace_bytes.argtypes = [dp]
    lib.epn_inter_bwd_data_f16x2_workspace_bytes.restype = _sz
    lib.epn_inter_bwd_data_f16x2_f32.argtypes = [dp, _vp, _vp, _vp, _vp, _ci, _vp, _sz, _vp]
    lib.epn_f16x2_overflow_count.restype = _ll
    lib.epn_inter_ungroup_cloud_ok.argtypes = [dp]
    lib.epn_inter_ungroup_cloud_ok.restype = _ci
    lib.epn_inter_ungroup_cloud_workspace_bytes.argtypes = [dp]
    lib.epn_inter_ungroup_cloud_workspace_bytes.restype = _sz
    lib.epn_inter_ungroup_cloud_f32.argtypes = [dp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]
    lib.epn_inter_ungroup_cloud_bf16.argtypes = [dp, _vp, _vp, _vp, _vp, _ci, _vp, _sz, _vp]
    lib.epn_inter_ungroup_cloud_range_count.argtypes = [_ci]
    lib.epn_inter_ungroup_cloud_range_count.restype = _ll
    lib.epn_gemm_nt_f16x2_workspace_bytes.argtypes = [_ci, gp]
    lib.epn_gemm_nt_f16x2_workspace_bytes.restype = ctypes.c_size_t
    lib.epn_gemm_nt_f16x2_f32.argtypes = [_ci, gp, pp, _vp, ctypes.c_size_t, _vp]
    lib.epn_gemm_tn_f16x2_f32.argtypes = [_vp, _ll, _vp, _ll, _vp, _ll, _ll, _ci, _ci, _vp, _vp, _vp, _sz, _vp]
    lib.epn_gemm_tn_grouped_f16x2.argtypes = [_ci, tp, pp, pp, _vp, _sz, _vp]
    lib.epn_gemm_tn_grouped.restype = _ci
    lib.epn_transpose_cast.argtypes = [_vp, _vp, _ci, _ci, _ci, _ci, _vp]
    lib.epn_cast.argtypes = [_vp, _vp, _sz, _ci, _ci, _vp]
    # the bf16 twins share their fp32 counterparts' signatures (void* feature pointers)
    lib.epn_gather_rows.argtypes = [_vp, _vp, _vp, _ci, _ci, _ci, _ll, _vp]
    lib.epn_scatter_rows.argtypes = [_vp, _vp, _vp, _ci, _ci, _ci, _ll, _vp]
    lib.epn_scatter_rows_add.argtypes = [_vp, _vp, _vp, _ci, _ci, _ci, _ll, _ci, _vp]
    sp = ctypes.POINTER(NormPairSide)
    lib.epn_norm_pair_workspace_bytes.argtypes = [_ci, _ll, _ci]
    lib.epn_norm_pair_workspace_bytes.restype = _sz
    lib.epn_norm_act_pair_fwd.argtypes = [_vp, _vp, _ci, _ll, _ci, sp, sp, _cf, _vp, _ci, _vp]
    lib.epn_norm_act_pair_bwd_reduce.argtypes = [_vp, _vp, _vp, _ci, _ll, _ci, sp, sp, _cf, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                                 _sz, _ci, _vp]
    lib.epn_norm_act_pair_bwd_apply.argtypes = [_vp, _vp, _vp, _ci, _ll, _ci, sp, sp, _cf, _vp, _vp, _vp, _vp, _ci, _vp]
    lib.epn_norm_act_pair_fwd_amax.argtypes = [_vp, _vp, _ci, _ll, _ci, sp, sp, _cf, _vp, _ci, _vp, _vp]
    lib.epn_norm_act_pair_bwd_apply_amax.argtypes = [_vp, _vp, _vp, _ci, _ll, _ci, sp, sp, _cf, _vp, _vp, _vp, _vp, _ci, _vp, _vp]
    lib.epn_norm_act_bwd_apply_amax_f32.argtypes = [_vp, _vp, _ci, _ll, _ci, _vp, _vp, _vp, _vp, _cf, _cf, _vp, _vp, _vp]
    for _n in ("epn_norm_act_pair_fwd", "epn_norm_act_pair_bwd_reduce", "epn_norm_act_pair_bwd_apply"):
        getattr(lib, _n).restype = _ci
    lib.epn_scatter_rows_add.restype = _ci
    lib.epn_conv1x1_c1_f32.argtypes = [_vp, _vp, _vp, _ll, _ci, _vp]
    lib.epn_conv1x1_c1_bwd_weight_f32.argtypes = [_vp, _vp, _vp, _ll, _ci, _vp]
    lib.epn_anchor_softmax_pool_fwd_f32.argtypes = [_vp, _vp, _vp, _vp, _ll, _ci, _ci, _vp]
    lib.epn_anchor_softmax_pool_bwd_f32.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _ll, _ci, _ci, _vp]
    lib.epn_gather_rows.restype = lib.epn_scatter_rows.restype = _ci
    lib.epn_inter_inverse_list.argtypes = [_vp, _ci, _ci, _ci, _ci, _vp, _vp, _vp]
    lib.epn_inter_inverse_list.restype = _ci
    lib.epn_inter_ungroup_det_f32.argtypes = [dp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _sz, _vp]
    lib.epn_inter_ungroup_det_bf16.argtypes = lib.epn_inter_ungroup_det_f32.argtypes
    lib.epn_inter_group_bf16.argtypes = lib.epn_inter_group_f32.argtypes
    lib.epn_inter_ungroup_bf16.argtypes = lib.epn_inter_ungroup_f32.argtypes
    lib.epn_intra_group_bf16.argtypes = lib.epn_intra_group_f32.argtypes
    lib.epn_so3_basis_bf16.argtypes = lib.epn_so3_basis_f32.argtypes
    lib.epn_chan_stats_bf16.argtypes = lib.epn_chan_stats_f32.argtypes
    lib.epn_norm_act_fwd_bf16.argtypes = lib.epn_norm_act_fwd_f32.argtypes
    lib.epn_norm_act_bwd_reduce_bf16.argtypes = lib.epn_norm_act_bwd_reduce_f32.argtypes
    lib.epn_norm_act_bwd_apply_bf16.argtypes = lib.epn_norm_act_bwd_apply_f32.argtypes
    for name in EXPORTS:
        getattr(lib, name)  # AttributeError here = header/library mismatch
        if name.endswith("_f32") or name.endswith("_bf16") or name in ("epn_transpose_cast", "epn_cast"):
            getattr(lib, name).restype = _ci
    lib.epn_gemm_tn_workspace_bytes.restype = _sz
    _lib = _LibProxy(lib)
    return _lib


# Host-only entry points (no stream argument, nothing launched): handed out unwrapped.
_HOST_ONLY = {"epn_inter_bwd_data_f16x2_ok", "epn_inter_ungroup_cloud_ok", "epn_inter_ungroup_cloud_range_count", "epn_version", "epn_strerror", "epn_set_kernel_policy", "epn_last_kernel", "epn_f16x2_overflow_count", "epn_inter_is_fused", "epn_inter_split_ok", "epn_inter_c1_ok",
              "epn_inter_split_saved_bytes",
              "epn_intra_is_fused", "epn_inter_onchip_ok", "epn_inter_group_packed_ok", "epn_inter_packed_position"}
CALL_HOOK = None      # ops.profile_begin(): callable(name, fn, args) -> rc, brackets every launching call with HIP events


class _StreamArg(ctypes.c_void_p):
    """epn_stream_t argument that remembers its device, and the device to switch back to after the call (stream_of)."""
    restore = None
    device = None


class _LibProxy:
    """The CDLL with every launching entry point wrapped: (i) the current device, if stream_of() had to switch it to the
    tensors' device for the launch, is restored afterwards -- a scoped guard like ATen's, not a permanent set_device;
    (ii) an optional hook (CALL_HOOK) sees every call, so a benchmark can time ALL of them, not only the ones a
    caller chose to bracket."""

    def __init__(self, lib):
        object.__setattr__(self, "_cdll", lib)

    def __getattr__(self, name):
        fn = getattr(self._cdll, name)
        if name in _HOST_ONLY or name.endswith("_workspace_bytes"):
            object.__setattr__(self, name, fn)
            return fn

        def call(*args):
            hook = CALL_HOOK
            st = args[-1] if args else None
            prev = getattr(st, "restore", None)
            dev = getattr(st, "device", None)
            if dev is not None and torch.cuda.current_device() != dev:
                # a stream argument used for a SECOND call (stream_of()'s switch was undone when the first returned): the
                # proxy makes the stream's device current itself and restores whatever was current
                prev = torch.cuda.current_device() if prev is None else prev
                torch.cuda.set_device(dev)
            try:
                return fn(*args) if hook is None else hook(name, fn, args)
            finally:
                if prev is not None:
                    torch.cuda.set_device(prev)
        call.__name__ = name
        object.__setattr__(self, name, call)
        return call


class kernel_policy:
    """`with _lib.kernel_policy(n):` -- epn_set_kernel_policy(n) for the block (1 = generic kernels, 2 = the first layer's VALU
    kernel instead of its matrix-pipe form); cross-checks only."""

    def __init__(self, policy):
        self.policy = int(policy)

    def __enter__(self):
        check(get_lib().epn_set_kernel_policy(self.policy), "set_kernel_policy")

    def __exit__(self, *exc):
        check(get_lib().epn_set_kernel_policy(0), "set_kernel_policy")


class generic_kernels:
    """`with _lib.generic_kernels():` -- run on the any-shape generic kernels (cross-check tests only)."""

    def __enter__(self):
        check(get_lib().epn_set_kernel_policy(1), "set_kernel_policy")

    def __exit__(self, *exc):
        check(get_lib().epn_set_kernel_policy(0), "set_kernel_policy")


def check(rc, what):
    if rc != 0:
        msg = get_lib().epn_strerror(int(rc)).decode()
        raise RuntimeError(f"{what} failed: {msg} (code {rc})")


def stream_of(t):
    """Current torch stream of t's device, as the epn_stream_t argument of the call being made.  The HIP runtime launches
    on ITS current device, so that is switched to t's device first when it differs (a tensor on cuda:1 while cuda:0 is
    current must not launch on cuda:0); the returned argument carries the device to switch back to, and the library
    proxy restores it when the call returns -- the scoped guard the reference's extensions get from ATen."""
    dev = t.device
    arg = _StreamArg(torch.cuda.current_stream(dev).cuda_stream)
    arg.device = dev.index
    if dev.index is not None and torch.cuda.current_device() != dev.index:
        arg.restore = torch.cuda.current_device()
        torch.cuda.set_device(dev)
    return arg


def same_device(*tensors):
    """All device tensors of one call must live on one device (checked where several tensors meet)."""
    devs = {t.device for t in tensors if t is not None}
    if len(devs) > 1:
        raise RuntimeError(f"tensors of one call are on different devices: {sorted(str(d) for d in devs)}")


def float_dtype(t, name):
    """float32 or float64: the dtypes the reference's index / gather extensions dispatch on."""
    if t.dtype not in (torch.float32, torch.float64):
        raise TypeError(f"{name} must be float32 or float64, got {t.dtype}")
    return t.dtype


def dev_ptr(t, name, dtype=torch.float32):
    """Pointer of a device tensor; mirrors the reference's CHECK_INPUT (grouping_cuda.cpp:66-68)."""
    if t is None:
        return ctypes.c_void_p(0)
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")  # ROCm devices are 'cuda' in torch
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    return ctypes.c_void_p(t.data_ptr())
