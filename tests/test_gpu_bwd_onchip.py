"""-m gpu: the data gradient of InterSO3Conv with the grouped-feature gradient kept on chip (csrc/inter_bwd_f2.hip,
epn_inter_bwd_data_f16x2_f32, EPN_INTER_BWD_DATA=onchip) -- autograd's transpose of BasicSO3Conv's matmul
(vgtk/vgtk/so3conv/modules.py:48-55) chained with the scatter-add that is the backward of the grouping's gather
(vgtk/vgtk/spconv/functional.py:372-390) in one kernel.  Checked against the CPU oracle's autograd gradient (1e-3 of the
gradient's scale, as the split form's) and against the split form on the same inputs."""
import numpy as np
import pytest
import torch

from conftest import unit_ball_cloud
from oracle import so3conv_ref as R

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _layer(gpu, b, n, cin, cout, K, stride, radius=0.45, sigma=0.09, seed=5):
    from epn_pointcloud_amd import ops
    from epn_pointcloud_amd.vgtk import pc as pctk
    from epn_pointcloud_amd.vgtk.so3conv import functional as L
    from epn_pointcloud_amd.vgtk import functional as fr
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    xyz = T(unit_ball_cloud(rng, b, n)).to(gpu)
    anchors = T(L.get_anchors(60)).to(gpu)
    kernels = R.scaled_kernel_points(T(fr.kernel_points_raw(24)), radius).to(gpu)
    _, new_xyz = pctk.furthest_sample(xyz, n // stride, stride == 1)
    idx = pctk.ball_query_index(new_xyz, xyz, radius, K)
    geo = ops.InterGeometry(xyz, new_xyz, idx, anchors, kernels, sigma)
    feats = torch.randn(b, cin, n, 60, device=gpu)
    W = torch.randn(cout, cin * 24, device=gpu) / (cin * 24) ** 0.5
    gy = torch.randn(b, cout, n // stride, 60, device=gpu) * 1e-3        # gradient-sized: exercises the power-of-two scales
    return geo, feats, W, gy, (xyz, new_xyz, idx, anchors, kernels, sigma)


def _grad(ops, mode, geo, feats, W, gy, monkeypatch, side=None):
    monkeypatch.setenv("EPN_INTER_BWD_DATA", mode)
    f = feats.clone().requires_grad_(True)
    w = W.clone().requires_grad_(True)
    if side is None:
        out = ops.InterSO3ConvSplitFn.apply(f * 1.0, w, geo)
        gf, gw = torch.autograd.grad(out, [f, w], gy)
    else:                                                                  # the shared-input form: the scatter accumulates
        monkeypatch.setenv("EPN_SHARE_INPUT_GRAD", "1")
        out, h2, _ = ops.inter_so3conv(f * 1.0, w, geo, share_input=True)
        gf, gw = torch.autograd.grad([out, h2], [f, w], [gy, side])
    return gf, gw


@pytest.mark.parametrize("cin,cout,K,stride", [(64, 64, 16, 1), (64, 128, 32, 2), (128, 128, 16, 1), (128, 256, 32, 2),
                                               (256, 256, 16, 1), (16, 64, 16, 1), (32, 128, 20, 2), (48, 64, 9, 1)])
def test_onchip_data_gradient_vs_split_and_oracle(gpu, monkeypatch, cin, cout, K, stride):
    from epn_pointcloud_amd import gemm, ops
    assert gemm.FP32_MODE == "f16x2"
    b, n = 2, 192
    geo, feats, W, gy, raw = _layer(gpu, b, n, cin, cout, K, stride)
    d = geo.desc(cin, cout)
    import ctypes
    from epn_pointcloud_amd import _lib
    assert _lib.get_lib().epn_inter_bwd_data_f16x2_ok(ctypes.byref(d)) == 1
    gs, _ = _grad(ops, "split", geo, feats, W, gy, monkeypatch)
    go, gwo = _grad(ops, "onchip", geo, feats, W, gy, monkeypatch)
    scale = gs.abs().max().item()
    assert scale > 0
    # same arithmetic form on both sides (two-piece fp16 contraction, fp32 weights and scatter): they differ by the order of the
    # fp32 atomics, the tensor-wide instead of per-row scale of W and the kernel-point order of the tail's contraction
    assert (go - gs).abs().max().item() <= 2e-5 * scale, ((go - gs).abs().max().item(), scale)
    # the oracle's autograd gradient (materialising reference algorithm, CPU)
    xyz, new_xyz, idx, anchors, kernels, sigma = raw
    fc = feats.cpu().clone().requires_grad_(True)
    grouped = R.group_nd(R.add_shadow_point(xyz.cpu()), idx.cpu()) - new_xyz.cpu().unsqueeze(3)
    o_w = R.inter_weights(grouped, anchors.cpu(), kernels.cpu(), sigma)
    out = R.basic_conv(W.cpu(), R.inter_feat_grouping(idx.cpu(), o_w, R.add_shadow_feature(fc)))
    (g_ref,) = torch.autograd.grad(out, fc, gy.cpu())
    assert (go.cpu() - g_ref).abs().max().item() <= 1e-3 * g_ref.abs().max().item()
    # the accumulating form (the skip branch's gradient of the shared block input is added by the same scatter)
    side = torch.randn_like(feats) * scale
    a_s, _ = _grad(ops, "split", geo, feats, W, gy, monkeypatch, side)
    a_o, _ = _grad(ops, "onchip", geo, feats, W, gy, monkeypatch, side)
    assert (a_o - a_s).abs().max().item() <= 2e-5 * a_s.abs().max().item()
    assert (a_o - go - side).abs().max().item() <= 1e-5 * a_s.abs().max().item()


def test_onchip_data_gradient_is_taken_and_checks_its_arguments(gpu, monkeypatch):
    import ctypes
    from epn_pointcloud_amd import _lib, ops
    geo, feats, W, gy, _ = _layer(gpu, 2, 128, 64, 64, 16, 1)
    lib = _lib.get_lib()
    d = geo.desc(64, 64)
    calls = []
    real = ops._launch
    monkeypatch.setattr(ops, "_launch", lambda kind, *a: (calls.append(kind), real(kind, *a))[1])
    _grad(ops, "onchip", geo, feats, W, gy, monkeypatch)
    assert "inter_bwd_data_f2" in calls and "inter_gemm_dg" not in calls and "inter_ungroup" not in calls
    # shapes it does not take fall back to the split form: ks != 24 is not reachable through this geometry; cout = 96 is
    d2 = geo.desc(64, 96)
    assert lib.epn_inter_bwd_data_f16x2_ok(ctypes.byref(d2)) == 0
    assert lib.epn_inter_bwd_data_f16x2_workspace_bytes(ctypes.byref(d2)) == 0
    g = torch.zeros(2, 64, 128, 60, device=gpu).contiguous(memory_format=torch.channels_last)
    am = torch.ones(1, device=gpu)
    ws = torch.empty(int(lib.epn_inter_bwd_data_f16x2_workspace_bytes(ctypes.byref(d))), dtype=torch.uint8, device=gpu)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    st = _lib.stream_of(g)
    assert lib.epn_inter_bwd_data_f16x2_f32(ctypes.byref(d), vp(g), vp(W), None, vp(g), 0, vp(ws), ws.numel(), st) == -3       # EPN_ENULL
    assert lib.epn_inter_bwd_data_f16x2_f32(ctypes.byref(d), vp(g), vp(W), vp(am), vp(g), 0, vp(ws), 16, st) == -2           # EPN_EWORKSPACE
    assert lib.epn_inter_bwd_data_f16x2_f32(ctypes.byref(d2), vp(g), vp(W), vp(am), vp(g), 0, vp(ws), ws.numel(), st) == -1   # EPN_EINVAL
