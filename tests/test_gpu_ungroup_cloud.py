"""-m gpu: the transpose of the grouping with a cloud's gradient rows resident in LDS (csrc/inter_ungroup_cloud.hip,
epn_inter_ungroup_cloud_{f32,bf16}, the default data-gradient path of InterSO3Conv where it is faster) -- the backward of
inter_zpconv_grouping_naive's gather (vgtk/vgtk/spconv/functional.py:372-390) accumulated as 64-bit fixed point, no global
atomics.  Checked against the CPU oracle's autograd gradient, against the atomic scatter (EPN_INTER_BWD_DATA=split) on the same
inputs, for bitwise repeatability, and for what happens when the range contract is broken."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import unit_ball_cloud
from oracle import so3conv_ref as R

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _layer(gpu, b, n, cin, cout, K, stride, radius=0.45, sigma=0.09, seed=5, dtype=torch.float32):
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
    feats = torch.randn(b, cin, n, 60, device=gpu).to(dtype)
    W = torch.randn(cout, cin * 24, device=gpu) / (cin * 24) ** 0.5
    gy = (torch.randn(b, cout, n // stride, 60, device=gpu) * 1e-3).to(dtype)
    return geo, feats, W, gy, (xyz, new_xyz, idx, anchors, kernels, sigma)


def _grad(ops, mode, geo, feats, W, gy, monkeypatch, side=None):
    monkeypatch.setenv("EPN_INTER_BWD_DATA", mode)
    f = feats.clone().requires_grad_(True)
    w = W.clone().requires_grad_(True)
    if side is None:
        out = ops.InterSO3ConvSplitFn.apply(f * 1.0, w, geo)
        gf, gw = torch.autograd.grad(out, [f, w], gy)
    else:
        monkeypatch.setenv("EPN_SHARE_INPUT_GRAD", "1")
        out, h2, _ = ops.inter_so3conv(f * 1.0, w, geo, share_input=True)
        gf, gw = torch.autograd.grad([out, h2], [f, w], [gy, side])
    return gf, gw


def _oracle_grad(raw, feats, W, gy):
    xyz, new_xyz, idx, anchors, kernels, sigma = raw
    fc = feats.float().cpu().clone().requires_grad_(True)
    grouped = R.group_nd(R.add_shadow_point(xyz.cpu()), idx.cpu()) - new_xyz.cpu().unsqueeze(3)
    o_w = R.inter_weights(grouped, anchors.cpu(), kernels.cpu(), sigma)
    out = R.basic_conv(W.cpu(), R.inter_feat_grouping(idx.cpu(), o_w, R.add_shadow_feature(fc)))
    (g_ref,) = torch.autograd.grad(out, fc, gy.float().cpu())
    return g_ref


SHAPES = [(64, 64, 16, 1), (64, 128, 32, 2), (128, 128, 16, 1), (256, 256, 32, 2), (32, 64, 64, 2), (16, 64, 16, 1),
          (32, 128, 20, 2), (48, 64, 9, 1), (64, 96, 40, 2)]


@pytest.mark.parametrize("cin,cout,K,stride", SHAPES)
def test_cloud_transpose_vs_atomic_scatter_and_oracle_f32(gpu, monkeypatch, cin, cout, K, stride):
    from epn_pointcloud_amd import _lib, ops
    b, n = 2, 192
    geo, feats, W, gy, raw = _layer(gpu, b, n, cin, cout, K, stride)
    d = geo.desc(cin, cout)
    assert _lib.get_lib().epn_inter_ungroup_cloud_ok(ctypes.byref(d)) == 1
    calls = []
    real = ops._launch
    monkeypatch.setattr(ops, "_launch", lambda kind, *a: (calls.append(kind), real(kind, *a))[1])
    gs, gws = _grad(ops, "split", geo, feats, W, gy, monkeypatch)
    gc, gwc = _grad(ops, "cloud", geo, feats, W, gy, monkeypatch)
    assert "inter_ungroup" in calls and "inter_gemm_dg" in calls
    scale = gs.abs().max().item()
    assert scale > 0
    # same dG (same GEMM), same per-contribution arithmetic: the two differ by the rounding of fp32 atomics in arrival order
    # against one rounding of an exact integer sum
    assert (gc - gs).abs().max().item() <= 2e-6 * scale, ((gc - gs).abs().max().item(), scale)
    assert torch.equal(gwc, gws)
    g_ref = _oracle_grad(raw, feats, W, gy)
    assert (gc.cpu() - g_ref).abs().max().item() <= 1e-3 * g_ref.abs().max().item()
    # bitwise repeatable (integer accumulation does not depend on the order in which waves arrive)
    gc2, _ = _grad(ops, "cloud", geo, feats, W, gy, monkeypatch)
    assert torch.equal(gc, gc2)
    # the shared-input form: the other branch's gradient is folded into the write-out
    side = torch.randn_like(feats) * scale
    a_c, _ = _grad(ops, "cloud", geo, feats, W, gy, monkeypatch, side)
    assert (a_c - gc - side).abs().max().item() <= 1e-6 * (scale + side.abs().max().item())


@pytest.mark.parametrize("cin,cout,K,stride", [(32, 32, 32, 1), (32, 64, 64, 2), (64, 64, 32, 1), (128, 128, 64, 2), (128, 128, 16, 1)])
def test_cloud_transpose_bf16(gpu, monkeypatch, cin, cout, K, stride):
    from epn_pointcloud_amd import ops
    b, n = 2, 192
    geo, feats, W, gy, raw = _layer(gpu, b, n, cin, cout, K, stride, dtype=torch.bfloat16)
    gs, _ = _grad(ops, "split", geo, feats, W, gy, monkeypatch)
    gc, _ = _grad(ops, "cloud", geo, feats, W, gy, monkeypatch)
    assert gc.dtype == torch.bfloat16 and gs.dtype == torch.bfloat16
    scale = gs.float().abs().max().item()
    # both round the same sums to bf16 once (the atomic scatter accumulates in fp32 and converts): one bf16 ulp of the largest value
    assert (gc.float() - gs.float()).abs().max().item() <= 2 ** -7 * scale
    g_ref = _oracle_grad(raw, feats, W, gy)
    assert (gc.float().cpu() - g_ref).abs().max().item() <= 3e-2 * g_ref.abs().max().item()      # bf16 operands: as the split form's bound
    gc2, _ = _grad(ops, "cloud", geo, feats, W, gy, monkeypatch)
    assert torch.equal(gc, gc2)
    side = (torch.randn_like(feats.float()) * scale).to(torch.bfloat16)
    a_c, _ = _grad(ops, "cloud", geo, feats, W, gy, monkeypatch, side)
    a_s, _ = _grad(ops, "split", geo, feats, W, gy, monkeypatch, side)
    assert (a_c.float() - a_s.float()).abs().max().item() <= 2 ** -6 * a_s.float().abs().max().item()


def test_auto_takes_the_cloud_form_where_it_is_faster(gpu, monkeypatch):
    from epn_pointcloud_amd import ops
    took = []
    real = ops._ungroup_cloud_takes
    monkeypatch.setattr(ops, "_ungroup_cloud_takes", lambda *a: (took.append(real(*a)), took[-1])[1])
    for (cin, cout, K, dtype, want) in [(64, 64, 16, torch.float32, True), (64, 64, 32, torch.bfloat16, True),
                                         (32, 64, 64, torch.bfloat16, True), (32, 64, 64, torch.float32, False)]:
        geo, feats, W, gy, _ = _layer(gpu, 2, 128, cin, cout, K, 1, dtype=dtype)
        del took[:]
        _grad(ops, "auto", geo, feats, W, gy, monkeypatch)
        assert took == [want], (cin, cout, K, dtype, took)
    # deterministic mode: the cloud form is bitwise repeatable by construction and replaces the slab-based kernels where it applies
    monkeypatch.setenv("EPN_DETERMINISTIC", "1")
    geo, feats, W, gy, _ = _layer(gpu, 2, 128, 32, 64, 64, 1)
    del took[:]
    g1, _ = _grad(ops, "auto", geo, feats, W, gy, monkeypatch)
    assert took == [True]
    g2, _ = _grad(ops, "auto", geo, feats, W, gy, monkeypatch)
    assert torch.equal(g1, g2)


def _call(lib, d, dG, amax, out, add, ws):
    from epn_pointcloud_amd import _lib, ops
    args = [ctypes.byref(d), ctypes.c_void_p(dG.data_ptr()), None if amax is None else ctypes.c_void_p(amax.data_ptr()),
            ops._cl_ptr(out), None if add is None else ops._cl_ptr(add)]
    if dG.dtype == torch.bfloat16:
        args.append(1 if out.dtype == torch.float32 else 0)
        return lib.epn_inter_ungroup_cloud_bf16(*args, ctypes.c_void_p(ws.data_ptr()), ws.numel(), _lib.stream_of(dG))
    return lib.epn_inter_ungroup_cloud_f32(*args, ctypes.c_void_p(ws.data_ptr()), ws.numel(), _lib.stream_of(dG))


def test_cloud_entry_arguments_range_contract_and_edge_cases(gpu):
    from epn_pointcloud_amd import _lib, gemm, ops
    lib = _lib.get_lib()
    b, n, cin, cout, K = 2, 160, 32, 64, 32
    geo, feats, W, gy, _ = _layer(gpu, b, n, cin, cout, K, 1)
    d = geo.desc(cin, cout)
    cols = b * n * 60
    dG = torch.randn(cols, cin * 24, device=gpu) * 1e-2
    ws = torch.empty(int(lib.epn_inter_ungroup_cloud_workspace_bytes(ctypes.byref(d))), dtype=torch.uint8, device=gpu)
    out = ops.empty_cl(b, cin, n, 60, gpu)
    amax = dG.abs().max().reshape(1)
    assert _call(lib, d, dG, amax, out, None, ws) == 0
    ref = out.clone()
    # no maximum supplied: the entry takes it itself -- same result, bit for bit
    out.fill_(7.0)
    assert _call(lib, d, dG, None, out, None, ws) == 0
    assert torch.equal(out, ref)
    # a larger (power-of-two) reported maximum only coarsens the unit: same result to fp32 rounding
    assert _call(lib, d, dG, amax * 64, out, None, ws) == 0
    assert (out - ref).abs().max().item() <= 1e-6 * ref.abs().max().item()
    # add may be the output itself (accumulate in place)
    acc = torch.randn_like(ref).contiguous(memory_format=torch.channels_last)
    start = acc.clone()
    assert _call(lib, d, dG, amax, acc, acc, ws) == 0
    assert (acc - start - ref).abs().max().item() <= 1e-6 * (ref.abs().max().item() + start.abs().max().item())
    # all-zero gradient (maximum 0): zeros, no range event
    z = torch.zeros_like(dG)
    assert _call(lib, d, z, z.abs().max().reshape(1), out, None, ws) == 0
    assert out.abs().max().item() == 0.0
    assert gemm.fixed_point_range_count(reset=True) == 0
    # bf16 gradient in, fp32 out (the contract of epn_inter_ungroup_bf16: what the composed split backward uses)
    dGb = dG.to(torch.bfloat16)
    outb = ops.empty_cl(b, cin, n, 60, gpu, torch.bfloat16)
    assert _call(lib, d, dGb, amax, outb, None, ws) == 0 and _call(lib, d, dGb, amax, out, None, ws) == 0
    assert (out.to(torch.bfloat16).float() - outb.float()).abs().max().item() <= 2 ** -7 * ref.abs().max().item()
    assert (out - ref).abs().max().item() <= 2 ** -6 * ref.abs().max().item()
    # argument checks
    assert _call(lib, d, dG, amax, out, None, ws[:64]) == -2                 # EPN_EWORKSPACE
    d2 = geo.desc(40, cout)                                                  # cin % 16 != 0: not this kernel's
    assert lib.epn_inter_ungroup_cloud_ok(ctypes.byref(d2)) == 0 and lib.epn_inter_ungroup_cloud_workspace_bytes(ctypes.byref(d2)) == 0


@pytest.mark.nonfinite_inputs
def test_understated_maximum_is_loud(gpu):
    """A reported max|dG| 2^20 x too small: contributions leave the 2^50-unit range.  The workgroups that see one write NaN to
    their rows (a wrapped integer sum would pass for a gradient) and the sticky counter says how many."""
    from epn_pointcloud_amd import _lib, gemm, ops
    lib = _lib.get_lib()
    b, n, cin, cout, K = 2, 128, 32, 64, 16
    geo, *_ = _layer(gpu, b, n, cin, cout, K, 1)
    d = geo.desc(cin, cout)
    dG = torch.randn(b * n * 60, cin * 24, device=gpu)
    ws = torch.empty(int(lib.epn_inter_ungroup_cloud_workspace_bytes(ctypes.byref(d))), dtype=torch.uint8, device=gpu)
    out = ops.empty_cl(b, cin, n, 60, gpu)
    gemm.fixed_point_range_count(reset=True)
    assert _call(lib, d, dG, dG.abs().max().reshape(1) * 2.0 ** -20, out, None, ws) == 0
    assert gemm.fixed_point_range_count(reset=True) > 0
    assert torch.isnan(out).any()
    # a non-finite gradient is reported the same way
    dG[5, 7] = float("inf")
    assert _call(lib, d, dG, torch.ones(1, device=gpu) * 8.0, out, None, ws) == 0
    assert gemm.fixed_point_range_count(reset=True) > 0 and torch.isnan(out).any()


@pytest.mark.parametrize("mode", ["f16x2", "split", "native", "bf16"])
def test_gemm_epilogue_maximum(gpu, mode):
    """gemm_nt(..., c_amax=True): max|C| from the accumulators of the kernel that writes C -- what the cloud transpose takes as
    dg_amax -- equals the maximum of the stored tensor, on full and ragged tiles and on the generic path."""
    from epn_pointcloud_amd import gemm
    old = gemm.FP32_MODE
    try:
        if mode != "bf16":
            gemm.set_fp32_mode(mode)
        dt = torch.bfloat16 if mode == "bf16" else torch.float32
        for (M, N, K) in [(4096, 1536, 64), (1000, 200, 128), (300, 96, 64), (77, 40, 24)]:
            A = (torch.randn(M, K, device=gpu) * 3e-3).to(dt)
            B = torch.randn(N, K, device=gpu).to(dt)
            C, cm = gemm.gemm_nt(A, B, c_amax=True)
            assert float(cm.item()) == float(C.float().abs().max().item()), (mode, M, N, K)
            assert gemm.amax_tag(C) is cm
    finally:
        gemm.set_fp32_mode(old)
