"""The composed split-form entry points (include/epn_so3conv.h: epn_inter_so3conv_{fwd,bwd}_split_{f32,bf16}): what
INTEGRATION.md B.2 tells a maintainer to bind -- four calls, caller-owned buffers, no torch types -- must (i) compute what the
Python autograd Function the benchmark times computes, against the oracle, and (ii) reach its speed: the documented calls ARE
the benchmarked kernel chain, not the round-1 fused kernels."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import unit_ball_cloud
from oracle import so3conv_ref as R

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _layer(gpu, b, n, stride, cin, cout, K, radius, sigma, dtype, seed=3):
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
    feats = ops.to_cl(torch.randn(b, cin, n, 60, device=gpu).to(dtype))
    W = torch.randn(cout, cin * 24, device=gpu) / (cin * 24) ** 0.5
    gout = ops.to_cl(torch.randn(b, cout, n // stride, 60, device=gpu).to(dtype))
    return geo, feats, W, gout


class Composed:
    """The documented binding: size queries, two caller-owned buffers, one call per direction."""

    def __init__(self, geo, cin, cout, dtype):
        from epn_pointcloud_amd import _lib
        self.lib, self._lib = _lib.get_lib(), _lib
        self.bf = int(dtype == torch.bfloat16)
        self.d = geo.desc(cin, cout)
        self.dtype, self.dev = dtype, geo.device
        dref = ctypes.byref(self.d)
        assert self.lib.epn_inter_split_ok(dref) == 1
        self.saved = torch.empty(int(self.lib.epn_inter_split_saved_bytes(dref, self.bf)), dtype=torch.uint8, device=self.dev)
        self.ws_f = torch.empty(int(self.lib.epn_inter_split_workspace_bytes(dref, self.bf, 0)), dtype=torch.uint8, device=self.dev)
        self.ws_b = torch.empty(int(self.lib.epn_inter_split_workspace_bytes(dref, self.bf, 1)), dtype=torch.uint8, device=self.dev)
        sfx = "bf16" if self.bf else "f32"
        self.fwd = getattr(self.lib, f"epn_inter_so3conv_fwd_split_{sfx}")
        self.bwd = getattr(self.lib, f"epn_inter_so3conv_bwd_split_{sfx}")

    def forward(self, feats, W, out=None, stats=None):
        d = self.d
        if out is None:
            out = torch.empty((d.b, d.cout, d.p2, d.na), dtype=self.dtype, device=self.dev, memory_format=torch.channels_last)
        self._lib.check(self.fwd(ctypes.byref(d), feats.data_ptr(), W.data_ptr(), out.data_ptr(),
                                 stats.data_ptr() if stats is not None else None, self.saved.data_ptr(), self.saved.numel(),
                                 self.ws_f.data_ptr(), self.ws_f.numel(), self._lib.stream_of(feats)), "fwd_split")
        return out

    def backward(self, gout, W, gf=None, gW=None, accumulate=0, want_f=True, want_w=True):
        d = self.d
        if want_f and gf is None:
            gf = torch.empty((d.b, d.cin, d.p1, d.na), dtype=torch.float32, device=self.dev, memory_format=torch.channels_last)
        if want_w and gW is None:
            gW = torch.empty((d.cout, d.cin * d.ks), dtype=torch.float32, device=self.dev)
        self._lib.check(self.bwd(ctypes.byref(d), gout.data_ptr(), W.data_ptr(), self.saved.data_ptr(), self.saved.numel(),
                                 gf.data_ptr() if want_f else None, accumulate, gW.data_ptr() if want_w else None,
                                 self.ws_b.data_ptr(), self.ws_b.numel(), self._lib.stream_of(gout)), "bwd_split")
        return gf, gW


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("cfg", [(2, 128, 2, 32, 48, 16), (2, 96, 1, 64, 64, 32), (1, 128, 2, 16, 32, 64)],
                         ids=["s2_c32_k16", "s1_c64_k32", "s2_c16_k64"])
def test_composed_split_entry_points_vs_oracle_and_autograd_function(gpu, vgtk_alias, cfg, dt):
    from epn_pointcloud_amd import ops
    dtype = torch.bfloat16 if dt == "bf16" else torch.float32
    b, n, stride, cin, cout, K = cfg
    geo, feats, W, gout = _layer(gpu, b, n, stride, cin, cout, K, 0.45, 0.09, dtype)
    c = Composed(geo, cin, cout, dtype)
    cols = b * (n // stride) * 60
    stats = torch.zeros((cols // 32, cout, 2), device=gpu) if cols % 32 == 0 else None
    out = c.forward(feats, W, stats=stats)
    base = torch.randn(b, cin, n, 60, device=gpu).contiguous(memory_format=torch.channels_last)
    gf, gW = c.backward(gout, W)
    gf_acc, _ = c.backward(gout, W, gf=base.clone(memory_format=torch.channels_last), accumulate=1, want_w=False)
    _, gW_only = c.backward(gout, W, want_f=False)
    torch.cuda.synchronize()

    # (i) the autograd Function the benchmark runs -- same kernels, same order: equal up to the atomics' order
    f = feats.clone(memory_format=torch.channels_last).requires_grad_(True)
    w = W.clone().requires_grad_(True)
    o2 = ops.InterSO3ConvSplitFn.apply(f, w, geo)
    o2.backward(gout)
    assert torch.equal(out, o2.detach())
    scale = gf.abs().max().item() + 1e-12
    assert (gf - f.grad.float()).abs().max().item() <= (2e-2 if dt == "bf16" else 1e-4) * scale
    assert (gW - w.grad).abs().max().item() <= 1e-5 * w.grad.abs().max().item()
    assert torch.equal(gW, gW_only)
    assert (gf_acc - (base + gf)).abs().max().item() <= 1e-4 * scale + (1e-5 if dt == "f32" else 1e-3)
    if stats is not None:          # epilogue statistics: column sums of the stored tensor per 32-row block
        o2d = out.permute(0, 2, 3, 1).reshape(cols, cout).float()
        want = torch.stack([o2d.view(cols // 32, 32, cout).sum(1), (o2d ** 2).view(cols // 32, 32, cout).sum(1)], dim=-1)
        assert torch.allclose(stats, want, rtol=2e-3 if dt == "bf16" else 1e-4, atol=1e-3)

    # (ii) the oracle (materialising restatement of the reference), fed the same (bf16-rounded) operands
    fo = feats.float().cpu().requires_grad_(True)
    wo = (W.to(dtype).float() if dt == "bf16" else W).cpu().requires_grad_(True)
    grouped = R.group_nd(R.add_shadow_point(geo.xyz.cpu()), geo.ball_idx.cpu()) - geo.new_xyz.cpu().unsqueeze(3)
    w_ref = R.inter_weights(grouped, geo.anchors.cpu(), geo.kernels.cpu(), geo.sigma)
    G = R.inter_feat_grouping(geo.ball_idx.cpu(), w_ref, R.add_shadow_feature(fo))
    oo = R.basic_conv(wo, G)
    goo = gout.float().cpu()
    dWo, dFo = torch.autograd.grad(oo, [wo, fo], goo)
    tol = 8e-3 if dt == "bf16" else 1e-3
    assert (out.float().cpu() - oo.detach()).abs().max().item() <= tol * max(1.0, oo.abs().max().item() if dt == "bf16" else 1.0)
    assert (gf.cpu() - dFo).abs().max().item() <= tol * max(1.0, dFo.abs().max().item() if dt == "bf16" else 1.0)
    assert ((gW.cpu() - dWo).norm() / dWo.norm()).item() <= (2e-2 if dt == "bf16" else 1e-3)


def test_composed_entry_points_reject_what_they_do_not_take(gpu, vgtk_alias):
    from epn_pointcloud_amd import _lib
    lib = _lib.get_lib()
    geo, feats, W, gout = _layer(gpu, 1, 64, 1, 16, 16, 16, 0.45, 0.09, torch.float32)
    d = geo.desc(12, 16)                      # cin % 16 != 0: the fused / generic entry points' job
    assert lib.epn_inter_split_ok(ctypes.byref(d)) == 0
    assert lib.epn_inter_split_saved_bytes(ctypes.byref(d), 0) == 0
    d = geo.desc(16, 16)
    buf = torch.empty(1024, dtype=torch.uint8, device=gpu)
    out = torch.empty((1, 16, 64, 60), device=gpu).contiguous(memory_format=torch.channels_last)
    rc = lib.epn_inter_so3conv_fwd_split_f32(ctypes.byref(d), feats.data_ptr(), W.data_ptr(), out.data_ptr(), None,
                                             buf.data_ptr(), buf.numel(), buf.data_ptr(), buf.numel(), _lib.stream_of(out))
    assert rc == -2 or "workspace" in lib.epn_strerror(rc).decode()          # EPN_EWORKSPACE: buffers too small


@pytest.mark.parametrize("dt,cfg", [("f32", (16, 512, 1, 64, 64, 16)), ("bf16", (16, 512, 2, 64, 128, 64))],
                         ids=["cls_64to64_k16_f32", "reg_64to128_k64_bf16"])
def test_documented_calls_reach_the_benchmarked_layer_time(gpu, vgtk_alias, dt, cfg):
    """A production-shaped layer (half the benchmark's batch), forward + backward: the four documented C calls against
    ops.inter_so3conv under autograd (what bench.py times).  >= 90 % of its speed -- in fact the same kernels, minus the
    Python / autograd / allocator work between them."""
    from epn_pointcloud_amd import ops
    dtype = torch.bfloat16 if dt == "bf16" else torch.float32
    b, n, stride, cin, cout, K = cfg
    geo, feats, W, gout = _layer(gpu, b, n, stride, cin, cout, K, 0.2828 if stride == 1 else 0.4, 0.04 if stride == 1 else 0.08,
                                 dtype)
    c = Composed(geo, cin, cout, dtype)
    out = torch.empty((b, cout, n // stride, 60), dtype=dtype, device=gpu).contiguous(memory_format=torch.channels_last)
    gf = torch.empty((b, cin, n, 60), device=gpu).contiguous(memory_format=torch.channels_last)
    gW = torch.empty_like(W)

    def composed():
        c.forward(feats, W, out=out)
        c.backward(gout, W, gf=gf, gW=gW)

    f = feats.clone(memory_format=torch.channels_last).requires_grad_(True)
    w = W.clone().requires_grad_(True)

    def autograd():
        f.grad = w.grad = None
        ops.inter_so3conv(f, w, geo).backward(gout)

    def timed(fn, reps=5):
        fn(); fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    t_auto = min(timed(autograd) for _ in range(3))
    t_comp = min(timed(composed) for _ in range(3))
    print(f"{dt} {cfg}: autograd Function {t_auto:.3f} ms, composed C calls {t_comp:.3f} ms")
    assert t_comp <= t_auto / 0.9, (t_comp, t_auto)
