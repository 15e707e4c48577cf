"""Oracle comparison AT BASELINE configs[1] SIZE (ModelNet40 classification, B=32, N=1024, A=60, full widths): every
convolution of the schedule runs on the GPU on the whole batch of 32 clouds and is compared with the CPU oracle on two
of them (clouds are independent inside a convolution, so a slice of the batch is a complete check of the kernels at
their production launch geometry: grid size, XCD tile remap, split-K factors, 256-row GEMM tiles).  Features and data
gradients within 1e-3 absolute (north_star).  Weight gradients sum over all 32 clouds and are not sliceable: their
contraction kernels are checked at the PRODUCTION row counts (R = b*p2*na of every layer) against fp64 on non-negative
(post-activation-like) operands, where the bf16 MFMA adder's truncation bias does not average out
(test_weight_gradient_gemm_at_production_rows).  Configs 3 and 4 (bf16): every convolution of the rotation / 3DMatch
schedules on the whole B=64 batch against the fp32 oracle on two clouds, bf16-rounded inputs
(test_bf16_inter_conv_full_size_slice_vs_oracle).  Also: the caches of index-derived tables follow the live tensor."""
import math

import numpy as np
import pytest
import torch

from oracle import so3conv_ref as R

pytestmark = pytest.mark.gpu
TOL = 1e-3
PICK = [0, 31]


def _mods(vgtk_alias):
    import vgtk.so3conv as sptk
    import vgtk.spconv as zptk
    return sptk, zptk


@pytest.fixture(scope="module")
def pyramid(gpu, vgtk_alias):
    """xyz of the 32 synthetic clouds at every resolution of the cls schedule (FPS chain, as the network computes it)."""
    from epn_pointcloud_amd import schedule as S
    import vgtk.pc as pctk
    layers = S.cls_so3net_schedule(1024)
    pts = S.synthetic_clouds(32, 1024, gpu, seed=2913)
    xyz = pts.permute(0, 2, 1).contiguous()
    levels = {1024: xyz}
    cur = xyz
    for l in layers:
        if l.stride > 1:
            _, cur = pctk.furthest_sample(cur, math.ceil(cur.shape[2] / l.stride), l.lazy)
            levels[cur.shape[2]] = cur
    return layers, levels


@pytest.mark.parametrize("bwd_data", ["auto", "split"])   # auto: the fixed-point transpose of the grouping (csrc/inter_ungroup_cloud.hip); split: the atomic scatter
@pytest.mark.parametrize("li", [0, 1, 2, 3, 4, 5, 6])     # li = 0: the cin = 1 kernels (inter_c1_*) at B=32, N=1024, K=32
def test_inter_conv_full_size_slice_vs_oracle(gpu, vgtk_alias, pyramid, li, bwd_data, monkeypatch):
    if bwd_data == "split" and li == 0:
        pytest.skip("the cin = 1 layer has no grouped-feature gradient")
    monkeypatch.setenv("EPN_INTER_BWD_DATA", bwd_data)
    sptk, zptk = _mods(vgtk_alias)
    layers, levels = pyramid
    l = layers[li]
    p1 = 1024
    for k in range(li):
        p1 = math.ceil(p1 / layers[k].stride)
    xyz = levels[p1]
    torch.manual_seed(100 + li)
    conv = sptk.InterSO3Conv(l.cin, l.cout, 1, l.stride, l.radius, l.sigma, l.nn, lazy_sample=l.lazy).to(gpu)
    feats = torch.randn(32, l.cin, p1, 60, device=gpu).mul_(0.5).requires_grad_(True)
    iidx, _, sidx, y = conv(zptk.SphericalPointCloud(xyz, feats, None))
    gy = torch.randn_like(y.feats).mul_(0.1)
    (dF,) = torch.autograd.grad(y.feats, [feats], gy)
    # oracle on two clouds of the batch
    xs, fs = xyz[PICK].cpu(), feats.detach()[PICK].cpu().requires_grad_(True)
    o_idx, _, o_sidx, _, oy = R.inter_so3conv(xs, fs, conv.basic_conv.W.detach().cpu(), conv.anchors.cpu(),
                                              conv.kernels.cpu(), l.stride, l.radius, l.sigma, l.nn, l.lazy)
    (odF,) = torch.autograd.grad(oy, [fs], gy[PICK].cpu())
    assert torch.equal(iidx[PICK].cpu(), o_idx)
    if sidx is not None:
        assert torch.equal(sidx[PICK].cpu(), o_sidx)
    assert (y.feats.detach()[PICK].cpu() - oy.detach()).abs().max().item() < TOL
    assert (dF[PICK].cpu() - odF).abs().max().item() < TOL


@pytest.mark.parametrize("mode", ["f16x2", "split", "native"])
@pytest.mark.parametrize("li", [1, 2, 3, 4, 5, 6])
def test_weight_gradient_gemm_at_production_rows(gpu, pyramid, li, mode):
    """dW = dOut^T G of every cls layer at its production contraction length R = b*p2*na (983 040 ... 122 880 rows) on
    NON-NEGATIVE operands (|randn|: the post-activation case, where the truncating adder of the bf16 MFMA gives the split
    form a systematic bias that grows with R instead of averaging out) against fp64: relative L2 error < 1e-4, the mean
    signed error (bias) reported and bounded at 1e-4 of the mean result."""
    from epn_pointcloud_amd import gemm
    layers, _ = pyramid
    l = layers[li]
    p2 = 1024
    for k in range(li + 1):
        p2 = math.ceil(p2 / layers[k].stride)
    R_ = 32 * p2 * 60
    torch.manual_seed(300 + li)
    X = torch.randn(R_, l.cout, device=gpu).abs_()
    Y = torch.randn(R_, l.cin * 24, device=gpu).abs_()
    old = gemm.FP32_MODE
    gemm.set_fp32_mode(mode)
    try:
        C = gemm.gemm_tn(X, Y)
    finally:
        gemm.set_fp32_mode(old)
    ref = torch.zeros(l.cout, l.cin * 24, dtype=torch.float64, device=gpu)
    step = 1 << 16
    for r0 in range(0, R_, step):                                   # fp64 reference in row slabs (bounded memory)
        ref += X[r0:r0 + step].double().t() @ Y[r0:r0 + step].double()
    err = C.double() - ref
    rel = (err.norm() / ref.norm()).item()
    bias = (err.mean() / ref.mean()).item()
    print(f"layer {li} R={R_} {l.cout}x{l.cin * 24} mode={mode}: rel L2 {rel:.2e}, bias {bias:+.2e}")
    assert rel < 1e-4, (rel, bias)
    assert abs(bias) < 1e-4, (rel, bias)


@pytest.mark.parametrize("c,p", [(64, 512), (128, 256), (256, 128), (256, 64)])
def test_intra_conv_full_size_slice_vs_oracle(gpu, vgtk_alias, c, p):
    sptk, zptk = _mods(vgtk_alias)
    torch.manual_seed(c + p)
    conv = sptk.IntraSO3Conv(c, c).to(gpu)
    feats = torch.randn(32, c, p, 60, device=gpu).mul_(0.5).requires_grad_(True)
    y = conv(zptk.SphericalPointCloud(torch.zeros(32, 3, p, device=gpu), feats, None))
    gy = torch.randn_like(y.feats).mul_(0.1)
    (dF,) = torch.autograd.grad(y.feats, [feats], gy)
    fs = feats.detach()[PICK].cpu().requires_grad_(True)
    oy = R.intra_so3conv(fs, conv.basic_conv.W.detach().cpu(), conv.intra_idx.cpu())
    (odF,) = torch.autograd.grad(oy, [fs], gy[PICK].cpu())
    assert (y.feats.detach()[PICK].cpu() - oy.detach()).abs().max().item() < TOL
    assert (dF[PICK].cpu() - odF).abs().max().item() < TOL


def test_index_table_caches_follow_the_live_tensor(gpu, vgtk_alias):
    """ops.inverse_intra_idx / ops.spectral_basis cache per LIVE tensor object: a table that is freed and replaced by
    another one of the same shape at the same address must not be served the old entry (round-1 keyed on data_ptr)."""
    from epn_pointcloud_amd import ops
    sptk, _ = _mods(vgtk_alias)
    base = torch.from_numpy(sptk.get_intra_idx()).int()

    def check(idx, inv):
        a = torch.arange(60, device=gpu).view(60, 1).expand(60, 12)
        assert torch.equal(inv.long().gather(0, idx.long()), a)     # inv[idx[a,k], k] == a

    idx_a = base.to(gpu)
    inv_a = ops.inverse_intra_idx(idx_a)
    check(idx_a, inv_a)
    assert ops.inverse_intra_idx(idx_a) is inv_a                     # cached while alive and unmodified
    basis_a = ops.spectral_basis(idx_a)
    ptr = idx_a.data_ptr()
    del idx_a
    perm = torch.randperm(60, generator=torch.Generator().manual_seed(1))
    other = perm[base.long()][torch.argsort(perm)].int()             # the same group action with relabelled anchors
    idx_b = other.to(gpu)                                            # the caching allocator hands back the freed block
    inv_b = ops.inverse_intra_idx(idx_b)
    check(idx_b, inv_b)
    assert not torch.equal(inv_b, inv_a) or idx_b.data_ptr() != ptr
    basis_b = ops.spectral_basis(idx_b)
    assert basis_b is not basis_a
    # in-place modification bumps the version: the entry is rebuilt
    idx_b.copy_(base.to(gpu))
    inv_c = ops.inverse_intra_idx(idx_b)
    assert torch.equal(inv_c, inv_a)


# ------------------------------------------------------------------------------------------------ configs 3 and 4 (bf16)
BF16_TOL = 8e-3       # of the tensor's largest magnitude (measured at production geometry: 2.9e-3 .. 4.1e-3; tests/test_gpu_bf16.py derives the bound)


def _r16(t):
    return t.to(torch.bfloat16).float()


@pytest.fixture(scope="module", params=["reg", "inv"])
def pyramid_bf16(request, gpu, vgtk_alias):
    """Layer schedule + xyz at every resolution of BASELINE configs[2] (rotation estimation, 64 clouds of 1024 points) /
    configs[3] (3DMatch descriptor, 64 patches of 2048 points, radius 0.4)."""
    from epn_pointcloud_amd import schedule as S
    import vgtk.pc as pctk
    if request.param == "reg":
        layers, n, scale = S.reg_so3net_schedule(1024), 1024, 1.0
    else:
        layers, n, scale = S.inv_so3net_schedule(2048), 2048, 0.4
    pts = S.synthetic_clouds(64, n, gpu, seed=2913, scale=scale)
    cur = pts.permute(0, 2, 1).contiguous()
    levels = [cur]
    for l in layers:
        if l.stride > 1:
            _, cur = pctk.furthest_sample(cur, math.ceil(cur.shape[2] / l.stride), l.lazy)
        levels.append(cur)
    return request.param, layers, levels


@pytest.mark.parametrize("li", range(8))
def test_bf16_inter_conv_full_size_slice_vs_oracle(gpu, vgtk_alias, pyramid_bf16, li):
    """Every InterSO3Conv of the rotation / 3DMatch schedules at its production geometry (B=64, K = 32 / 64 / 128, the
    4-wave kernels of the K > 32 layers, the scatter's point groups) in bf16, against the fp32 oracle fed the same
    bf16-rounded inputs on two clouds of the batch: output and data gradient within BF16_TOL of the tensor's largest
    magnitude, indices bit-exact."""
    sptk, zptk = _mods(vgtk_alias)
    name, layers, levels = pyramid_bf16
    if li >= len(layers):
        pytest.skip(f"{name} schedule has {len(layers)} layers")
    l = layers[li]
    xyz = levels[li]
    p1 = xyz.shape[2]
    torch.manual_seed(500 + li)
    conv = sptk.InterSO3Conv(l.cin, l.cout, 1, l.stride, l.radius, l.sigma, l.nn, lazy_sample=l.lazy).to(gpu)
    conv.basic_conv.W.data = _r16(conv.basic_conv.W.data)
    conv.feat_dtype = torch.bfloat16
    f32 = _r16(torch.randn(64, l.cin, p1, 60, device=gpu).mul_(0.5))
    feats = (f32 if l.cin == 1 else f32.bfloat16()).requires_grad_(True)     # the cin = 1 layer takes fp32 occupancy features
    iidx, _, sidx, y = conv(zptk.SphericalPointCloud(xyz, feats, None))
    assert y.feats.dtype == torch.bfloat16
    gy = _r16(torch.randn(y.feats.shape, device=gpu).mul_(0.1))
    (dF,) = torch.autograd.grad(y.feats, [feats], gy.to(y.feats.dtype))
    xs, fs = xyz[PICK].cpu(), f32[PICK].cpu().requires_grad_(True)
    o_idx, _, o_sidx, _, oy = R.inter_so3conv(xs, fs, conv.basic_conv.W.detach().cpu(), conv.anchors.cpu(),
                                              conv.kernels.cpu(), l.stride, l.radius, l.sigma, l.nn, l.lazy)
    (odF,) = torch.autograd.grad(oy, [fs], gy[PICK].cpu())
    assert torch.equal(iidx[PICK].cpu(), o_idx)
    if sidx is not None:
        assert torch.equal(sidx[PICK].cpu(), o_sidx)
    ey = (y.feats.detach()[PICK].float().cpu() - oy.detach()).abs().max().item() / oy.detach().abs().max().item()
    ed = (dF[PICK].float().cpu() - odF).abs().max().item() / odF.abs().max().item()
    print(f"{name} layer {li} {l.cin}->{l.cout} K={l.nn}: out {ey:.2e}, dF {ed:.2e} of the largest magnitude")
    assert ey < BF16_TOL and ed < BF16_TOL, (ey, ed)


@pytest.mark.parametrize("c,p", [(64, 256), (128, 128), (128, 64)])
def test_bf16_intra_conv_full_size_slice_vs_oracle(gpu, vgtk_alias, c, p):
    """IntraSO3Conv widths of the rotation / 3DMatch schedules that take the block-diagonal form (c % 64 == 0) on the whole
    B=64 bf16 batch; the 32-channel layers (split form) are covered at b <= 2 by tests/test_gpu_bf16.py."""
    sptk, zptk = _mods(vgtk_alias)
    torch.manual_seed(c + p)
    conv = sptk.IntraSO3Conv(c, c).to(gpu)
    conv.basic_conv.W.data = _r16(conv.basic_conv.W.data)
    f32 = _r16(torch.randn(64, c, p, 60, device=gpu).mul_(0.5))
    feats = f32.bfloat16().requires_grad_(True)
    y = conv(zptk.SphericalPointCloud(torch.zeros(64, 3, p, device=gpu), feats, None))
    gy = _r16(torch.randn(y.feats.shape, device=gpu).mul_(0.1))
    (dF,) = torch.autograd.grad(y.feats, [feats], gy.to(y.feats.dtype))
    fs = f32[PICK].cpu().requires_grad_(True)
    oy = R.intra_so3conv(fs, conv.basic_conv.W.detach().cpu(), conv.intra_idx.cpu())
    (odF,) = torch.autograd.grad(oy, [fs], gy[PICK].cpu())
    ey = (y.feats.detach()[PICK].float().cpu() - oy.detach()).abs().max().item() / oy.detach().abs().max().item()
    ed = (dF[PICK].float().cpu() - odF).abs().max().item() / odF.abs().max().item()
    assert ey < BF16_TOL and ed < BF16_TOL, (ey, ed)


# ------------------------------------------------------------------------------------------------ on-chip form
@pytest.mark.parametrize("li", [1, 2, 3, 4, 5, 6])
def test_onchip_inter_conv_full_size_slice_vs_oracle(gpu, vgtk_alias, pyramid, li):
    """epn_inter_so3conv_fwd_onchip_f32 (csrc/inter_fx.hip: no [cols, cin*ks] tensor) on every cls layer at B=32 against
    the CPU oracle on two clouds: 1e-3 absolute, like the split form."""
    from epn_pointcloud_amd import ops
    import vgtk.pc as pctk
    sptk, zptk = _mods(vgtk_alias)
    layers, levels = pyramid
    l = layers[li]
    p1 = 1024
    for k in range(li):
        p1 = math.ceil(p1 / layers[k].stride)
    xyz = levels[p1]
    p2 = math.ceil(p1 / l.stride)
    torch.manual_seed(100 + li)
    conv = sptk.InterSO3Conv(l.cin, l.cout, 1, l.stride, l.radius, l.sigma, l.nn, lazy_sample=l.lazy).to(gpu)
    feats = ops.to_cl(torch.randn(32, l.cin, p1, 60, device=gpu).mul_(0.5))
    _, new_xyz = pctk.furthest_sample(xyz, p2, l.lazy)
    idx = pctk.ball_query_index(new_xyz, xyz, l.radius, l.nn)
    geo = ops.InterGeometry(xyz, new_xyz, idx, conv.anchors, conv.kernels, conv.sigma)
    W = conv.basic_conv.W.detach().contiguous()
    assert ops.inter_onchip_ok(feats, W, geo)
    y = ops.inter_onchip_fwd(feats, W, geo)
    _, _, _, _, oy = R.inter_so3conv(xyz[PICK].cpu(), feats[PICK].cpu(), W.cpu(), conv.anchors.cpu(), conv.kernels.cpu(),
                                     l.stride, l.radius, l.sigma, l.nn, l.lazy)
    assert (y[PICK].cpu() - oy).abs().max().item() < TOL


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_pointnet_head_at_production_size(gpu, monkeypatch, dt):
    """The aggregation head of the classification / rotation networks at its production size (32 clouds x 64 points x 60
    anchors, 256 -> 256 channels: 122 880 feature rows).  The GEMM-composed form (what the benchmark runs) against the
    oracle on two clouds (fp32: 1e-3, north_star) and against the fused fp32 kernels on the whole batch: outputs within
    fp32 / bf16 rounding; gradients in rel-L2 -- the two forms sum in different orders, so a few of the 491 520 maxima
    over 64 points flip between near-tied points and move whole gradient rows (3.5e-3 rel-L2 measured in fp32)."""
    from epn_pointcloud_amd import ops
    b, c, co, p, na = 32, 256, 256, 64, 60
    g = torch.Generator().manual_seed(5)
    xyz = (torch.rand(b, 3, p, generator=g) - 0.5)
    f = torch.randn(b, c, p, na, generator=g).to(dt).float()
    anchors = torch.linalg.qr(torch.randn(na, 3, 3, generator=g))[0].contiguous()
    w = torch.randn(co, c + 3, 1, 1, generator=g) / (c ** 0.5)
    if dt == torch.bfloat16:
        w[:, :c] = w[:, :c].bfloat16().float()
    bias, gy = torch.randn(co, generator=g), torch.randn(b, co, na, generator=g)

    def run(form, dtype):
        monkeypatch.setenv("EPN_POINTNET", form)
        fg = f.to(gpu).to(dtype).requires_grad_(True)
        wg, bg = w.to(gpu).requires_grad_(True), bias.to(gpu).requires_grad_(True)
        y = ops.pointnet_so3conv(fg, xyz.to(gpu), anchors.to(gpu), wg, bg)
        return [y.detach().float().cpu()] + [t.float().cpu() for t in torch.autograd.grad(y, [fg, wg, bg], gy.to(gpu))]

    got = run("gemm", dt)
    ref = run("fused", torch.float32)                # the fused kernels take fp32 features (the same bf16-rounded values)
    tol = 1e-3 if dt == torch.float32 else 2e-3
    assert (got[0] - ref[0]).abs().max().item() < tol
    for u, v, n in zip(got[1:], ref[1:], ("dF", "dW", "dB")):
        assert ((u - v).norm() / v.norm()).item() < (1e-2 if dt == torch.float32 else 3e-2), n
    # oracle on two clouds (fp32 forward value)
    yo = R.pointnet_so3conv(xyz[PICK], f[PICK], anchors, w, bias)
    assert (got[0][PICK] - yo).abs().max().item() < tol
