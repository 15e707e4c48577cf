"""Oracle comparison AT BASELINE configs[1] SIZE (ModelNet40 classification, B=32, N=1024, A=60, full widths): every
convolution of the schedule runs on the GPU on the whole batch of 32 clouds and is compared with the CPU oracle on two
of them (clouds are independent inside a convolution, so a slice of the batch is a complete check of the kernels at
their production launch geometry: grid size, XCD tile remap, split-K factors, 256-row GEMM tiles).  Features and data
gradients within 1e-3 absolute (north_star), weight gradients (which sum over all 32 clouds) are not sliceable and are
covered by the small-size oracle tests.  Also: the caches of index-derived tables follow the live tensor."""
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


@pytest.mark.parametrize("li", [1, 2, 3, 4, 5, 6])
def test_inter_conv_full_size_slice_vs_oracle(gpu, vgtk_alias, pyramid, li):
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
