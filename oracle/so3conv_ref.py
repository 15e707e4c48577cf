"""oracle/so3conv_ref.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Pure-torch CPU restatement of the reference's SO(3) separable convolution *as
written* (materialised gather -> einsum -> matmul), used as

* the parity oracle for the fused HIP kernels (tests/, smoke()), and
* bench.py's ``cpu_baseline`` leg (kind "port").

Pinned against the imported reference through tests/golden/*.npz (generated
by tests/golden/gen_golden.py, which imports /root/reference in the build
container; see tests/test_oracle_golden.py).  Index kernels come from
oracle/index_ref.py (C restatement; parity vs CUDA unpinned).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.  Every function cites the reference lines it follows.
"""
import math

import torch
import torch.nn.functional as F

from . import index_ref


# ---------------------------------------------------------------- grouping helpers
def group_nd(pc, idx):
    """vgtk/vgtk/pc/sample.py:46-50 -> vgtk/vgtk/utils.py:25-27: gather along the last axis with a
    flattened [b, m1*m2..] index, result reshaped to [b, c, m1, m2, ...]; output forced to float32
    (gathering_cuda.cpp:38-39)."""
    b = idx.shape[0]
    flat = index_ref.gather_points_forward(pc, idx.reshape(b, -1).int())
    return flat.view(b, -1, *idx.shape[1:])


def furthest_sample(xyz, n_sample, lazy_sample=True):
    """vgtk/vgtk/pc/sample.py:63-77: arange when lazy or nothing to drop, else FPS."""
    if xyz.shape[2] == n_sample or lazy_sample:
        idx = torch.arange(n_sample).view(1, -1).expand(xyz.shape[0], -1).int().contiguous()
    else:
        idx = index_ref.furthest_point_sampling(xyz, n_sample)
    return idx, group_nd(xyz, idx)


def add_shadow_point(x):
    """vgtk/vgtk/spconv/functional.py:83-87: append one column of 1e4."""
    b, c, _ = x.shape
    return torch.cat((x, torch.full((b, c, 1), 1e4, dtype=torch.float32)), dim=2).contiguous()


def add_shadow_feature(x):
    """vgtk/vgtk/spconv/functional.py:91-95: append one zero row at index P."""
    b, c, _, a = x.shape
    return torch.cat((x, torch.zeros(b, c, 1, a, dtype=torch.float32)), dim=2).contiguous()


def ball_query(query, support, radius, n_sample):
    """vgtk/vgtk/spconv/functional.py:340-349: indices + grouped support coords."""
    idx = index_ref.ball_query(query, support, radius, n_sample)
    return idx, group_nd(add_shadow_point(support), idx)


def inter_grouping_ball(xyz, stride, radius, n_neighbor, lazy_sample=True):
    """vgtk/vgtk/spconv/functional.py:412-421."""
    n_sample = math.ceil(xyz.shape[2] / stride)
    sample_idx, new_xyz = furthest_sample(xyz, n_sample, lazy_sample)
    ball_idx, grouped = ball_query(new_xyz, xyz, radius, n_neighbor)
    grouped = grouped - new_xyz.unsqueeze(3)
    return grouped, ball_idx, sample_idx, new_xyz


def batched_index_select(x, dim, index):
    """vgtk/vgtk/spconv/functional.py:361-369: torch.gather with the index broadcast over the
    remaining axes."""
    shape = list(x.shape)
    view = [1] * x.dim()
    view[0] = index.shape[0]
    view[dim] = index.shape[1]
    shape[0] = -1
    shape[dim] = -1
    return torch.gather(x, dim, index.view(view).expand(shape))


# ---------------------------------------------------------------- inter conv
def inter_weights(grouped_xyz, anchors, kernels, sigma):
    """vgtk/vgtk/so3conv/functional.py:180-218: w[b,p,a,k,n] = relu(1 - |g[b,:,p,n] - R_a kappa_k|^2 / sigma)."""
    rk = torch.matmul(anchors, kernels.t()).permute(1, 0, 2).contiguous()      # [3, A, ks]
    diff = grouped_xyz[..., None, None, :] - rk[None, :, None, :, :, None]     # [b,3,p,A,ks,n]
    d2 = torch.sum(diff ** 2, dim=1)
    return F.relu(1.0 - d2 / sigma)


def inter_feat_grouping(inter_idx, inter_w, feats_shadow):
    """vgtk/vgtk/spconv/functional.py:372-390: G[b,c,k,p,a] = sum_n F[b,c,idx[b,p,n],a] w[b,p,a,k,n]."""
    b, p, nn = inter_idx.shape
    a = feats_shadow.shape[3]
    g = batched_index_select(feats_shadow, 2, inter_idx.long().view(b, -1)).view(b, -1, p, nn, a)
    return torch.einsum('bcpna,bpakn->bckpa', g, inter_w).contiguous()


def inter_grouping(xyz, feats, stride, n_neighbor, anchors, kernels, radius, sigma,
                   inter_idx=None, inter_w=None, lazy_sample=True):
    """vgtk/vgtk/so3conv/functional.py:118-178 with pooling=None (the only mode the shipped models use)."""
    if inter_idx is None:
        grouped, inter_idx, sample_idx, new_xyz = inter_grouping_ball(xyz, stride, radius, n_neighbor, lazy_sample)
        inter_w = inter_weights(grouped, anchors, kernels, sigma)
    else:
        sample_idx, new_xyz = None, xyz
    new_feats = inter_feat_grouping(inter_idx, inter_w, add_shadow_feature(feats))
    return inter_idx, inter_w, new_xyz, new_feats, sample_idx


def inter_pooling_naive(inter_idx, sample_idx, feats, alpha=0.5):
    """vgtk/vgtk/spconv/functional.py:393-399: alpha * feats[sampled] + (1 - alpha) * mean over the ball."""
    b, p, nn = inter_idx.shape
    a = feats.shape[3]
    new_feats = batched_index_select(feats, 2, sample_idx.long())
    g = batched_index_select(add_shadow_feature(feats), 2, inter_idx.long().view(b, -1)).view(b, -1, p, nn, a)
    return alpha * new_feats + (1 - alpha) * g.mean(3)


def inter_blurring_naive(inter_idx, feats, alpha=0.5):
    """vgtk/vgtk/spconv/functional.py:402-407: the same blend without sub-sampling."""
    b, p, nn = inter_idx.shape
    a = feats.shape[3]
    g = batched_index_select(add_shadow_feature(feats), 2, inter_idx.long().view(b, -1)).view(b, -1, p, nn, a)
    return alpha * feats + (1 - alpha) * g.mean(3)


def inter_blurring(xyz, feats, n_neighbor, radius, stride, inter_idx=None, lazy_sample=True):
    """vgtk/vgtk/so3conv/functional.py:108-116 (inter_so3conv_blurring)."""
    if inter_idx is None:
        _, inter_idx, sample_idx, sample_xyz = inter_grouping_ball(xyz, stride, radius, n_neighbor, lazy_sample)
    if stride == 1:
        return inter_blurring_naive(inter_idx, feats), xyz
    return inter_pooling_naive(inter_idx, sample_idx, feats), sample_xyz


def basic_conv(W, g):
    """vgtk/vgtk/so3conv/modules.py:48-55: W[Cout, Cin*ks] @ G.view(B, Cin*ks, P*A), no bias."""
    b, _, _, p, a = g.shape
    return torch.matmul(W, g.reshape(b, W.shape[1], p * a)).view(b, W.shape[0], p, a)


def inter_so3conv(xyz, feats, W, anchors, kernels, stride, radius, sigma, n_neighbor,
                  lazy_sample=True, inter_idx=None, inter_w=None):
    """InterSO3Conv.forward, vgtk/vgtk/so3conv/modules.py:157-174.
    Returns (inter_idx, inter_w, sample_idx, new_xyz, out_feats)."""
    inter_idx, inter_w, new_xyz, g, sample_idx = inter_grouping(
        xyz, feats, stride, n_neighbor, anchors, kernels, radius, sigma, inter_idx, inter_w, lazy_sample)
    return inter_idx, inter_w, sample_idx, new_xyz, basic_conv(W, g)


# ---------------------------------------------------------------- intra conv
def intra_grouping(intra_idx, feats):
    """vgtk/vgtk/so3conv/functional.py:221-233: G[b,c,k,p,a] = F[b,c,p,intra_idx[a,k]]."""
    b, c, p, a = feats.shape
    k = intra_idx.shape[1]
    return feats.index_select(3, intra_idx.reshape(-1)).view(b, c, p, a, k).permute(0, 1, 4, 2, 3).contiguous()


def intra_so3conv(feats, W, intra_idx):
    """IntraSO3Conv.forward, vgtk/vgtk/so3conv/modules.py:197-200."""
    return basic_conv(W, intra_grouping(intra_idx, feats))


# ---------------------------------------------------------------- kernels / synthetic data
def scaled_kernel_points(raw_kernel_points, radius, ratio=0.7):
    """vgtk/vgtk/so3conv/functional.py:86-96 with KERNEL_CONDENSE_RATIO (modules.py:16): rescale the
    kpsphere points so the largest norm equals ratio*radius."""
    r = (raw_kernel_points ** 2).sum(1).max().sqrt()
    return (raw_kernel_points * (ratio * radius) / r).float()


def pointnet_so3conv(xyz, feats, anchors, weight, bias):
    """PointnetSO3Conv.forward (vgtk/vgtk/so3conv/modules.py:219-235): centre xyz, rotate into every anchor frame
    (einsum 'aji,bjn->bina'), concatenate behind the features, 1x1 embed, max over the point axis -> [b, co, a]."""
    na = feats.shape[3]
    xyz = xyz - xyz.mean(2, keepdim=True)
    if na == 1:
        ext = xyz[..., None]
    else:
        ext = torch.einsum('aji,bjn->bina', anchors, xyz)
    z = torch.nn.functional.conv2d(torch.cat([feats, ext], 1), weight.reshape(weight.shape[0], -1, 1, 1), bias)
    return z.max(2)[0]


# ---------------------------------------------------------------- legacy ZPConv grouping (vgtk.cuda.zpconv)
def zp_inter_forward(nbr, w, feats):
    """spherical_conv_forward_cuda_kernel, vgtk/vgtk/cuda/zpconv_cuda_kernel.cu:33-72 (wrapper zpconv_cuda.cpp:41-56):
    out[b,c,k,p,a] = sum_ni feats[b,c,nbr[b,p,a,k,ni],a] * w[b,p,a,k,ni].  Differentiable (autograd = the backward
    kernel :75-116)."""
    b, npts, na, ks, ann = nbr.shape
    nq = feats.shape[2]
    valid = ((nbr >= 0) & (nbr < nq)).to(feats.dtype)
    idx = nbr.long().clamp(0, nq - 1)
    f = feats.permute(0, 3, 2, 1)                                                   # [b, a, q, c]
    bi = torch.arange(b)[:, None, None, None, None]
    ai = torch.arange(na)[None, None, :, None, None]
    g = f[bi, ai, idx]                                                              # [b, p, a, k, ni, c]
    out = (g * (w * valid)[..., None]).sum(4)                                       # [b, p, a, k, c]
    return out.permute(0, 4, 3, 1, 2).contiguous()


def zp_intra_forward(nbr, w, feats):
    """intraspherical_conv_forward_cuda_kernel, zpconv_cuda_kernel.cu:119-155 (wrapper zpconv_cuda.cpp:77-93):
    out[b,c,k,p,ao] = sum_ni feats[b,c,p,nbr[ao,ni]] * w[ao,k,ni]."""
    na_in = feats.shape[3]
    valid = ((nbr >= 0) & (nbr < na_in)).to(feats.dtype)
    g = feats[..., nbr.long().clamp(0, na_in - 1)]                                  # [b, c, p, ao, ni]
    return torch.einsum('bcpan,akn->bckpa', g, w * valid[:, None, :]).contiguous()
