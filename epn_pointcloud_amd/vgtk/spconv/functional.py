"""Grouping helpers shared by the SO(3) path (vgtk/vgtk/spconv/functional.py).  The fused modules in
vgtk.so3conv never materialise what these return; they exist so code written against the reference's
functional API keeps working, and they too run on the HIP library (no CPU path)."""
import math

import torch

from .. import pc as pctk
from ... import ops

__all__ = ["add_shadow_point", "add_shadow_feature", "ball_query", "batched_index_select",
           "inter_zpconv_grouping_naive", "inter_pooling_naive", "inter_blurring_naive",
           "inter_zpconv_grouping_ball"]


def add_shadow_point(x):
    """[b,c,n] -> [b,c,n+1], extra column 1e4   (spconv/functional.py:83-87)"""
    b, c, _ = x.shape
    return torch.cat((x, torch.full((b, c, 1), 1e4, dtype=torch.float32, device=x.device)), dim=2).contiguous()


def add_shadow_feature(x):
    """[b,c,n,a] -> [b,c,n+1,a], extra zero row   (spconv/functional.py:91-95)"""
    b, c, _, a = x.shape
    return torch.cat((x, torch.zeros(b, c, 1, a, dtype=torch.float32, device=x.device)), dim=2).contiguous()


def ball_query(query_points, support_points, radius, n_sample, support_feats=None):
    """spconv/functional.py:340-349 -> (idx [b,m,k] int32, grouped xyz [b,3,m,k](, grouped feats))."""
    idx = pctk.ball_query_index(query_points, support_points, radius, n_sample)
    support_points = add_shadow_point(support_points)
    if support_feats is None:
        return idx, pctk.group_nd(support_points, idx)
    return idx, pctk.group_nd(support_points, idx), pctk.group_nd(support_feats, idx)


def batched_index_select(input, dim, index):
    """spconv/functional.py:361-369 (used by SPConvNets for the strided skip connection)."""
    for ii in range(1, len(input.shape)):
        if ii != dim:
            index = index.unsqueeze(ii)
    expanse = list(input.shape)
    expanse[0] = -1
    expanse[dim] = -1
    return torch.gather(input, dim, index.expand(expanse))


def inter_zpconv_grouping_naive(inter_idx, inter_w, feats):
    """spconv/functional.py:372-390 -> G[b,c,ks,p,a] = sum_n feats[b,c,idx[b,p,n],a] w[b,p,a,k,n].
    `feats` carries the shadow row (index n_points) as in the reference.  Runs the HIP inter kernel with
    an identity BasicSO3Conv weight, i.e. the same device code path as the fused module."""
    b, p, nn = inter_idx.shape
    _, c, q, a = feats.shape
    w = inter_w.dense() if isinstance(inter_w, ops.InterGeometry) else inter_w
    ks = w.shape[3]
    handle = ops.DenseInterWeights(inter_idx.int().contiguous(), w, q)
    eye = torch.eye(c * ks, dtype=torch.float32, device=feats.device)
    out = ops.inter_so3conv(feats, eye, handle)              # [b, c*ks, p, a]
    return out.view(b, c, ks, p, a).contiguous()


def inter_pooling_naive(inter_idx, sample_idx, feats, alpha=0.5):
    """spconv/functional.py:393-399 (pooling='stride'; unused by the shipped models)."""
    b, p, pnn = inter_idx.shape
    a = feats.shape[3]
    new_feats = batched_index_select(feats, 2, sample_idx.long())
    grouped = batched_index_select(add_shadow_feature(feats), 2, inter_idx.long().view(b, -1)).view(b, -1, p, pnn, a)
    return alpha * new_feats + (1 - alpha) * grouped.mean(3)


def inter_blurring_naive(inter_idx, feats, alpha=0.5):
    """spconv/functional.py:402-407 (pooling='no-stride'; unused by the shipped models)."""
    b, p, pnn = inter_idx.shape
    _, c, q, a = feats.shape
    assert p == q
    grouped = batched_index_select(add_shadow_feature(feats), 2, inter_idx.long().view(b, -1)).view(b, -1, p, pnn, a)
    return alpha * feats + (1 - alpha) * grouped.mean(3)


def inter_zpconv_grouping_ball(xyz, stride, radius, n_neighbor, lazy_sample=True):
    """spconv/functional.py:412-421 -> (grouped_xyz [b,3,p2,nn], ball_idx, sample_idx, sample_xyz)."""
    n_sample = math.ceil(xyz.shape[2] / stride)
    idx, sample_xyz = pctk.furthest_sample(xyz, n_sample, lazy_sample)
    ball_idx, grouped_xyz = ball_query(sample_xyz, xyz, radius, n_neighbor)
    grouped_xyz = grouped_xyz - sample_xyz.unsqueeze(3)
    return grouped_xyz, ball_idx, idx, sample_xyz
