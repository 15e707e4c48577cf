"""Grouping helpers shared by the SO(3) path (vgtk/vgtk/spconv/functional.py).  The fused modules in
vgtk.so3conv never materialise what these return; they exist so code written against the reference's
functional API keeps working, and they too run on the HIP library (no CPU path)."""
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
    """spconv/functional.py:361-369: out[b, ..., j, ...] = input[b, ..., index[b, j], ...] along `dim` (used by SPConvNets for
    the strided skip connection).  The [b, c, p, a] feature case is whole (anchor, channel) rows moving: the library's row
    gather (ops.gather_rows: epn_gather_rows forward, the accumulating epn_scatter_rows_add backward); any other rank, axis
    or dtype is torch.gather with the index laid along `dim` and broadcast over the remaining axes."""
    if dim == 2 and input.dim() == 4 and index.dim() == 2 and input.is_cuda and input.dtype in ops.FEATURE_DTYPES:
        return ops.gather_rows(input, index)
    lead = [1] * input.dim()
    lead[0], lead[dim] = index.shape[0], index.shape[1]
    full = [-1 if ax in (0, dim) else n for ax, n in enumerate(input.shape)]
    return torch.gather(input, dim, index.reshape(lead).expand(full))


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


def _neighbour_mean(inter_idx, feats):
    """Mean of every output point's neighbour rows, [b, c, p, a]: the shadow row (index n_points, zeros) stands in for the
    slots the ball query left empty, exactly as the reference's gather-then-mean does -- so the divisor is the slot count."""
    b, p, slots = inter_idx.shape
    padded = add_shadow_feature(feats)
    rows = batched_index_select(padded, 2, inter_idx.reshape(b, p * slots).long())      # [b, c, p * slots, a]
    return rows.reshape(b, feats.shape[1], p, slots, feats.shape[3]).mean(dim=3)


def inter_pooling_naive(inter_idx, sample_idx, feats, alpha=0.5):
    """spconv/functional.py:393-399 (pooling='stride'; unused by the shipped models): the sampled point's own features blended
    with the mean over its ball."""
    own = batched_index_select(feats, 2, sample_idx.long())
    return torch.lerp(_neighbour_mean(inter_idx, feats), own, alpha)


def inter_blurring_naive(inter_idx, feats, alpha=0.5):
    """spconv/functional.py:402-407 (pooling='no-stride'; unused by the shipped models): as above with every point its own
    sample, so the ball query must have been made for all of them."""
    if inter_idx.shape[1] != feats.shape[2]:
        raise AssertionError(f"inter_blurring_naive: {inter_idx.shape[1]} neighbourhoods for {feats.shape[2]} points")
    return torch.lerp(_neighbour_mean(inter_idx, feats), feats, alpha)


def inter_zpconv_grouping_ball(xyz, stride, radius, n_neighbor, lazy_sample=True):
    """spconv/functional.py:412-421 -> (grouped_xyz [b,3,p2,nn] relative to the ball centres, ball_idx, sample_idx, sample_xyz):
    FPS picks ceil(n / stride) centres, the ball query groups the cloud around them."""
    centres_idx, centres = pctk.furthest_sample(xyz, int(-(-xyz.shape[2] // stride)), lazy_sample)
    ball_idx, neighbours = ball_query(centres, xyz, radius, n_neighbor)
    return neighbours - centres[..., None], ball_idx, centres_idx, centres
