"""Functional API of the SO(3) convolutions (vgtk/vgtk/so3conv/functional.py).

`inter_so3conv_grouping` / `intra_so3conv_grouping` keep the reference's materialising signatures for
API compatibility; the nn.Modules in .modules call the fused HIP kernels directly instead."""
import math

import numpy as np
import torch

from .. import functional as fr
from .. import spconv as zpconv
from ... import ops

__all__ = ["get_occupancy_features", "get_sphereical_kernel_points_from_ply", "initial_anchor_query",
           "inter_so3conv_blurring", "inter_so3conv_grouping", "inter_so3conv_grouping_anchor",
           "intra_so3conv_grouping", "select_anchor", "get_anchors", "get_intra_idx", "get_canonical_relative",
           "inter_so3conv_feat_grouping", "batched_index_select"]

inter_so3conv_feat_grouping = zpconv.inter_zpconv_grouping_naive   # so3conv/functional.py:21
batched_index_select = zpconv.batched_index_select                 # so3conv/functional.py:22

GAMMA_SIZE = 3
Rs, R_idx = fr.icosahedron_so3_tables()                            # so3conv/functional.py:271-278
canonical_relative = None


def get_occupancy_features(pc, n_anchor, use_center=False):
    """pc [nb,np,3] -> ones [nb,1,np,na]   (so3conv/functional.py:25-44; the normals branch of the
    reference is broken -- `ns.anchors` typo -- and unreachable from the shipped configs)."""
    nb, npts, nd = pc.shape
    if nd == 6:
        raise NotImplementedError("normals input: unreachable/broken in the reference (functional.py:36)")
    features = torch.ones(nb, 1, npts, n_anchor, dtype=torch.float32, device=pc.device)
    if use_center:
        features[:, :, 0, :] = 0.0
    return features


def get_sphereical_kernel_points_from_ply(radius, kernel_size):
    """so3conv/functional.py:86-96: kpsphere{24,30,66} rescaled so the largest norm equals `radius`."""
    assert kernel_size <= 3 and kernel_size > 0
    mapping = {1: 24, 2: 30, 3: 66}
    ply = fr.kernel_points_raw(mapping[kernel_size]).astype('float32')
    r = np.sqrt((ply ** 2).sum(1).max())
    return ply * radius / r


def initial_anchor_query(frag, centers, kernels, r, sigma):
    from ..cuda import grouping as cuda_nn
    return cuda_nn.initial_anchor_query(centers, frag, kernels, r, sigma)


def inter_so3conv_blurring(xyz, feats, n_neighbor, radius, stride, inter_idx=None, lazy_sample=True,
                           radius_expansion=1.0):
    """so3conv/functional.py:108-116 (pooling modes; unused by the shipped models)."""
    if inter_idx is None:
        _, inter_idx, sample_idx, sample_xyz = zpconv.inter_zpconv_grouping_ball(
            xyz, stride, radius * radius_expansion, n_neighbor, lazy_sample)
    if stride == 1:
        return zpconv.inter_blurring_naive(inter_idx, feats), xyz
    return zpconv.inter_pooling_naive(inter_idx, sample_idx, feats), sample_xyz


def inter_so3conv_grouping_anchor(grouped_xyz, anchors, kernels, sigma, interpolate='linear'):
    """grouped_xyz [b,3,p2,nn] -> w [b,p2,na,ks,nn] = relu(1 - |g - R_a kappa_k|^2 / sigma)
    (so3conv/functional.py:180-218), materialised by the HIP kernel."""
    if interpolate != 'linear':
        raise NotImplementedError("kernel function %s is not implemented!" % interpolate)
    b, _, p2, nn = grouped_xyz.shape
    dev = grouped_xyz.device
    flat = grouped_xyz.reshape(b, 3, p2 * nn).contiguous()
    idx = torch.arange(p2 * nn, dtype=torch.int32, device=dev).view(1, p2, nn).expand(b, -1, -1).contiguous()
    zero = torch.zeros(b, 3, p2, dtype=torch.float32, device=dev)
    geo = ops.InterGeometry(flat, zero, idx, anchors.contiguous().float(), kernels.contiguous().float(), sigma)
    return geo.dense()


def inter_so3conv_grouping(xyz, feats, stride, n_neighbor, anchors, kernels, radius, sigma,
                           inter_idx=None, inter_w=None, lazy_sample=True, radius_expansion=1.0, pooling=None):
    """so3conv/functional.py:118-178, materialising form:
    -> (inter_idx, inter_w [b,p2,na,ks,nn], new_xyz, new_feats [b,c,ks,p2,na], sample_idx)."""
    if pooling is not None and stride > 1 and feats.shape[1] > 1:
        if pooling == 'stride':
            pool_stride = stride
            stride_nn = int(n_neighbor * pool_stride ** 0.5)
            stride = 1
        elif pooling == 'no-stride':
            pool_stride = 1
            stride_nn = n_neighbor
        else:
            raise NotImplementedError(f"Pooling mode {pooling} is not implemented!")
        feats, xyz = inter_so3conv_blurring(xyz, feats, stride_nn, radius, pool_stride, inter_idx, lazy_sample)
        inter_idx = None
    if inter_idx is None:
        grouped_xyz, inter_idx, sample_idx, new_xyz = zpconv.inter_zpconv_grouping_ball(
            xyz, stride, radius * radius_expansion, n_neighbor, lazy_sample)
        inter_w = inter_so3conv_grouping_anchor(grouped_xyz, anchors, kernels, sigma)
    else:
        sample_idx = None
        new_xyz = xyz
    feats = zpconv.add_shadow_feature(feats)
    new_feats = inter_so3conv_feat_grouping(inter_idx, inter_w, feats)
    return inter_idx, inter_w, new_xyz, new_feats, sample_idx


def intra_so3conv_grouping(intra_idx, feature):
    """so3conv/functional.py:221-233 -> G[b,c,k,p,a] = feature[b,c,p,intra_idx[a,k]] (HIP intra kernel with
    an identity weight: same device code path as the fused module)."""
    nb, c_in, nq, na = feature.shape
    pnn = intra_idx.shape[1]
    eye = torch.eye(c_in * pnn, dtype=torch.float32, device=feature.device)
    out = ops.intra_so3conv(feature, eye, intra_idx.int().contiguous())   # [b, c*pnn, p, a]
    return out.view(nb, c_in, pnn, nq, na).contiguous()


def select_anchor(anchors, k):
    """so3conv/functional.py:281-289: 1 -> identity anchor, 20 -> one per face, 40 -> two per face,
    anything else -> all 60."""
    if k == 1:
        return anchors[29][None]
    elif k == 20:
        return anchors[::3]
    elif k == 40:
        return anchors.reshape(20, 3, 3, 3)[:, :2].reshape(-1, 3, 3)
    return anchors


def get_anchors(k=60):
    return select_anchor(Rs, k)


def get_intra_idx():
    return R_idx


def get_canonical_relative():
    return canonical_relative
