"""Index-based point-cloud operators (vgtk/vgtk/pc/sample.py:46-77)."""
import torch

from ..cuda import grouping as cuda_nn
from .. import utils


def group_nd(pc, idx):
    """[b,c,n] x [b,m1(,m2,...)] -> [b,c,m1(,m2,...)]   (sample.py:46-50)"""
    b = idx.shape[0]
    pc = utils.batch_gather(pc, idx.view(b, -1).contiguous(), dim=2)
    return pc.view(b, -1, *idx.shape[1:])


def ball_query_index(query_points, support_points, radius, n_sample):
    """[b,3,m] x [b,3,n] -> int32 [b,m,k]   (sample.py:54-59)"""
    return cuda_nn.ball_query(query_points, support_points, radius, n_sample)


def furthest_sample_index(pc, n_sample, lazy_sample):
    """sample.py:63-72: arange when nothing is dropped or lazy_sample, FPS kernel otherwise."""
    if pc.shape[2] == n_sample or lazy_sample:
        nb = pc.shape[0]
        return torch.arange(n_sample, dtype=torch.int32, device=pc.device).view(1, -1).expand(nb, -1).contiguous()
    return cuda_nn.furthest_point_sampling(pc, n_sample)


def furthest_sample(pc, n_sample, lazy_sample=True):
    """[b,3,n] -> ([b,m] int32, [b,3,m])   (sample.py:75-77)"""
    idx = furthest_sample_index(pc, n_sample, lazy_sample)
    return idx, group_nd(pc, idx)
