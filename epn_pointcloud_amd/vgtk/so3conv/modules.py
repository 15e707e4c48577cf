"""nn.Modules of the SO(3) separable convolution (vgtk/vgtk/so3conv/modules.py) -- same constructor
signatures, return tuples and state_dict keys (anchors, kernels, intra_idx, basic_conv.W); forward
runs the fused HIP kernels through epn_pointcloud_amd.ops instead of gather -> einsum -> matmul."""
import math

import numpy as np
import torch
import torch.nn as nn

from ..spconv import SphericalPointCloud
from .. import pc as pctk
from . import functional as L
from ... import gemm, ops

__all__ = ["BasicSO3Conv", "KernelPropagation", "InterSO3Conv", "IntraSO3Conv", "PointnetSO3Conv",
           "KERNEL_CONDENSE_RATIO"]

KERNEL_CONDENSE_RATIO = 0.7


class BasicSO3Conv(nn.Module):
    """[b,c1,k,p,a] -> [b,c2,p,a]: W[c2, c1*k] applied at every (p,a); no bias (modules.py:21-55).
    Inside Inter/IntraSO3Conv only `W` is used (the contraction is fused into the HIP kernel); calling
    the module on a materialised tensor keeps the reference's semantics."""

    def __init__(self, dim_in, dim_out, kernel_size, debug=False):
        super(BasicSO3Conv, self).__init__()
        self.dim_in = dim_in
        self.dim_out = dim_out
        self.kernel_size = kernel_size
        if debug:
            self.register_buffer('W', torch.ones(self.dim_out, self.dim_in * self.kernel_size))
        else:
            W = torch.empty(self.dim_out, self.dim_in, self.kernel_size)
            nn.init.xavier_normal_(W, gain=nn.init.calculate_gain('relu'))
            self.register_parameter('W', nn.Parameter(W.view(self.dim_out, self.dim_in * self.kernel_size)))

    def forward(self, x):
        """x [b, dim_in, ks, p, a] -> [b, dim_out, p, a] (modules.py:48-55).  Device tensors run on the library's own
        GEMM (gemm.matmul_nt: rows = (b, p, a) columns, fp32 contractions in the default split form); host tensors
        -- module construction tests only -- on torch."""
        bs, npt, na = x.shape[0], x.shape[3], x.shape[4]
        ck = self.dim_in * self.kernel_size
        if x.is_cuda and x.dtype in ops.FEATURE_DTYPES:
            rows = x.reshape(bs, ck, npt * na).permute(0, 2, 1).reshape(bs * npt * na, ck)
            y = gemm.matmul_nt(rows, self.W)
            return y.view(bs, npt, na, self.dim_out).permute(0, 3, 1, 2)
        x = x.reshape(bs, ck, npt * na)
        return torch.matmul(self.W, x).view(bs, self.dim_out, npt, na)


class KernelPropagation(nn.Module):
    """modules.py:57-119.  Buffers/parameters as in the reference; forward = FPS centres, `initial_anchor_query` (HIP,
    atomic-free), normalisation by the point count, BasicSO3Conv.  No shipped model reaches it."""

    def __init__(self, dim_in, dim_out, n_center, kernel_size, radius, sigma, kanchor=60):
        super(KernelPropagation, self).__init__()
        kernels = L.get_sphereical_kernel_points_from_ply(KERNEL_CONDENSE_RATIO * radius, kernel_size)
        anchors = L.get_anchors(kanchor)
        kernels = np.ascontiguousarray(np.transpose(anchors @ kernels.T, (2, 0, 1)))   # [ks, na, 3], contiguous for the kernel
        self.radius = radius
        self.sigma = sigma
        self.n_center = n_center
        self.register_buffer('anchors', torch.from_numpy(anchors))
        self.register_buffer('kernels', torch.from_numpy(kernels))
        self.basic_conv = BasicSO3Conv(dim_in, dim_out, kernels.shape[0])

    def forward(self, frag, clouds):
        if clouds.shape[2] == self.n_center:
            centers = clouds
        else:
            _, centers = pctk.furthest_sample(clouds, self.n_center, False)
        wts, nnctn = L.initial_anchor_query(frag, centers, self.kernels, self.radius, self.sigma)
        wts = wts / (nnctn + 1.0)
        feats = self.basic_conv(wts.unsqueeze(1))
        return SphericalPointCloud(centers, feats, self.anchors)


class InterSO3Conv(nn.Module):
    """[b,c1,p1,a] -> [b,c2,p2,a]: convolution over the K spatial neighbours under every anchor rotation
    (modules.py:125-174).  forward(x, inter_idx=None, inter_w=None) ->
    (inter_idx, inter_w, sample_idx, SphericalPointCloud).  `inter_w` is returned as a lazy
    ops.InterGeometry (call .dense() for the reference tensor); a dense tensor is accepted on input."""

    def __init__(self, dim_in, dim_out, kernel_size, stride, radius, sigma, n_neighbor,
                 lazy_sample=True, pooling=None, kanchor=60):
        super(InterSO3Conv, self).__init__()
        kernels = L.get_sphereical_kernel_points_from_ply(KERNEL_CONDENSE_RATIO * radius, kernel_size)
        anchors = L.get_anchors(kanchor)
        self.dim_in = dim_in
        self.dim_out = dim_out
        self.kernel_size = kernels.shape[0]
        self.stride = stride
        self.radius = radius
        self.sigma = sigma
        self.n_neighbor = n_neighbor
        self.lazy_sample = lazy_sample
        self.pooling = pooling
        self.basic_conv = BasicSO3Conv(dim_in, dim_out, self.kernel_size)
        self.register_buffer('anchors', torch.from_numpy(np.ascontiguousarray(anchors)))
        self.register_buffer('kernels', torch.from_numpy(np.ascontiguousarray(kernels)))
        self.feat_dtype = None     # output feature dtype; None = the input's (set by schedule.set_feature_dtype)

    def forward(self, x, inter_idx=None, inter_w=None):
        xyz, feats = x.xyz, x.feats
        stride = self.stride
        if self.pooling is not None and stride > 1 and feats.shape[1] > 1:
            # low-pass blurring before the strided conv (functional.py:133-148); torch glue, never taken
            # by the shipped models (xyz_pooling=None)
            if self.pooling == 'stride':
                pool_stride, stride_nn, stride = stride, int(self.n_neighbor * stride ** 0.5), 1
            elif self.pooling == 'no-stride':
                pool_stride, stride_nn = 1, self.n_neighbor
            else:
                raise NotImplementedError(f"Pooling mode {self.pooling} is not implemented!")
            feats, xyz = L.inter_so3conv_blurring(xyz, feats, stride_nn, self.radius, pool_stride, inter_idx,
                                                  self.lazy_sample)
            inter_idx = None
        if inter_idx is None:
            n_sample = math.ceil(xyz.shape[2] / stride)
            sample_idx, new_xyz = pctk.furthest_sample(xyz, n_sample, self.lazy_sample)
            inter_idx = pctk.ball_query_index(new_xyz, xyz, self.radius, self.n_neighbor)
            inter_w = ops.InterGeometry(xyz, new_xyz, inter_idx, self.anchors, self.kernels, self.sigma)
            handle = inter_w
        else:
            sample_idx, new_xyz = None, xyz
            if isinstance(inter_w, ops.InterGeometry):
                handle = inter_w
            else:
                handle = ops.DenseInterWeights(inter_idx.int().contiguous(), inter_w, xyz.shape[2])
        if getattr(self, "share_input_grad", False) and feats is x.feats:
            # (set by a block whose skip branch reads x.feats too: see ops.InterSO3ConvSplitFn.forward)
            out, self._shared_input, self._out_stats = ops.inter_so3conv(feats, self.basic_conv.W, handle, self.feat_dtype,
                                                                         share_input=True)
        else:
            out = ops.inter_so3conv(feats, self.basic_conv.W, handle, self.feat_dtype)
        return inter_idx, inter_w, sample_idx, SphericalPointCloud(new_xyz, out, self.anchors)


class IntraSO3Conv(nn.Module):
    """[b,c1,p,a] -> [b,c2,p,a]: convolution over the 12 nearest rotation anchors (modules.py:177-200).
    Like the reference it always uses the 60-anchor table (get_anchors() without k)."""

    def __init__(self, dim_in, dim_out):
        super(IntraSO3Conv, self).__init__()
        anchors = L.get_anchors()
        intra_idx = L.get_intra_idx()
        self.dim_in = dim_in
        self.dim_out = dim_out
        self.kernel_size = intra_idx.shape[1]
        self.basic_conv = BasicSO3Conv(dim_in, dim_out, self.kernel_size)
        self.register_buffer('anchors', torch.from_numpy(anchors))
        self.register_buffer('intra_idx', torch.from_numpy(intra_idx).long())

    def _idx32(self):
        """int32 copy of the index buffer, made once per (buffer, device): the derived tables (inverse permutation,
        block-diagonalising basis) are cached per tensor, and nothing is re-derived inside a captured graph."""
        src = self.intra_idx
        key = (src.data_ptr(), src._version, str(src.device))
        if getattr(self, "_idx32_key", None) != key:
            self._idx32_cache, self._idx32_key = src.int().contiguous(), key
        return self._idx32_cache

    def forward(self, x, pre_norm=None, pre_part=None):
        """pre_norm (extension, used by schedule.FusedSeparableBlock): the norm module whose leaky_relu(norm(x.feats)) is the
        input -- folded into the convolution's basis change when it takes the block-diagonal form; pre_part: partial
        per-channel statistics of x.feats from the epilogue of the GEMM that produced it."""
        if getattr(self, "want_out_stats", False):       # set by a block whose norm follows: see ops.intra_so3conv
            feats, self._out_stats = ops.intra_so3conv(x.feats, self.basic_conv.W, self._idx32(), pre_norm=pre_norm,
                                                       pre_part=pre_part, out_stats=True)
        else:
            feats = ops.intra_so3conv(x.feats, self.basic_conv.W, self._idx32(), pre_norm=pre_norm, pre_part=pre_part)
        return SphericalPointCloud(x.xyz, feats, self.anchors)

    def takes_spectral_form(self, is_cuda=True):
        return ops.intra_takes_spectral(self.dim_in, self.dim_out, self._idx32(), is_cuda)


class PointnetSO3Conv(nn.Module):
    """Equivariant pointnet aggregation over points (modules.py:203-235), the tail of every model: centre xyz, rotate
    it into each anchor frame, concatenate to the features, 1x1 `embed` convolution, max over points -- one fused HIP
    pass (epn_pointnet_so3conv_*_f32); same parameters / state_dict (embed.weight [co, c+3, 1, 1], embed.bias,
    anchors buffer) and the same [nb, nc, na] result."""

    def __init__(self, dim_in, dim_out, kanchor=60):
        super(PointnetSO3Conv, self).__init__()
        anchors = L.get_anchors(kanchor)
        self.dim_in = dim_in + 3
        self.dim_out = dim_out
        self.embed = nn.Conv2d(self.dim_in, self.dim_out, 1)
        self.register_buffer('anchors', torch.from_numpy(np.ascontiguousarray(anchors)))

    def forward(self, x):
        return ops.pointnet_so3conv(x.feats, x.xyz, self.anchors, self.embed.weight, self.embed.bias)
