"""Layer schedules of the shipped models' backbones and a minimal backbone module over the hot path.

The reference builds its networks from parameter dicts (SPConvNets/models/cls_so3net_pn.py:43-167); the
numbers below restate that arithmetic (SURVEY.md 8a "per-layer schedule", verified there against the
instantiated reference builders).  `HotPathBackbone` chains InterSO3Conv -> IntraSO3Conv exactly as
SeparableSO3ConvBlock does (SPConvNets/utils/base_so3conv.py:168-212: inter conv, norm, leaky_relu,
intra conv, InstanceNorm, leaky_relu, strided skip + 1x1 conv + norm + leaky_relu, add) so that bench.py
and the tests exercise the real call pattern; the norm / activation / skip glue are plain torch ops
(fusing them is a "next" row), the convolutions are the fused HIP kernels.
"""
import math
from collections import namedtuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .vgtk import so3conv as sptk
from .vgtk import spconv as zptk

Layer = namedtuple("Layer", "cin cout stride radius sigma nn lazy")


def cls_so3net_schedule(input_num=1024, mlps=((64, 64), (128, 128), (256, 256), (256,)), strides=(2, 2, 2, 2),
                        initial_radius_ratio=0.2, sampling_ratio=0.4, sampling_density=0.5, sigma_ratio=0.5,
                        input_radius=1.0):
    """cls_so3net_pn.build_model (SPConvNets/models/cls_so3net_pn.py:43-150), ModelNet40 classification."""
    strides = list(strides)
    if input_num > 1024:
        sampling_ratio /= (input_num / 1024)
        strides[0] = int(2 * (input_num / 1024))
    mult = [2 ** i for i in range(len(mlps) + 1)]
    num_centers = [int(input_num / m) for m in mult]
    radius_ratio = [initial_radius_ratio * m ** sampling_density for m in mult]
    radii = [r * input_radius for r in radius_ratio]
    sigma = [sigma_ratio * radii[0] ** 2]
    for i in range(len(strides)):
        sigma.append(sigma[i] * 2)
    layers, dim_in = [], 1
    for i, block in enumerate(mlps):
        for j, dim_out in enumerate(block):
            neighbor = int(sampling_ratio * num_centers[i] * radius_ratio[i] ** (1 / sampling_density))
            if j == 0:
                stride, nidx, neighbor = strides[i], (i if i == 0 else i + 1), neighbor * 2
            else:
                stride, nidx = 1, i + 1
            layers.append(Layer(dim_in, dim_out, stride, radii[nidx], sigma[nidx], neighbor, i != 0 or j != 0))
            dim_in = dim_out
    return layers


def scaled(layers, width_div):
    """Same geometry, channel widths divided (tests / smoke)."""
    out, cin = [], 1
    for l in layers:
        cout = max(l.cout // width_div, 1)
        out.append(l._replace(cin=cin, cout=cout))
        cin = cout
    return out


class SeparableBlock(nn.Module):
    """One SeparableSO3ConvBlock of the cls model (norm='BatchNorm2d', activation='leaky_relu',
    SPConvNets/utils/base_so3conv.py:168-212).  The norm / activation / skip glue are stock torch modules (MIOpen
    batch-norm and 1x1 convolution: measured faster than hand-composed channels-last torch ops; fusing them into the
    conv epilogues is SURVEY 8f.1, "next")."""

    def __init__(self, l, kanchor=60):
        super().__init__()
        self.stride = l.stride
        self.inter = sptk.InterSO3Conv(l.cin, l.cout, 1, l.stride, l.radius, l.sigma, l.nn,
                                       lazy_sample=l.lazy, kanchor=kanchor)
        self.inter_norm = nn.BatchNorm2d(l.cout)
        self.intra = sptk.IntraSO3Conv(l.cout, l.cout)
        self.intra_norm = nn.InstanceNorm2d(l.cout, affine=False)
        self.skip_conv = nn.Conv2d(l.cin, l.cout, 1)
        self.norm = nn.BatchNorm2d(l.cout)

    def forward(self, x):
        skip = x.feats
        _, _, sample_idx, y = self.inter(x)
        feat = F.leaky_relu(self.inter_norm(y.feats))
        z = self.intra(zptk.SphericalPointCloud(y.xyz, feat, y.anchors))
        feat = F.leaky_relu(self.intra_norm(z.feats))
        if self.stride > 1:
            skip = zptk.functional.batched_index_select(skip, 2, sample_idx.long())
        skip = F.leaky_relu(self.norm(self.skip_conv(skip)))
        return zptk.SphericalPointCloud(z.xyz, feat + skip, z.anchors)


class HotPathBackbone(nn.Module):
    """preprocess_input (ones features) -> chain of separable blocks.  Input [b, n, 3] point clouds."""

    def __init__(self, layers, kanchor=60):
        super().__init__()
        self.kanchor = kanchor
        self.blocks = nn.ModuleList([SeparableBlock(l, kanchor) for l in layers])

    def forward(self, pts):
        xyz = pts.permute(0, 2, 1).contiguous()
        feats = torch.ones(pts.shape[0], 1, pts.shape[1], self.kanchor, dtype=torch.float32, device=pts.device)
        x = zptk.SphericalPointCloud(xyz, feats, None)
        for blk in self.blocks:
            x = blk(x)
        return x


def synthetic_clouds(b, n, device, seed=2913, scale=1.0):
    """N points uniform in the unit ball, centred, max-norm 1 -> [b, n, 3] float32."""
    gen = torch.Generator(device="cpu").manual_seed(seed)
    g = torch.randn(b, n, 3, generator=gen, dtype=torch.float64)
    u = torch.rand(b, n, 1, generator=gen, dtype=torch.float64)
    p = g / g.norm(dim=2, keepdim=True) * u ** (1.0 / 3.0)
    p = p - p.mean(dim=1, keepdim=True)
    p = p / p.norm(dim=2).amax(dim=1)[:, None, None]
    return (scale * p).float().to(device)


def hot_path_flops(layers, b, n, na=60, ks=24, kn=12):
    """Algorithmic flops of one forward of the hot path: per inter layer weight generation
    9*B*P2*A*ks*K, grouping 2*B*Cin*ks*P2*A*K, GEMM 2*B*P2*A*Cout*Cin*ks; per intra layer 2*B*P*A*C*C*12."""
    p, out = n, []
    for l in layers:
        p2 = math.ceil(p / l.stride)
        cols = b * p2 * na
        out.append(dict(wgen=9.0 * cols * ks * l.nn, group=2.0 * cols * l.cin * ks * l.nn,
                        gemm=2.0 * cols * l.cout * l.cin * ks, intra=2.0 * cols * l.cout * l.cout * kn))
        p = p2
    return out
