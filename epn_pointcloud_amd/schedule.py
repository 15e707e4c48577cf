"""Layer schedules of the shipped models' backbones and a minimal backbone module over the hot path.

The reference builds its networks from parameter dicts (SPConvNets/models/cls_so3net_pn.py:43-167); the
numbers below restate that arithmetic (SURVEY.md 8a "per-layer schedule", verified there against the
instantiated reference builders).  `HotPathBackbone` chains InterSO3Conv -> IntraSO3Conv exactly as
SeparableSO3ConvBlock does (SPConvNets/utils/base_so3conv.py:168-212: inter conv, norm, leaky_relu,
intra conv, InstanceNorm, leaky_relu, strided skip + 1x1 conv + norm + leaky_relu, add) so that bench.py
and the tests exercise the real call pattern; the norm / activation / skip glue run
on the HIP block-glue kernels (csrc/glue.hip) in FusedSeparableBlock, stock torch modules in SeparableBlock.
"""
import math
from collections import namedtuple

import torch

from ._ab import ab
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .vgtk import so3conv as sptk
from .vgtk import spconv as zptk

Layer = namedtuple("Layer", "cin cout stride radius sigma nn lazy stage", defaults=(0,))


def _schedule(input_num, mlps, strides, initial_radius_ratio, sampling_ratio, sampling_density, sigma_ratio,
              input_radius, sigma_by_stride, scale_first_neighbor):
    """Common arithmetic of the three model builders (SPConvNets/models/{cls_so3net_pn,reg_so3net,inv_so3net_pn}.py):
    radii / sigmas / neighbour counts per layer from a handful of ratios."""
    strides = list(strides)
    if input_num > 1024:
        sampling_ratio /= (input_num / 1024)
        strides[0] = int(2 * (input_num / 1024))
    mult = [2 ** i for i in range(len(mlps) + 1)]
    num_centers = [int(input_num / m) for m in mult]
    radius_ratio = [initial_radius_ratio * m ** sampling_density for m in mult]
    radii = [r * input_radius for r in radius_ratio]
    sigma = [sigma_ratio * radii[0] ** 2]
    for i, st in enumerate(strides):
        sigma.append(sigma[i] * (st if sigma_by_stride else 2))
    layers, dim_in = [], 1
    for i, block in enumerate(mlps):
        for j, dim_out in enumerate(block):
            neighbor = int(sampling_ratio * num_centers[i] * radius_ratio[i] ** (1 / sampling_density))
            if scale_first_neighbor and i == 0 and j == 0:
                neighbor *= int(input_num / 1024)
            if j == 0:
                stride, nidx, neighbor = strides[i], (i if i == 0 else i + 1), neighbor * 2
            else:
                stride, nidx = 1, i + 1
            layers.append(Layer(dim_in, dim_out, stride, radii[nidx], sigma[nidx], neighbor, i != 0 or j != 0, i))
            dim_in = dim_out
    return layers


def cls_so3net_schedule(input_num=1024, mlps=((64, 64), (128, 128), (256, 256), (256,)), strides=(2, 2, 2, 2)):
    """cls_so3net_pn.build_model (SPConvNets/models/cls_so3net_pn.py:43-150), ModelNet40 classification."""
    return _schedule(input_num, mlps, strides, 0.2, 0.4, 0.5, 0.5, 1.0, False, False)


def reg_so3net_schedule(input_num=1024, mlps=((32, 32), (64, 64), (128, 128), (256,)), strides=(2, 2, 2, 2)):
    """reg_so3net.build_model (SPConvNets/models/reg_so3net.py:52-150), ModelNet40 rotation estimation (pairs of
    clouds are concatenated along the batch axis, reg_so3net.py:31-33)."""
    return _schedule(input_num, mlps, strides, 0.2, 0.8, 0.5, 0.5, 1.0, False, False)


def inv_so3net_schedule(input_num=2048, search_radius=0.4, mlps=((32, 32), (64, 64), (128, 128), (128, 128)),
                        strides=(2, 2, 2, 2)):
    """inv_so3net_pn.build_model (SPConvNets/models/inv_so3net_pn.py:43-150), 3DMatch local-patch descriptor."""
    return _schedule(input_num, mlps, strides, 0.2, 0.8, 0.5, 0.5, search_radius, True, True)


def scaled(layers, width_div):
    """Same geometry, channel widths divided (tests / smoke)."""
    out, cin = [], 1
    for l in layers:
        cout = max(l.cout // width_div, 1)
        out.append(l._replace(cin=cin, cout=cout))
        cin = cout
    return out


class _ConvNorm(nn.Module):
    """Attribute layout of InterSO3ConvBlock / IntraSO3ConvBlock (base_so3conv.py:32-62, 93-126): `.conv`, `.norm`."""

    def __init__(self, conv, norm):
        super().__init__()
        self.conv = conv
        self.norm = norm


class SeparableBlock(nn.Module):
    """One SeparableSO3ConvBlock (SPConvNets/utils/base_so3conv.py:168-212) with the reference's module tree, so a
    reference checkpoint's keys (`inter_conv.conv.basic_conv.W`, `inter_conv.norm.*`, `intra_conv.conv.*`,
    `skip_conv.*`, `norm.*`) load unchanged.  This class runs the glue (norm, leaky_relu, skip) on stock torch modules
    around the fused HIP convolutions; FusedSeparableBlock below moves the glue onto HIP kernels as well.
    forward(x, inter_idx=None, inter_w=None) -> (inter_idx, inter_w, sample_idx, x_out) like the reference."""

    def __init__(self, l, kanchor=60, norm="BatchNorm2d", dropout_rate=0.0):
        super().__init__()
        # norm=None -> InstanceNorm2d(affine=False), the default of InterSO3ConvBlock / SeparableSO3ConvBlock
        # (base_so3conv.py:107,191) used by the rotation and 3DMatch models; the cls model passes 'BatchNorm2d'
        mk = (lambda c: nn.InstanceNorm2d(c, affine=False)) if norm is None else getattr(nn, norm)
        self.stride = l.stride
        self.inter_conv = _ConvNorm(sptk.InterSO3Conv(l.cin, l.cout, 1, l.stride, l.radius, l.sigma, l.nn,
                                                      lazy_sample=l.lazy, kanchor=kanchor), mk(l.cout))
        self.use_intra = kanchor > 1           # SeparableSO3ConvBlock.use_intra (base_so3conv.py:177)
        if self.use_intra:
            if kanchor != 60:
                raise NotImplementedError(
                    f"SeparableSO3ConvBlock with kanchor={kanchor}: IntraSO3Conv always uses the 60-anchor table "
                    "(vgtk/vgtk/so3conv/modules.py:186); the reference builders pick 'inter_block' for kanchor != 60")
            self.intra_conv = _ConvNorm(sptk.IntraSO3Conv(l.cout, l.cout), nn.InstanceNorm2d(l.cout, affine=False))
        self.skip_conv = nn.Conv2d(l.cin, l.cout, 1)
        self.norm = mk(l.cout)
        self.dropout = nn.Dropout(dropout_rate) if dropout_rate > 0 else None

    def _drop(self, t):
        return self.dropout(t) if (self.dropout is not None and self.training) else t

    def forward(self, x, inter_idx=None, inter_w=None):
        skip = x.feats
        inter_idx, inter_w, sample_idx, y = self.inter_conv.conv(x, inter_idx, inter_w)
        feat = self._drop(F.leaky_relu(self.inter_conv.norm(y.feats)))
        z = y
        if self.use_intra:
            z = self.intra_conv.conv(zptk.SphericalPointCloud(y.xyz, feat, y.anchors))
            feat = self._drop(F.leaky_relu(self.intra_conv.norm(z.feats)))
        if self.stride > 1:
            skip = zptk.functional.batched_index_select(skip, 2, sample_idx.long())
        skip = F.leaky_relu(self.norm(self.skip_conv(skip.float())))       # stock modules hold fp32 parameters
        return inter_idx, inter_w, sample_idx, zptk.SphericalPointCloud(z.xyz, (feat.float() + skip).to(feat.dtype),
                                                                        z.anchors)


class FusedSeparableBlock(SeparableBlock):
    """Same module tree / state_dict as SeparableBlock, with the glue on the HIP "block glue" kernels:
    norm + leaky_relu (+ the residual add) are two streaming passes each, the 1x1 skip convolution runs on the intra
    GEMM kernel (one anchor "neighbour", identity index) -- everything stays channels-last, no layout copies.
    Training-mode semantics (batch statistics); eval mode and dropout fall back to the stock modules."""

    def forward(self, x, inter_idx=None, inter_w=None):
        c_out = self.inter_conv.conv.dim_out
        if (not self.training) or self.dropout is not None or not ops.norm_act_supported(c_out) or not self.use_intra:
            return super().forward(x, inter_idx, inter_w)
        import os
        conv = self.inter_conv.conv
        # x.feats feeds the inter convolution AND the skip branch: the convolution hands back the tensor for the second use
        # and folds that branch's gradient into its own data gradient (ops.InterSO3ConvSplitFn; EPN_SHARE_INPUT_GRAD=0: off)
        conv.share_input_grad = True
        # two-piece fp16 GEMMs: max|x| once for both consumers of the block input (the grouped features' bound K max|x| and
        # the skip convolution; a strided block's gathered rows are a subset: the same scalar bounds them)
        x_amax = None
        if x.feats.is_cuda and ops.gemm.f16x2_on(x.feats) and x.feats.shape[1] >= 16:
            x_amax = ops.gemm.absmax_cached(ops.to_cl(x.feats))
        try:
            inter_idx, inter_w, sample_idx, y = conv(x, inter_idx, inter_w)
        finally:
            conv.share_input_grad = False
        skip = conv.__dict__.pop("_shared_input", None)
        if skip is None:
            skip = x.feats
        # per-channel statistics of the two GEMM outputs that a norm follows (inter convolution, skip convolution) come from
        # the GEMMs' epilogues (block partials; EPN_EPILOGUE_STATS=0: separate passes over the tensors)
        epi = ab("EPN_EPILOGUE_STATS") == "1"
        y_part = conv.__dict__.pop("_out_stats", None)
        y_part = y_part if epi else None

        pair = ab("EPN_NORM_PAIR") == "1"     # skip norm folded into the block's final pass

        def skip_branch():
            sk = skip
            if self.stride > 1:                                    # batched_index_select(skip, 2, sample_idx) on rows
                sk = ops.gather_rows(sk, sample_idx)
            if pair and epi:
                return ops.conv1x1(sk, self.skip_conv.weight, None, col_stats=True, x_amax=x_amax)   # (tensor, partial statistics)
            sk = ops.conv1x1(sk, self.skip_conv.weight, None, x_amax=x_amax)    # the norm cancels the bias: see ops.norm_act
            return (sk, None) if pair else (ops.norm_act(sk, self.norm, conv_bias=self.skip_conv.bias), None)

        side = None
        if os.environ.get("EPN_SKIP_STREAM", "1") == "1" and skip.is_cuda:
            # the skip branch (row gather, 1x1 conv, norm: small HBM-bound kernels) is independent of the main branch
            # until the final add: issue it on a second stream so it can fill in under the GEMMs
            main = torch.cuda.current_stream(skip.device)
            side = ops._side_stream(skip.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                s, s_part = skip_branch()
        if ab("EPN_NORM_ON_LOAD") == "1" and self.intra_conv.conv.takes_spectral_form(y.feats.is_cuda):
            # norm + leaky_relu of the inter convolution applied as the intra convolution's basis change loads its
            # rows: the normalised tensor is never written
            iconv = self.intra_conv.conv
            iconv.want_out_stats = epi and pair
            try:
                z = iconv(zptk.SphericalPointCloud(y.xyz, y.feats, y.anchors), pre_norm=self.inter_conv.norm, pre_part=y_part)
            finally:
                iconv.want_out_stats = False
            z_part = iconv.__dict__.pop("_out_stats", None)
        else:
            feat = ops.norm_act(y.feats, self.inter_conv.norm)
            z = self.intra_conv.conv(zptk.SphericalPointCloud(y.xyz, feat, y.anchors))
            z_part = None
        if side is None:
            s, s_part = skip_branch()
        else:
            main.wait_stream(side)
            s.record_stream(main)
            if s_part is not None:
                s_part.record_stream(main)
        if pair:
            # leaky(IN(z)) + leaky(norm(skip conv)) in ONE pass: the skip branch's normalised tensor is never written, the
            # backward reads the output gradient once per pass for both norms
            out = ops.norm_act_pair(z.feats, self.intra_conv.norm, s, self.norm, conv_bias_b=self.skip_conv.bias,
                                    part_b=s_part, part_a=z_part)
        else:
            out = ops.norm_act(z.feats, self.intra_conv.norm, residual=s)   # leaky(IN(z)) + skip in the same pass
        return inter_idx, inter_w, sample_idx, zptk.SphericalPointCloud(z.xyz, out, z.anchors)


class InterBlock(nn.Module):
    """InterSO3ConvBlock (SPConvNets/utils/base_so3conv.py:87-126): InterSO3Conv -> norm -> leaky_relu (-> dropout), no
    intra convolution and no skip branch -- the block type the reference builders choose when kanchor != 60
    (cls_so3net_pn.py:127 `na < 60`, reg_so3net.py:139 / inv_so3net_pn.py:139 `na != 60`).  state_dict: `conv.*`, `norm.*`."""

    def __init__(self, l, kanchor=60, norm="BatchNorm2d", dropout_rate=0.0):
        super().__init__()
        self.stride = l.stride
        self.conv = sptk.InterSO3Conv(l.cin, l.cout, 1, l.stride, l.radius, l.sigma, l.nn, lazy_sample=l.lazy,
                                      kanchor=kanchor)
        self.norm = nn.InstanceNorm2d(l.cout, affine=False) if norm is None else getattr(nn, norm)(l.cout)
        self.dropout = nn.Dropout(dropout_rate) if dropout_rate > 0 else None

    def forward(self, x, inter_idx=None, inter_w=None):
        inter_idx, inter_w, sample_idx, y = self.conv(x, inter_idx, inter_w)
        if self.training and self.dropout is None and y.feats.is_cuda and ops.norm_act_supported(y.feats.shape[1]):
            feat = ops.norm_act(y.feats, self.norm)
        else:
            feat = F.leaky_relu(self.norm(y.feats))
            if self.training and self.dropout is not None:
                feat = self.dropout(feat)
        return inter_idx, inter_w, sample_idx, zptk.SphericalPointCloud(y.xyz, feat, y.anchors)


def block_type(kanchor, model="cls"):
    """'inter_block' | 'separable_block' exactly as the reference builders decide (cls_so3net_pn.py:127: na < 60;
    reg_so3net.py:139, inv_so3net_pn.py:139: na != 60)."""
    inter_only = kanchor < 60 if model == "cls" else kanchor != 60
    return "inter_block" if inter_only else "separable_block"


class BasicBlock(nn.Module):
    """One resolution stage = BasicSO3ConvBlock (base_so3conv.py:129-166): `.blocks`, the (inter_idx, inter_w) of a
    block handed to the next one until a strided block resets them."""

    def __init__(self, layers, kanchor=60, norm="BatchNorm2d", fused_glue=True, dropout_rate=0.0,
                 btype="separable_block"):
        super().__init__()
        if btype == "inter_block":
            blk = InterBlock
        elif btype == "separable_block":
            blk = FusedSeparableBlock if fused_glue else SeparableBlock
        else:
            raise ValueError(f"No such type of SO3Conv {btype}")
        self.blocks = nn.ModuleList([blk(l, kanchor, norm, dropout_rate) for l in layers])

    def forward(self, x):
        inter_idx, inter_w = None, None
        for blk in self.blocks:
            inter_idx, inter_w, _, x = blk(x, inter_idx, inter_w)
            if blk.stride > 1:
                inter_idx, inter_w = None, None
        return x


def set_feature_dtype(model, dtype):
    """Feature storage dtype of a network built from the blocks above: torch.float32 (default) or torch.bfloat16
    (BASELINE configs 3-4: bf16 features, fp32 accumulation).  Parameters, coordinates, indices and the kernel-influence
    weights stay fp32; every InterSO3Conv emits features in `dtype` (the first layer computes in fp32 and converts),
    everything downstream follows its input's dtype, heads / PointnetSO3Conv convert back to fp32."""
    if dtype not in ops.FEATURE_DTYPES:
        raise TypeError(f"feature dtype must be float32 or bfloat16, got {dtype}")
    for m in model.modules():
        if isinstance(m, sptk.InterSO3Conv):
            m.feat_dtype = dtype
    return model


def stages(layers):
    """Split a flat schedule into its resolution stages (Layer.stage = index of the mlps row it came from)."""
    out, last = [], None
    for l in layers:
        if last is None or l.stage != last:
            out.append([])
            last = l.stage
        out[-1].append(l)
    return out


def preprocess_input(pts, kanchor):
    """preprocess_input(x, na, add_center=False) (base_so3conv.py:16-23): [b, n, 3] -> SphericalPointCloud with the
    all-ones occupancy feature [b, 1, n, na] (get_occupancy_features, vgtk/vgtk/so3conv/functional.py:25-44)."""
    xyz = pts[:, :, :3].permute(0, 2, 1).contiguous()
    feats = torch.ones(pts.shape[0], 1, pts.shape[1], kanchor, dtype=torch.float32, device=pts.device)
    return zptk.SphericalPointCloud(xyz, feats, None)


class HotPathBackbone(nn.Module):
    """preprocess_input (ones features) -> stages of separable blocks, with the reference models' `backbone.{i}.blocks.{j}`
    module tree.  Input [b, n, 3] point clouds."""

    def __init__(self, layers, kanchor=60, norm="BatchNorm2d", fused_glue=True, dropout_rate=0.0, model="cls"):
        super().__init__()
        self.kanchor = kanchor
        self.backbone = nn.ModuleList([BasicBlock(st, kanchor, norm, fused_glue, dropout_rate,
                                                  block_type(kanchor, model)) for st in stages(layers)])

    def forward(self, pts):
        x = preprocess_input(pts, self.kanchor)
        for stage in self.backbone:
            x = stage(x)
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
