"""The three shipped networks over the hot path: backbone stages + output heads, with the reference's module tree
(`backbone.{i}.blocks.{j}.{inter_conv,intra_conv,skip_conv,norm}`, `outblock.*`) so that reference checkpoints load
unchanged.

    ClsSO3ConvModel   SPConvNets/models/cls_so3net_pn.py:15-40    ModelNet40 classification   (ClsOutBlockPointnet)
    RegSO3ConvModel   SPConvNets/models/reg_so3net.py:16-48        relative rotation estimation (RelSO3OutBlockR)
    InvSO3ConvModel   SPConvNets/models/inv_so3net_pn.py:15-41     3DMatch invariant descriptor (InvOutBlockMVD)

The backbones are schedule.BasicBlock stages (fused HIP convolutions + HIP block glue); every head aggregates over
points with PointnetSO3Conv, which is one fused HIP pass (epn_pointnet_so3conv_*_f32); the heads' 1x1 convolutions
and Linear layers (attention logits, class logits, the pair head's 60 x 60 anchor-pair MLP) run on the library's GEMMs
(ops.conv1x1 / ops.linear).  What remains on torch are elementwise tails on [b, c, a]-sized tensors (BatchNorm1d,
softmax, relu).
"""
import os

import torch

from ._ab import ab
import torch.nn as nn
import torch.nn.functional as F

from . import ops, schedule as S
from .vgtk import so3conv as sptk
from .vgtk import spconv as zptk


class ClsOutBlockPointnet(nn.Module):
    """1x1 conv + BatchNorm2d + relu per `mlp` entry, PointnetSO3Conv over points, BatchNorm1d + relu, pooling over
    the anchors ('attention' | 'max' | 'mean'), Linear to k classes (SPConvNets/utils/base_so3conv.py:358-448).
    forward(x) -> (logits [b, k], anchor attention logits [b, na] (or the pre-pointnet features for max / mean))."""

    def __init__(self, params):
        super().__init__()
        c_in, na = params['dim_in'], params['kanchor']
        self.linear, self.norm = nn.ModuleList(), nn.ModuleList()
        for c in params['mlp']:
            self.linear.append(nn.Conv2d(c_in, c, 1))
            self.norm.append(nn.BatchNorm2d(c))
            c_in = c
        self.pooling_method = params.get('pooling', 'max')
        if self.pooling_method == 'attention':
            self.temperature = params['temperature']
            self.attention_layer = nn.Conv1d(c_in, 1, 1)
        self.pointnet = sptk.PointnetSO3Conv(c_in, c_in, na)
        self.norm.append(nn.BatchNorm1d(c_in))
        self.fc2 = nn.Linear(c_in, params['k'])

    def forward(self, x, label=None):
        f = x.feats
        for lin, norm in zip(self.linear, self.norm):
            if self.training and ops.norm_act_supported(lin.out_channels):
                f = ops.conv1x1(f, lin.weight, None)                       # BatchNorm cancels the bias (ops.norm_act)
                f = ops.norm_act(f, norm, slope=0.0, conv_bias=lin.bias)   # relu(BatchNorm2d(.)) on the HIP glue
            else:
                f = F.relu(norm(ops.conv1x1(f, lin.weight, lin.bias)))
        out_feat = f
        y = self.pointnet(zptk.SphericalPointCloud(x.xyz, f, x.anchors))   # [b, c, a]
        y = F.relu(self.norm[len(self.linear)](y))
        if self.pooling_method == 'mean':
            y = y.mean(dim=2)
        elif self.pooling_method == 'max':
            y = y.max(2)[0]
        elif self.pooling_method.startswith('attention'):
            nb, c, na = y.shape                                             # Conv1d(c, 1, 1) as a GEMM over (b, a) rows
            att = self.attention_layer
            out_feat = ops.linear(y.transpose(1, 2).reshape(nb * na, c), att.weight, att.bias).view(nb, 1, na)
            y = (y * F.softmax(out_feat * self.temperature, dim=2)).sum(-1)
        else:
            raise NotImplementedError(f"Pooling mode {self.pooling_method} is not implemented!")
        return ops.linear(y, self.fc2.weight, self.fc2.bias), out_feat.squeeze()


class InvOutBlockMVD(nn.Module):
    """Per-point attention over the anchors (two 1x1 convs, softmax over A), then PointnetSO3Conv with a single
    (identity) anchor and L2 normalisation (base_so3conv.py:572-613).  forward(x) -> (descriptor [b, c_out], attention)."""

    def __init__(self, params):
        super().__init__()
        c_in, c_out = params['dim_in'], params['mlp'][-1]
        self.temperature = params['temperature']
        self.attention_layer = nn.Sequential(nn.Conv2d(c_in, c_in, 1), nn.ReLU(inplace=True), nn.Conv2d(c_in, c_in, 1))
        self.pooling_method = params.get('pooling', 'max')
        self.pointnet = sptk.PointnetSO3Conv(c_in, c_out, params['kanchor'])

    def forward(self, x):
        nb = x.feats.shape[0]
        a0, a2 = self.attention_layer[0], self.attention_layer[2]
        attn = ops.conv1x1(F.relu(ops.conv1x1(x.feats, a0.weight, a0.bias)), a2.weight, a2.bias)
        attn, pooled = ops.anchor_softmax_pool(x.feats, attn)              # softmax over the anchors, [b, c, p, 1]
        y = self.pointnet(zptk.SphericalPointCloud(x.xyz, pooled, None)).reshape(nb, -1)
        return F.normalize(y, p=2, dim=1), attn


class RelSO3OutBlockR(nn.Module):
    """Relative-rotation head: PointnetSO3Conv + relu per cloud, every (target anchor, source anchor) pair of the two
    [b, c, a] codes concatenated to [b, 2c, a, a], 1x1 MLP, attention over the target anchors and the per-pair rotation
    regressor (base_so3conv.py:661-731).  forward(f1, f2, x1, x2) -> (confidence [b, a, a], y [b, n_out, a, a])."""

    def __init__(self, params):
        super().__init__()
        c_in, na = params['dim_in'], params['kanchor']
        self.pointnet = sptk.PointnetSO3Conv(c_in, c_in, na)
        c_in *= 2
        self.temperature = params['temperature']
        rp = params['representation']
        if rp == 'quat':
            self.out_channel = 4
        elif rp == 'ortho6d':
            self.out_channel = 6
        else:
            raise KeyError("Unrecognized representation of rotation: %s" % rp)
        mlp = params['mlp']
        self.attention_layer = nn.Conv2d(mlp[-1], 1, (1, 1))
        self.regressor_layer = nn.Conv2d(mlp[-1], self.out_channel, (1, 1))
        self.linear = nn.ModuleList()
        for c in mlp:
            self.linear.append(nn.Conv2d(c_in, c, (1, 1)))
            c_in = c

    def _pooling(self, xyz, feats):
        return F.relu(self.pointnet(zptk.SphericalPointCloud(xyz, feats, None)))

    def forward(self, f1, f2, x1, x2):
        # bf16 feature networks: the 60 x 60 anchor-pair tensor ([b, 2c, a, a]: 115 200 rows x 512 channels, the largest
        # tensor of the head) and its 1x1 MLP follow the backbone's storage format -- bf16 values, fp32 accumulation in the
        # GEMMs, fp32 again from the two output convolutions on (softmax, rotation regressor); EPN_REG_MLP_BF16=0: all fp32
        low = f1.dtype == torch.bfloat16 and f1.is_cuda and ab("EPN_REG_MLP_BF16") == "1"
        f1, f2 = self._pooling(x1, f1), self._pooling(x2, f2)
        nb, _, na = f1.shape
        if low:
            f1t, f2t = f1.permute(0, 2, 1).bfloat16(), f2.permute(0, 2, 1).bfloat16()
            pair = torch.cat((f1t.unsqueeze(1).expand(-1, na, -1, -1), f2t.unsqueeze(2).expand(-1, -1, na, -1)), 3)
            pair = pair.permute(0, 3, 1, 2)
            for lin in self.linear:
                pair = F.relu(ops.conv1x1(pair, lin.weight.flatten(1)) + lin.bias.to(pair.dtype).view(1, -1, 1, 1))
            att = ops.conv1x1(pair, self.attention_layer.weight.flatten(1)).float() + self.attention_layer.bias.view(1, -1, 1, 1)
            confidence = F.softmax(att.reshape(nb, na, na) * self.temperature, dim=1)
            y = ops.conv1x1(pair, self.regressor_layer.weight.flatten(1)).float() + self.regressor_layer.bias.view(1, -1, 1, 1)
            return confidence, y
        if f1.is_cuda:
            # the reference's pair tensor cat(f1[b,:,None,j], f2[b,:,i,None]) built directly channels-last ([b][i][j][2c]),
            # so that the 1x1 MLP on the 60 x 60 anchor pairs is a chain of NT GEMMs on this library's kernels (through
            # nn.Conv2d it lands on MIOpen's direct-convolution kernels: 43 ms per step at B=32 pairs, measured)
            f1t, f2t = f1.permute(0, 2, 1), f2.permute(0, 2, 1)
            pair = torch.cat((f1t.unsqueeze(1).expand(-1, na, -1, -1), f2t.unsqueeze(2).expand(-1, -1, na, -1)), 3)
            pair = pair.permute(0, 3, 1, 2)                                   # logical [b, 2c, a, a], channels-last memory
            for lin in self.linear:
                pair = F.relu(ops.conv1x1(pair, lin.weight.flatten(1), lin.bias))
            att = ops.conv1x1(pair, self.attention_layer.weight.flatten(1), self.attention_layer.bias)
            confidence = F.softmax(att.reshape(nb, na, na) * self.temperature, dim=1)
            return confidence, ops.conv1x1(pair, self.regressor_layer.weight.flatten(1), self.regressor_layer.bias)
        pair = torch.cat((f1.unsqueeze(-2).expand(-1, -1, na, -1), f2.unsqueeze(-1).expand(-1, -1, -1, na)), 1)
        for lin in self.linear:
            pair = F.relu(lin(pair))
        confidence = F.softmax(self.attention_layer(pair).view(nb, na, na) * self.temperature, dim=1)
        return confidence, self.regressor_layer(pair)


class _SO3ConvModel(nn.Module):
    MODEL = "cls"

    def __init__(self, layers, kanchor, norm, fused_glue, dropout_rate):
        super().__init__()
        btype = S.block_type(kanchor, self.MODEL)
        self.backbone = nn.ModuleList([S.BasicBlock(st, kanchor, norm, fused_glue, dropout_rate, btype)
                                       for st in S.stages(layers)])
        self.na_in = kanchor
        self.invariance = True

    HEAD_TAKES_BF16 = False     # the head's first operation is PointnetSO3Conv, which reads bf16 features itself

    def features(self, pts):
        x = S.preprocess_input(pts, self.na_in)
        for stage in self.backbone:
            x = stage(x)
        if x.feats.dtype != torch.float32 and not (self.HEAD_TAKES_BF16 and x.feats.is_cuda
                                                   and ab("EPN_HEAD_BF16") == "1"):
            # bf16 feature path: the heads run in fp32 (their inputs are small)
            x = zptk.SphericalPointCloud(x.xyz, ops.cast_feats(x.feats, torch.float32), x.anchors)
        return x

    def get_anchor(self):
        last = self.backbone[-1].blocks[-1]
        return (last.inter_conv.conv if hasattr(last, "inter_conv") else last.conv).anchors


class ClsSO3ConvModel(_SO3ConvModel):
    """forward(x [b, n, 3]) -> (logits [b, 40], anchor attention [b, na])."""

    def __init__(self, layers, out_mlps=(256,), k=40, kanchor=60, pooling='attention', temperature=3.0,
                 fused_glue=True, dropout_rate=0.0):
        super().__init__(layers, kanchor, 'BatchNorm2d', fused_glue, dropout_rate)
        self.outblock = ClsOutBlockPointnet(dict(dim_in=layers[-1].cout, mlp=list(out_mlps), fc=[64], k=k,
                                                 pooling=pooling, temperature=temperature, kanchor=kanchor))

    def forward(self, x, rlabel=None):
        return self.outblock(self.features(x), rlabel)


class InvSO3ConvModel(_SO3ConvModel):
    """forward(x [b, n, 3]) -> (unit descriptor [b, c_out], per-point anchor attention)."""
    MODEL = "inv"

    def __init__(self, layers, out_mlps=(128, 64), kanchor=60, pooling='attention', temperature=3.0, fused_glue=True,
                 dropout_rate=0.0):
        super().__init__(layers, kanchor, None, fused_glue, dropout_rate)
        self.outblock = InvOutBlockMVD(dict(dim_in=layers[-1].cout, mlp=list(out_mlps), pooling=pooling,
                                            temperature=temperature, kanchor=kanchor))

    def forward(self, x):
        return self.outblock(self.features(x))


class RegSO3ConvModel(_SO3ConvModel):
    """forward(x [b, 2, n, 3]) -> (confidence [b, na, na], rotations [b, 4 | 6, na, na]); the two clouds of a pair go
    through the backbone as one batch of 2b (reg_so3net.py:31-33)."""
    MODEL = "reg"
    HEAD_TAKES_BF16 = True      # RelSO3OutBlockR starts with PointnetSO3Conv: bf16 GEMMs on the bf16 features, fp32 from its
                                # max over points onwards (ops.pointnet_so3conv; its fused fall-back converts by itself)

    def __init__(self, layers, out_mlps=(256, 128, 64), kanchor=60, representation='quat', temperature=3.0,
                 fused_glue=True, dropout_rate=0.0):
        super().__init__(layers, kanchor, None, fused_glue, dropout_rate)
        self.outblock = RelSO3OutBlockR(dict(dim_in=layers[-1].cout, mlp=list(out_mlps), fc=[64], k=40,
                                             kanchor=kanchor, representation=representation, temperature=temperature))

    def forward(self, x):
        x = self.features(torch.cat((x[:, 0], x[:, 1]), dim=0))
        f1, f2 = torch.chunk(x.feats, 2, dim=0)
        x1, x2 = torch.chunk(x.xyz, 2, dim=0)
        return self.outblock(f1, f2, x1, x2)


def build_cls(input_num=1024, width_div=1, **kw):
    """cls_so3net_pn.build_model defaults (mlps [[64,64],[128,128],[256,256],[256]], out_mlps [256])."""
    layers = S.scaled(S.cls_so3net_schedule(input_num), width_div)
    return ClsSO3ConvModel(layers, out_mlps=(max(256 // width_div, 1),), **kw)


def build_reg(input_num=1024, width_div=1, **kw):
    """reg_so3net.build_model defaults (mlps [[32,32],[64,64],[128,128],[256]], out_mlps [256,128,64])."""
    layers = S.scaled(S.reg_so3net_schedule(input_num), width_div)
    return RegSO3ConvModel(layers, out_mlps=tuple(max(c // width_div, 1) for c in (256, 128, 64)), **kw)


def build_inv(input_num=2048, search_radius=0.4, width_div=1, **kw):
    """inv_so3net_pn.build_model defaults (mlps [[32,32],[64,64],[128,128],[128,128]], out_mlps [128,64])."""
    layers = S.scaled(S.inv_so3net_schedule(input_num, search_radius), width_div)
    return InvSO3ConvModel(layers, out_mlps=tuple(max(c // width_div, 1) for c in (128, 64)), **kw)
