"""oracle/backbone_ref.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU networks built from the oracle's materialising restatement (oracle/so3conv_ref.py): the separable block glue of
SPConvNets/utils/base_so3conv.py:168-212, the stage loop of :129-166 and the three output heads (:358-448, :572-613,
:661-731), with the reference's module tree so that a product (or reference) state_dict loads key for key.
Used by tests (block / model level parity) and by bench.py's cpu_baseline leg (kind "port": the reference has no
CPU path for FPS / ball query / gather and its Python cannot travel to the GPU box)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import so3conv_ref as R

_BUFFERS = ("anchors", "kernels", "intra_idx")


class _W(nn.Module):
    def __init__(self, rows, cols):
        super().__init__()
        self.W = nn.Parameter(torch.zeros(rows, cols))


class _Conv(nn.Module):
    def __init__(self, rows, cols):
        super().__init__()
        self.basic_conv = _W(rows, cols)


class _ConvNorm(nn.Module):
    def __init__(self, rows, cols, norm):
        super().__init__()
        self.conv = _Conv(rows, cols)
        self.norm = norm


def _mk_norm(norm):
    return (lambda c: nn.InstanceNorm2d(c, affine=False)) if norm is None else getattr(nn, norm)


class RefSeparableBlock(nn.Module):
    def __init__(self, l, tables, norm="BatchNorm2d"):
        super().__init__()
        anchors, kernels_raw, intra_idx = tables
        self.l = l
        self.tab = (anchors, R.scaled_kernel_points(kernels_raw, l.radius), intra_idx)
        mk = _mk_norm(norm)
        self.inter_conv = _ConvNorm(l.cout, l.cin * kernels_raw.shape[0], mk(l.cout))
        self.intra_conv = _ConvNorm(l.cout, l.cout * intra_idx.shape[1], nn.InstanceNorm2d(l.cout, affine=False))
        self.skip_conv = nn.Conv2d(l.cin, l.cout, 1)
        self.norm = mk(l.cout)

    def forward(self, xyz, feats):
        l = self.l
        anchors, kernels, intra_idx = self.tab
        _, _, sample_idx, new_xyz, y = R.inter_so3conv(xyz, feats, self.inter_conv.conv.basic_conv.W, anchors, kernels,
                                                       l.stride, l.radius, l.sigma, l.nn, l.lazy)
        y = F.leaky_relu(self.inter_conv.norm(y))
        y = R.intra_so3conv(y, self.intra_conv.conv.basic_conv.W, intra_idx)
        y = F.leaky_relu(self.intra_conv.norm(y))
        skip = feats
        if l.stride > 1:
            skip = R.batched_index_select(skip, 2, sample_idx.long())
        skip = F.leaky_relu(self.norm(self.skip_conv(skip)))
        return new_xyz, y + skip


class _Stage(nn.Module):
    def __init__(self, layers, tables, norm):
        super().__init__()
        self.blocks = nn.ModuleList([RefSeparableBlock(l, tables, norm) for l in layers])


def _stages(layers):
    out, last = [], None
    for l in layers:
        if last is None or l.stage != last:
            out.append([])
            last = l.stage
        out[-1].append(l)
    return out


class _Embed(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.embed = nn.Conv2d(cin + 3, cout, 1)


class RefBackbone(nn.Module):
    """tables = (anchors [a,3,3], raw kernel points [24,3], intra_idx [a,12] long)."""

    def __init__(self, layers, anchors, kernels_raw, intra_idx, norm="BatchNorm2d"):
        super().__init__()
        self.tables = (anchors, kernels_raw, intra_idx)
        self.backbone = nn.ModuleList([_Stage(st, self.tables, norm) for st in _stages(layers)])

    def load_from_product(self, product_state_dict):
        """Copy the parameters of a product (epn_pointcloud_amd.schedule / .models) or reference state_dict."""
        sd = {k: v.detach().cpu() for k, v in product_state_dict.items() if not k.endswith(_BUFFERS)}
        missing, unexpected = self.load_state_dict(sd, strict=False)
        assert not unexpected and not missing, (missing, unexpected)

    def features(self, pts):
        xyz = pts.permute(0, 2, 1).contiguous()
        feats = torch.ones(pts.shape[0], 1, pts.shape[1], self.tables[0].shape[0])
        for st in self.backbone:
            for blk in st.blocks:
                xyz, feats = blk(xyz, feats)
        return xyz, feats

    def forward(self, pts):
        return self.features(pts)


class RefClsModel(RefBackbone):
    """ClsSO3ConvModel with ClsOutBlockPointnet (cls_so3net_pn.py:15-40, base_so3conv.py:358-448)."""

    def __init__(self, layers, tables, out_mlps=(256,), k=40, pooling="attention", temperature=3.0):
        super().__init__(layers, *tables, norm="BatchNorm2d")
        ob = nn.Module()
        ob.linear, ob.norm = nn.ModuleList(), nn.ModuleList()
        c = layers[-1].cout
        for m in out_mlps:
            ob.linear.append(nn.Conv2d(c, m, 1)); ob.norm.append(nn.BatchNorm2d(m)); c = m
        if pooling == "attention":
            ob.attention_layer = nn.Conv1d(c, 1, 1)
        ob.pointnet = _Embed(c, c)
        ob.norm.append(nn.BatchNorm1d(c))
        ob.fc2 = nn.Linear(c, k)
        self.outblock, self.pooling, self.temperature = ob, pooling, temperature

    def forward(self, pts):
        xyz, f = self.features(pts)
        ob = self.outblock
        for lin, norm in zip(ob.linear, ob.norm):
            f = F.relu(norm(lin(f)))
        e = ob.pointnet.embed
        y = F.relu(ob.norm[len(ob.linear)](R.pointnet_so3conv(xyz, f, self.tables[0], e.weight, e.bias)))
        if self.pooling == "attention":
            att = ob.attention_layer(y)
            return ob.fc2((y * F.softmax(att * self.temperature, dim=2)).sum(-1)), att.squeeze()
        y = y.max(2)[0] if self.pooling == "max" else y.mean(2)
        return ob.fc2(y), f


class RefInvModel(RefBackbone):
    """InvSO3ConvModel with InvOutBlockMVD (inv_so3net_pn.py:15-41, base_so3conv.py:572-613)."""

    def __init__(self, layers, tables, out_mlps=(128, 64)):
        super().__init__(layers, *tables, norm=None)
        ob = nn.Module()
        c = layers[-1].cout
        ob.attention_layer = nn.Sequential(nn.Conv2d(c, c, 1), nn.ReLU(), nn.Conv2d(c, c, 1))
        ob.pointnet = _Embed(c, out_mlps[-1])
        self.outblock = ob

    def forward(self, pts):
        xyz, f = self.features(pts)
        attn = F.softmax(self.outblock.attention_layer(f), dim=3)
        pooled = (f * attn).sum(-1, keepdim=True)
        e = self.outblock.pointnet.embed
        y = R.pointnet_so3conv(xyz, pooled, None, e.weight, e.bias).reshape(f.shape[0], -1)
        return F.normalize(y, p=2, dim=1), attn


class RefRegModel(RefBackbone):
    """RegSO3ConvModel with RelSO3OutBlockR (reg_so3net.py:16-48, base_so3conv.py:661-731)."""

    def __init__(self, layers, tables, out_mlps=(256, 128, 64), n_out=4, temperature=3.0):
        super().__init__(layers, *tables, norm=None)
        ob = nn.Module()
        c = layers[-1].cout
        ob.pointnet = _Embed(c, c)
        ob.attention_layer = nn.Conv2d(out_mlps[-1], 1, 1)
        ob.regressor_layer = nn.Conv2d(out_mlps[-1], n_out, 1)
        ob.linear = nn.ModuleList()
        c *= 2
        for m in out_mlps:
            ob.linear.append(nn.Conv2d(c, m, 1)); c = m
        self.outblock, self.temperature = ob, temperature

    def forward(self, pairs):
        xyz, f = self.features(torch.cat((pairs[:, 0], pairs[:, 1]), 0))
        f1, f2 = torch.chunk(f, 2, 0)
        x1, x2 = torch.chunk(xyz, 2, 0)
        ob = self.outblock
        e = ob.pointnet.embed
        c1 = F.relu(R.pointnet_so3conv(x1, f1, self.tables[0], e.weight, e.bias))
        c2 = F.relu(R.pointnet_so3conv(x2, f2, self.tables[0], e.weight, e.bias))
        nb, _, na = c1.shape
        x = torch.cat((c1.unsqueeze(-2).expand(-1, -1, na, -1), c2.unsqueeze(-1).expand(-1, -1, -1, na)), 1)
        for lin in ob.linear:
            x = F.relu(lin(x))
        conf = F.softmax(ob.attention_layer(x).view(nb, na, na) * self.temperature, dim=1)
        return conf, ob.regressor_layer(x)
