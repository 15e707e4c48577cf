"""oracle/backbone_ref.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU backbone built from the oracle's materialising restatement (oracle/so3conv_ref.py) with the same
block glue as epn_pointcloud_amd.schedule.SeparableBlock (SPConvNets/utils/base_so3conv.py:168-212).
Used by tests (block-level parity) and by bench.py's cpu_baseline leg (kind "port": the reference has no
CPU path for FPS / ball query / gather and its Python cannot travel to the GPU box)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import so3conv_ref as R


class RefSeparableBlock(nn.Module):
    def __init__(self, l, anchors, kernels_raw, intra_idx):
        super().__init__()
        self.l = l
        self.register_buffer("anchors", anchors)
        self.register_buffer("kernels", R.scaled_kernel_points(kernels_raw, l.radius))
        self.register_buffer("intra_idx", intra_idx)
        self.W_inter = nn.Parameter(torch.zeros(l.cout, l.cin * kernels_raw.shape[0]))
        self.W_intra = nn.Parameter(torch.zeros(l.cout, l.cout * intra_idx.shape[1]))
        self.inter_norm = nn.BatchNorm2d(l.cout)
        self.intra_norm = nn.InstanceNorm2d(l.cout, affine=False)
        self.skip_conv = nn.Conv2d(l.cin, l.cout, 1)
        self.norm = nn.BatchNorm2d(l.cout)

    def forward(self, xyz, feats):
        l = self.l
        _, _, sample_idx, new_xyz, y = R.inter_so3conv(xyz, feats, self.W_inter, self.anchors, self.kernels,
                                                       l.stride, l.radius, l.sigma, l.nn, l.lazy)
        y = F.leaky_relu(self.inter_norm(y))
        y = R.intra_so3conv(y, self.W_intra, self.intra_idx)
        y = F.leaky_relu(self.intra_norm(y))
        skip = feats
        if l.stride > 1:
            skip = R.batched_index_select(skip, 2, sample_idx.long())
        skip = F.leaky_relu(self.norm(self.skip_conv(skip)))
        return new_xyz, y + skip


class RefBackbone(nn.Module):
    def __init__(self, layers, anchors, kernels_raw, intra_idx):
        super().__init__()
        self.blocks = nn.ModuleList([RefSeparableBlock(l, anchors, kernels_raw, intra_idx) for l in layers])

    def load_from_product(self, product_state_dict):
        """Copy parameters of an epn_pointcloud_amd.schedule.HotPathBackbone state_dict."""
        sd = {}
        for k, v in product_state_dict.items():
            v = v.detach().cpu()
            if k.endswith("inter.basic_conv.W"):
                sd[k.replace("inter.basic_conv.W", "W_inter")] = v
            elif k.endswith("intra.basic_conv.W"):
                sd[k.replace("intra.basic_conv.W", "W_intra")] = v
            elif ".inter." in k or ".intra." in k:
                continue
            else:
                sd[k] = v
        missing, unexpected = self.load_state_dict(sd, strict=False)
        assert not unexpected, unexpected
        assert all(m.endswith(("anchors", "kernels", "intra_idx")) for m in missing), missing

    def forward(self, pts):
        xyz = pts.permute(0, 2, 1).contiguous()
        feats = torch.ones(pts.shape[0], 1, pts.shape[1], self.blocks[0].anchors.shape[0])
        for blk in self.blocks:
            xyz, feats = blk(xyz, feats)
        return xyz, feats
