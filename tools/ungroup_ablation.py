"""Where the time of inter_ungroup_shared_kernel goes: the kernel with parts switched off (tuning library,
epn_set_kernel_policy(0x800 | bits): 1 no atomics, 2 no gather-sum phase, 4 no LDS stores of the per-slot tile, 8 no
contraction MFMAs, 16 dG fragments loaded once).  Results are WRONG by construction; only the times mean anything.
python tools/ungroup_ablation.py [cls|reg|inv] [f32|bf16] bits..."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _tuning import use_tuning_lib

use_tuning_lib()
import torch

import epn_pointcloud_amd
from epn_pointcloud_amd import ops, schedule as S, _lib
from group_probe import timeit

vgtk = epn_pointcloud_amd.install_vgtk_alias()
import vgtk.pc as pctk
import vgtk.so3conv as sptk


def main():
    model = sys.argv[1]
    dt = torch.bfloat16 if sys.argv[2] == "bf16" else torch.float32
    exps = [int(v) for v in sys.argv[3:]] or [0]
    dev = torch.device("cuda:0")
    if model == "cls":
        layers, b, n, scale = S.cls_so3net_schedule(1024), 32, 1024, 1.0
    elif model == "reg":
        layers, b, n, scale = S.reg_so3net_schedule(1024), 64, 1024, 1.0
    else:
        layers, b, n, scale = S.inv_so3net_schedule(2048), 64, 2048, 0.4
    xyz = S.synthetic_clouds(b, n, dev, seed=2913, scale=scale).permute(0, 2, 1).contiguous()
    lib = _lib.get_lib()
    tot = {e: 0.0 for e in exps}
    for li, l in enumerate(layers):
        p1 = xyz.shape[2]
        p2 = math.ceil(p1 / l.stride)
        _, new_xyz = pctk.furthest_sample(xyz, p2, l.lazy)
        if l.cin >= 16:
            conv = sptk.InterSO3Conv(l.cin, l.cout, 1, l.stride, l.radius, l.sigma, l.nn, lazy_sample=l.lazy).to(dev)
            idx = pctk.ball_query_index(new_xyz, xyz, l.radius, l.nn)
            geo = ops.InterGeometry(xyz, new_xyz, idx, conv.anchors, conv.kernels, conv.sigma)
            f = ops.to_cl(torch.randn(b, l.cin, p1, 60, device=dev).mul_(0.5).to(dt)).requires_grad_(True)
            G = ops.inter_group(f, geo)
            dG = torch.randn_like(G)
            row = f"L{li} {l.cin:3d} K={l.nn:3d} p2={p2:4d}:"
            for e in exps:
                assert lib.epn_set_kernel_policy((0x800 | e) if e else 0) == 0
                t = timeit(lambda: torch.autograd.grad(G, f, dG, retain_graph=True), 8)
                tot[e] += t
                row += f"  [{e}] {t:.3f}"
            lib.epn_set_kernel_policy(0)
            print(row, flush=True)
            del G, dG
        xyz = new_xyz
    print("total ms: " + "  ".join(f"[{e}] {v:.2f}" for e, v in tot.items()))


if __name__ == "__main__":
    main()
