"""Where the time of the packed grouping kernel goes (tuning library, epn_set_kernel_policy(0x800 | bits): 1 no stores,
2 neighbour rows gathered once per segment, 4 no contraction MFMAs).  Results are WRONG by construction.
python tools/group_ablation.py [cls|reg|inv] [f32|bf16] bits..."""
import ctypes
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
        if l.cin >= 32:
            conv = sptk.InterSO3Conv(l.cin, l.cout, 1, l.stride, l.radius, l.sigma, l.nn, lazy_sample=l.lazy).to(dev)
            idx = pctk.ball_query_index(new_xyz, xyz, l.radius, l.nn)
            geo = ops.InterGeometry(xyz, new_xyz, idx, conv.anchors, conv.kernels, conv.sigma)
            f = ops.to_cl(torch.randn(b, l.cin, p1, 60, device=dev).mul_(0.5).to(dt))
            d = geo.desc(l.cin, 16)
            G = torch.empty((b * p2 * 60, l.cin * d.ks), dtype=dt, device=dev)
            ws, wsp, wsn = ops._group_workspace(lib, d, dev)
            ent = ops._entry(lib, "inter_group_packed", dt)
            row = f"L{li} {l.cin:3d} K={l.nn:3d} p2={p2:4d}:"
            for e in exps:
                assert lib.epn_set_kernel_policy((0x800 | e) if e else 0) == 0
                t = timeit(lambda: ent(ctypes.byref(d), ops._cl_ptr(f), G.data_ptr(), wsp, wsn, _lib.stream_of(f)), 8)
                tot[e] += t
                row += f"  [{e}] {t:.3f}"
            lib.epn_set_kernel_policy(0)
            print(row, flush=True)
            del G
        xyz = new_xyz
    print("total ms: " + "  ".join(f"[{e}] {v:.2f}" for e, v in tot.items()))


if __name__ == "__main__":
    main()
