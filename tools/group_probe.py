"""epn_inter_group / epn_inter_ungroup per layer of a schedule: ms per call.  python tools/group_probe.py [cls|reg|inv] [f32|bf16]"""
import ctypes
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import epn_pointcloud_amd
from epn_pointcloud_amd import _lib, ops, schedule as S

vgtk = epn_pointcloud_amd.install_vgtk_alias()
import vgtk.pc as pctk
import vgtk.so3conv as sptk


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    model = sys.argv[1] if len(sys.argv) > 1 else "cls"
    dt = torch.bfloat16 if (len(sys.argv) > 2 and sys.argv[2] == "bf16") else torch.float32
    dev = torch.device("cuda:0")
    if model == "cls":
        layers, b, n, scale = S.cls_so3net_schedule(1024), 32, 1024, 1.0
    elif model == "reg":
        layers, b, n, scale = S.reg_so3net_schedule(1024), 64, 1024, 1.0
    else:
        layers, b, n, scale = S.inv_so3net_schedule(2048), 64, 2048, 0.4
    xyz = S.synthetic_clouds(b, n, dev, seed=2913, scale=scale).permute(0, 2, 1).contiguous()
    tg = tu = tp = 0.0
    lib = _lib.get_lib()
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
            t_g = timeit(lambda: ops.inter_group(f.detach(), geo))
            t_p = float("nan")
            d = geo.desc(l.cin, 16)
            if lib.epn_inter_group_packed_ok(ctypes.byref(d)):      # the split convolution's own (packed) column order
                ws, wsp, wsn = ops._group_workspace(lib, d, dev)
                fc, ent = ops.to_cl(f.detach()), ops._entry(lib, "inter_group_packed", dt)
                t_p = timeit(lambda: ent(ctypes.byref(d), ops._cl_ptr(fc), G.data_ptr(), wsp, wsn, _lib.stream_of(fc)))
            t_u = timeit(lambda: torch.autograd.grad(G, f, dG, retain_graph=True))
            gb = G.numel() * G.element_size() / 1e9
            print(f"L{li} {l.cin:3d} K={l.nn:3d} p1={p1:4d} p2={p2:4d}: group {t_g:.3f} ms ({gb / t_g:.2f} TB/s of G), packed "
                  f"{t_p:.3f} ms ({gb / t_p:.2f} TB/s)   ungroup {t_u:.3f} ms", flush=True)
            tg += t_g; tu += t_u; tp += t_p
            del G, dG
        xyz = new_xyz
    print(f"total: group {tg:.2f} ms (packed {tp:.2f}), ungroup {tu:.2f} ms")


if __name__ == "__main__":
    main()
