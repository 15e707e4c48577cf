"""Per-layer A/B of the two transposes of the grouping: inter_ungroup_shared_kernel (LDS pre-reduction over 8-16 output points +
fp32 atomics, epn_inter_ungroup_*) against inter_ungroup_cloud_kernel (a cloud's gradient rows resident in LDS as 64-bit
fixed point, epn_inter_ungroup_cloud_*), on the layers of a schedule: time, agreement, repeatability.
python tools/ungroup_cloud_probe.py [reg|inv|cls] [bf16|f32]"""
import ctypes
import math
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import epn_pointcloud_amd
from epn_pointcloud_amd import _lib, ops, schedule as S

vgtk = epn_pointcloud_amd.install_vgtk_alias()
import vgtk.pc as pctk
import vgtk.so3conv as sptk


def timed(fn, n=6):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    model = sys.argv[1] if len(sys.argv) > 1 else "reg"
    dt = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else torch.bfloat16
    dev = torch.device("cuda:0")
    if model == "cls":
        layers, b, n, scale = S.cls_so3net_schedule(1024), 32, 1024, 1.0
    elif model == "reg":
        layers, b, n, scale = S.reg_so3net_schedule(1024), 64, 1024, 1.0
    else:
        layers, b, n, scale = S.inv_so3net_schedule(2048), 64, 2048, 0.4
    xyz = S.synthetic_clouds(b, n, dev, seed=2913, scale=scale).permute(0, 2, 1).contiguous()
    lib = _lib.get_lib()
    bf = dt == torch.bfloat16
    print(f"{model} {dt}")
    print(f"{'layer':34s} {'pair ms':>8s} {'(+cast)':>8s} {'cloud ms':>9s} {'dG TB/s':>8s} {'max err / max|dF|':>18s} {'repeatable':>10s} {'range':>6s}")
    tot = [0.0, 0.0]
    for li, l in enumerate(layers):
        p1 = xyz.shape[2]
        p2 = math.ceil(p1 / l.stride)
        _, new_xyz = pctk.furthest_sample(xyz, p2, l.lazy)
        if l.cin >= 16:
            conv = sptk.InterSO3Conv(l.cin, l.cout, 1, l.stride, l.radius, l.sigma, l.nn, lazy_sample=l.lazy).to(dev)
            idx = pctk.ball_query_index(new_xyz, xyz, l.radius, l.nn)
            geo = ops.InterGeometry(xyz, new_xyz, idx, conv.anchors, conv.kernels, conv.sigma)
            d = geo.desc(l.cin, l.cout)
            G = (torch.randn(b * p2 * 60, l.cin * 24, device=dev) * 1e-3).to(dt)
            amax = G.abs().max().float().reshape(1)
            gf = ops.empty_cl(b, l.cin, p1, 60, dev)
            ws = torch.empty(max(int(lib.epn_inter_group_workspace_bytes(ctypes.byref(d))), 16), dtype=torch.uint8, device=dev)
            fn = lib.epn_inter_ungroup_bf16 if bf else lib.epn_inter_ungroup_f32

            def pair():
                _lib.check(fn(ctypes.byref(d), G.data_ptr(), ops._cl_ptr(gf), ws.data_ptr(), ws.numel(), _lib.stream_of(G)), "ungroup")
            t_pair = timed(pair)
            t_cast = timed(lambda: (pair(), ops.cast_feats(gf, dt))) if bf else t_pair
            if not lib.epn_inter_ungroup_cloud_ok(ctypes.byref(d)):
                print(f"L{li} {l.cin:3d}->{l.cout:3d} K={l.nn:2d} p1={p1:4d} p2={p2:4d}: not taken")
                xyz = new_xyz
                continue
            out = ops.empty_cl(b, l.cin, p1, 60, dev, dt)
            out2 = ops.empty_cl(b, l.cin, p1, 60, dev, dt)
            ws2 = torch.empty(int(lib.epn_inter_ungroup_cloud_workspace_bytes(ctypes.byref(d))), dtype=torch.uint8, device=dev)
            fc = lib.epn_inter_ungroup_cloud_bf16 if bf else lib.epn_inter_ungroup_cloud_f32

            def cloud(o=out):
                extra = (0,) if bf else ()
                _lib.check(fc(ctypes.byref(d), G.data_ptr(), amax.data_ptr(), ops._cl_ptr(o), None, *extra, ws2.data_ptr(), ws2.numel(),
                              _lib.stream_of(G)), "ungroup_cloud")
            t_cloud = timed(cloud)
            cloud(out2)
            torch.cuda.synchronize()
            ref = gf.float()
            err = (out.float() - ref).abs().max().item() / ref.abs().max().item()
            rep = bool(torch.equal(out, out2))
            rng = int(lib.epn_inter_ungroup_cloud_range_count(1))
            tot[0] += t_cast; tot[1] += t_cloud
            print(f"L{li} {l.cin:3d}->{l.cout:3d} K={l.nn:2d} p1={p1:4d} p2={p2:4d}     {t_pair:8.3f} {t_cast:8.3f} {t_cloud:9.3f} "
                  f"{G.numel() * G.element_size() / t_cloud / 1e9:8.2f} {err:18.2e} {str(rep):>10s} {rng:6d}", flush=True)
            del G, gf, out, out2
        xyz = new_xyz
    print(f"sum: pair(+cast) {tot[0]:.3f} ms, cloud {tot[1]:.3f} ms")


if __name__ == "__main__":
    main()
