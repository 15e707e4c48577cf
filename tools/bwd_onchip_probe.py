"""Per-layer A/B of the data gradient of InterSO3Conv (review item 1a, round 5): the split pair -- dG = dOut W on the two-piece
GEMM, then the LDS-reduced transpose of the grouping -- against the on-chip kernel (csrc/inter_bwd_f2.hip) that never writes
dG.  Same inputs, same process, one box; times by HIP events around the C-ABI calls, max|difference| of the two gradients.
python tools/bwd_onchip_probe.py [cls|reg|inv]   (fp32 features)"""
import ctypes
import math
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
ABL = [int(v, 0) for v in sys.argv[2:]]          # ablation bits of the on-chip kernel (tuning library): 1 no tail, 2 no MFMAs /
if ABL:                                          # fragment reads, 4 no staging, 8 no atomics -- times only, results are wrong
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _tuning import use_tuning_lib
    use_tuning_lib()
import torch

import epn_pointcloud_amd
from epn_pointcloud_amd import _lib, gemm, ops, schedule as S

vgtk = epn_pointcloud_amd.install_vgtk_alias()
import vgtk.pc as pctk
import vgtk.so3conv as sptk


def timeit(fn, n=6):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    model = sys.argv[1] if len(sys.argv) > 1 else "cls"
    dev = torch.device("cuda:0")
    if model == "cls":
        layers, b, n, scale = S.cls_so3net_schedule(1024), 32, 1024, 1.0
    elif model == "reg":
        layers, b, n, scale = S.reg_so3net_schedule(1024), 64, 1024, 1.0
    else:
        layers, b, n, scale = S.inv_so3net_schedule(2048), 64, 2048, 0.4
    xyz = S.synthetic_clouds(b, n, dev, seed=2913, scale=scale).permute(0, 2, 1).contiguous()
    lib = _lib.get_lib()
    tot = [0.0, 0.0, 0.0]
    print(f"{'layer':34s} {'dG GEMM':>9s} {'transpose':>9s} {'pair':>8s} {'on-chip':>8s} {'ratio':>6s}  max|diff|/max|g|  kernel")
    for li, l in enumerate(layers):
        p1 = xyz.shape[2]
        p2 = math.ceil(p1 / l.stride)
        _, new_xyz = pctk.furthest_sample(xyz, p2, l.lazy)
        if l.cin >= 16:
            conv = sptk.InterSO3Conv(l.cin, l.cout, 1, l.stride, l.radius, l.sigma, l.nn, lazy_sample=l.lazy).to(dev)
            idx = pctk.ball_query_index(new_xyz, xyz, l.radius, l.nn)
            geo = ops.InterGeometry(xyz, new_xyz, idx, conv.anchors, conv.kernels, conv.sigma)
            d = geo.desc(l.cin, l.cout)
            W = conv.basic_conv.W.detach().contiguous()
            cols, ck = b * p2 * 60, l.cin * 24
            g = ops.to_cl(torch.randn(b, l.cout, p2, 60, device=dev) * 1e-3)
            g2d = g.permute(0, 2, 3, 1).reshape(cols, l.cout)
            am = gemm.absmax(g)
            Wt = gemm.transpose_cast(W, torch.float32)
            gws = torch.empty(max(int(lib.epn_inter_group_workspace_bytes(ctypes.byref(d))), 16), dtype=torch.uint8, device=dev)
            gf_s = ops.empty_cl(b, l.cin, p1, 60, dev)
            gf_o = ops.empty_cl(b, l.cin, p1, 60, dev)
            dG = torch.empty(cols, ck, device=dev)

            def gemm_dg():
                gemm.gemm_nt(g2d, Wt, out=dG, a_amax=am)

            def ungroup():
                _lib.check(lib.epn_inter_ungroup_f32(ctypes.byref(d), dG.data_ptr(), ops._cl_ptr(gf_s), gws.data_ptr(), gws.numel(),
                                                     _lib.stream_of(g)), "ungroup")
            row = f"L{li} {l.cin:3d}->{l.cout:3d} K={l.nn:2d} p2={p2:4d} cols={cols:7d}"
            t_g, t_u = timeit(gemm_dg), timeit(ungroup)
            t_o, diff, kern = float("nan"), float("nan"), "-"
            if lib.epn_inter_bwd_data_f16x2_ok(ctypes.byref(d)):
                ows = torch.empty(int(lib.epn_inter_bwd_data_f16x2_workspace_bytes(ctypes.byref(d))), dtype=torch.uint8, device=dev)

                def onchip():
                    _lib.check(lib.epn_inter_bwd_data_f16x2_f32(ctypes.byref(d), ops._cl_ptr(g), W.data_ptr(), am.data_ptr(),
                                                                ops._cl_ptr(gf_o), 0, ows.data_ptr(), ows.numel(), _lib.stream_of(g)),
                               "bwd_data_f16x2")
                t_o = timeit(onchip)
                kern = lib.epn_last_kernel().decode().split("::")[-1]
                for e in ABL:
                    assert lib.epn_set_kernel_policy(0x900 | e) == 0
                    kern += f"  [{e}] {timeit(onchip):.3f}"
                    lib.epn_set_kernel_policy(0)
                    onchip()
                diff = ((gf_o - gf_s).abs().max() / gf_s.abs().max()).item()
                tot[2] += t_o
            tot[0] += t_g; tot[1] += t_u
            print(f"{row:34s} {t_g:9.3f} {t_u:9.3f} {t_g + t_u:8.3f} {t_o:8.3f} {t_o / (t_g + t_u):6.2f}  {diff:.2e}  {kern}", flush=True)
            del dG, gf_s, gf_o
        xyz = new_xyz
    print(f"total ms: dG GEMM {tot[0]:.2f} + transpose {tot[1]:.2f} = {tot[0] + tot[1]:.2f}   on-chip {tot[2]:.2f}")
    print("f16x2 overflow count:", gemm.f16x2_overflow_count())


if __name__ == "__main__":
    main()
