"""On-chip InterSO3Conv (csrc/inter_fx.hip) against the split form (inter_group + gemm_nt) on the schedules' layers:
max |difference| and time per call.  python tools/fx_probe.py [cls|reg|inv] [f32|bf16]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import epn_pointcloud_amd
from epn_pointcloud_amd import ops, schedule as S

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
    p1 = n
    tot_s = tot_o = 0.0
    for li, l in enumerate(layers):
        p2 = math.ceil(p1 / l.stride)
        if l.cin >= 16:
            torch.manual_seed(li)
            conv = sptk.InterSO3Conv(l.cin, l.cout, 1, l.stride, l.radius, l.sigma, l.nn, lazy_sample=l.lazy).to(dev)
            f = ops.to_cl(torch.randn(b, l.cin, p1, 60, device=dev).mul_(0.5).to(dt))
            _, new_xyz = pctk.furthest_sample(xyz, p2, l.lazy)
            idx = pctk.ball_query_index(new_xyz, xyz, l.radius, l.nn)
            geo = ops.InterGeometry(xyz, new_xyz, idx, conv.anchors, conv.kernels, conv.sigma)
            W = conv.basic_conv.W.detach()
            ys = ops.InterSO3ConvSplitFn.apply(f, W, geo)
            ok = ops.inter_onchip_ok(f, W, geo)
            if ok:
                yo = ops.inter_onchip_fwd(f, W.contiguous(), geo)
                err = (yo.float() - ys.float()).abs().max().item()
                ref = ys.float().abs().max().item()
                ts = timeit(lambda: ops.InterSO3ConvSplitFn.apply(f, W, geo))
                to = timeit(lambda: ops.inter_onchip_fwd(f, W.contiguous(), geo))
                tot_s += ts; tot_o += to
                cols = b * p2 * 60
                fl = 2.0 * cols * l.cout * l.cin * 24
                print(f"L{li} {l.cin:3d}->{l.cout:3d} K={l.nn:3d} p1={p1:4d} p2={p2:4d}: max|onchip-split| {err:.2e} (max|y| {ref:.2f})  "
                      f"split {ts:.3f} ms  onchip {to:.3f} ms  ({fl / to / 1e9:.0f} TFLOP/s gemm-equivalent)", flush=True)
            else:
                extra = ""
                if os.environ.get("FX_ABL"):
                    from epn_pointcloud_amd import _lib
                    for pol, nm in ((0x401, "no-gather"), (0x402, "no-table"), (0x403, "no-loads"), (0x407, "no-prod"), (0x408, "no-gemm"), (0x40f, "nothing")):
                        _lib.get_lib().epn_set_kernel_policy(pol)
                        extra += f" {nm} {timeit(lambda: ops.inter_onchip_fwd(f, W.contiguous(), geo)):.3f}"
                    _lib.get_lib().epn_set_kernel_policy(0)
                print(extra, end="")
                print(f"L{li}: not served by the on-chip form")
        if l.stride > 1:
            _, xyz = pctk.furthest_sample(xyz, p2, l.lazy)
        p1 = p2
    print(f"total: split {tot_s:.2f} ms, onchip {tot_o:.2f} ms")


if __name__ == "__main__":
    main()
