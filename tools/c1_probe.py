"""First layer (cin = 1) of the three benchmark networks, timed cold: forward kernel and the two forms of its weight
gradient (dedicated kernel vs the library's TN GEMM over the saved grouped values).  `python tools/c1_probe.py`"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epn_pointcloud_amd import ops, schedule  # noqa: E402
from epn_pointcloud_amd.vgtk.so3conv import modules as M  # noqa: E402


def timeit(fns, reps=5):
    """Device time per call, plain stream timing (these launches are 0.1-1.5 ms each)."""
    for f in fns:
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for f in fns:
            f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * len(fns))


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    print("EPN_C1_MFMA =", os.environ.get("EPN_C1_MFMA", "1"), flush=True)
    for name, b, n, sched in [("cls", 32, 1024, schedule.cls_so3net_schedule()), ("reg", 64, 1024, schedule.reg_so3net_schedule()),
                              ("inv", 64, 2048, schedule.inv_so3net_schedule())]:
        l = sched[0]
        nn, cout = l.nn, l.cout
        pts = schedule.synthetic_clouds(b, n, dev, scale=0.4 if name == "inv" else 1.0)
        conv = M.InterSO3Conv(l.cin, l.cout, 1, l.stride, l.radius, l.sigma, l.nn, lazy_sample=l.lazy).to(dev)
        sp = schedule.preprocess_input(pts, 60)
        W = conv.basic_conv.W.detach().requires_grad_(True)
        with torch.no_grad():
            _, geo, _, _ = conv(sp)
        outs = {}
        g = None
        for mode in ("gemm", "kernel"):
            os.environ["EPN_C1_DW"] = mode
            Wm = W.detach().clone().requires_grad_(True)
            out = ops.InterSO3ConvFn.apply(sp.feats, Wm, geo)
            g = torch.randn_like(out) if g is None else g
            outs[mode] = torch.autograd.grad(out, [Wm], g)[0].clone()

            def fb(Wm=Wm, g=g):
                o = ops.InterSO3ConvFn.apply(sp.feats, Wm, geo)
                torch.autograd.grad(o, [Wm], g)

            def fw(Wm=Wm):
                with torch.no_grad():
                    ops.InterSO3ConvFn.apply(sp.feats, Wm, geo)
            t_fb = timeit([fb] * 4)
            t_f = timeit([fw] * 4)
            print(f"{name} b={b} n={n} nn={nn} cout={cout} dW={mode}: fwd {t_f:.3f} ms  fwd+dW {t_fb:.3f} ms  -> dW {t_fb - t_f:.3f} ms")
        ref = outs["kernel"]
        print(f"   max|dW_gemm - dW_kernel| / max|dW| = {((outs['gemm'] - ref).abs().max() / ref.abs().max()).item():.2e}")


if __name__ == "__main__":
    main()
