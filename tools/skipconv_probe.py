#!/usr/bin/env python3
"""1x1 skip convolution of every block of the cls schedule: the intra GEMM kernel (kn = 1) against the BLAS library on
the channels-last 2-D view, forward / data gradient / weight gradient separately (perf iteration tool)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epn_pointcloud_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


tot = {}
for cin, cout, p in [(64, 64, 512), (64, 128, 256), (128, 128, 256), (128, 256, 128), (256, 256, 128), (256, 256, 64)]:
    x = torch.randn(32, cin, p, 60, device=dev).contiguous(memory_format=torch.channels_last)
    W = torch.randn(cout, cin, device=dev) / cin ** 0.5
    ident = torch.arange(60, dtype=torch.int32, device=dev).view(60, 1)
    x2 = x.permute(0, 2, 3, 1).reshape(-1, cin)
    y = ops.IntraSO3ConvFn.apply(x, W, ident)
    g = torch.randn_like(y)
    g2 = g.permute(0, 2, 3, 1).reshape(-1, cout)
    xr, Wr = x.clone().requires_grad_(True), W.clone().requires_grad_(True)
    o1 = ops.IntraSO3ConvFn.apply(xr, W, ident)
    o2 = ops.IntraSO3ConvFn.apply(x, Wr, ident)
    r = {"k_fwd": timeit(lambda: ops.IntraSO3ConvFn.apply(x, W, ident)),
         "k_dX": timeit(lambda: torch.autograd.grad(o1, xr, g, retain_graph=True)),
         "k_dW": timeit(lambda: torch.autograd.grad(o2, Wr, g, retain_graph=True)),
         "b_fwd": timeit(lambda: torch.mm(x2, W.t())), "b_dX": timeit(lambda: torch.mm(g2, W)),
         "b_dW": timeit(lambda: torch.mm(g2.t(), x2))}
    print(f"{cin:3d}->{cout:3d} p={p:3d} " + " ".join(f"{k} {v:6.3f}" for k, v in r.items()), flush=True)
    for k, v in r.items():
        tot[k] = tot.get(k, 0.0) + v
print("total", {k: round(v, 2) for k, v in tot.items()})
