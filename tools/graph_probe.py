#!/usr/bin/env python3
"""Does a whole training step (forward + backward of the cls network through the C ABI + library GEMMs) capture into
a HIP graph, and what does replaying it save over eager launches?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epn_pointcloud_amd import models as M, schedule as S  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = M.ClsSO3ConvModel(S.cls_so3net_schedule(1024), out_mlps=(256,), pooling="attention").to(dev).train()
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.Adam(params, lr=1e-3, capturable=True)
pts = S.synthetic_clouds(32, 1024, dev)
labels = torch.arange(32, device=dev) % 40


def fwd_bwd():
    loss = torch.nn.functional.cross_entropy(model(pts)[0], labels)
    loss.backward()
    return loss


def timeit(fn, n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        fwd_bwd()
        opt.step()
torch.cuda.current_stream().wait_stream(side)


def eager():
    opt.zero_grad(set_to_none=True)
    fwd_bwd()
    opt.step()


print(f"eager step {timeit(eager):.1f} ms")
g = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    static_loss = fwd_bwd()
    opt.step()
print(f"graph replay {timeit(g.replay):.1f} ms, loss {static_loss.item():.4f}")
