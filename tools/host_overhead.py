#!/usr/bin/env python3
"""Host-side cost of one training step (time for Python to enqueue it) vs its GPU time: headroom check for the
one-process-per-GPU launch with few CPU cores per rank."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epn_pointcloud_amd import models as M, schedule as S  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = M.ClsSO3ConvModel(S.cls_so3net_schedule(1024), out_mlps=(256,), pooling="attention").to(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
pts = S.synthetic_clouds(32, 1024, dev)
labels = torch.arange(32, device=dev) % 40


def step():
    opt.zero_grad(set_to_none=True)
    loss = torch.nn.functional.cross_entropy(model(pts)[0], labels)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
if len(sys.argv) > 1:
    torch.set_num_threads(int(sys.argv[1]))
host, total = [], []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
print(f"host enqueue {sum(host) / len(host):.1f} ms/step, step {sum(total) / len(total):.1f} ms "
      f"(cpus visible {len(os.sched_getaffinity(0))})")
