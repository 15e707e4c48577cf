#!/usr/bin/env python3
"""Which torch (aten) kernels one training step of the cls network still launches, by op and input shape:
`python tools/aten_ops.py [cls|reg|inv]` (torch.profiler, one eager step)."""
import os
import sys

import torch
from torch.profiler import profile, ProfilerActivity

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epn_pointcloud_amd import models as M, schedule as S  # noqa: E402

dev = torch.device("cuda", 0)
name = sys.argv[1] if len(sys.argv) > 1 else "cls"      # cls (fp32) | reg | inv (bf16 features)
if name == "cls":
    model = M.ClsSO3ConvModel(S.cls_so3net_schedule(1024), out_mlps=(256,), pooling="attention").to(dev).train()
    pts = S.synthetic_clouds(32, 1024, dev)
    labels = torch.arange(32, device=dev) % 40
elif name == "reg":
    model = S.set_feature_dtype(M.RegSO3ConvModel(S.reg_so3net_schedule(1024)).to(dev).train(), torch.bfloat16)
    pts = S.synthetic_clouds(64, 1024, dev).view(32, 2, 1024, 3)
else:
    model = S.set_feature_dtype(M.InvSO3ConvModel(S.inv_so3net_schedule(2048)).to(dev).train(), torch.bfloat16)
    pts = S.synthetic_clouds(64, 2048, dev, scale=0.4)


def step():
    out = model(pts)
    if name == "cls":
        torch.nn.functional.cross_entropy(out[0], labels).backward()
    elif name == "reg":
        (out[0].square().mean() + out[1].square().mean()).backward()
    else:
        (out[0] @ out[0].t()).square().mean().backward()


step(); step(); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") and e.device_time_total > 0]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:45]:
    print(f"{e.count:4d} x {e.device_time_total / max(e.count, 1):7.1f} us  {e.key:28s} {str(e.input_shapes)[:110]}")
