#!/usr/bin/env python3
"""Per-call breakdown of one training step of the cls network (HIP events around every C-ABI call and library GEMM):
which layer / shape costs what.  usage: [EPN_SB_DTYPE=bf16] [EPN_SB_POLICY=0x401] tools/step_breakdown.py [substring filter]"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("EPN_SB_POLICY"):
    from _tuning import use_tuning_lib
    use_tuning_lib()                          # A/B policies: the -DEPN_TUNING library
from epn_pointcloud_amd import models as M, schedule as S, ops, _lib  # noqa: E402

if os.environ.get("EPN_SB_POLICY"):
    _lib.check(_lib.get_lib().epn_set_kernel_policy(int(os.environ["EPN_SB_POLICY"], 0)), "set_kernel_policy")

flt = sys.argv[1] if len(sys.argv) > 1 else ""
dev = torch.device("cuda", 0)
model = M.ClsSO3ConvModel(S.cls_so3net_schedule(1024), out_mlps=(256,), pooling="attention").to(dev).train()
if os.environ.get("EPN_SB_DTYPE") == "bf16":
    S.set_feature_dtype(model, torch.bfloat16)
pts = S.synthetic_clouds(32, 1024, dev)
labels = torch.arange(32, device=dev) % 40


def step():
    torch.nn.functional.cross_entropy(model(pts)[0], labels).backward()


step(); step(); torch.cuda.synchronize()
ops.profile_begin(); step(); torch.cuda.synchronize()
rec = ops.profile_end()
agg, fam = collections.defaultdict(lambda: [0, 0.0]), collections.defaultdict(float)
for kind, key, fl, e0, e1, _kernel in rec:
    ms = e0.elapsed_time(e1)
    k = (kind,) + tuple(key)[:8]
    agg[k][0] += 1; agg[k][1] += ms; fam[kind] += ms
    agg[k].append(fl) if len(agg[k]) == 2 else None
print({k: round(v, 2) for k, v in sorted(fam.items(), key=lambda kv: -kv[1])}, "sum", round(sum(fam.values()), 1))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if flt in k[0]:
        tf = v[2] * v[0] / (v[1] * 1e-3) / 1e12 if len(v) > 2 and v[1] > 0 else 0.0
        print(f"{v[1]:7.3f} ms x{v[0]} {tf:6.1f} TF  {k}")
