"""Which fp32 <-> bf16 feature casts does one bf16 training step make?  (shape, direction, caller)"""
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from epn_pointcloud_amd import models as M, ops, schedule as S, gemm  # noqa: E402

dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "reg"
points = 2048 if name == "inv" else 1024
layers = {"reg": S.reg_so3net_schedule, "inv": S.inv_so3net_schedule, "cls": S.cls_so3net_schedule}[name](points)
torch.manual_seed(2913)
model = {"reg": M.RegSO3ConvModel, "inv": M.InvSO3ConvModel}[name](layers)
model = S.set_feature_dtype(model.to(dev).train(), torch.bfloat16)
b = 8
pts = S.synthetic_clouds(b, points, dev, scale=0.4 if name == "inv" else 1.0)
if name == "reg":
    pts = pts.view(b // 2, 2, points, 3)
seen = collections.Counter()
orig_f, orig_c = ops.CastFn.forward, gemm.cast


def where():
    st = [f for f in traceback.extract_stack()[:-2] if "epn_pointcloud_amd" in f.filename]
    return " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in st[-3:])


def fwd(ctx, x, dtype):
    seen[(tuple(x.shape), str(x.dtype), "feat", where())] += 1
    return orig_f(ctx, x, dtype)


def cst(t, dtype):
    if t.numel() > 1 << 20:
        seen[(tuple(t.shape), str(t.dtype), "gemm.cast", where())] += 1
    return orig_c(t, dtype)


ops.CastFn.forward = staticmethod(fwd)
gemm.cast = cst
out = model(pts)
loss = out[0].square().mean() + (out[1].square().mean() if name == "reg" else 0)
loss.backward()
torch.cuda.synchronize()
for k, v in sorted(seen.items(), key=lambda kv: -torch.Size(kv[0][0]).numel()):
    print(v, k)
