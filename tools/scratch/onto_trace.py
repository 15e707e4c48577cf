"""Per InterSO3ConvSplitFn.backward of one cls step: does the data gradient accumulate in place into the shared input's gradient?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from epn_pointcloud_amd import models as M, ops, schedule as S  # noqa: E402

dev = torch.device("cuda", 0)
model = M.ClsSO3ConvModel(S.cls_so3net_schedule(1024), out_mlps=(256,), pooling="attention").to(dev).train()
pts = S.synthetic_clouds(8, 1024, dev)
labels = torch.arange(8, device=dev) % 40
orig = ops.InterSO3ConvSplitFn._may_write_into


def traced(ctx, gs):
    r = orig(ctx, gs)
    b = gs._base
    print("grad_shared", tuple(gs.shape), "base" if b is not None else "own", None if b is None else (tuple(b.shape), getattr(b, "_epn_private", None)),
          "->", r, flush=True)
    return r


ops.InterSO3ConvSplitFn._may_write_into = staticmethod(traced)
obw = ops.InterSO3ConvSplitFn.backward


def bw(ctx, grad_out, grad_shared=None, _gp=None):
    print("backward: share_input", ctx.share_input, "grad_out", None if grad_out is None else tuple(grad_out.shape),
          "grad_shared", None if grad_shared is None else tuple(grad_shared.shape), flush=True)
    return obw(ctx, grad_out, grad_shared, _gp)


ops.InterSO3ConvSplitFn.backward = staticmethod(bw)
torch.nn.functional.cross_entropy(model(pts)[0], labels).backward()
torch.cuda.synchronize()
