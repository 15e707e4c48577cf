"""Which tensors does a classification step still scan for max|x| (gemm.absmax passes) -- shape, MB and the call site.
One eager fwd+bwd of the cls network; producer-side maxima (ops._tag_amax) do not show up here.
usage (GPU box): python tools/amax_trace.py [cls|reg]"""
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epn_pointcloud_amd import gemm, models as M, schedule as S  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "cls"
    dev = torch.device("cuda:0")
    layers = {"cls": S.cls_so3net_schedule, "reg": S.reg_so3net_schedule}[which](1024)
    torch.manual_seed(2913)
    model = (M.ClsSO3ConvModel(layers, out_mlps=(256,), pooling="attention") if which == "cls" else M.RegSO3ConvModel(layers))
    model = S.set_feature_dtype(model.to(dev).train(), torch.float32)
    batch = 32
    pts = S.synthetic_clouds(batch, 1024, dev, seed=2913)
    if which == "reg":
        pts = pts.view(batch // 2, 2, 1024, 3)
    labels = torch.arange(batch, device=dev) % 40
    seen = collections.OrderedDict()
    orig = gemm.absmax

    def traced(t):
        fr = [f for f in traceback.extract_stack()[:-1] if "epn_pointcloud_amd" in f.filename and "absmax" not in f.name]
        site = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}:{f.name}" for f in fr[-3:][::-1])
        k = (tuple(t.shape), site)
        seen[k] = seen.get(k, 0) + 1
        return orig(t)

    gemm.absmax = traced
    for it in range(2):
        seen.clear()
        out = model(pts)
        loss = torch.nn.functional.cross_entropy(out[0], labels) if which == "cls" else out[0].square().mean() + out[1].square().mean()
        loss.backward()
    torch.cuda.synchronize()
    tot = 0.0
    for (shape, site), n in seen.items():
        mb = 4.0
        for s in shape:
            mb *= s
        mb /= 1e6
        tot += mb * n
        print(f"{n} x {str(shape):28s} {mb:8.1f} MB  {site}")
    print(f"python-side passes: {sum(seen.values())}, {tot:.0f} MB scanned")


if __name__ == "__main__":
    main()
