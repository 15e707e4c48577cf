import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from epn_pointcloud_amd import models as M, schedule as S
model_name = sys.argv[1] if len(sys.argv) > 1 else "reg"
dev = torch.device("cuda:0")
torch.manual_seed(0)
if model_name == "reg":
    layers = S.reg_so3net_schedule(1024); model = M.RegSO3ConvModel(layers); dt = torch.bfloat16; B = 8
else:
    layers = S.cls_so3net_schedule(1024); model = M.ClsSO3ConvModel(layers, out_mlps=(256,), pooling="attention"); dt = torch.float32; B = 4
model = S.set_feature_dtype(model.to(dev).train(), dt)
pts = S.synthetic_clouds(B, 1024, dev, seed=1)
if model_name == "reg":
    pts = pts.view(B // 2, 2, 1024, 3)
def step():
    for p in model.parameters(): p.grad = None
    out = model(pts)
    loss = out[0].square().mean() + (out[1].square().mean() if model_name == "reg" else 0)
    loss.backward()
step(); torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::fill_", "aten::zero_", "aten::zeros", "aten::zeros_like", "aten::full"):
        st = [s for s in (e.stack or []) if "epn_pointcloud_amd" in s or "autograd" in s.lower()][:3]
        shp = str(e.input_shapes)[:40]
        cnt[(e.name, shp, tuple(s.split("/")[-1][:70] for s in st))] += 1
for (k, v) in cnt.most_common(40):
    print(v, k)
