import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from epn_pointcloud_amd import models as M, schedule as S
gpu = torch.device("cuda:0")
torch.manual_seed(11)
model = sys.argv[1] if len(sys.argv) > 1 else "inv"
points, batch = (2048, 64) if model == "inv" else (1024, 64)
net = (M.build_inv if model == "inv" else M.build_reg)(points).to(gpu).train()
pts = S.synthetic_clouds(batch, points, gpu, seed=77, scale=0.4 if model == "inv" else 1.0)
inp = pts.view(batch // 2, 2, points, 3) if model == "reg" else pts
def run():
    for p in net.parameters(): p.grad = None
    out = net(inp)
    loss = (out[0].float() @ out[0].float().t()).square().mean() if model == "inv" else out[0].float().square().mean() + out[1].float().square().mean()
    loss.backward()
    return loss.item(), out[0].detach().float().cpu(), {n: p.grad.detach().float().cpu().clone() for n, p in net.named_parameters() if p.grad is not None}
l32, f32, g32 = run()
l32b, f32b, g32b = run()      # fp32 run-to-run (atomics order)
S.set_feature_dtype(net, torch.bfloat16)
l16, f16, g16 = run()
rel = lambda a, b: ((a - b).norm() / (b.norm() + 1e-30)).item()
print("loss", l32, l16, "out rel", rel(f16, f32), "fp32 rerun out rel", rel(f32b, f32))
for n in g32:
    print(f"{n:60s} |g| {g32[n].norm().item():.3e}  bf16 rel {rel(g16[n], g32[n]):.4f}  fp32-rerun rel {rel(g32b[n], g32[n]):.2e}")

# ---- control: the fp32 network with ONLY the backbone's output rounded to bf16 (the head's input perturbed the way a bf16
# backbone perturbs it): how much do the head's gradients move?
if model == "inv":
    S.set_feature_dtype(net, torch.float32)
    ob = net.outblock
    orig = ob.forward
    def fwd(x):
        from epn_pointcloud_amd.vgtk import spconv as zptk
        f = (x.feats.bfloat16().float() - x.feats).detach() + x.feats      # straight-through bf16 rounding
        return orig(zptk.SphericalPointCloud(x.xyz, f, x.anchors))
    ob.forward = fwd
    l3, f3, g3 = run()
    print("control (fp32 net, head input rounded to bf16): out rel", rel(f3, f32))
    for n in g32:
        if n.startswith("outblock") or "blocks.1.intra" in n:
            print(f"{n:60s} control rel {rel(g3[n], g32[n]):.4f}   bf16 rel {rel(g16[n], g32[n]):.4f}")
