"""Debug: the pack form of dp.GradBuckets against plain autograd gradients on the cls network, step by step (eager)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epn_pointcloud_amd import dp, models as M, schedule as S


def build(dev):
    layers = S.cls_so3net_schedule(1024)
    torch.manual_seed(2913)
    m = M.ClsSO3ConvModel(layers, out_mlps=(256,), pooling="attention")
    return S.set_feature_dtype(m.to(dev).train(), torch.float32)


def main():
    dev = torch.device("cuda:0")
    B = int(os.environ.get("B", "32"))
    pts = S.synthetic_clouds(B, 1024, dev, seed=2913)
    labels = torch.arange(B, device=dev) % 40
    ma, mb = build(dev), build(dev)
    mb.load_state_dict(ma.state_dict())
    pa = [p for p in ma.parameters() if p.requires_grad]
    pb = [p for p in mb.parameters() if p.requires_grad]
    oa, ob = torch.optim.Adam(pa, lr=1e-3), torch.optim.Adam(pb, lr=1e-3)
    buckets = dp.GradBuckets(dp.stage_buckets(mb), 1, hooks=False, collect="pack", force_collectives=True)
    names = [n for n, p in ma.named_parameters() if p.requires_grad]
    side = torch.cuda.Stream(device=dev)
    use_side = os.environ.get("SIDE", "1") == "1"
    for it in range(int(os.environ.get("STEPS", "8"))):
        ctxm = torch.cuda.stream(side) if use_side else torch.cuda.stream(torch.cuda.current_stream())
        if use_side:
            side.wait_stream(torch.cuda.current_stream())
        with ctxm:
            for p in pa:
                p.grad = None
            la = torch.nn.functional.cross_entropy(ma(pts)[0], labels)
            la.backward()
            buckets.zero()
            lb = torch.nn.functional.cross_entropy(mb(pts)[0], labels)
            lb.backward()
            buckets.pack()
            torch.cuda.synchronize()
            worst, wn, nan = 0.0, None, []
            for n, p, q in zip(names, pa, pb):
                if p.grad is None or q.grad is None:
                    if (p.grad is None) != (q.grad is None):
                        print("  grad None mismatch", n, p.grad is None, q.grad is None)
                    continue
                if not torch.isfinite(q.grad).all():
                    nan.append(n)
                e = ((p.grad - q.grad).norm() / p.grad.norm().clamp_min(1e-30)).item()
                if e > worst:
                    worst, wn = e, n
            print(f"step {it}: loss plain {la.item():.6g} pack {lb.item():.6g}  worst rel grad diff {worst:.3e} ({wn})  non-finite: {nan[:4]}", flush=True)
            oa.step(); ob.step()
        if use_side:
            torch.cuda.current_stream().wait_stream(side)


if __name__ == "__main__":
    main()
