"""Weight-gradient (TN) GEMMs of the benchmark schedules timed COLD: every launch reads a different copy of its operands,
enough copies that the set exceeds the 256 MB infinity cache (an isolated timing loop over one buffer pair reports the
narrow 1x1-convolution shapes 2-3x faster than they run inside a training step).
`python tools/tn_probe.py [--dtype bf16|f32] [--set conv|spectral|inter|all]`  (f32 = the split form used by the benchmark)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epn_pointcloud_amd import gemm  # noqa: E402


def timeit(fns, reps=3):
    """Device time per call: the calls are captured into one graph (no host launch gaps between these 50-500 us kernels)."""
    for f in fns:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for f in fns:
            f()
        st.synchronize()
        with torch.cuda.graph(g, stream=st):
            for f in fns:
                f()
        g.replay()
        st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps):
            g.replay()
        e1.record(st)
        st.synchronize()
    return e0.elapsed_time(e1) / (reps * len(fns))


def copies(nbytes):
    return max(2, min(12, int(600e6 // max(nbytes, 1)) + 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--set", default="all")
    a = ap.parse_args()
    dt = torch.float32 if a.dtype == "f32" else torch.bfloat16
    esz = 4 if a.dtype == "f32" else 2
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    if a.set in ("conv", "all"):
        for (R, N1, N2) in [(983040, 64, 64), (1966080, 32, 32), (983040, 64, 32), (491520, 128, 128), (491520, 128, 64),
                            (245760, 256, 128), (245760, 256, 256), (115200, 256, 512), (115200, 128, 256)]:
            nb = R * (N1 + N2) * esz
            n = copies(nb)
            Xs = [torch.randn(R, N1, device=dev).to(dt) for _ in range(n)]
            Ys = [torch.randn(R, N2, device=dev).to(dt) for _ in range(n)]
            C = gemm.gemm_tn(Xs[0], Ys[0])
            ref = Xs[0].float().t() @ Ys[0].float()
            err = (C - ref).abs().max().item() / ref.abs().max().item()
            t = timeit([lambda X=X, Y=Y: gemm.gemm_tn(X, Y, out=C) for X, Y in zip(Xs, Ys)])
            print(f"conv1x1 dW {R}x{N1}x{N2}: {t:.3f} ms  {nb / t / 1e6:7.1f} GB/s  {2.0 * R * N1 * N2 / t / 1e9:6.1f} TF  err {err:.1e}"
                  f"  ({n} copies)", flush=True)
            del Xs, Ys
    if a.set in ("spectral", "all"):
        for pts, c in [(32768, 32), (16384, 64), (8192, 128), (4096, 256), (2048, 512)]:
            nb = pts * 60 * 2 * c * esz
            n = copies(nb)
            sets, fl = [], 0.0
            for i in range(n):
                probs = []
                for d in (1, 3, 3, 4, 5):
                    probs.append((torch.randn(pts * d, d * c, device=dev).to(dt), torch.randn(pts * d, d * c, device=dev).to(dt)))
                sets.append(probs)
            fl = sum(2.0 * pts * d * d * c * d * c for d in (1, 3, 3, 4, 5))
            outs = gemm.gemm_tn_grouped(sets[0])
            err = 0.0
            for (X, Y), C in zip(sets[0], outs):
                ref = X.float().t() @ Y.float()
                err = max(err, (C - ref).abs().max().item() / ref.abs().max().item())
            t = timeit([lambda p=p: gemm.gemm_tn_grouped(p, outs) for p in sets])
            print(f"spectral dW pts={pts} c={c}: {t:.3f} ms  {nb / t / 1e6:7.1f} GB/s  {fl / t / 1e9:6.1f} TF  err {err:.1e}  ({n} copies)",
                  flush=True)
            del sets
    if a.set in ("inter", "all"):
        for (R, N1, N2) in [(245760, 256, 3072), (491520, 128, 3072), (491520, 128, 1536), (983040, 64, 1536), (245760, 256, 6144),
                            (983040, 64, 768), (1966080, 32, 768)]:
            nb = R * (N1 + N2) * esz
            n = copies(nb)
            Xs = [torch.randn(R, N1, device=dev).to(dt) for _ in range(n)]
            Ys = [torch.randn(R, N2, device=dev).to(dt) for _ in range(n)]
            C = gemm.gemm_tn(Xs[0], Ys[0])
            t = timeit([lambda X=X, Y=Y: gemm.gemm_tn(X, Y, out=C) for X, Y in zip(Xs, Ys)])
            print(f"inter dW {R}x{N1}x{N2}: {t:.3f} ms  {nb / t / 1e6:7.1f} GB/s  {2.0 * R * N1 * N2 / t / 1e9:6.1f} TF  ({n} copies)", flush=True)
            del Xs, Ys


if __name__ == "__main__":
    main()
