#!/usr/bin/env python3
"""Library fp32 GEMM ceiling (rocBLAS / hipBLASLt through torch.mm) on the GEMM shapes hidden inside the fused
convolutions -- a yardstick for the hand-written kernels, not part of the product path."""
import torch

dev = torch.device("cuda", 0)
torch.backends.cuda.matmul.allow_tf32 = False


def bench(m, k, n, ta=False, label=""):
    a = torch.randn((k, m) if ta else (m, k), device=dev)
    b = torch.randn(k, n, device=dev)
    f = (lambda: torch.mm(a.t(), b)) if ta else (lambda: torch.mm(a, b))
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{label:28s} M={m:7d} K={k:7d} N={n:5d} ta={int(ta)}  {ms:7.3f} ms  {2.0 * m * k * n / ms / 1e9:7.1f} TFLOP/s", flush=True)


# forward weight contraction: out[cols][cout] = G[cols][cin*24] @ W^T
bench(983040, 1536, 64, label="fwd  L1 64->64")
bench(491520, 3072, 128, label="fwd  L3 128->128")
bench(245760, 6144, 256, label="fwd  L5 256->256")
# weight gradient: dW[cout][cin*24] = dOut^T[cout][cols] @ G[cols][cin*24]
bench(64, 983040, 1536, ta=True, label="dW   L1 64->64")
bench(128, 491520, 3072, ta=True, label="dW   L3 128->128")
bench(256, 245760, 6144, ta=True, label="dW   L5 256->256")
# data gradient: dG[cols][cin*24] = dOut[cols][cout] @ W
bench(983040, 64, 1536, label="dG   L1 64->64")
bench(245760, 256, 6144, label="dG   L5 256->256")
# square reference point
bench(8192, 8192, 8192, label="square 8192")

# precision check: is this true fp32 accumulation of fp32 products?
a = torch.randn(4096, 1536, device=dev); b = torch.randn(1536, 256, device=dev)
ref = (a.double() @ b.double())
err = ((a @ b).double() - ref).abs().max().item() / ref.abs().max().item()
print(f"max rel err of fp32 mm vs fp64 (K=1536): {err:.3e}   (fp32 chain ~1e-6, tf32/xf32 ~1e-3)")
import os
print({k: v for k, v in os.environ.items() if "TF32" in k or "BLAS" in k or "TENSILE" in k})
