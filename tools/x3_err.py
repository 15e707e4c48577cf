"""Error of the fp32 contractions vs fp64 as a function of the contraction length: native fp32 MFMA, the lossless three-piece
bf16 form ("split"), the two-piece fp16 form ("f16x2"), rocBLAS.  rms = rms error / rms of the result; mean = signed mean
error / mean |result| (the 16-bit MFMAs' truncating adder shows up as a small negative bias that grows with K).
Second table: what the two-piece form does to ROWS far below the tensor's largest magnitude (its window of full relative
precision is 2^17 below max|x|; beneath it the absolute error stays <= max|x| 2^-39): per-row relative error of rows scaled by
1, 1e-2 ... 1e-8 against the unscaled rows of the same matrix."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epn_pointcloud_amd import gemm
dev = torch.device("cuda:0"); torch.manual_seed(0)
MODES = ("native", "split", "f16x2")


def rel(C, ref):
    d = C.double() - ref
    return f"rms {(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item():.2e} mean {(d.mean() / ref.abs().mean()).item():+.2e}"


for kind in ("randn", "|randn|", "randn x 1e-7", "randn x lognormal rows"):
    for K in (512, 2048, 8192, 32768, 131072):
        A = torch.randn(4096, K, device=dev); B = torch.randn(128, K, device=dev)
        if kind == "|randn|": A, B = A.abs(), B.abs()
        if kind == "randn x 1e-7": A = A * 1e-7
        if kind == "randn x lognormal rows": A = A * torch.exp(torch.randn(4096, 1, device=dev))
        ref = A.double() @ B.double().t()
        out = []
        for mode in MODES:
            gemm.set_fp32_mode(mode)
            out.append(f"{mode}: " + rel(gemm.gemm_nt(A, B), ref))
        out.append("torch: " + rel(torch.mm(A, B.t()), ref))
        print(f"{kind} K={K}: " + " | ".join(out), flush=True)

print("\nper-row relative rms error of a row scaled by s inside a tensor whose other rows are N(0,1) (K = 2048):")
K = 2048
A = torch.randn(4096, K, device=dev); B = torch.randn(128, K, device=dev)
for e in (0, -2, -4, -5, -6, -7, -8, -10):
    As = A.clone(); As[:256] *= 10.0 ** e
    ref = As[:256].double() @ B.double().t()
    out = []
    for mode in MODES:
        gemm.set_fp32_mode(mode)
        C = gemm.gemm_nt(As, B)[:256]
        out.append(f"{mode} {((C.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item():.2e}")
    print(f"  s = 1e{e}: " + " | ".join(out), flush=True)
gemm.set_fp32_mode("split")
