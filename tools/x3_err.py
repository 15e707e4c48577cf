"""Error of the fp32 contractions vs fp64 as a function of the contraction length: native fp32 MFMA, split form, rocBLAS.
rms = rms error / rms of the result; mean = signed mean error / mean |result| (the bf16 MFMA's truncating adder shows up
as a small negative bias that grows linearly with K)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epn_pointcloud_amd import gemm
dev = torch.device("cuda:0"); torch.manual_seed(0)
for pos in (False, True):
    for K in (512, 2048, 8192, 32768, 131072):
        A = torch.randn(4096, K, device=dev); B = torch.randn(128, K, device=dev)
        if pos: A, B = A.abs(), B.abs()
        ref = A.double() @ B.double().t()
        out = []
        for mode in ("native", "split"):
            gemm.set_fp32_mode(mode)
            C = gemm.gemm_nt(A, B)
            d = C.double() - ref
            out.append(f"{mode}: rms {(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item():.2e} mean {(d.mean() / ref.abs().mean()).item():+.2e}")
        t = torch.mm(A, B.t()); d = t.double() - ref
        out.append(f"torch: rms {(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item():.2e} mean {(d.mean() / ref.abs().mean()).item():+.2e}")
        print(f"pos={pos} K={K}: " + " | ".join(out), flush=True)
