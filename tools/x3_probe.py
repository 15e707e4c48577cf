"""Split form (csrc/gemm_x3.hip) vs the native fp32 MFMA kernels on the schedule's contraction shapes: time and error vs
fp64.  `python tools/x3_probe.py [cfg,...]`: cfg 0 = native, 1 = split with the launcher's tile rule, 0x121.. = split with
a tile override (launch_gemm_nt_x3).  X3_PROBE=nt|tn restricts to one GEMM form."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:                     # tile overrides: the -DEPN_TUNING library
    from _tuning import use_tuning_lib
    use_tuning_lib()
from epn_pointcloud_amd import gemm, _lib  # noqa: E402
from gemm_bench import timeit  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    shapes = [(983040, 64, 1536), (491520, 128, 1536), (245760, 256, 3072), (245760, 256, 6144), (245760, 6144, 256), (245760, 320, 320)]
    cfgs = [int(c, 0) for c in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "0x1", "0x2"])]
    for (M, N, K) in (shapes if "nt" in os.environ.get("X3_PROBE", "nt,tn") else []):
        A = torch.randn(M, K, device=dev) * torch.exp(torch.randn(M, 1, device=dev))
        B = torch.randn(N, K, device=dev)
        ref = A[:2048].double() @ B.double().t()
        C = torch.empty(M, N, device=dev)
        for cfg in cfgs:
            gemm.set_fp32_mode("native" if cfg == 0 else ("f16x2" if cfg == 2 or (cfg >> 12) == 2 else "split"))     # 0x2 / 0x21xx: two-piece fp16 form
            _lib.check(_lib.get_lib().epn_set_kernel_policy((cfg & 0xfff) if cfg > 2 else 0), "policy")
            C.zero_()
            am = gemm.absmax(A) if gemm.FP32_MODE == "f16x2" else None      # the producer supplies max|A| in deployment
            gemm.gemm_nt(A, B, out=C, a_amax=am)
            err = ((C[:2048].double() - ref).abs().max() / ref.abs().max()).item()
            rms = ((C[:2048].double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
            tail = (C[-2048:].double() - A[-2048:].double() @ B.double().t()).abs().max().item()
            t = timeit(lambda: gemm.gemm_nt(A, B, out=C, a_amax=am))
            print(f"NT {M}x{N}x{K} cfg {cfg:#x}: {t:.3f} ms {2.0 * M * N * K / t / 1e9:.1f} TF  max {err:.2e} rms {rms:.2e} tail {tail:.1e}", flush=True)
        _lib.get_lib().epn_set_kernel_policy(0)
        del A, B, C, ref
    for (R, N1, N2) in [(983040, 64, 1536), (491520, 128, 1536), (245760, 256, 3072), (245760, 256, 6144), (491520, 128, 128), (245760, 256, 256)]:
        X = torch.randn(R, N1, device=dev) * torch.exp(torch.randn(1, N1, device=dev))
        Y = torch.randn(R, N2, device=dev)
        ref = X[:, :32].double().t() @ Y[:, :256].double()
        for mode in ("native", "split", "f16x2"):
            gemm.set_fp32_mode(mode)
            xa, ya = (gemm.absmax(X), gemm.absmax(Y)) if mode == "f16x2" else (None, None)
            C = gemm.gemm_tn(X, Y, x_amax=xa, y_amax=ya)
            d = C[:32, :256].double() - ref
            err = (d.abs().max() / ref.abs().max()).item()
            rms = (d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
            t = timeit(lambda: gemm.gemm_tn(X, Y, out=C, x_amax=xa, y_amax=ya))
            print(f"TN {R}x{N1}x{N2} {mode}: {t:.3f} ms {2.0 * R * N1 * N2 / t / 1e9:.1f} TF  max {err:.2e} rms {rms:.2e}", flush=True)
        del X, Y, C, ref
    _lib.get_lib().epn_set_kernel_policy(0)
    gemm.set_fp32_mode("native")


if __name__ == "__main__":
    main()
