"""NT GEMMs of the cls step in the two-piece fp16 form (max|A| supplied, as in the step): forward shapes (long K) and
data-gradient / 1x1 shapes (short K, output-heavy).  A/B over library builds: EPN_LIB=... python tools/nt_epi_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epn_pointcloud_amd import gemm  # noqa: E402
from gemm_bench import timeit  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    print("lib:", os.path.basename(os.environ.get("EPN_LIB", "default")), "mode:", gemm.FP32_MODE)
    shapes = [(245760, 6144, 256), (245760, 3072, 256), (491520, 3072, 128), (491520, 1536, 128), (983040, 1536, 64),
              (245760, 256, 6144), (245760, 256, 3072), (491520, 128, 1536), (983040, 64, 1536),
              (983040, 64, 64), (491520, 128, 128), (245760, 256, 256)]
    for (M, N, K) in shapes:
        A = torch.randn(M, K, device=dev)
        B = torch.randn(N, K, device=dev)
        C = torch.empty(M, N, device=dev)
        am = gemm.absmax(A)
        gemm.gemm_nt(A, B, out=C, a_amax=am)
        ref = A[:1024].double() @ B.double().t()
        err = ((C[:1024].double() - ref).abs().max() / ref.abs().max()).item()
        tail = ((C[-1024:].double() - A[-1024:].double() @ B.double().t()).abs().max() / ref.abs().max()).item()
        t = min(timeit(lambda: gemm.gemm_nt(A, B, out=C, a_amax=am)) for _ in range(2))
        print(f"NT {M}x{N}x{K}: {t:.3f} ms {2.0 * M * N * K / t / 1e9:6.1f} TF {M * (N + K) * 4 / t / 1e6:6.0f} GB/s  err {err:.1e} tail {tail:.1e}", flush=True)
        del A, B, C


if __name__ == "__main__":
    main()
