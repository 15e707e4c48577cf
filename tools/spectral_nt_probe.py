"""Grouped spectral NT GEMMs (five ragged problems per IntraSO3Conv: Z^rho = Y^rho What^rho) of the cls schedule, two-piece
fp16 form with the maximum supplied, under tile overrides of the tuning library (epn_set_kernel_policy 0x100 | cfg).
  python tools/spectral_nt_probe.py [cfg,...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _tuning import use_tuning_lib
use_tuning_lib()
from epn_pointcloud_amd import gemm, _lib  # noqa: E402
from tn_probe import timeit, copies  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    cfgs = [int(c, 0) for c in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "0x123", "0x125", "0x126", "0x121"])]
    lib = _lib.get_lib()
    am = torch.full((1,), 6.0, device=dev)
    for pts, c in [(16384, 64), (8192, 128), (4096, 256)]:
        nb = pts * 60 * 2 * c * 4
        n = copies(nb)
        sets = []
        for i in range(n):
            sets.append([(torch.randn(pts * d, d * c, device=dev), torch.randn(d * c, d * c, device=dev),
                          torch.empty(pts * d, d * c, device=dev)) for d in (1, 3, 3, 4, 5)])
        fl = sum(2.0 * pts * d * d * c * d * c for d in (1, 3, 3, 4, 5))
        row = f"spectral NT pts={pts} c={c}:"
        for cfg in cfgs:
            assert lib.epn_set_kernel_policy((0x100 | (cfg & 0xff)) if cfg else 0) == 0
            t = timeit([lambda s=s: gemm.gemm_nt_grouped(s, a_amax=[am] * 5) for s in sets])
            row += f"  [{cfg:#x}] {t:.3f} ms {fl / t / 1e9:5.0f} TF"
        lib.epn_set_kernel_policy(0)
        print(row, flush=True)
        del sets
    # single narrow problems of the step: cout = 64 forward GEMM, 1x1 convolutions
    for (M, N, K) in [(983040, 64, 1536), (983040, 64, 64), (491520, 128, 128), (491520, 64, 128)]:
        n = copies(M * (N + K) * 4)
        As = [torch.randn(M, K, device=dev) for _ in range(n)]
        B = torch.randn(N, K, device=dev)
        C = torch.empty(M, N, device=dev)
        row = f"NT {M}x{N}x{K}:"
        for cfg in cfgs:
            assert lib.epn_set_kernel_policy((0x100 | (cfg & 0xff)) if cfg else 0) == 0
            t = timeit([lambda A=A: gemm.gemm_nt(A, B, out=C, a_amax=am) for A in As])
            row += f"  [{cfg:#x}] {t:.3f} ms {M * (N + K) * 4 / t / 1e6:5.0f} GB/s"
        lib.epn_set_kernel_policy(0)
        print(row, flush=True)
        del As


if __name__ == "__main__":
    main()
