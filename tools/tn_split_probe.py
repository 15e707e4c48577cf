"""Single TN (weight-gradient) GEMMs of the cls step, two-piece fp16 form with maxima supplied, cold, under split-count targets
(epn_set_kernel_policy 0x200 | v: about 256 v workgroups).  python tools/tn_split_probe.py [v,...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _tuning import use_tuning_lib
use_tuning_lib()
from epn_pointcloud_amd import gemm, _lib  # noqa: E402
from tn_probe import timeit  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    vs = [int(c, 0) for c in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "1", "2", "3", "4", "6", "8"])]
    lib = _lib.get_lib()
    am = torch.full((1,), 6.0, device=dev)
    for (R, N1, N2) in [(245760, 256, 6144), (245760, 256, 3072), (491520, 128, 3072), (491520, 128, 1536), (983040, 64, 1536),
                        (245760, 256, 256), (491520, 128, 128), (983040, 64, 64), (122880, 256, 256)]:
        n = max(2, min(4, int(600e6 // (R * N2 * 4)) + 1))
        Xs = [torch.randn(R, N1, device=dev) for _ in range(n)]
        Ys = [torch.randn(R, N2, device=dev) for _ in range(n)]
        C = torch.empty(N1, N2, device=dev)
        row = f"TN {R}x{N1}x{N2}:"
        for v in vs:
            assert lib.epn_set_kernel_policy((0x200 | v) if v else 0) == 0
            t = min(timeit([lambda X=X, Y=Y: gemm.gemm_tn(X, Y, out=C, x_amax=am, y_amax=am) for X, Y in zip(Xs, Ys)]) for _ in range(2))
            row += f"  [{v}] {t:.3f} {2.0 * R * N1 * N2 / t / 1e9:4.0f}TF"
        lib.epn_set_kernel_policy(0)
        print(row, flush=True)
        del Xs, Ys


if __name__ == "__main__":
    main()
