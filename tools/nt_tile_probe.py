"""Single NT GEMMs of the cls step in the two-piece fp16 form (maximum supplied) under the tile overrides of the tuning library:
forward shapes (long K), data-gradient shapes (short K, output-heavy).  python tools/nt_tile_probe.py [cfg,...]"""
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
    cfgs = [int(c, 0) for c in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "0x121", "0x122", "0x12b", "0x127", "0x128", "0x129", "0x123"])]
    lib = _lib.get_lib()
    am = torch.full((1,), 6.0, device=dev)
    shapes = [(491520, 128, 1536), (491520, 128, 3072), (245760, 256, 3072), (245760, 256, 6144),
              (983040, 1536, 64), (491520, 1536, 128), (491520, 3072, 128), (245760, 3072, 256), (245760, 6144, 256), (122880, 6144, 256)]
    for (M, N, K) in shapes:
        n = max(2, min(4, int(600e6 // (M * K * 4)) + 1))
        As = [torch.randn(M, K, device=dev) for _ in range(n)]
        B = torch.randn(N, K, device=dev)
        C = torch.empty(M, N, device=dev)
        row = f"NT {M}x{N}x{K}:"
        for cfg in cfgs:
            assert lib.epn_set_kernel_policy((0x100 | (cfg & 0xff)) if cfg else 0) == 0
            t = min(timeit([lambda A=A: gemm.gemm_nt(A, B, out=C, a_amax=am) for A in As]) for _ in range(2))
            row += f"  [{cfg:#x}] {t:.3f} {2.0 * M * N * K / t / 1e9:4.0f}TF"
        lib.epn_set_kernel_policy(0)
        print(row, flush=True)
        del As, B, C


if __name__ == "__main__":
    main()
