"""Short-contraction NT GEMMs of the step -- the data-gradient GEMMs dG = dOut . W (K = cout = 64 .. 256, N = cin ks) and the
1x1 skip convolutions (K = N = cin) -- under the tile overrides of the split form (gemm_x3.hip) and, with --bf16, of the
bf16 kernels.  These tiles run 2..8 K steps: pipeline fill, epilogue and workgroup turnover weigh as much as the MFMAs, and
the 256 x 256 / 256 x 128 tiles own a CU's whole LDS (one workgroup per CU: nothing overlaps them).
  python tools/nt_shortk_probe.py [--bf16] [cfg,...]      cfg 1 = the launcher's rule, 0x121.. / 0x101.. = tile overrides"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _tuning import use_tuning_lib  # noqa: E402
use_tuning_lib()
from epn_pointcloud_amd import gemm, _lib  # noqa: E402
from gemm_bench import timeit  # noqa: E402


def main():
    bf = "--bf16" in sys.argv
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    if bf:      # rotation network (64 clouds): dG GEMMs + 1x1 convolutions
        shapes = [(491520, 3072, 128), (983040, 1536, 64), (1966080, 768, 32), (245760, 3072, 256), (491520, 1536, 128),
                  (983040, 768, 64), (1966080, 32, 32), (983040, 64, 64), (491520, 128, 128)]
        cfgs = args[0].split(",") if args else ["1", "0x101", "0x102", "0x103", "0x104", "0x105", "0x106"]
    else:       # cls network (32 clouds)
        shapes = [(245760, 6144, 256), (491520, 3072, 128), (122880, 6144, 256), (245760, 3072, 256), (983040, 1536, 64),
                  (491520, 1536, 128), (491520, 128, 128), (983040, 64, 64), (245760, 256, 256), (491520, 128, 64)]
        cfgs = args[0].split(",") if args else ["1", "0x121", "0x122", "0x123", "0x125", "0x127", "0x128", "0x129"]
    cfgs = [int(c, 0) for c in cfgs]
    gemm.set_fp32_mode("split")
    dt = torch.bfloat16 if bf else torch.float32
    esz = 2 if bf else 4
    for (M, N, K) in shapes:
        A = torch.randn(M, K, device=dev).to(dt)
        B = torch.randn(N, K, device=dev).to(dt)
        C = torch.empty(M, N, device=dev, dtype=dt)
        line = []
        for cfg in cfgs:
            _lib.check(_lib.get_lib().epn_set_kernel_policy(cfg if cfg > 1 else 0), "policy")
            gemm.gemm_nt(A, B, out=C)
            t = min(timeit(lambda: gemm.gemm_nt(A, B, out=C)) for _ in range(2))
            line.append(f"{cfg:#x}: {t:.3f} ms {2.0 * M * N * K / t / 1e9:6.1f} TF {M * (N + K) * esz / t / 1e6:6.0f} GB/s")
        _lib.get_lib().epn_set_kernel_policy(0)
        print(f"NT {M}x{N}x{K} " + " | ".join(line), flush=True)
        del A, B, C


if __name__ == "__main__":
    main()
