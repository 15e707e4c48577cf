"""Review item 3 (round 5): the split-form NT GEMM with BOTH operands pre-split into bf16 planes (csrc/gemm_pp.hip, tuning
library only) against gemm_nt_x3_kernel (A split in registers) on the schedule's wide shapes: time, fp32-equivalent TFLOP/s,
error vs fp64.  The split pass that produces A's planes is NOT timed (the producer kernel would emit them).
python tools/pp_probe.py [cfg,cfg,...] [layout,...]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _tuning import use_tuning_lib  # noqa: E402
use_tuning_lib()
from epn_pointcloud_amd import gemm, _lib  # noqa: E402
from gemm_bench import timeit  # noqa: E402

KS_OF = {0: 16, 1: 32, 2: 16, 3: 32, 4: 16, 5: 16, 6: 32}
NAME = {0: "256x256 k16 3stg", 1: "256x128 k32 2stg", 2: "256x256 k16 2stg", 3: "128x256 k32 2stg", 4: "256x128/4w k16 4stg",
        5: "256x128/4w k16 2stg", 6: "256x64/4w k32 2stg"}


def main():
    dev = torch.device("cuda:0")
    lib = _lib.get_lib()
    vp, ll, ci = ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int
    # (argtypes go on the CDLL's functions: the proxy hands out wrappers)
    lib._cdll.epn_lab_pp_split.argtypes = [vp, ll, ll, ci, vp, ci, ci, vp]
    lib._cdll.epn_lab_gemm_nt_pp.argtypes = [vp, vp, vp, ll, ci, ci, ll, ci, ci, vp]
    cfgs = [int(c) for c in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,1,2,3,4,5".split(","))]
    layouts = [int(c) for c in (sys.argv[2].split(",") if len(sys.argv) > 2 else "0,1".split(","))]
    torch.manual_seed(0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    shapes = [(245760, 256, 3072), (245760, 256, 6144), (491520, 128, 1536), (983040, 64, 1536), (245760, 6144, 256),
              (491520, 3072, 128)]
    for (M, N, K) in shapes:
        A = torch.randn(M, K, device=dev) * torch.exp(torch.randn(M, 1, device=dev))
        B = torch.randn(N, K, device=dev)
        ref = A[:1024].double() @ B.double().t()
        C = torch.empty(M, N, device=dev)
        gemm.set_fp32_mode("split")
        t = timeit(lambda: gemm.gemm_nt(A, B, out=C))
        rms = ((C[:1024].double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
        print(f"NT {M}x{N}x{K} x3 (A split in registers): {t:.3f} ms {2.0 * M * N * K / t / 1e9:.1f} TF  rms {rms:.2e}", flush=True)
        Ap = torch.empty(3 * M * K, dtype=torch.bfloat16, device=dev)
        Bp = torch.empty(3 * N * K, dtype=torch.bfloat16, device=dev)
        for cfg in cfgs:
            ks = KS_OF[cfg]
            for layout in layouts:
                _lib.check(lib.epn_lab_pp_split(A.data_ptr(), K, M, K, Ap.data_ptr(), layout, ks, st), "split A")
                _lib.check(lib.epn_lab_pp_split(B.data_ptr(), K, N, K, Bp.data_ptr(), layout, ks, st), "split B")
                C.zero_()

                def run():
                    _lib.check(lib.epn_lab_gemm_nt_pp(Ap.data_ptr(), Bp.data_ptr(), C.data_ptr(), M, N, K, N, layout, cfg, st), "pp")
                run()
                rms = ((C[:1024].double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
                tail = (C[-512:].double() - A[-512:].double() @ B.double().t()).abs().max().item() / ref.abs().max().item()
                t = timeit(run)
                print(f"   pp cfg {cfg} [{NAME[cfg]}] layout {layout}: {t:.3f} ms {2.0 * M * N * K / t / 1e9:.1f} TF  rms {rms:.2e} "
                      f"tail {tail:.1e}", flush=True)
        del A, B, C, Ap, Bp, ref


if __name__ == "__main__":
    main()
