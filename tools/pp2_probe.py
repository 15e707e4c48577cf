"""Round-5 probe: fp32 GEMM as THREE fp16 MFMA products of a two-piece split (x 2^s = h + l; hh + hl + lh on
v_mfma_f32_32x32x16_f16) -- csrc/gemm_pp.hip, tuning library.  Reports time / fp32-equivalent TFLOP/s of the VALU-free kernel
and the error against fp64 beside the native fp32 MFMA kernel and the lossless 3 x bf16 kernel on the same operands, for
operand magnitudes from O(1) down to gradient-sized values (the per-tensor power-of-two scale puts max|x| at 2^14)."""
import ctypes
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _tuning import use_tuning_lib  # noqa: E402
use_tuning_lib()
from epn_pointcloud_amd import gemm, _lib  # noqa: E402
from gemm_bench import timeit  # noqa: E402

KS_OF = {0: 16, 1: 32, 2: 16, 3: 32, 4: 32}
NAME = {0: "256x256 k16 3stg", 1: "256x256 k32 2stg", 2: "256x256 k16 4stg", 3: "256x128 k32 2stg", 4: "256x128 k32 3stg"}


def pow2_scale(t):
    m = float(t.abs().max())
    return 2.0 ** (14 - math.ceil(math.log2(m))) if m > 0 else 1.0


def main():
    dev = torch.device("cuda:0")
    lib = _lib.get_lib()
    vp, ll, ci, cf = ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_float
    lib._cdll.epn_lab_pp_split2.argtypes = [vp, ll, ll, ci, vp, ci, ci, cf, vp]
    lib._cdll.epn_lab_gemm_nt_pp2.argtypes = [vp, vp, vp, ll, ci, ci, ll, ci, ci, cf, vp]
    cfgs = [int(c) for c in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,1,2".split(","))]
    torch.manual_seed(0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def err(C, ref):
        d = C[:ref.shape[0]].double() - ref
        return (d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item(), (d.mean() / ref.abs().mean()).item()

    shapes = [(245760, 256, 3072, "randn x lognormal rows", 1.0), (245760, 256, 6144, "randn", 1.0),
              (245760, 256, 3072, "|randn| (post-activation)", 1.0), (245760, 256, 3072, "randn x 1e-6 (gradient-sized)", 1e-6),
              (245760, 256, 3072, "rows spanning 12 decades", 1.0), (491520, 128, 1536, "randn", 1.0), (245760, 6144, 256, "randn", 1.0)]
    for (M, N, K, kind, mag) in shapes:
        A = torch.randn(M, K, device=dev)
        if "lognormal" in kind:
            A *= torch.exp(torch.randn(M, 1, device=dev))
        if "|randn|" in kind:
            A = A.abs()
        if "decades" in kind:
            A *= 10.0 ** (torch.rand(M, 1, device=dev) * 12 - 6)
        A *= mag
        B = torch.randn(N, K, device=dev)
        nr = 1024
        ref = A[:nr].double() @ B.double().t()
        C = torch.empty(M, N, device=dev)
        print(f"NT {M}x{N}x{K} [{kind}]", flush=True)
        for mode in ("native", "split"):
            gemm.set_fp32_mode(mode)
            gemm.gemm_nt(A, B, out=C)
            e = err(C, ref)
            t = timeit(lambda: gemm.gemm_nt(A, B, out=C))
            print(f"   {mode:7s}: {t:.3f} ms {2.0 * M * N * K / t / 1e9:6.1f} TF  rms {e[0]:.2e} bias {e[1]:+.1e}", flush=True)
        sa, sb = pow2_scale(A), pow2_scale(B)
        Ap = torch.empty(2 * M * K, dtype=torch.float16, device=dev)
        Bp = torch.empty(2 * N * K, dtype=torch.float16, device=dev)
        for cfg in cfgs:
            ks = KS_OF[cfg]
            for layout in (0, 1):
                _lib.check(lib.epn_lab_pp_split2(A.data_ptr(), K, M, K, Ap.data_ptr(), layout, ks, sa, st), "split A")
                _lib.check(lib.epn_lab_pp_split2(B.data_ptr(), K, N, K, Bp.data_ptr(), layout, ks, sb, st), "split B")
                C.zero_()

                def run():
                    _lib.check(lib.epn_lab_gemm_nt_pp2(Ap.data_ptr(), Bp.data_ptr(), C.data_ptr(), M, N, K, N, layout, cfg,
                                                       1.0 / (sa * sb), st), "pp2")
                run()
                e = err(C, ref)
                t = timeit(run)
                print(f"   f16x2 cfg {cfg} [{NAME[cfg]}] layout {layout}: {t:.3f} ms {2.0 * M * N * K / t / 1e9:6.1f} TF  rms {e[0]:.2e} "
                      f"bias {e[1]:+.1e}", flush=True)
        del A, B, C, Ap, Bp, ref
    gemm.set_fp32_mode("split")


if __name__ == "__main__":
    main()
