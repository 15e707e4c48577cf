"""Correctness + throughput probe of csrc/gemm.hip on the contraction shapes of the ModelNet40 B=32 schedule.
Compares with torch.mm (rocBLAS/hipBLASLt) in the same process: `python tools/gemm_bench.py [--dtype f32|bf16] [--quick]`."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epn_pointcloud_amd import gemm  # noqa: E402


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--one", default="", help="profile mode: 'nt:M,N,K' or 'tn:R,N1,N2' -> 3 launches of that problem (own + torch)")
    ap.add_argument("--set", default="all", help="all | small (the narrow 1x1-convolution shapes) | dg (short-K data-gradient GEMMs)")
    ap.add_argument("--cfg", type=int, default=0, help="NT tile override (csrc/gemm.hip launch_nt_typed), 0 = heuristic")
    a = ap.parse_args()
    if a.cfg:
        from _tuning import use_tuning_lib
        lib_path = use_tuning_lib()           # tile overrides: the -DEPN_TUNING library (before the library is loaded)
        from epn_pointcloud_amd import _lib
        _lib.LIB_PATH = lib_path
        _lib.check(_lib.get_lib().epn_set_kernel_policy(a.cfg if a.cfg >= 0x100 else 0x100 | a.cfg), "set_kernel_policy")
    dt = torch.float32 if a.dtype == "f32" else torch.bfloat16
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    tol = 2e-3 if dt == torch.float32 else 3e-2

    if a.one:
        kind, dims = a.one.split(":")
        d0, d1, d2 = (int(v) for v in dims.split(","))
        if kind == "nt":
            A = torch.randn(d0, d2, device=dev).to(dt); B = torch.randn(d1, d2, device=dev).to(dt)
            for _ in range(3):
                gemm.gemm_nt(A, B)
                torch.mm(A, B.t())
        else:
            X = torch.randn(d0, d1, device=dev).to(dt); Y = torch.randn(d0, d2, device=dev).to(dt)
            for _ in range(3):
                gemm.gemm_tn(X, Y)
                torch.mm(X.t(), Y)
        torch.cuda.synchronize()
        return

    # ---- small correctness cases vs fp64 (edges: ragged M, N not a tile multiple)
    for (M, N, K) in [(1000, 64, 256), (513, 192, 320), (256, 32, 64), (777, 320, 128), (100, 24, 40)]:
        A = torch.randn(M, K, device=dev).to(dt)
        B = torch.randn(N, K, device=dev).to(dt)
        C = gemm.gemm_nt(A, B)
        ref = A.double() @ B.double().t()
        err = (C.double() - ref).abs().max().item() / ref.abs().max().item()
        print(f"NT  {M}x{N}x{K}: rel err {err:.2e}", "OK" if err < tol else "FAIL")
    for (R, N1, N2) in [(4096, 64, 512), (2048, 128, 200 if dt == torch.float32 else 192), (960, 32, 768), (1024, 256, 256), (100, 20, 36)]:
        X = torch.randn(R, N1, device=dev).to(dt)
        Y = torch.randn(R, N2, device=dev).to(dt)
        C = gemm.gemm_tn(X, Y)
        ref = X.double().t() @ Y.double()
        err = (C.double() - ref).abs().max().item() / ref.abs().max().item()
        print(f"TN  {R}x{N1}x{N2}: rel err {err:.2e}", "OK" if err < tol else "FAIL")
    W = torch.randn(96, 200, device=dev)
    print("transpose_cast:", torch.equal(gemm.transpose_cast(W, torch.float32), W.t().contiguous()),
          torch.equal(gemm.transpose_cast(W, torch.bfloat16), W.t().contiguous().bfloat16()))
    if a.quick:
        return

    small = a.set in ("small", "dg")
    nt_shapes = [(491520, 1536, 128), (491520, 3072, 128), (245760, 3072, 256), (245760, 6144, 256), (122880, 6144, 256),
                 (983040, 1536, 64)] if a.set == "dg" else [(983040, 64, 64), (491520, 128, 128), (245760, 128, 256), (491520, 64, 128), (245760, 256, 256),
                 (122880, 256, 256), (245760, 256, 128), (491520, 128, 64)] if small else \
                [(983040, 64, 1536), (491520, 128, 1536), (491520, 128, 3072), (245760, 256, 3072), (245760, 256, 6144),
                 (122880, 256, 6144), (245760, 6144, 256), (983040, 64, 64), (1966080, 32, 768), (1966080, 768, 32), (1966080, 32, 32)]
    for (M, N, K) in nt_shapes:
        A = torch.randn(M, K, device=dev).to(dt)
        B = torch.randn(N, K, device=dev).to(dt)
        C = gemm.gemm_nt(A, B)
        ref = torch.mm(A, B.t())
        err = (C.float() - ref.float()).abs().max().item() / ref.float().abs().max().item()
        t0 = timeit(lambda: gemm.gemm_nt(A, B, out=C))
        t1 = timeit(lambda: torch.mm(A, B.t(), out=ref))
        fl = 2.0 * M * N * K
        print(f"NT  {M}x{N}x{K}: own {t0:.3f} ms {fl / t0 / 1e9:.1f} TF | torch {t1:.3f} ms {fl / t1 / 1e9:.1f} TF | diff {err:.1e}")
        del A, B, C, ref
    tn_shapes = [] if a.set == "dg" else [(491520, 128, 128), (245760, 256, 256), (245760, 256, 128), (491520, 128, 64), (983040, 64, 64),
                 (122880, 256, 256)] if small else \
                [(983040, 64, 1536), (491520, 128, 1536), (491520, 128, 3072), (245760, 256, 3072), (245760, 256, 6144),
                 (122880, 256, 6144), (983040, 64, 64)]
    for (R, N1, N2) in tn_shapes:
        X = torch.randn(R, N1, device=dev).to(dt)
        Y = torch.randn(R, N2, device=dev).to(dt)
        C = gemm.gemm_tn(X, Y)
        ref = torch.mm(X.t(), Y)
        err = (C.float() - ref.float()).abs().max().item() / ref.float().abs().max().item()
        t0 = timeit(lambda: gemm.gemm_tn(X, Y, out=C))
        t1 = timeit(lambda: torch.mm(X.t(), Y))
        fl = 2.0 * R * N1 * N2
        print(f"TN  {R}x{N1}x{N2}: own {t0:.3f} ms {fl / t0 / 1e9:.1f} TF | torch {t1:.3f} ms {fl / t1 / 1e9:.1f} TF | diff {err:.1e}")
        del X, Y, C, ref
    if small:
        return
    # grouped spectral blocks (cin = cout = c): M = pts*d, K = N = d*c
    for pts, c in [(16384, 64), (8192, 128), (4096, 256)]:
        probs, fl = [], 0.0
        for d in (1, 3, 3, 4, 5):
            A = torch.randn(pts * d, d * c, device=dev).to(dt)
            B = torch.randn(d * c, d * c, device=dev).to(dt)
            probs.append((A, B, torch.empty(pts * d, d * c, device=dev, dtype=dt)))
            fl += 2.0 * pts * d * d * c * d * c
        t0 = timeit(lambda: gemm.gemm_nt_grouped(probs))
        t1 = timeit(lambda: [torch.mm(A, B.t(), out=C) for A, B, C in probs])
        print(f"grouped spectral pts={pts} c={c}: own {t0:.3f} ms {fl / t0 / 1e9:.1f} TF | torch {t1:.3f} ms {fl / t1 / 1e9:.1f} TF")


if __name__ == "__main__":
    main()
