"""MFMA ceiling of the bf16 NT kernel: operands small enough to stay in the 256 MB infinity cache (no HBM bound)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from epn_pointcloud_amd import gemm
dev = torch.device("cuda:0")
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (M, N, K) in [(245760, 512, 3072), (245760, 512, 1536), (122880, 1024, 3072), (245760, 256, 3072)]:
    A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16()
    C = gemm.gemm_nt(A, B)
    t = timeit(lambda: gemm.gemm_nt(A, B, out=C)); t2 = timeit(lambda: torch.mm(A, B.t()))
    print(f"NT bf16 {M}x{N}x{K}: own {t*1e3:.1f} us {2.0*M*N*K/t/1e9:.0f} TF | torch {t2*1e3:.1f} us {2.0*M*N*K/t2/1e9:.0f} TF", flush=True)
    X = torch.randn(M, N, device=dev).bfloat16()
    t = timeit(lambda: gemm.gemm_tn(X, A)); t2 = timeit(lambda: torch.mm(X.t(), A))
    print(f"TN bf16 {M}x{N}x{K}: own {t*1e3:.1f} us {2.0*M*N*K/t/1e9:.0f} TF | torch {t2*1e3:.1f} us {2.0*M*N*K/t2/1e9:.0f} TF", flush=True)
