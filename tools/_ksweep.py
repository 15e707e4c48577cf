import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epn_pointcloud_amd import gemm, _lib
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
dev = torch.device("cuda:0")
for (M, N) in [(245760, 1536), (245760, 256), (245760, 6144)]:
    for K in [64, 128, 256, 512, 1024, 2048]:
        A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev)
        t = timeit(lambda: gemm.gemm_nt(A, B, out=C)); t2 = timeit(lambda: torch.mm(A, B.t(), out=C))
        tiles = (M // 256) * ((N + 255) // 256)
        print(f"M={M} N={N} K={K}: own {t:.3f} ms {2.0*M*N*K/t/1e9:.1f} TF  us/tile/CU {t*1e3*256/tiles:.1f} | torch {t2:.3f} ms {2.0*M*N*K/t2/1e9:.1f} TF")
        del A, B, C
