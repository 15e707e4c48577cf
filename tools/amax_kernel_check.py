"""Do the producer-side maxima equal max|output|?  epn_norm_act_bwd_apply_amax_f32 called directly, repeatedly, on random data."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epn_pointcloud_amd import _lib

lib = _lib.get_lib()
dev = torch.device("cuda:0")
torch.manual_seed(0)
bad = 0
for it in range(40):
    b, c, p, a = 32, (64, 128, 256)[it % 3], (512, 256, 128)[it % 3], 60
    groups, rows = 1, b * p * a
    x = torch.randn(b * p * a, c, device=dev)
    dy = torch.randn(b * p * a, c, device=dev) * (10.0 ** torch.randn(1, device=dev))
    dy[torch.randint(0, rows, (1,)), torch.randint(0, c, (1,))] *= 50.0      # one outlier somewhere
    sums = torch.stack([x.sum(0), (x * x).sum(0)], 1).reshape(1, c, 2).contiguous()
    dsums = torch.randn(1, c, 2, device=dev)
    gamma, beta = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
    dx = torch.empty_like(x)
    amax = torch.full((1,), 123.0, device=dev)
    rc = lib.epn_norm_act_bwd_apply_amax_f32(x.data_ptr(), dy.data_ptr(), groups, ctypes.c_longlong(rows), c, sums.data_ptr(),
                                             dsums.data_ptr(), gamma.data_ptr(), beta.data_ptr(), ctypes.c_float(1e-5),
                                             ctypes.c_float(0.01), dx.data_ptr(), amax.data_ptr(), _lib.stream_of(x))
    assert rc == 0, rc
    t, m = dx.abs().max().item(), amax.item()
    ok = abs(t - m) <= 1e-6 * t
    bad += not ok
    if not ok or it < 3:
        print(f"c={c} rows={rows}: true {t:.6g} tag {m:.6g} {'OK' if ok else 'MISMATCH'}")
print("mismatches:", bad)
from epn_pointcloud_amd import gemm
badm = 0
for it in range(20):
    t = torch.randn(245760 * (1 + it % 3), 256, device=dev) * (10.0 ** torch.randn(1, device=dev))
    t[torch.randint(0, t.shape[0], (1,)), torch.randint(0, 256, (1,))] *= 40.0
    a, m = gemm.absmax(t).item(), t.abs().max().item()
    badm += abs(a - m) > 1e-6 * m
    if abs(a - m) > 1e-6 * m:
        print(f"absmax: true {m:.6g} got {a:.6g}")
print("absmax mismatches:", badm)
