import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epn_pointcloud_amd import _lib
lib = _lib.get_lib()
dev = torch.device("cuda:0")
torch.manual_seed(1)
for c, rows in ((64, 983040), (256, 245760), (64, 4096), (64, 100000)):
    x = torch.randn(rows, c, device=dev)
    dy = torch.randn(rows, c, device=dev)
    sums = torch.stack([x.sum(0), (x * x).sum(0)], 1).reshape(1, c, 2).contiguous()
    dsums = torch.zeros(1, c, 2, device=dev)
    dx = torch.full_like(x, float("nan"))
    amax = torch.zeros(1, device=dev)
    rc = lib.epn_norm_act_bwd_apply_amax_f32(x.data_ptr(), dy.data_ptr(), 1, ctypes.c_longlong(rows), c, sums.data_ptr(),
                                             dsums.data_ptr(), None, None, ctypes.c_float(1e-5), ctypes.c_float(0.01),
                                             dx.data_ptr(), amax.data_ptr(), _lib.stream_of(x))
    torch.cuda.synchronize()
    print(f"c={c} rows={rows}: rc {rc} unwritten {torch.isnan(dx).sum().item()} tag {amax.item():.6g} true {dx.abs().max().item():.6g}")
    a = dx.abs()
    rowmax = a.max(1).values
    colmax = a.max(0).values
    t = amax.item()
    # which rows / columns have maxima above the tag?
    over_r = (rowmax > t * 1.000001).nonzero().flatten()
    over_c = (colmax > t * 1.000001).nonzero().flatten()
    print("  rows above tag:", over_r.numel(), over_r[:12].tolist(), " cols above tag:", over_c.numel(), over_c[:16].tolist())
    if over_r.numel():
        print("  rows mod 4:", torch.bincount(over_r % 4, minlength=4).tolist(), " mod 16:", torch.bincount(over_r % 16, minlength=16).tolist())
        print("  cols mod 4:", torch.bincount(over_c % 4, minlength=4).tolist())
