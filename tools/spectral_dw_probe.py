"""Grouped spectral weight-gradient GEMMs (five ragged TN problems per IntraSO3Conv) of the cls schedule, two-piece fp16
form, maxima supplied (as in the step), timed cold over several operand copies.  A/B over library builds:
  EPN_LIB=.../libepn_so3conv_<tag>.so python tools/spectral_dw_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epn_pointcloud_amd import gemm  # noqa: E402
from tn_probe import timeit, copies  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    print("lib:", os.environ.get("EPN_LIB", "default"), "mode:", gemm.FP32_MODE)
    for pts, c in [(16384, 64), (8192, 128), (4096, 256), (2048, 256)]:
        nb = pts * 60 * 2 * c * 4
        n = copies(nb)
        sets = []
        for i in range(n):
            sets.append([(torch.randn(pts * d, d * c, device=dev), torch.randn(pts * d, d * c, device=dev)) for d in (1, 3, 3, 4, 5)])
        fl = sum(2.0 * pts * d * d * c * d * c for d in (1, 3, 3, 4, 5))
        am = torch.full((1,), 6.0, device=dev)
        outs = gemm.gemm_tn_grouped(sets[0], x_amax=[am] * 5, y_amax=[am] * 5)
        err = 0.0
        for (X, Y), C in zip(sets[0], outs):
            ref = X.double().t() @ Y.double()
            err = max(err, ((C - ref).abs().max() / ref.abs().max()).item())
        t = timeit([lambda s=s: gemm.gemm_tn_grouped(s, outs, x_amax=[am] * 5, y_amax=[am] * 5) for s in sets])
        print(f"spectral dW pts={pts} c={c}: {t:.3f} ms  {nb / t / 1e6:7.1f} GB/s  {fl / t / 1e9:6.1f} TF  err {err:.1e}  ({n} copies)", flush=True)
        del sets


if __name__ == "__main__":
    main()
