"""Achievable HBM write / copy / read bandwidth of this GPU with plain torch kernels (calibration for the memory-bound
kernels of DESIGN.md 3.x): fill, copy and sum of a 6 GB fp32 tensor."""
import torch

dev = torch.device("cuda:0")
n = 6 * (1 << 30) // 4
x = torch.empty(n, device=dev)
y = torch.empty(n, device=dev)


def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


gb = n * 4 / 1e9
ms = t(lambda: x.fill_(1.0)); print(f"fill   {gb:.1f} GB: {ms:.3f} ms  {gb / ms:.2f} TB/s written")
ms = t(lambda: y.copy_(x)); print(f"copy   {gb:.1f} GB: {ms:.3f} ms  {2 * gb / ms:.2f} TB/s read+written")
ms = t(lambda: x.sum()); print(f"sum    {gb:.1f} GB: {ms:.3f} ms  {gb / ms:.2f} TB/s read")
ms = t(lambda: torch.cuda.memset if False else x.zero_()); print(f"zero_  {gb:.1f} GB: {ms:.3f} ms  {gb / ms:.2f} TB/s written")
