"""Price the transpose of the grouping (inter_ungroup_shared_kernel) against the roof that bounds it at K = 64: L2 atomic
operations, not HBM bytes (review item 3, round 5: "so the row stops reading 0.18 of HBM forever").

For every InterSO3Conv layer of a schedule: the kernel is timed on a random dG of the layer's shape, and its fp32 atomics are
COUNTED on the device from the layer's own geometry, replicating the kernel's grouping: Morton order of the output points
(10 bits per axis, ties by index), GP consecutive points per workgroup, the distinct valid destinations U of their neighbour
slots (cyclic repeats of the ball query and shadow indices excluded, as load_hood does).  Per anchor step and chunk group a
workgroup issues ceil(U * 16 CW / 64) wave-level atomic instructions, each covering 64 consecutive floats of a destination row
= 64 / (16 CW) destinations = that many 128-byte-aligned pieces of 64 CW bytes (CW = 2: two whole lines per instruction).
The roofs come from tools/atomic_rate_probe.hip on the same part (profiles/r06_atomic_rate_probe.txt): 5.2 G atomic
instructions/s, 21 G (instruction x line) operations/s.
python tools/ungroup_atomic_pricing.py [reg|inv|cls] [bf16|f32]"""
import ctypes
import math
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import epn_pointcloud_amd
from epn_pointcloud_amd import _lib, ops, schedule as S

vgtk = epn_pointcloud_amd.install_vgtk_alias()
import vgtk.pc as pctk
import vgtk.so3conv as sptk

ROOF_INSTR, ROOF_LINE = 5.2e9, 21.0e9


def spread3(v):
    v = v & 0x3ff
    v = (v | (v << 16)) & 0x030000ff
    v = (v | (v << 8)) & 0x0300f00f
    v = (v | (v << 4)) & 0x030c30c3
    v = (v | (v << 2)) & 0x09249249
    return v


def morton_order(new_xyz):                       # [b, 3, p2] -> order [b, p2] (as morton_order_kernel)
    lo = new_xyz.amin(dim=2, keepdim=True)
    hi = new_xyz.amax(dim=2, keepdim=True)
    scale = torch.where(hi > lo, 1023.0 / (hi - lo), torch.zeros_like(hi))
    q = ((new_xyz - lo) * scale).clamp(0, 1023).to(torch.int64)
    code = spread3(q[:, 0]) | (spread3(q[:, 1]) << 1) | (spread3(q[:, 2]) << 2)
    p2 = new_xyz.shape[2]
    key = code * 8192 + torch.arange(p2, device=new_xyz.device)
    return key.argsort(dim=1)


def group_points(nt, p2):                        # ungroup_group_points + the launcher's overrides (cin % 64 == 0 assumed)
    return 8 if nt <= 1 else ((16 if p2 % 16 == 0 else 8) if nt <= 2 else ((8 if p2 % 8 == 0 else 4) if nt <= 4 else 2))


def main():
    model = sys.argv[1] if len(sys.argv) > 1 else "reg"
    dt = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else torch.bfloat16
    dev = torch.device("cuda:0")
    if model == "cls":
        layers, b, n, scale = S.cls_so3net_schedule(1024), 32, 1024, 1.0
    elif model == "reg":
        layers, b, n, scale = S.reg_so3net_schedule(1024), 64, 1024, 1.0
    else:
        layers, b, n, scale = S.inv_so3net_schedule(2048), 64, 2048, 0.4
    xyz = S.synthetic_clouds(b, n, dev, seed=2913, scale=scale).permute(0, 2, 1).contiguous()
    lib = _lib.get_lib()
    print(f"{model} {dt}: roofs {ROOF_INSTR / 1e9:.1f} G atomic instr/s, {ROOF_LINE / 1e9:.0f} G (instr x line)/s")
    print(f"{'layer':30s} {'ms':>7s} {'GP':>3s} {'CW':>3s} {'slots/U':>8s} {'M instr':>8s} {'G instr/s':>10s} {'frac':>6s} {'G line-ops/s':>13s} {'frac':>6s}  {'dG TB/s':>8s}")
    for li, l in enumerate(layers):
        p1 = xyz.shape[2]
        p2 = math.ceil(p1 / l.stride)
        _, new_xyz = pctk.furthest_sample(xyz, p2, l.lazy)
        if l.cin >= 16:
            conv = sptk.InterSO3Conv(l.cin, l.cout, 1, l.stride, l.radius, l.sigma, l.nn, lazy_sample=l.lazy).to(dev)
            idx = pctk.ball_query_index(new_xyz, xyz, l.radius, l.nn)
            geo = ops.InterGeometry(xyz, new_xyz, idx, conv.anchors, conv.kernels, conv.sigma)
            d = geo.desc(l.cin, l.cout)
            nt = (l.nn + 15) // 16
            gp = group_points(nt, p2)
            if nt == 2 and l.cin % 64 != 0 and l.cin % 32 == 0 and p2 % 8 == 0:
                gp = 8
            wide = l.cin % 64 == 0 and nt <= 2
            wide2 = (not wide) and l.cin % 32 == 0 and ((2 < nt <= 4 and gp == 8) or nt == 2)
            cw = 4 if wide else (2 if wide2 else 1)
            # distinct valid destinations per workgroup
            row = idx.long()                                                    # [b, p2, nn]
            first = row[:, :, :1]
            rep = (row == first) & (torch.arange(l.nn, device=dev) > 0)
            cnt = torch.where(rep.any(2), rep.float().argmax(2), torch.full_like(rep[:, :, 0], l.nn, dtype=torch.long))
            valid = (torch.arange(l.nn, device=dev)[None, None] < cnt[:, :, None]) & (row >= 0) & (row < p1)
            order = morton_order(new_xyz)
            rows = torch.gather(row, 1, order[:, :, None].expand(-1, -1, l.nn)).view(b, p2 // gp, gp * l.nn)
            vals = torch.gather(valid, 1, order[:, :, None].expand(-1, -1, l.nn)).view(b, p2 // gp, gp * l.nn)
            key = torch.where(vals, rows, torch.full_like(rows, p1))            # invalid slots -> one extra bucket
            srt = key.sort(dim=2).values
            distinct = (srt[:, :, 1:] != srt[:, :, :-1]).sum(2) + 1 - (srt[:, :, -1] == p1).long()
            U = distinct.float()
            slots = vals.sum(2).float()
            groups = l.cin // (16 * cw)
            instr = (torch.ceil(U * 16 * cw / 64) * 60 * groups).sum().item()
            lines = (U * 60 * groups).sum().item() * max(1, (64 * cw) // 128)   # pieces of 64 CW bytes; CW = 4: two lines per destination
            G = torch.randn(b * p2 * 60, l.cin * 24, device=dev).to(dt)
            gf = ops.empty_cl(b, l.cin, p1, 60, dev)
            ws = torch.empty(max(int(lib.epn_inter_group_workspace_bytes(ctypes.byref(d))), 16), dtype=torch.uint8, device=dev)
            fn = lib.epn_inter_ungroup_bf16 if dt == torch.bfloat16 else lib.epn_inter_ungroup_f32

            def run():
                _lib.check(fn(ctypes.byref(d), G.data_ptr(), ops._cl_ptr(gf), ws.data_ptr(), ws.numel(), _lib.stream_of(G)), "ungroup")
            run(); run(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(6):
                run()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 6          # (includes the zero fill of the target and the Morton / table helpers)
            kern = lib.epn_last_kernel().decode().split("::")[-1]
            print(f"L{li} {l.cin:3d} K={l.nn:3d} p2={p2:4d} b={b:2d}     {ms:7.3f} {gp:3d} {cw:3d} {slots.sum().item() / U.sum().item():8.2f} "
                  f"{instr / 1e6:8.2f} {instr / ms / 1e6:10.2f} {instr / ms / 1e-3 / ROOF_INSTR:6.2f} {lines / ms / 1e6:13.2f} "
                  f"{lines / ms / 1e-3 / ROOF_LINE:6.2f}  {G.numel() * G.element_size() / ms / 1e9:8.2f}  {kern}", flush=True)
            del G, gf
        xyz = new_xyz


if __name__ == "__main__":
    main()
