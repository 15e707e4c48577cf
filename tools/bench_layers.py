#!/usr/bin/env python3
"""Per-layer timing of the fused kernels on the ModelNet cls schedule (B=32, N=1024, A=60): the perf iteration
tool.  Prints ms and algorithmic TFLOP/s per C-ABI call.   usage: tools/bench_layers.py [--iters 10] [--only fwd]"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epn_pointcloud_amd import ops, schedule as S  # noqa: E402
from epn_pointcloud_amd.vgtk import so3conv as sptk, spconv as zptk, pc as pctk  # noqa: E402


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--only", default="")
    ap.add_argument("--split", action="store_true", help="also time the split form: grouping kernels + BLAS GEMMs")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    layers = S.cls_so3net_schedule(1024)
    fl = S.hot_path_flops(layers, args.batch, 1024)
    pts = S.synthetic_clouds(args.batch, 1024, dev)
    xyz = pts.permute(0, 2, 1).contiguous()
    tot = {}
    for li, (l, f) in enumerate(zip(layers, fl)):
        conv = sptk.InterSO3Conv(l.cin, l.cout, 1, l.stride, l.radius, l.sigma, l.nn, lazy_sample=l.lazy).to(dev)
        intra = sptk.IntraSO3Conv(l.cout, l.cout).to(dev)
        p1 = xyz.shape[2]
        feats = torch.randn(args.batch, l.cin, p1, 60, device=dev).contiguous(memory_format=torch.channels_last)
        n_sample = math.ceil(p1 / l.stride)
        sidx, new_xyz = pctk.furthest_sample(xyz, n_sample, l.lazy)
        idx = pctk.ball_query_index(new_xyz, xyz, l.radius, l.nn)
        geo = ops.InterGeometry(xyz, new_xyz, idx, conv.anchors, conv.kernels, l.sigma)
        W = conv.basic_conv.W.detach()
        out = ops.InterSO3ConvFn.apply(feats, W, geo)
        gout = torch.randn_like(out)
        fr = feats.clone().requires_grad_(True)
        Wr = W.clone().requires_grad_(True)
        inter_fl = f["wgen"] + f["group"] + f["gemm"]
        res = {}

        def run_bwd(need_f, need_w):
            o = ops.InterSO3ConvFn.apply(fr if need_f else feats, Wr if need_w else W, geo)
            return o

        if not args.only or "fwd" in args.only:
            res["inter_fwd"] = (timeit(lambda: ops.InterSO3ConvFn.apply(feats, W, geo), args.iters), inter_fl)
        if l.cin >= 16 and (not args.only or "bwd" in args.only):
            o = ops.InterSO3ConvFn.apply(fr, W, geo)
            t_f = timeit(lambda: torch.autograd.grad(o, fr, gout, retain_graph=True), args.iters)
            o = ops.InterSO3ConvFn.apply(feats, Wr, geo)
            t_w = timeit(lambda: torch.autograd.grad(o, Wr, gout, retain_graph=True), args.iters)
            res["inter_bwd_data"] = (t_f, inter_fl)
            res["inter_bwd_weight"] = (t_w, inter_fl)
        if args.split and l.cin >= 16:
            import ctypes
            from epn_pointcloud_amd import _lib
            lib = _lib.get_lib()
            d = geo.desc(l.cin, l.cout)
            cols, ck = d.b * d.p2 * d.na, l.cin * d.ks
            G = torch.empty(cols, ck, device=dev)
            ws, wsp, wsn = ops._group_workspace(lib, d, dev)
            g2d = gout.permute(0, 2, 3, 1).reshape(cols, l.cout)
            gf = torch.empty_like(feats)
            grp = f["wgen"] + f["group"]
            res["group"] = (timeit(lambda: lib.epn_inter_group_f32(ctypes.byref(d), ops._cl_ptr(feats), G.data_ptr(), wsp, wsn,
                                                                   _lib.stream_of(feats)), args.iters), grp)
            res["gemm_out"] = (timeit(lambda: torch.mm(G, W.t()), args.iters), f["gemm"])
            res["gemm_dW"] = (timeit(lambda: torch.mm(g2d.t(), G), args.iters), f["gemm"])
            dG = torch.mm(g2d, W)
            res["gemm_dG"] = (timeit(lambda: torch.mm(g2d, W), args.iters), f["gemm"])
            res["ungroup"] = (timeit(lambda: lib.epn_inter_ungroup_f32(ctypes.byref(d), dG.data_ptr(), ops._cl_ptr(gf), wsp, wsn,
                                                                       _lib.stream_of(feats)), args.iters), grp)
            del G, dG
        fi = out.detach()
        Wi = intra.basic_conv.W.detach()
        i32 = intra.intra_idx.int()
        if not args.only or "intra" in args.only:
            res["intra_fwd"] = (timeit(lambda: ops.IntraSO3ConvFn.apply(fi, Wi, i32), args.iters), f["intra"])
            fir = fi.clone().requires_grad_(True)
            Wir = Wi.clone().requires_grad_(True)
            o = ops.IntraSO3ConvFn.apply(fir, Wi, i32)
            res["intra_bwd_data"] = (timeit(lambda: torch.autograd.grad(o, fir, gout, retain_graph=True), args.iters),
                                     f["intra"])
            o = ops.IntraSO3ConvFn.apply(fi, Wir, i32)
            res["intra_bwd_weight"] = (timeit(lambda: torch.autograd.grad(o, Wir, gout, retain_graph=True),
                                              args.iters), f["intra"])
        line = f"L{li} {l.cin:3d}->{l.cout:3d} s{l.stride} K{l.nn:2d} P{p1:4d}->{n_sample:4d} |"
        for k, (ms, flops) in res.items():
            line += f" {k} {ms:6.2f}ms {flops / ms / 1e9:5.1f}TF |"
            tot[k] = tot.get(k, 0.0) + ms
        print(line, flush=True)
        xyz = new_xyz
    print("total ms/step:", {k: round(v, 2) for k, v in tot.items()}, "sum", round(sum(tot.values()), 2))


if __name__ == "__main__":
    main()
