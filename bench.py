#!/usr/bin/env python3
"""bench.py -- point-clouds/sec, forward + backward, of the EPN hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): the ModelNet40 classification network cls_so3net_pn, B=32 clouds per GPU, N=1024
points, K=32/16 neighbours, A=60 anchors, fp32 -- 7 separable blocks (FPS -> ball query -> gather -> InterSO3Conv ->
IntraSO3Conv + the block's norm/activation/skip glue) and the ClsOutBlockPointnet head (1x1 conv, PointnetSO3Conv,
attention over the anchors, 40-way logits), random-init weights, synthetic unit-ball clouds and labels already resident
in HBM.  One step = forward + cross-entropy + backward (+ gradient all-reduce over RCCL when N > 1) + Adam update.
--backbone-only drops the head (loss = mean square of the last feature map).  Weak scaling: per-GPU batch fixed.  Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"

BLAS_FAMILY = "rocBLAS fp32 GEMM (Cijk_*)"   # the library kernels torch.mm dispatches to (Tensile "Cijk_..." names)
KERNEL_OF = {  # C-ABI call -> device kernel family it launches on this workload (csrc/*.hip; template variants summed)
    "inter_fwd": "epn::inter_fwd8_kernel", "inter_bwd_data": "epn::inter_bwd_data8_kernel",
    "inter_bwd_weight": "epn::inter_bwd_weight8_kernel", "intra_fwd": "epn::intra_gemm_kernel",
    "intra_bwd_data": "epn::intra_gemm_kernel", "intra_bwd_weight": "epn::intra_bwd_weight_pt_kernel",
    "inter_group": "epn::inter_group_kernel", "inter_ungroup": "epn::inter_ungroup_kernel",
    "inter_gemm": "epn::gemm_nt_kernel", "intra_gemm": "epn::gemm_nt_kernel",
    "inter_gemm_dw": "epn::gemm_tn_f32_kernel", "intra_gemm_dw": "epn::gemm_tn_f32_kernel",
    "intra_group": "epn::intra_group_kernel", "so3_basis": "epn::so3_basis_kernel",
    "pointnet_fwd": "epn::pointnet_fwd_kernel", "pointnet_bwd_data": "epn::pointnet_bwd_data_kernel",
    "pointnet_bwd_weight": "epn::pointnet_bwd_weight_kernel",
}
PMC_FILE = os.path.join(ROOT, "profiles", "r01_pmc_per_kernel.json")   # tools/collect_profiles.sh + pmc_summary.py


def recorded_traffic(kernel_family):
    """HBM bytes per launch of one kernel family from the committed rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE
    in separate runs, x1024, read side doubled per MI355X_MICROARCH.md): PMC counters cannot be read live here."""
    try:
        pmc = json.load(open(PMC_FILE))
    except OSError:
        return None
    tot = n = 0.0
    for name, e in pmc.items():
        fam = BLAS_FAMILY if name.startswith("Cijk_") else name.split("<")[0]
        if fam == kernel_family and "hbm_bytes_per_launch" in e:
            tot += e["hbm_bytes_per_launch"] * e["launches"]
            n += e["launches"]
    return round(tot / n) if n else None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="clouds per GPU")
    ap.add_argument("--points", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-clouds", type=int, default=2, help="sample size of the CPU baseline")
    ap.add_argument("--cpu-threads", type=int, default=16,
                    help="torch CPU threads of the baseline (16 measured fastest of {16,48,256} on the 2x EPYC 9575F "
                         "GPU host: the materialising reference algorithm is memory-bound and slows down with more)")
    ap.add_argument("--forward-only", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of replaying a captured HIP graph")
    ap.add_argument("--backbone-only", action="store_true", help="without the output head")
    ap.add_argument("--model", default="cls", choices=["cls", "reg", "inv"],
                    help="cls = BASELINE configs[1] (the headline metric); reg / inv = the layer schedules of configs[2] / "
                         "[3] in fp32 (their bf16 variants are not implemented yet)")
    return ap.parse_args()


def cpu_baseline(layers, product_sd, n_points, n_clouds, threads=16, head=True):
    """The oracle's materialising restatement of the same network (kind "port"), fwd+bwd, on the host cores."""
    from epn_pointcloud_amd import schedule as S
    from epn_pointcloud_amd.vgtk.so3conv import functional as L
    from epn_pointcloud_amd.vgtk import functional as fr
    from oracle import backbone_ref
    cores = max(1, min(threads, os.cpu_count() or 1))
    torch.set_num_threads(cores)
    tables = (torch.from_numpy(L.get_anchors(60)), torch.from_numpy(fr.kernel_points_raw(24)),
              torch.from_numpy(L.get_intra_idx()).long())
    ref = (backbone_ref.RefClsModel(layers, tables, out_mlps=(256,), pooling="attention") if head
           else backbone_ref.RefBackbone(layers, *tables))
    ref.load_from_product(product_sd)
    ref.train()
    pts = S.synthetic_clouds(n_clouds, n_points, "cpu", seed=2913)
    labels = torch.arange(n_clouds) % 40
    t0 = time.perf_counter()
    if head:
        logits, _ = ref(pts)
        torch.nn.functional.cross_entropy(logits, labels).backward()
    else:
        _, feats = ref(pts)
        feats.square().mean().backward()
    dt = time.perf_counter() - t0
    return {"value": n_clouds / dt, "unit": "point-clouds/s", "cores": cores, "kind": "port",
            "host_cpus": os.cpu_count(),
            "sample": f"{n_clouds} clouds N={n_points} A=60, fwd+bwd once, oracle/backbone_ref.py (torch CPU "
                      f"{torch.get_num_threads()} threads + C index kernels), {dt:.1f} s"}


def main():
    args = parse()
    from epn_pointcloud_amd import _lib, dp, models as M, ops, schedule as S
    _lib.get_lib()                                     # fail loudly if the HIP library is missing
    rank, local_rank, world = dp.init_from_env()
    assert world == args.gpus or world == 1 and args.gpus == 1, (world, args.gpus)
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    if args.model == "inv" and args.points == 1024:
        args.points = 2048                                  # 3DMatch patches (generate_eval.py:26,68)
    layers = {"cls": S.cls_so3net_schedule, "reg": S.reg_so3net_schedule,
              "inv": S.inv_so3net_schedule}[args.model](args.points)
    torch.manual_seed(2913)
    head = not args.backbone_only
    if not head:
        model = S.HotPathBackbone(layers, norm="BatchNorm2d" if args.model == "cls" else None)
    elif args.model == "cls":
        model = M.ClsSO3ConvModel(layers, out_mlps=(256,), pooling="attention")
    elif args.model == "reg":
        model = M.RegSO3ConvModel(layers)
    else:
        model = M.InvSO3ConvModel(layers)
    model = model.to(dev).train()
    dp.broadcast_parameters(model)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-3)
    scale = 0.4 if args.model == "inv" else 1.0            # 3DMatch search_radius (options.py:30)
    pts = S.synthetic_clouds(args.batch, args.points, dev, seed=2913 + rank, scale=scale)   # resident in HBM
    labels = (torch.arange(args.batch, device=dev) + rank) % 40
    if head and args.model == "reg":                        # pairs of clouds [b/2, 2, n, 3] (reg_so3net.py:31-33)
        pts = pts.view(args.batch // 2, 2, args.points, 3)

    def loss_of(out):
        if not head:
            return out.feats.square().mean()
        if args.model == "cls":
            return torch.nn.functional.cross_entropy(out[0], labels)
        if args.model == "inv":                             # descriptors are unit vectors: push them apart
            return (out[0] @ out[0].t()).square().mean()
        return out[0].square().mean() + out[1].square().mean()

    def compute():                      # the hot path: forward (+ loss + backward)
        if args.forward_only:
            with torch.no_grad():
                return loss_of(model(pts))
        loss = loss_of(model(pts))
        loss.backward()
        return loss

    def eager_step():
        opt.zero_grad(set_to_none=True)
        loss = compute()
        if not args.forward_only:
            dp.allreduce_gradients(params, world)
            opt.step()
        return loss

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # warm-up (also what torch requires before a capture: eager iterations on a side stream)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(max(args.warmup, 1) if not args.no_graph else args.warmup):
            eager_step()
    torch.cuda.current_stream(dev).wait_stream(side)
    fence()

    # The ~1300 launches of one step (kernels of this library, library GEMMs, small torch ops) are captured ONCE into a
    # HIP graph and replayed: same kernels, same work, no per-launch host latency / inter-kernel bubbles.  Gradient
    # all-reduce and the Adam update stay outside the graph (identical path for every world size).
    launch, graph, static_loss = "eager", None, None
    # single-process runs only: with a process group alive, RCCL's watchdog thread issues HIP calls of its own, which a
    # capture in progress does not tolerate on every stack -- not worth 1 % to the multi-GPU runs
    if not args.no_graph and world == 1:
        try:
            opt.zero_grad(set_to_none=True)      # gradients are (re)materialised inside the graph's memory pool
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                static_loss = compute()
            launch = "hipgraph"
        except Exception as e:                   # capture is an optimisation, never a requirement
            graph = None
            torch.cuda.synchronize()
            if rank == 0:
                print(f"[bench] HIP graph capture unavailable ({type(e).__name__}: {e}); eager launches", file=sys.stderr)

    def step():
        if graph is None:
            return eager_step()
        graph.replay()
        if not args.forward_only:
            dp.allreduce_gradients(params, world)
            opt.step()
        return static_loss

    if graph is not None:
        step()                                   # one untimed replay
    fence()
    if graph is None:
        ops.profile_begin()                      # eager: HIP events around every call of the timed region itself
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    fence()
    dt = time.perf_counter() - t0
    if graph is None:
        records, prof_steps = ops.profile_end(), args.steps
    else:
        # graph replays have no host-side launch points: for the roofline the same step runs eagerly, timed call by
        # call, right after the timed region (same process, same shapes, same kernels)
        prof_steps = min(args.steps, 3)
        opt.zero_grad(set_to_none=True)
        eager_step()                             # untimed: refills the eager allocator pool after the capture, so no
        fence()                                  # allocation stall sits between an event and the kernel it brackets
        ops.profile_begin()
        for _ in range(prof_steps):
            eager_step()
        fence()
        records = ops.profile_end()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    assert torch.isfinite(last.float()).all(), "non-finite output in the timed region"

    if rank == 0:
        # ---- roofline of the dominant kernel, from HIP events recorded around its launches in the timed region
        agg = {}
        for kind, key, flops, e0, e1 in records:
            k = KERNEL_OF.get(kind, kind)
            if kind.startswith("inter") and key[6] == 1:      # cin = 1 (first layer): dedicated kernels
                k = {"inter_fwd": "epn::inter_c1_fwd_kernel", "inter_bwd_weight": "epn::inter_c1_bwd_weight_kernel"}.get(kind, k)
            a = agg.setdefault(k, {"ms": 0.0, "flops": 0.0, "launches": 0})
            a["ms"] += e0.elapsed_time(e1)
            a["flops"] += flops
            a["launches"] += 1
        def roof(k):
            d = agg[k]
            ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
            return {"bound": "mfma", "kernel": k, "achieved": round(ach, 2), "peak": MFMA_F32_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(ach / MFMA_F32_PEAK_TFLOPS, 4),
                    "traffic": recorded_traffic(k) if args.model == "cls" and args.batch == 32 else None,
                    "launches": d["launches"], "avg_launch_ms": round(d["ms"] / d["launches"], 4)}

        dom = max(agg, key=lambda k: agg[k]["ms"])
        own = max((k for k in agg if k.startswith("epn::")), key=lambda k: agg[k]["ms"])
        roofline = roof(dom)
        roofline["traffic_note"] = ("HBM bytes/launch (avg over the family's launches) from the committed rocprofv3 "
                                    "--pmc passes, profiles/r01_pmc_per_kernel.json")
        roofline["own_kernel"] = roof(own)      # the dominant kernel of THIS library (the GEMM family is rocBLAS)
        roofline["per_kernel_ms_per_step"] = {k: round(v["ms"] / prof_steps, 3) for k, v in sorted(agg.items())}
        roofline["measured"] = (f"HIP events around every call of {prof_steps} eager step(s) "
                                + ("run right after the timed graph replays" if graph is not None else "= the timed region"))
        out = {
            "metric": (f"point-clouds/sec {'fwd' if args.forward_only else 'fwd+bwd'}, "
                       + ("ModelNet40" if args.model != "inv" else "3DMatch") + f" N={args.points} A=60"),
            "value": round(args.batch * world * args.steps / dt, 3), "unit": "point-clouds/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": {"cls": "ModelNet40 classification (cls_so3net_pn: 7 separable SO3 blocks",
                                    "reg": "ModelNet40 relative rotation (reg_so3net: 7 separable SO3 blocks",
                                    "inv": "3DMatch descriptor (inv_so3net_pn: 8 separable SO3 blocks"}[args.model]
                                   + ({"cls": " + ClsOutBlockPointnet head)", "reg": " + RelSO3OutBlockR head)",
                                       "inv": " + InvOutBlockMVD head)"}[args.model] if head else ", backbone only)")
                                   + f", B={args.batch}/GPU N={args.points} K={'/'.join(str(k) for k in sorted({l.nn for l in layers}, reverse=True))} "
                                   + f"A=60 fp32, {'fwd' if args.forward_only else 'fwd+bwd+Adam'}",
                       "global_batch": args.batch * world, "points": args.points, "anchors": 60, "launch": launch,
                       "hbm_peak_gb": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1),
                       "parallelism": f"dp{world}"},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline and args.model == "cls":
            out["cpu_baseline"] = cpu_baseline(layers, model.state_dict(), args.points, args.cpu_clouds,
                                               args.cpu_threads, head)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
