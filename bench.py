#!/usr/bin/env python3
"""bench.py -- point-clouds/sec of the EPN hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--model cls|reg|inv] [--dtype f32|bf16] [--forward-only]

With N > 1 and no launcher environment the script starts its own N ranks (one process per GPU, dp.launch);
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...` works
the same way.

Workloads (BASELINE.json `configs`):
  cls (default, the headline metric = configs[1]): the ModelNet40 classification network cls_so3net_pn, B=32 clouds per
      GPU, N=1024, K=32/16 neighbours, A=60 anchors, fp32 -- 7 separable blocks (FPS -> ball query -> grouping ->
      InterSO3Conv -> IntraSO3Conv + the block's norm/activation/skip glue) and the ClsOutBlockPointnet head;
  reg --dtype bf16 (configs[2]): ModelNet40 rotation estimation, 32 pairs = 64 clouds, N=1024, bf16 features;
  inv --dtype bf16 (configs[3]): 3DMatch descriptor, 64 patches, N=2048, bf16 features.
Random-init weights, synthetic unit-ball clouds and labels already resident in HBM.  One step = forward + loss + backward
(+ gradient all-reduce over RCCL when N > 1) + Adam update.  Weak scaling: per-GPU batch fixed.  Prints ONE JSON line.
"""
import argparse
import json
import os
import re
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md: "Peak FP32 (matrix)", "Peak BF16/FP16 MFMA" (dense), HBM3E peak
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}
HBM_PEAK_GBS = 8000.0
# Prediction only (configs.cls_dp_rank.predicted_eff_8gpu): bus bandwidth of a 31 MB ring all-reduce over 8 GPUs.  xGMI gives
# 7 links x ~153 GB/s per GPU; RCCL's measured bus bandwidth at tens of MB is well below the link sum -- 100 GB/s is a
# deliberately low figure, stated in the line as an assumption.
XGMI_ALLREDUCE_BUSBW_GBS = 100.0
XGMI_ALLREDUCE_LATENCY_MS = 0.05

def _latest_profile(suffix):
    """profiles/r0N_<suffix> of the latest round that committed one (tools/collect_round.sh)."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)))
    return found[-1] if found else os.path.join(ROOT, "profiles", "r05_" + suffix)


PMC_FILE = _latest_profile("pmc_per_kernel.json")                        # tools/collect_profiles.sh + pmc_summary.py
PMC_FILES = {"cls_f32": PMC_FILE,                                        # which committed pass profiled which workload
             "reg_bf16": _latest_profile("reg_bf16_pmc_per_kernel.json"),
             "inv_bf16": _latest_profile("inv_bf16_pmc_per_kernel.json")}


def recorded_traffic(kernel_family, pmc_file=PMC_FILE):
    """HBM bytes per launch of one kernel family from the committed rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE
    in separate runs, x1024, read side doubled per MI355X_MICROARCH.md): PMC counters cannot be read live here."""
    try:
        pmc = json.load(open(pmc_file))
    except OSError:
        return None
    tot = n = 0.0
    for name, e in pmc.items():
        if norm_kernel_name(name) == kernel_family and "hbm_bytes_per_launch" in e:
            tot += e["hbm_bytes_per_launch"] * e["launches"]
            n += e["launches"]
    return round(tot / n) if n else None


def recorded_step_traffic(pmc_file):
    """HBM GB per step of a profiled workload: sum over every kernel of the committed PMC pass of bytes/launch x launches,
    divided by the number of steps the profiled process ran (`_meta.step_equivalents`, written by tools/pmc_summary.py from
    the bench line's own count of compute() calls; the round-4 files carry no _meta: 10)."""
    try:
        pmc = json.load(open(pmc_file))
    except (OSError, TypeError):
        return None
    steps = (pmc.get("_meta") or {}).get("step_equivalents", 10)
    tot = sum(e["hbm_bytes_per_launch"] * e["launches"] for k, e in pmc.items()
              if k != "_meta" and isinstance(e, dict) and "hbm_bytes_per_launch" in e)
    return round(tot / steps / 1e9, 1) if tot else None


def algorithmic_work(layers, batch, points, esz, na=60, backward=True):
    """SURVEY 8(d)'s algorithmic work of one step of a schedule: (flops, minimal HBM bytes).  Flops per layer:
    schedule.hot_path_flops (weight generation + grouping + inter GEMM + intra GEMM); minimal bytes per separable block =
    B A (Cin P1 + 3 Cout P2) sizeof (read the input, write/read the inter output, write the intra output).  fwd+bwd = 3 x
    forward (dW + dX per contraction), as the survey prices it -- config 2 at B=32: 4.18 TF / 6.4 GB forward."""
    from epn_pointcloud_amd import schedule as S
    fl = sum(sum(d.values()) for d in S.hot_path_flops(layers, batch, points, na))
    by, p1 = 0.0, points
    for l in layers:
        p2 = -(-p1 // l.stride)
        by += batch * na * (l.cin * p1 + 3 * l.cout * p2) * esz
        p1 = p2
    m = 3.0 if backward else 1.0
    return fl * m, by * m


def norm_kernel_name(name):
    """rocprofv3's demangled kernel name -> the form epn_last_kernel() reports: no 'void ' prefix, no '(anonymous
    namespace)::', no argument list."""
    name = name.strip().strip('"')
    if name.startswith("void "):
        name = name[5:]
    name = name.replace("(anonymous namespace)::", "")
    depth = 0
    for i, ch in enumerate(name):            # cut the argument list: the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return name[:i]
    return name


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0, help="clouds per GPU (default: 32 for cls, 64 for reg / inv)")
    ap.add_argument("--points", type=int, default=0, help="points per cloud (default: 1024; 2048 for inv)")
    ap.add_argument("--dtype", default="", choices=["", "f32", "bf16"],
                    help="feature storage / GEMM operand type (default: f32 for cls, bf16 for reg / inv as BASELINE states)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-native-line", action="store_true",
                    help="skip the second measurement with the exact-f32 MFMA GEMMs (fp32, single rank, split form only)")
    ap.add_argument("--cpu-clouds", type=int, default=4, help="sample size of the CPU baseline (SURVEY 8d: B=4 chunks)")
    ap.add_argument("--cpu-samples", type=int, default=2, help="timed fwd+bwd passes of the CPU baseline (median reported)")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="torch CPU threads of the baseline; 0 (default) = sweep {16, 64, physical cores} on a one-cloud "
                         "forward pass and time the sample on the fastest")
    ap.add_argument("--forward-only", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of replaying a captured HIP graph")
    ap.add_argument("--backbone-only", action="store_true", help="without the output head")
    ap.add_argument("--model", default="cls", choices=["cls", "reg", "inv"])
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="skip the other single-GPU configs (cls forward only, reg bf16, inv bf16) that the default run "
                         "embeds under \"configs\"")
    ap.add_argument("--dp-path", action="store_true",
                    help="run the program ONE RANK of a multi-GPU job runs, on a single GPU: gradients gathered into the flat "
                         "GradBuckets buffer, the all-reduce issued on a 1-rank RCCL communicator after every replay, inside the "
                         "timed region (what --gpus N adds to a rank's step, minus the wire time)")
    ap.add_argument("--dp-collect", default="pack", choices=["pack", "accumulate"],
                    help="how gradients reach the flat buffer (dp.GradBuckets): one multi-tensor copy at the end of backward "
                         "(default) or autograd accumulating into views of it (round 4's form: one add per parameter + a fill)")
    ap.add_argument("--trace-loss", action="store_true",
                    help="diagnostics: print the loss of every warm-up / untimed / timed step to stderr (synchronises every step; "
                         "tools/nan_hunt.sh)")
    ap.add_argument("--policy", default="", help="A/B switch of the tuning tools: epn_set_kernel_policy value (e.g. 0x401), see "
                                                 "include/epn_so3conv.h; default = the library's own choices")
    return ap.parse_args()


def usable_cpus():
    """CPUs this process may run on (affinity mask / cgroup), not the host's count."""
    try:
        return len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return os.cpu_count() or 1


def cpu_baseline(layers, product_sd, n_points, n_clouds, threads=0, head=True, samples=2):
    """The oracle's materialising restatement of the same network (kind "port"), on the host cores: one warm-up
    forward of a single cloud, then `samples` timed forward + backward passes of `n_clouds` clouds (the median is
    reported); the forward share is reported separately (north_star states its >= 10x target on the forward pass).
    threads=0: BASELINE.md asks for "all physical cores" -- the thread count is swept over {16, 64, physical cores = half of
    the host's logical CPUs} with a one-cloud forward pass and the sample is timed on the fastest (`cores`); the sweep is in
    `sweep` (round 5, on the 2 x 64-core host of a GPU box: 16 threads 1.8 s, 64 threads 2.5 s, 256 threads 37 s -- the
    materialising algorithm is bound by memory traffic and allocator contention, not by cores)."""
    from epn_pointcloud_amd import schedule as S
    from epn_pointcloud_amd.vgtk.so3conv import functional as L
    from epn_pointcloud_amd.vgtk import functional as fr
    from oracle import backbone_ref
    avail = usable_cpus()
    cores = max(1, min(threads or 16, os.cpu_count() or 1))
    torch.set_num_threads(cores)
    tables = (torch.from_numpy(L.get_anchors(60)), torch.from_numpy(fr.kernel_points_raw(24)),
              torch.from_numpy(L.get_intra_idx()).long())
    ref = (backbone_ref.RefClsModel(layers, tables, out_mlps=(256,), pooling="attention") if head
           else backbone_ref.RefBackbone(layers, *tables))
    ref.load_from_product(product_sd)
    ref.train()
    pts = S.synthetic_clouds(n_clouds, n_points, "cpu", seed=2913)
    labels = torch.arange(n_clouds) % 40
    with torch.no_grad():                                   # warm-up: thread pool, allocator, C index library
        ref(pts[:1])
    sweep = {}
    if not threads:
        phys = max(1, min(avail, (os.cpu_count() or 2) // 2))      # SMT siblings are not "physical cores" 
        for t in sorted({min(16, avail), min(64, avail), phys}):
            torch.set_num_threads(t)
            with torch.no_grad():
                t0 = time.perf_counter()
                ref(pts[:1])
                sweep[t] = round(time.perf_counter() - t0, 2)
        cores = min(sweep, key=sweep.get)
        torch.set_num_threads(cores)
    runs = []
    for _ in range(max(1, samples)):
        ref.zero_grad(set_to_none=True)
        t0 = time.perf_counter()
        if head:
            logits, _ = ref(pts)
            loss = torch.nn.functional.cross_entropy(logits, labels)
        else:
            _, feats = ref(pts)
            loss = feats.square().mean()
        t1 = time.perf_counter()
        loss.backward()
        t2 = time.perf_counter()
        runs.append((t2 - t0, t1 - t0))
    runs.sort()
    tot, fwd = runs[(len(runs) - 1) // 2]                   # median (lower one of an even count)
    return {"value": round(n_clouds / tot, 4), "unit": "point-clouds/s", "cores": cores, "kind": "port",
            "samples": len(runs), "forward_only_value": round(n_clouds / fwd, 4),
            "sample": f"{n_clouds} clouds fwd+bwd x{len(runs)}, median {tot:.1f} s (fwd {fwd:.1f}); oracle/backbone_ref.py; "
                      f"threads = best of sweep_s; {avail}/{os.cpu_count()} host CPUs usable",
            "sweep": {"one_cloud_forward_s_by_threads": sweep}, "all_samples_s": [round(r[0], 2) for r in runs]}


def index_kernel_line(pts, layers, dev, reps=20):
    """FPS and ball query of the first layer, timed alone with HIP events (us per cloud and GB/s on
    the algorithmic bytes: FPS reads a cloud once (12 B/point) and writes m indices; the ball query reads support and
    query coordinates and writes K indices per query)."""
    from epn_pointcloud_amd.vgtk import pc as pctk
    b, n, _ = pts.shape
    l0 = layers[0]
    xyz = pts.permute(0, 2, 1).contiguous()
    m = -(-n // l0.stride)

    def timed(fn):
        fn()
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            out = fn()
        e1.record()
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) / reps * 1e3, out      # us per launch

    us_fps, (_, new_xyz) = timed(lambda: pctk.furthest_sample(xyz, m, False))
    us_bq, _ = timed(lambda: pctk.ball_query_index(new_xyz, xyz, l0.radius, l0.nn))
    fps_bytes = b * (n * 12 + m * 4)
    bq_bytes = b * (n * 12 + m * 12 + m * l0.nn * 4)
    return {"fps": {"n": n, "m": m, "us_per_launch": round(us_fps, 1), "us_per_cloud": round(us_fps / b, 2),
                    "GB/s": round(fps_bytes / us_fps / 1e3, 2), "bound": "latency (m-1 dependent rounds per cloud)"},
            "ball_query": {"queries": m, "support": n, "K": l0.nn, "us_per_launch": round(us_bq, 1),
                           "us_per_cloud": round(us_bq / b, 2), "GB/s": round(bq_bytes / us_bq / 1e3, 2)}}


WORKLOADS = {"cls": "ModelNet40 cls (cls_so3net_pn: 7 separable SO3 blocks",
             "reg": "ModelNet40 rotation (reg_so3net: 7 separable SO3 blocks",
             "inv": "3DMatch descriptor (inv_so3net_pn: 8 separable SO3 blocks"}
HEADS = {"cls": " + ClsOutBlockPointnet)", "reg": " + RelSO3OutBlockR)", "inv": " + InvOutBlockMVD)"}


def short_kernel(name):
    """Readable short form of a kernel name for the one-line report (the exact names -- rocprofv3's -- are in the detail
    file): 'epn::gemm_nt_x3_kernel<4, 2, 2, 4, 2>' -> 'gemm_nt_x3_kernel<4,2,2,4,2>'; names the C++ demangler leaves
    mangled (their template arguments contain __bf16 = 'DF16b') are decoded for the subset this library uses."""
    m = re.search(r"_GLOBAL__N_1(\d+)", name) if name.startswith("_Z") else None
    if m is None and name.startswith("_ZN3epn"):
        m = re.match(r"_ZN3epn(\d+)", name)
    if m:
        n = int(m.group(1))
        base, rest = name[m.end():m.end() + n], name[m.end() + n:]
        args = []
        if rest.startswith("I"):
            for tok in re.finditer(r"Li(\d+)E|Lb([01])E|(DF16b)|(f)|(d)|(E)", rest[1:]):
                if tok.group(6):
                    break
                args.append(tok.group(1) or ("true" if tok.group(2) == "1" else "false" if tok.group(2) else None)
                            or ("bf16" if tok.group(3) else "float" if tok.group(4) else "double"))
        return base + ("<" + ",".join(args) + ">" if args else "")
    name = norm_kernel_name(name)
    # (the C++ demangler predates 'DF16b' = __bf16 and renders some instances as 'bool _Accum': display only)
    name = name.replace("bool _Accum", "bf16")
    for pre in ("epn::", "at::native::", "at::cuda::"):
        name = name.replace(pre, "")
    return name.replace(", ", ",")[:72]


def algo_bytes(kind, key, esz):
    """Algorithmic HBM bytes of one call: the large operands read once + the result written
    once (weights and weight gradients are L2-resident / a few MB and not counted); the keys carry the dimensions."""
    if kind in ("inter_group", "inter_ungroup", "inter_ungroup_det"):
        b_, p1, p2, nn_, na, ks, cin, _ = key
        feats, grouped = b_ * p1 * na * cin, b_ * p2 * na * cin * ks
        return feats * (esz if kind == "inter_group" else 4) + grouped * esz
    if kind == "inter_bwd_data_f2":            # on-chip data gradient: dOut [cols, cout] read, dF [b p1 na, cin] accumulated into
        b_, p1, p2, nn_, na, ks, cin, cout = key
        return (b_ * p2 * na * cout + b_ * p1 * na * cin) * 4
    if kind in ("inter_gemm", "inter_gemm_dw", "inter_gemm_dg"):       # G [cols, cin ks] and out / dOut [cols, cout]
        b_, p1, p2, nn_, na, ks, cin, cout = key
        return b_ * p2 * na * (cin * ks + cout) * esz
    if kind in ("intra_gemm", "intra_gemm_dw"):
        if key and key[0] in ("spectral", "spectral_dw", "spectral_dA"):   # spectral buffers [pts 60, cin] / [pts 60, cout]
            _, pts, cin, cout = key
            return pts * 60 * (cin + cout) * esz
        b_, p_, na, kn, cin, cout = key
        return b_ * p_ * na * (kn * cin + cout) * esz
    if kind in ("conv1x1_gemm", "conv1x1_gemm_dw"):
        _, M, N, K = key
        return M * (N + K) * esz
    if kind == "intra_group":
        b_, p_, na, kn, c = key[:5]
        return b_ * p_ * na * c * esz * (1 + kn)
    if kind == "so3_basis":
        _, pts, c = key
        return 2 * pts * 60 * c * esz
    return 0


def aggregate(records, dtype_name):
    """Per device kernel: summed HIP-event time, algorithmic flops and bytes, launches.  A record's time is the pair of
    events around the call (tests pass a float of milliseconds and None instead)."""
    esz = 4 if dtype_name == "f32" else 2
    agg = {}
    for kind, key, flops, e0, e1, kname in records:
        k = kname or f"(host) {kind}"
        a = agg.setdefault(k, {"ms": 0.0, "flops": 0.0, "launches": 0, "bytes": 0.0})
        a["ms"] += e0.elapsed_time(e1) if e1 is not None else float(e0)
        a["flops"] += flops
        a["bytes"] += algo_bytes(kind, key, esz)
        a["launches"] += 1
    return agg


def split_factor(k):
    """Matrix instructions' worth of flops a kernel EXECUTES per algorithmic flop: 6 for the lossless three-piece bf16 form of
    an fp32 contraction, 3 for the two-piece fp16 form, 1 otherwise -- from the template arguments the profiler prints
    (gemm_nt_x3_kernel<WGM,WGN,TM,TN,NSTG[,NPL]>, gemm_tn_x3_kernel<..,BR[,NPL]>, gemm_tn_f32_kernel<..,BR,X3>)."""
    m = re.search(r"gemm_(nt_x3|tn_x3|tn_f32)_kernel<([^>]*)>", k)
    if m:
        a = [v.strip() for v in m.group(2).split(",")]
        last = a[5] if len(a) > 5 else ("3" if m.group(1) != "tn_f32" else "0")
        return {"3": 6, "true": 6, "2": 3}.get(last, 1)
    if "_x3_" in k or (k.startswith("epn::inter_fx") and "<float" in k):
        return 6
    return 1


def is_split_kernel(k):
    return split_factor(k) > 1


def price(k, d, dtype_name, traffic=None):
    """One kernel against the roof that bounds it: the larger of (algorithmic bytes / 8 TB/s) and (executed flops / the
    dense MFMA peak of the pipe it runs on) decides `bound`; `achieved` / `frac` are stated against that roof."""
    sec = d["ms"] * 1e-3
    split = is_split_kernel(k)
    # split form: every fp32 multiply-add is six bf16 MFMA multiply-adds (fp32 accumulate, csrc/gemm_x3.hip): the roof is
    # the bf16 matrix pipe and `achieved` the flops it EXECUTES; the fp32-equivalent rate is reported beside it
    peak = PEAK_TFLOPS["bf16"] if split else PEAK_TFLOPS[dtype_name]
    exec_flops = d["flops"] * split_factor(k)
    t_mfma, t_hbm = exec_flops / (peak * 1e12), d["bytes"] / (HBM_PEAK_GBS * 1e9)
    common = {"kernel": short_kernel(k), "traffic": traffic, "launches": d["launches"],
              "avg_launch_ms": round(d["ms"] / d["launches"], 4)}
    if t_hbm > t_mfma or d["flops"] == 0:
        ach = d["bytes"] / sec / 1e9
        r = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": round(ach / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": round(d["bytes"] / d["launches"])}
    else:
        ach = exec_flops / sec / 1e12
        r = {"bound": "mfma", "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4)}
    if split:
        r["algorithmic_fp32_tflops"] = round(d["flops"] / sec / 1e12, 2)
        r["vs_fp32_mfma_peak"] = round(d["flops"] / sec / 1e12 / PEAK_TFLOPS["f32"], 3)
    r.update(common)
    return r


def roofline_of(records, prof_steps, dtype_name, with_traffic, graph_mode):
    """(compact roofline object of the one-line report, per-kernel detail for the side file) of one measured workload from
    the per-call HIP-event records (ops.profile_end()): every record carries the device kernel the library itself reported
    for that call (epn_last_kernel), so the detail's names are the names rocprofv3 prints (profiles/r04_kernel_stats.csv,
    normalised by norm_kernel_name)."""
    agg = aggregate(records, dtype_name)

    def traffic(k):
        return recorded_traffic(k, with_traffic) if with_traffic else None

    priced = [k for k in agg if agg[k]["flops"] > 0 or agg[k]["bytes"] > 0]
    dom = max(priced or agg, key=lambda k: agg[k]["ms"])
    roofline = price(dom, agg[dom], dtype_name, traffic(dom))
    # the other roof of the pair: the heaviest kernel bound by it
    all_priced = {k: price(k, agg[k], dtype_name, traffic(k)) for k in priced}
    other = "hbm" if roofline["bound"] == "mfma" else "mfma"
    # (the cin = 1 first layer and the point-wise head are VALU / latency kernels: never the representative of a roof)
    cand = [k for k in priced if all_priced[k]["bound"] == other and "inter_c1" not in k and "pointnet" not in k]
    if cand:
        k2 = max(cand, key=lambda k: agg[k]["ms"])
        o = all_priced[k2]
        roofline["dominant_memory_bound_kernel" if other == "hbm" else "dominant_mfma_kernel"] = {
            x: o[x] for x in ("kernel", "achieved", "unit", "frac", "traffic", "avg_launch_ms")}
    # the same records per (call kind, layer dimensions): which LAYER a kernel family is slow on
    esz = 4 if dtype_name == "f32" else 2
    calls = {}
    for kind, key, flops, e0, e1, kname in records:
        if not key:
            continue
        c = calls.setdefault((kind, tuple(key), kname), [0.0, 0, flops, algo_bytes(kind, key, esz)])
        c[0] += e0.elapsed_time(e1) if e1 is not None else float(e0)
        c[1] += 1
    per_call = [{"kind": kind, "key": list(key), "kernel": short_kernel(kname or ""), "launches_per_step": round(n / prof_steps, 2),
                 "avg_ms": round(ms / n, 4), "GB/s": round(by / (ms / n) / 1e6, 1) if by else None,
                 "TFLOP/s": round(fl / (ms / n) / 1e9, 1) if fl else None}
                for (kind, key, kname), (ms, n, fl, by) in sorted(calls.items(), key=lambda kv: -kv[1][0])]
    detail = {
        "per_call": per_call,
        # keyed by the READABLE instance name (short_kernel: also decodes the names the C++ demangler leaves mangled -- every
        # bf16 instance, whose template arguments contain 'DF16b'); `kernel_exact` keeps the library's / rocprofv3's string
        "per_kernel": {(short_kernel(k) if not k.startswith("(host)") else k):
                       dict(all_priced.get(k, {}), kernel_exact=k, ms_per_step=round(v["ms"] / prof_steps, 3),
                            launches_per_step=round(v["launches"] / prof_steps, 1))
                       for k, v in sorted(agg.items())},
        "per_kernel_ms_sum": round(sum(v["ms"] for v in agg.values()) / prof_steps, 3),
        "dominant_kernel_exact": dom,
        "traffic_note": ("HBM bytes/launch (avg over the kernel's launches) from the committed rocprofv3 --pmc passes, "
                         + os.path.relpath(with_traffic or PMC_FILE, ROOT)
                         + ("" if with_traffic else " (this workload was not profiled: traffic null)")),
        "split_form_note": ("fp32 operands split losslessly into 3 bf16 pieces, 6 piece products per multiply on "
                            "v_mfma_f32_32x32x16_bf16, fp32 accumulate (csrc/gemm_x3.hip): fp32 accuracy; achieved = executed "
                            "bf16 flops = 6 x algorithmic"),
        "measured": (f"HIP events on the launch stream around EVERY C-ABI call of {prof_steps} eager step(s) "
                     + ("run right after the timed graph replays (same process, kernels and shapes; each step enqueued "
                        "behind a spinning kernel so that the brackets hold device time only; the skip branch's kernels "
                        "share the GPU from a second stream, so the sum can exceed the step)"
                        if graph_mode else "= the timed region")
                     + "; kernel names: the library's own report per call (epn_last_kernel), i.e. rocprofv3's names"),
    }
    return roofline, detail


_SLEEP_CYCLES_PER_MS = None


def hold_gpu(ms, dev):
    """Keep the current stream busy for about `ms` milliseconds (torch's spin kernel, calibrated once): enqueued in front
    of an eager step that is timed call by call, it lets the host run ahead, so that the HIP events around a call bracket
    device time only -- without it a host-bound stretch (hundreds of small launches) shows up inside the brackets."""
    global _SLEEP_CYCLES_PER_MS
    if _SLEEP_CYCLES_PER_MS is None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(100000)
        torch.cuda.synchronize(dev)
        e0.record()
        torch.cuda._sleep(2000000)
        e1.record()
        torch.cuda.synchronize(dev)
        _SLEEP_CYCLES_PER_MS = 2000000 / max(e0.elapsed_time(e1), 1e-3)
    torch.cuda._sleep(int(ms * _SLEEP_CYCLES_PER_MS))


def measure(cfg, rank, local_rank, world, dev, first=True):
    """One workload (cfg: model, dtype, forward_only, steps, warmup, batch, points, no_graph, backbone_only): builds the
    network, warms up, captures the step into a HIP graph, times exactly cfg.steps steps between barrier + synchronize
    fences (max over ranks), then profiles a few eager steps call by call.  Returns (json_dict, handles) on every rank."""
    from epn_pointcloud_amd import dp, models as M, ops, schedule as S
    from epn_pointcloud_amd import gemm as _gemm
    TRACE = bool(getattr(cfg, "trace_loss", False))
    dtype_name = cfg.dtype or ("f32" if cfg.model == "cls" else "bf16")
    split_gemm = dtype_name == "f32" and _gemm.FP32_MODE != "native"
    exec_x = {"split": 6, "f16x2": 3}.get(_gemm.FP32_MODE, 1) if dtype_name == "f32" else 1
    fdtype = torch.float32 if dtype_name == "f32" else torch.bfloat16
    batch = cfg.batch or (32 if cfg.model == "cls" else 64)
    points = cfg.points or (2048 if cfg.model == "inv" else 1024)   # 3DMatch patches (generate_eval.py:26,68)
    layers = {"cls": S.cls_so3net_schedule, "reg": S.reg_so3net_schedule,
              "inv": S.inv_so3net_schedule}[cfg.model](points)
    torch.manual_seed(2913)                                 # same seed on every rank: replicas start identical
    head = not cfg.backbone_only
    if not head:
        model = S.HotPathBackbone(layers, norm="BatchNorm2d" if cfg.model == "cls" else None, model=cfg.model)
    elif cfg.model == "cls":
        model = M.ClsSO3ConvModel(layers, out_mlps=(256,), pooling="attention")
    elif cfg.model == "reg":
        model = M.RegSO3ConvModel(layers)
    else:
        model = M.InvSO3ConvModel(layers)
    model = S.set_feature_dtype(model.to(dev).train(), fdtype)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-3)
    scale = 0.4 if cfg.model == "inv" else 1.0             # 3DMatch search_radius (options.py:30)
    pts = S.synthetic_clouds(batch, points, dev, seed=2913 + rank, scale=scale)   # resident in HBM
    flat_pts = pts
    labels = (torch.arange(batch, device=dev) + rank) % 40
    if head and cfg.model == "reg":                         # pairs of clouds [b/2, 2, n, 3] (reg_so3net.py:31-33)
        pts = pts.view(batch // 2, 2, points, 3)

    def loss_of(out):
        if not head:
            return out.feats.float().square().mean()
        if cfg.model == "cls":
            return torch.nn.functional.cross_entropy(out[0], labels)
        if cfg.model == "inv":                              # descriptors are unit vectors: push them apart
            return (out[0] @ out[0].t()).square().mean()
        return out[0].square().mean() + out[1].square().mean()

    # gradients live in one flat buffer cut into per-stage buckets (dp.GradBuckets): zero() instead of zero_grad, the
    # all-reduce runs on slices of it -- issued from backward hooks (eager) or right after the replayed graph
    # (world > 1 only: a single rank lets autograd hand its gradient tensors to p.grad directly -- no per-parameter
    # accumulate kernel, no zero fill)
    # --dp-path: the same on a world of one (forced collectives on a single-rank communicator): the rank program measured
    dp_path = bool(getattr(cfg, "dp_path", False)) and not cfg.forward_only
    collect = getattr(cfg, "dp_collect", "pack")
    buckets = (dp.GradBuckets(dp.stage_buckets(model), world, hooks=False, collect=collect, force_collectives=dp_path)
               if not cfg.forward_only and (world > 1 or dp_path) else None)
    calls = [0]

    def compute():                      # the hot path: forward (+ loss + backward)
        calls[0] += 1
        if cfg.forward_only:
            with torch.no_grad():
                return loss_of(model(pts))
        if buckets is not None:
            buckets.zero()
        else:
            for p in params:
                p.grad = None
        loss = loss_of(model(pts))
        loss.backward()
        if buckets is not None:
            buckets.pack()              # (pack form) the step's gradients -> the flat buffer, one multi-tensor copy
        return loss

    def finish():
        if not cfg.forward_only:
            if buckets is not None:
                # a replayed graph has no hook points: nothing to overlap with, ONE all-reduce of the whole buffer
                buckets.finish(one_collective=graph is not None)
            opt.step()

    def eager_step():
        loss = compute()
        finish()
        return loss

    # warm-up (also what torch requires before a capture: eager iterations on a side stream); single-rank so far --
    # the process group is created AFTER the capture (RCCL's watchdog thread issues HIP calls of its own, which a
    # capture in progress does not tolerate), replicas are identical by construction (same seed) and re-synchronised
    # by the broadcast below
    torch.cuda.reset_peak_memory_stats(dev)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(max(cfg.warmup, 1)):
            wl = compute()
            if not cfg.forward_only:
                opt.step()
            if TRACE and rank == 0:                  # diagnostics (--trace-loss): synchronises every step
                print(f"[bench] warm-up loss {float(wl):.6g}", file=sys.stderr, flush=True)
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize(dev)

    # The ~1300 launches of one step are captured ONCE into a HIP graph and replayed: same kernels, same work, no
    # per-launch host latency.  Gradient all-reduce and the Adam update stay outside the graph.
    launch, graph, static_loss = "eager", None, None
    if not cfg.no_graph:
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                static_loss = compute()
            launch = "hipgraph"
            calls[0] -= 1                        # captured, not executed: not a step's worth of kernels
        except Exception as e:                   # capture is an optimisation, never a requirement
            graph = None
            torch.cuda.synchronize()
            if rank == 0:
                print(f"[bench] HIP graph capture unavailable ({type(e).__name__}: {e}); eager launches", file=sys.stderr)

    if first or dp_path:
        dp.init_from_env(force=dp_path)          # RCCL communicator (world > 1 or --dp-path), after the capture
    dp.broadcast_parameters(model)
    if graph is None and buckets is not None:    # eager: per-stage all-reduce from backward hooks
        buckets = dp.GradBuckets(dp.stage_buckets(model), world, hooks=True, force_collectives=dp_path)

    def step():
        if graph is None:
            return eager_step()
        graph.replay()
        calls[0] += 1                            # a replay runs a step's kernels without calling compute()
        finish()
        return static_loss

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    ul = step()                                  # one untimed step on the final path (RCCL lazy init included)
    fence()
    if TRACE and rank == 0:
        print(f"[bench] untimed-step loss {float(ul):.6g}", file=sys.stderr, flush=True)
    f2_mode = dtype_name == "f32" and _gemm.FP32_MODE == "f16x2"
    if f2_mode:
        _gemm.f16x2_overflow_count(reset=True)   # the timed region must leave the overflow sentinel at zero (checked below)
    _gemm.fixed_point_range_count(reset=True)    # ... and the range sentinel of the fixed-point transpose of the grouping
    if buckets is not None:
        buckets.record_timing, buckets.timings = True, []
    if graph is None:
        ops.profile_begin()                      # eager: HIP events around every call of the timed region itself
    t0 = time.perf_counter()
    for _ in range(cfg.steps):
        last = step()
        if TRACE and rank == 0:
            print(f"[bench] loss {float(last):.6g}", file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0          # this rank's own time, before it waits for the others
    fence()
    dt = time.perf_counter() - t0
    # ---- what a multi-rank line needs to explain itself (world > 1, or the rank program on one GPU): per-rank step time,
    # the collective phase as the compute stream sees it, its bus bandwidth, and the same step WITHOUT the collective timed
    # right here on every rank at once (so the comparison shares this run's clocks, power state and neighbours)
    dp_info = None
    if buckets is not None:
        ar_ms = buckets.collective_ms()
        buckets.record_timing = False
        nbytes = buckets.flat.numel() * 4
        ar = sum(ar_ms) / max(len(ar_ms), 1)
        dp_info = {"ms_per_step_local": round(dt_local / cfg.steps * 1e3, 3),
                   "allreduce_ms": round(ar, 3), "allreduce_mb": round(nbytes / 2 ** 20, 1),
                   "allreduce_per_step": len(ar_ms) // max(cfg.steps, 1),
                   # ring convention (rccl-tests' busbw): 2 (N - 1) / N of the buffer crosses every link
                   "bus_gbps": round(2.0 * (world - 1) / max(world, 1) * nbytes / max(ar, 1e-6) / 1e6, 1) if world > 1 else None}

        dp_info["no_comm_ms_per_step_local"] = None
        if graph is not None:                    # (eager mode issues its collectives from backward hooks: no comm-free form)
            fence()
            t1 = time.perf_counter()
            for _ in range(cfg.steps):           # the rank program minus its collective (replicas drift: timing only)
                graph.replay()
                opt.step()
            torch.cuda.synchronize()
            dp_info["no_comm_ms_per_step_local"] = round((time.perf_counter() - t1) / cfg.steps * 1e3, 3)
            fence()
        if world > 1:                            # every rank's two figures on rank 0
            mine = torch.tensor([dp_info["ms_per_step_local"], dp_info["no_comm_ms_per_step_local"] or 0.0, dp_info["allreduce_ms"]],
                                dtype=torch.float64, device=dev)
            allr = [torch.zeros_like(mine) for _ in range(world)]
            torch.distributed.all_gather(allr, mine)
            rows = torch.stack(allr).cpu()
            dp_info["per_rank_ms_per_step"] = [round(float(rows[:, 0].min()), 3), round(float(rows[:, 0].max()), 3)]
            if graph is not None:
                dp_info["per_rank_no_comm_ms"] = [round(float(rows[:, 1].min()), 3), round(float(rows[:, 1].max()), 3)]
            dp_info["per_rank_allreduce_ms"] = [round(float(rows[:, 2].min()), 3), round(float(rows[:, 2].max()), 3)]
            dp.broadcast_parameters(model)       # re-synchronise the replicas the comm-free steps let drift
    if graph is None:
        records, prof_steps = ops.profile_end(), cfg.steps
    else:
        # graph replays have no host-side launch points: for the roofline the same step runs eagerly, timed call by
        # call, right after the timed region (same process, same shapes, same kernels)
        prof_steps = min(cfg.steps, 3)
        t_h = time.perf_counter()
        eager_step()                             # untimed: refills the eager allocator pool after the capture, so no
        host_ms0 = (time.perf_counter() - t_h) * 1e3     # allocation stall sits between an event and the kernel it brackets
        fence()
        host_ms = host_ms0
        for _ in range(prof_steps):
            t_h = time.perf_counter()
            hold_gpu(100.0 + 3.0 * host_ms, dev)              # the host enqueues the whole step behind a spinning kernel
            ops.profile_begin() if _ == 0 else None
            eager_step()
            host_ms = max(host_ms, (time.perf_counter() - t_h) * 1e3)
            fence()
        records = ops.profile_end()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    assert torch.isfinite(last.float()).all(), "non-finite output in the timed region"
    f2_overflow = _gemm.f16x2_overflow_count(reset=True) if f2_mode else None
    if f2_overflow:
        raise RuntimeError(f"{f2_overflow} wave(s) of two-piece fp16 GEMMs ended with a non-finite accumulator during the timed "
                           "region: a reported max|operand| was too small (epn_f16x2_overflow_count) -- the measurement is void")

    fx_range = _gemm.fixed_point_range_count(reset=True)
    if fx_range:
        raise RuntimeError(f"{fx_range} workgroup(s) of the fixed-point transpose of the grouping saw a contribution beyond the range "
                           "its reported max|dG| allows (epn_inter_ungroup_cloud_range_count) -- the measurement is void")

    nn_desc = "/".join(str(k) for k in sorted({l.nn for l in layers}, reverse=True))
    out = {
        "metric": (f"point-clouds/sec {'fwd' if cfg.forward_only else 'fwd+bwd'}, "
                   + ("ModelNet40" if cfg.model != "inv" else "3DMatch") + f" N={points} A=60"),
        "value": round(batch * world * cfg.steps / dt, 3), "unit": "point-clouds/s",
        "n_gpus": world, "steps": cfg.steps, "warmup": cfg.warmup,
        "ms_per_step": round(dt / cfg.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": dtype_name, "data": "synthetic",
        "config": {"workload": WORKLOADS[cfg.model] + (HEADS[cfg.model] if head else ", backbone only)")
                               + f", B={batch}/GPU N={points} K={nn_desc} A=60 "
                               + ({"split": "fp32 (contractions: lossless 3xbf16 split, fp32 accumulate)",
                                   "f16x2": "fp32 (contractions: 2-piece fp16 split x 3 MFMA products, fp32 accumulate)"}
                                  .get(_gemm.FP32_MODE, "fp32") if dtype_name == "f32" else "bf16 features / fp32 accumulate")
                               + f", {'fwd' if cfg.forward_only else 'fwd+bwd+Adam'}",
                   "global_batch": batch * world, "launch": launch,
                   "hbm_peak_gb": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1),
                   "fp32_gemm": _gemm.FP32_MODE if dtype_name == "f32" else None,
                   "parallelism": f"dp{world}"},
    }
    if os.environ.get("EPN_INTER_MODE", "auto") != "auto":
        out["config"]["inter_mode"] = os.environ["EPN_INTER_MODE"]
    if dp_path:
        out["config"]["dp_path"] = f"{collect}+1 all-reduce" if graph is not None else "hooks"
    if f2_overflow is not None:
        out["f16x2_overflow"] = f2_overflow      # the sentinel of the two-piece kernels over the timed region (0, or the run raised)
    out["fixed_point_range"] = fx_range          # ... of the fixed-point transpose of the grouping (0, or the run raised)
    if dp_info is not None:
        # efficiency against the rank program measured in THIS run: value / (N x the comm-free rate of the slowest rank)
        slow = dp_info.get("per_rank_no_comm_ms", [None, dp_info["no_comm_ms_per_step_local"]])[1]
        dp_info["eff_vs_rank_program"] = round(slow / (dt / cfg.steps * 1e3), 4) if slow else None
        out["dp"] = dp_info
    if rank == 0:
        # PMC passes exist for the cls fp32 step (B=32) and the rotation network's bf16 step (B=64): their kernels' traffic
        profiled = (not cfg.forward_only) and ((cfg.model, batch, dtype_name) in (("cls", 32, "f32"), ("reg", 64, "bf16"),
                                                                                  ("inv", 64, "bf16")))
        pmc_file = PMC_FILES.get(f"{cfg.model}_{dtype_name}") if profiled else None
        out["roofline"], detail = roofline_of(records, prof_steps, dtype_name, pmc_file, graph is not None)
        if head:
            # what the STEP achieves (SURVEY 8d's algorithmic work of the config / the measured step time) beside the
            # dominant kernel's figure; HBM GB per step from the committed PMC pass of the same workload
            fl, by = algorithmic_work(layers, batch, points, 4 if dtype_name == "f32" else 2, backward=not cfg.forward_only)
            tf = fl / (dt / cfg.steps) / 1e12
            st = {"algorithmic_tflops": round(tf, 1), "frac_fp32_matrix": round(tf / PEAK_TFLOPS["f32"], 3),
                  (f"frac_bf16_pipe_x{exec_x}" if exec_x > 1 else "frac_bf16_pipe"):
                      round(tf * exec_x / PEAK_TFLOPS["bf16"], 3),
                  "algorithmic_gb": round(by / 1e9, 1)}
            gb = recorded_step_traffic(pmc_file) if pmc_file else None
            if gb:
                st["hbm_gb"], st["hbm_over_algorithmic"] = gb, round(gb * 1e9 / by, 1)
            out["roofline"]["step"] = st
        detail["compute_calls"] = calls[0]
        print(f"[bench] compute_calls {calls[0]}", file=sys.stderr)
    else:
        detail = None
    handles = dict(detail=detail, model=model, layers=layers, flat_pts=flat_pts, compute=compute, finish=finish, opt=opt, graph=graph,
                   head=head, points=points, batch=batch, split_gemm=split_gemm)
    return out, handles


DETAIL_FILE = os.environ.get("EPN_BENCH_DETAIL") or os.path.join(ROOT, "bench_detail.json")
LINE_LIMIT = 3000            # bytes: the driver keeps ~9 KB of stdout; round 3's 35 KB line could not be parsed


def compact_config(o):
    """An embedded config of the one-line report: the numbers, the priced dominant kernel and the step-level figures only
    (the workload text of each is in the detail file and in DESIGN.md 5; the key names the BASELINE config)."""
    if "error" in o:
        return {"error": o["error"][:120]}
    r = o.get("roofline", {})
    c = {"value": o["value"], "ms_per_step": o["ms_per_step"], "steps": o["steps"]}
    if "overhead_ms" in o:          # the rank program: the headline's kernels, what matters is its difference to the headline
        c.update({k: o[k] for k in ("overhead_ms", "vs_headline", "collect", "allreduce_ms", "no_comm_ms", "predicted_eff_8gpu",
                                    "wire") if k in o})     # (`assumes`, the long form of `wire`, is in the detail file)
        return c
    c.update(bound=r.get("bound"), frac=r.get("frac"), kernel=r.get("kernel"))
    if r.get("traffic"):
        c["traffic"] = r["traffic"]
    for k in ("dominant_memory_bound_kernel", "dominant_mfma_kernel"):
        if k in r:
            c["other_roof"] = {"kernel": r[k]["kernel"], "frac": r[k]["frac"]}        # (the roof `bound` does not name)
    if "step" in r:
        st = r["step"]
        c["step"] = {"tflops": st["algorithmic_tflops"]}
        if st.get("hbm_gb"):
            c["step"].update(hbm_gb=st["hbm_gb"], hbm_x=st["hbm_over_algorithmic"])
    if "vs_cpu_forward" in o:
        c["vs_cpu_forward"] = o["vs_cpu_forward"]
    return c


def compact_line(out):
    """The ONE stdout line (< LINE_LIMIT bytes): the contract's keys, `roofline` of the dominant kernel, `cpu_baseline`,
    the native-fp32 comparison and the other single-GPU configs reduced to their numbers; everything per-kernel lives in the
    detail file named by "detail"."""
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                "scaling", "vs_baseline", "dtype", "data", "config") if k in out}
    if "roofline" in out:
        line["roofline"] = out["roofline"]
    for k in ("f16x2_overflow", "fixed_point_range", "dp"):          # the overflow sentinel over the timed region; the multi-rank self-explanation
        if k in out:
            line[k] = out[k]
    if "cpu_baseline" in out:
        line["cpu_baseline"] = {k: out["cpu_baseline"][k] for k in
                                ("value", "unit", "cores", "kind", "samples", "forward_only_value", "sample")}
        sw = out["cpu_baseline"].get("sweep", {}).get("one_cloud_forward_s_by_threads")
        if sw:
            line["cpu_baseline"]["sweep_s"] = sw
    if "native_fp32_mfma" in out:
        n = out["native_fp32_mfma"]
        line["native_fp32_mfma"] = ({"value": n["value"], "ms_per_step": n["ms_per_step"]} if "value" in n
                                    else {"error": n["error"][:120]})
    if out.get("fp32_modes"):
        line["fp32_modes"] = {k: ({"value": v["value"], "ms_per_step": v["ms_per_step"]} if "value" in v else {"error": v["error"][:80]})
                              for k, v in out["fp32_modes"].items()}
    if "index_kernels" in out:
        line["index_kernels"] = {k + "_us": v["us_per_launch"] for k, v in out["index_kernels"].items()}
    if "configs" in out:
        line["configs"] = {k: compact_config(v) for k, v in out["configs"].items()}
    line["detail"] = os.path.relpath(DETAIL_FILE, ROOT) if DETAIL_FILE.startswith(ROOT) else DETAIL_FILE
    return line


def fit_line(line):
    """json text of the line, trimmed (optional objects first) until it is under LINE_LIMIT: a line the driver cannot
    parse is worth nothing, so nothing optional may push the contract's keys out of its window."""
    def drop_other_roofs(l):
        for c in l.get("configs", {}).values():
            c.pop("other_roof", None)

    def drop_workloads(l):
        for c in l.get("configs", {}).values():
            c.pop("traffic", None)
    def drop_roofs_of_line(l):
        r = l.get("roofline", {})
        r.pop("dominant_memory_bound_kernel", None)
        r.pop("dominant_mfma_kernel", None)

    def drop_config_steps(l):
        for c in l.get("configs", {}).values():
            c.pop("step", None)
            c.pop("assumes", None)
    trims = [drop_workloads, drop_other_roofs, lambda l: l.pop("index_kernels", None),
             lambda l: l.get("cpu_baseline", {}).pop("sweep_s", None), drop_config_steps, drop_roofs_of_line,
             lambda l: l.pop("configs", None), lambda l: l.pop("native_fp32_mfma", None), lambda l: l.pop("fp32_modes", None)]
    text = json.dumps(line)
    for t in trims:
        if len(text) < LINE_LIMIT:
            break
        try:                                     # a line is ALWAYS emitted: a failing trim is skipped, not fatal
            t(line)
        except Exception:
            pass
        text = json.dumps(line)
    return text


def emit(out, detail):
    """stdout: the compact line; stderr + DETAIL_FILE: the complete record (every kernel of every config)."""
    full = dict(out, detail=detail)
    try:
        with open(DETAIL_FILE, "w") as f:
            json.dump(full, f, indent=1)
    except OSError as e:
        print(f"[bench] cannot write {DETAIL_FILE}: {e}", file=sys.stderr)
    print("[bench] detail: " + json.dumps(full), file=sys.stderr, flush=True)
    flush_native_stdout()
    print(fit_line(compact_line(out)), flush=True)


def flush_native_stdout():
    """Native libraries write to C stdio, which is block-buffered when stdout is a pipe or a file: RCCL's five-line version
    banner (printed when a communicator is created) would otherwise be flushed at exit, AFTER the JSON line -- and the driver
    parses the LAST stdout line.  Flush C stdio now, so that such output precedes the line."""
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except (OSError, AttributeError):
        pass


def stdout_to_stderr():
    """After the JSON line: whatever native code still prints at teardown (communicator destruction) goes to stderr."""
    flush_native_stdout()
    try:
        os.dup2(2, 1)
    except OSError:
        pass


def main():
    args = parse()
    from epn_pointcloud_amd import _lib, dp
    rank, local_rank, world = dp.env_world()
    if args.gpus > 1 and world == 1 and "EPN_DP_CHILD" not in os.environ:
        sys.exit(dp.launch(args.gpus, timeout=float(os.environ.get("EPN_DP_TIMEOUT", "1500"))))   # no launcher: start the ranks ourselves
    assert world == args.gpus, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    if rank != 0:
        stdout_to_stderr()                              # under torchrun all ranks share stdout: only rank 0 owns it
    _lib.get_lib()                                      # fail loudly if the HIP library is missing
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    dev = dp.local_device(local_rank)
    torch.cuda.set_device(dev)
    if args.policy:      # needs a -DEPN_TUNING library (python -m epn_pointcloud_amd.build --tuning; EPN_LIB=...)
        _lib.check(_lib.get_lib().epn_set_kernel_policy(int(args.policy, 0)), "set_kernel_policy")

    out, H = measure(args, rank, local_rank, world, dev)
    detail = {"headline": H["detail"]}
    cpu = None
    if rank == 0:
        from epn_pointcloud_amd import gemm as _gemm
        if world == 1 and H["split_gemm"] and not args.no_native_line:
            # the same step with the weight contractions / basis change in the OTHER fp32 forms (gemm.FP32_MODES: the fp32
            # matrix instruction v_mfma_f32_32x32x2_f32, the lossless three-piece bf16 form, the two-piece fp16 form), measured
            # right here: same box, same process, same inputs
            compute, finish, graph = H["compute"], H["finish"], H["graph"]
            mode0 = _gemm.FP32_MODE
            out["fp32_modes"] = {}
            for mode in [m for m in ("native", "split", "f16x2") if m != mode0]:
                try:
                    _gemm.set_fp32_mode(mode)
                    compute()                                   # eager once (allocations), then its own graph
                    if not args.forward_only:
                        H["opt"].step()
                    torch.cuda.synchronize()
                    g2 = None
                    if graph is not None:
                        g2 = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g2, capture_error_mode="thread_local"):
                            compute()

                    def other_step():
                        if g2 is None:
                            compute()
                        else:
                            g2.replay()
                        finish()
                    other_step()
                    torch.cuda.synchronize()
                    tn0 = time.perf_counter()
                    for _ in range(args.steps):
                        other_step()
                    torch.cuda.synchronize()
                    dtn = time.perf_counter() - tn0
                    rec = {"value": round(H["batch"] * args.steps / dtn, 3), "unit": "point-clouds/s",
                           "ms_per_step": round(dtn / args.steps * 1e3, 3), "steps": args.steps}
                    del g2
                except Exception as e:                          # a report, never a requirement
                    rec = {"error": f"{type(e).__name__}: {e}"}
                finally:
                    _gemm.set_fp32_mode(mode0)
                if mode == "native":
                    out["native_fp32_mfma"] = dict(rec, note="same step, fp32 contractions on v_mfma_f32_32x32x2_f32 (DESIGN.md "
                                                             "3.2b); value / this = speed-up of the headline's form")
                else:
                    out["fp32_modes"][mode] = rec
        if world == 1:
            out["index_kernels"] = index_kernel_line(H["flat_pts"], H["layers"], dev)
        if world == 1 and not args.no_cpu_baseline and args.model == "cls" and not args.forward_only:
            cpu = (H["layers"], {k: v.detach().cpu() for k, v in H["model"].state_dict().items()}, H["points"], H["head"])
    default_run = (args.model == "cls" and not args.dtype and not args.forward_only and not args.backbone_only
                   and not args.batch and not args.points)
    H.clear()
    if world == 1 and default_run and not args.no_extra_configs:
        # The other single-GPU BASELINE configs, in the SAME run so that they carry the driver's clock: configs[1] forward
        # only (north_star states its >= 10x CPU target on the forward pass), configs[2] (rotation estimation, bf16) and
        # configs[3] (3DMatch descriptor, bf16): 10 graph replays each, own roofline objects.
        import copy
        import gc
        extras, detail["configs"] = {}, {}
        for name, model, fwd in (("cls_fwd", "cls", True), ("reg_bf16", "reg", False), ("inv_bf16", "inv", False)):
            gc.collect()
            torch.cuda.empty_cache()
            c = copy.copy(args)
            c.model, c.forward_only, c.dtype, c.batch, c.points = model, fwd, "", 0, 0
            c.steps, c.warmup = min(args.steps, 10), min(args.warmup, 2)
            try:
                o, h = measure(c, rank, local_rank, world, dev, first=False)
                detail["configs"][name] = h["detail"]
                h.clear()
                extras[name] = {k: o[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "dtype", "config", "roofline")}
            except Exception as e:                          # a report, never a requirement
                extras[name] = {"error": f"{type(e).__name__}: {e}"}
        if not args.dp_path:
            # The program ONE RANK of `--gpus N` runs (BASELINE configs[4]), measured on the single GPU a bench box has: flat
            # gradient buffer + all-reduce on a 1-rank RCCL communicator inside the timed region.  LAST of the embedded
            # configs: the communicator's watchdog thread must not exist while another workload captures its graph.
            gc.collect()
            torch.cuda.empty_cache()
            c = copy.copy(args)
            c.dp_path, c.warmup = True, min(args.warmup, 2)
            try:
                o, h = measure(c, rank, local_rank, world, dev, first=False)
                grad_mb = sum(p.numel() for p in h["model"].parameters() if p.requires_grad) * 4 / 1e6
                detail["configs"]["cls_dp_rank"] = h["detail"]
                h.clear()
                e = {k: o[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "dtype", "config", "roofline", "dp") if k in o}
                wire_ms = grad_mb * 1e6 * 2 * 7 / 8 / (XGMI_ALLREDUCE_BUSBW_GBS * 1e9) * 1e3 + XGMI_ALLREDUCE_LATENCY_MS
                e.update(overhead_ms=round(o["ms_per_step"] - out["ms_per_step"], 3),
                         vs_headline=round(o["value"] / out["value"], 4), collect=o["config"].get("dp_path"),
                         predicted_eff_8gpu=round(out["ms_per_step"] / (o["ms_per_step"] + wire_ms), 4),
                         # measured in that run: the 1-rank all-reduce + average as the compute stream sees it, and the same
                         # step without them (what `dp` reports per rank when world > 1)
                         allreduce_ms=o.get("dp", {}).get("allreduce_ms"), no_comm_ms=o.get("dp", {}).get("no_comm_ms_per_step_local"),
                         wire=f"ASSUMED {XGMI_ALLREDUCE_BUSBW_GBS:.0f} GB/s busbw",
                         assumes=f"t1/(t_rank+wire); wire {wire_ms:.2f} ms = {grad_mb:.1f} MB ring all-reduce, 8 GPUs, "
                                 f"{XGMI_ALLREDUCE_BUSBW_GBS:.0f} GB/s busbw + {XGMI_ALLREDUCE_LATENCY_MS} ms (ASSUMED, unmeasured), "
                                 f"no overlap")
                extras["cls_dp_rank"] = e
            except Exception as e:                          # a report, never a requirement
                extras["cls_dp_rank"] = {"error": f"{type(e).__name__}: {e}"}
            if torch.distributed.is_initialized():
                torch.distributed.destroy_process_group()
        out["configs"] = extras
    if rank == 0:
        if cpu is not None:
            out["cpu_baseline"] = cpu_baseline(cpu[0], cpu[1], cpu[2], args.cpu_clouds, args.cpu_threads, cpu[3],
                                               samples=args.cpu_samples)
            if "configs" in out and "cls_fwd" in out["configs"] and "value" in out["configs"]["cls_fwd"]:
                out["configs"]["cls_fwd"]["vs_cpu_forward"] = round(
                    out["configs"]["cls_fwd"]["value"] / out["cpu_baseline"]["forward_only_value"], 1)
        emit(out, detail)
    stdout_to_stderr()                                  # every rank: nothing follows the JSON line on stdout
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
