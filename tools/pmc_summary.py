#!/usr/bin/env python3
"""rocprofv3 --pmc CSV directories -> per-kernel JSON (per-launch averages).
usage: tools/pmc_summary.py <dir containing pmc_*/p_counter_collection.csv> <out.json>

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md section "HBM": FETCH_SIZE / WRITE_SIZE are in KB (x1024),
collected in separate passes; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, i.e. reads HALF the bytes of a wide
coalesced stream -> the read side is doubled (an upper bound for this path's 64-B gather segments, which the guide
lists as uncalibrated)."""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(n):
    m = re.match(r"(Cijk_[A-Za-z]+_[A-Za-z]+)_.*?_(MT\d+x\d+x\d+)_", n)   # Tensile (rocBLAS / hipBLASLt) GEMM kernels
    if m:
        return f"{m.group(1)}_{m.group(2)}"
    return n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]


root, out = sys.argv[1], sys.argv[2]
res = {}
# how many steps the profiled process ran: bench.py prints "[bench] compute_calls N" (every eager / captured / profiled step)
steps = None
for log in sorted(glob.glob(os.path.join(root, "pmc_*.log"))) + sorted(glob.glob(os.path.join(root, "stats.log"))):
    m = re.findall(r"\[bench\] compute_calls (\d+)", open(log, errors="replace").read())
    if m:
        steps = int(m[-1])
        break
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    f = os.path.join(d, "p_counter_collection.csv")
    if not os.path.exists(f):
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if not k.startswith(("epn::", "Cijk_", "_ZN3epn", "fps_", "ball_query", "gather_")):
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    for k in agg:
        e = res.setdefault(k, {})
        e["launches"] = len(disp[k])
        for c, v in agg[k].items():
            e[c] = v / len(disp[k])
for k, e in res.items():
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        e["hbm_bytes_per_launch"] = (2.0 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024.0
    if "SQ_WAVE_CYCLES" in e:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if c in e:
                e[c + "_frac"] = e[c] / e["SQ_WAVE_CYCLES"]
if steps:
    res["_meta"] = {"step_equivalents": steps, "note": "steps the profiled bench process ran: executed compute() calls (warm-up, eager steps) + graph replays; the capture runs no kernel"}
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
print(f"{len(res)} kernels -> {out}")
