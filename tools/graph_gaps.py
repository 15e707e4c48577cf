#!/usr/bin/env python3
"""Idle time inside the replayed training step: from a rocprofv3 --kernel-trace rocpd database, the last `steps` replays
(the kernels between the last spin-free stretch) -> device busy time (union of kernel intervals), idle time, number of gaps
and their distribution.  usage: tools/graph_gaps.py <results.db> [steps in the run]"""
import sqlite3
import sys

db = sys.argv[1]
con = sqlite3.connect(db)
cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tabs if "kernel_dispatch" in t and "rocpd" in t] or [t for t in tabs if "kernel" in t]
print("tables:", kt[:6])
t = kt[0]
cols = [r[1] for r in cur.execute(f"pragma table_info({t})")]
print("columns:", cols)
sc = "start" if "start" in cols else [c for c in cols if "start" in c][0]
ec = "end" if "end" in cols else [c for c in cols if "end" in c][0]
rows = list(cur.execute(f"select {sc}, {ec} from {t} order by {sc}"))
print(len(rows), "dispatches")
# the replayed steps are the longest run of dispatches without a gap > 0.3 ms (eager passes have host gaps and spin kernels)
runs, start = [], 0
for i in range(1, len(rows)):
    if rows[i][0] - max(r[1] for r in rows[max(start, i - 8):i]) > 300000:
        runs.append((start, i)); start = i
runs.append((start, len(rows)))
a0, a1 = max(runs, key=lambda r: r[1] - r[0])
iv = sorted(rows[a0:a1])
busy, gaps, cur_end, first = 0, [], None, iv[0][0]
for a, b in iv:
    if cur_end is None:
        cur_end = b; busy += b - a; continue
    if a > cur_end:
        gaps.append(a - cur_end); busy += b - a; cur_end = b
    elif b > cur_end:
        busy += b - cur_end; cur_end = b
span = cur_end - first
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
big = [g for g in gaps if g > 20000]
small = [g for g in gaps if g <= 20000]
hist = {}
for g in small:
    k = int(g // 1000)
    hist[k] = hist.get(k, 0) + 1
print(f"replay block: {len(iv)} kernels over {span / 1e6:.2f} ms = {steps:g} steps of {span / 1e6 / steps:.2f} ms; per step: busy "
      f"{busy / 1e6 / steps:.2f} ms, idle {sum(gaps) / 1e6 / steps:.3f} ms in {len(gaps) / steps:.0f} gaps ({len(iv) / steps:.0f} kernels); "
      f"gaps > 20 us: {len(big) / steps:.1f} totalling {sum(big) / 1e6 / steps:.3f} ms")
print("gap histogram (us: count per step):", {k: round(v / steps, 1) for k, v in sorted(hist.items())})
