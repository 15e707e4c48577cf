#!/usr/bin/env python3
"""One replayed step of bench.py's HIP graph as a TIMELINE (rocprofv3 --kernel-trace rocpd db): every kernel of the step in
start order with its queue, start offset, duration and the idle gap before it on the same queue; helper families (maxima,
zero fills, operand splits, statistics) summed.  The step shown is the fastest window between two FPS launches (FPS opens
every step), i.e. a graph replay, not an eager pass.
usage: tools/step_timeline.py <results.db> <out.csv>"""
import csv
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
con = sqlite3.connect(db)
cur = con.cursor()
names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in names else None
if view is None:
    print("no `kernels` view; objects:", names)
    sys.exit(1)
cols = [r[1] for r in cur.execute(f"pragma table_info({view})")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
sel = f"select name, start, end, {qcol or '0'} from {view} order by start"
rows = [(n, int(s), int(e), q) for n, s, e, q in cur.execute(sel)]
fps = [i for i, r in enumerate(rows) if "fps_wave_kernel" in r[0]]
if len(fps) < 3:
    print("fewer than three steps in the trace"); sys.exit(1)
wins = [i for i in range(len(fps) - 1) if rows[fps[i + 1]][1] - rows[fps[i]][1] > 5e6 and fps[i + 1] - fps[i] > 50]   # a whole step, not an index-kernel timing loop
if not wins:
    print("no step-sized window between FPS launches"); sys.exit(1)
best = min(wins, key=lambda i: rows[fps[i + 1]][1] - rows[fps[i]][1])
lo, hi = fps[best], fps[best + 1]
t0 = rows[lo][1]
period = rows[hi][1] - t0
step = rows[lo:hi]
last_end = {}
fam = {}
KEYS = ("absmax", "fillBuffer", "split_", "stats_", "octets", "pack_cols", "rk4_table", "rk_table", "morton", "cast")
busy = 0
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["i", "queue", "start_us", "dur_us", "gap_before_us", "name"])
    for i, (n, s, e, q) in enumerate(step):
        gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = max(e, last_end.get(q, 0))
        w.writerow([i, q, round((s - t0) / 1e3, 1), round((e - s) / 1e3, 1), round(gap, 1), n[:140]])
        busy += e - s
        for k in KEYS:
            if k in n:
                a = fam.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
# union of busy intervals (any queue)
iv = sorted((s, e) for _, s, e, _ in step)
un, cs, ce = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > ce:
        un += ce - cs; cs, ce = s, e
    else:
        ce = max(ce, e)
un += ce - cs
print(f"step period {period / 1e6:.3f} ms, {len(step)} kernels, kernel time {busy / 1e6:.3f} ms, device busy (union) {un / 1e6:.3f} ms, idle {(period - un) / 1e6:.3f} ms")
for k, (c, t) in sorted(fam.items(), key=lambda x: -x[1][1]):
    print(f"  {k:12s} {c:4d} launches {t / 1e3:7.3f} ms")
