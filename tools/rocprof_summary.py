#!/usr/bin/env python3
"""rocprofv3 (ROCm 7.2 writes a rocpd sqlite db) -> the `--stats`-style per-kernel summary as CSV.
usage: tools/rocprof_summary.py <results.db> <out.csv>"""
import csv
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
    for name, calls, total, avg, pct in rows:
        w.writerow([name, calls, round(total, 3), round(avg, 3), round(pct, 3)])
print(f"{len(rows)} kernels -> {out}")
