#!/bin/bash
# Per-replay kernel time of bench.py's HIP graph: two rocprofv3 kernel traces that differ only in --steps; the
# difference of the per-kernel totals divided by the extra steps is what ONE replayed step spends in each kernel
# (warm-up, capture and the 3 eager roofline passes of either run cancel).  usage (on the GPU box): tools/replay_profile.sh <tag> [bench args]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-replay}; shift || true
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for K in 3 13; do
  rocprofv3 --kernel-trace --stats -d $OUT/s$K -o s -- python $R/bench.py --steps $K --warmup 2 --no-cpu-baseline --no-native-line --no-extra-configs "$@" > $OUT/s$K.log 2>&1
  python $R/tools/rocprof_summary.py $(find $OUT/s$K -name "*.db" | head -1) $OUT/k$K.csv > /dev/null
  rm -rf $OUT/s$K
done
python - <<PY
import csv
def load(p):
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationUs"])) for r in csv.DictReader(open(p))}
a, b = load("$OUT/k3.csv"), load("$OUT/k13.csv")
rows = []
for n, (c, t) in b.items():
    c0, t0 = a.get(n, (0, 0.0))
    if c > c0:
        rows.append((n, (c - c0) / 10.0, (t - t0) / 10.0))
rows.sort(key=lambda r: -r[2])
tot = sum(r[2] for r in rows)
with open("$OUT/per_replay.csv", "w") as f:
    f.write("Name,CallsPerStep,UsPerStep,Percent\n")
    for n, c, t in rows:
        f.write('"%s",%.1f,%.1f,%.2f\n' % (n, c, t, 100 * t / tot))
import os
for f in ("k3.csv", "k13.csv"):
    os.remove(os.path.join("$OUT", f))
print("kernel time per replayed step: %.2f ms over %d kernels, %.0f launches" % (tot / 1e3, len(rows), sum(r[1] for r in rows)))
PY
grep -h '"metric"' $OUT/s13.log | tail -1 | cut -c1-220
