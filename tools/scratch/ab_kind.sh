#!/bin/bash
# usage: tools/scratch/ab_kind.sh "<bench args>" "<kind substring>" tag1 tag2 ...  -> value, ms/step and the per-call ms of that kind
ARGS="$1"; KIND="$2"; shift; shift
for t in "$@"; do
  L=$PWD/epn_pointcloud_amd/libepn_so3conv${t:+_$t}.so
  EPN_LIB=$L EPN_BENCH_DETAIL=/tmp/d_$t.json python bench.py $ARGS --steps 10 --warmup 3 --no-cpu-baseline --no-native-line --no-extra-configs 2>/dev/null > /tmp/o_$t.json
  python - "$t" "$KIND" <<'PY'
import json,sys
t,kind=sys.argv[1],sys.argv[2]
o=json.load(open(f"/tmp/o_{t}.json")); pc=json.load(open(f"/tmp/d_{t}.json"))["detail"]["headline"]["per_call"]
sel=[(r["kind"], round(r["avg_ms"],3), r["launches_per_step"]) for r in pc if kind in r["kind"]]
print(f"{t or 'default':8} {o['value']:8.1f} {o['ms_per_step']:7.2f} | {sel}")
PY
done
