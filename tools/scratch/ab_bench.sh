#!/bin/bash
# usage: tools/scratch/ab_bench.sh "<bench args>" tag1 tag2 ...   (libs epn_pointcloud_amd/libepn_so3conv_<tag>.so; "" = default lib)
ARGS="$1"; shift
for t in "$@"; do
  L=$PWD/epn_pointcloud_amd/libepn_so3conv${t:+_$t}.so
  EPN_LIB=$L EPN_BENCH_DETAIL=/tmp/d_$t.json python bench.py $ARGS --steps 10 --warmup 3 --no-cpu-baseline --no-native-line --no-extra-configs 2>/dev/null > /tmp/o_$t.json
  python - "$t" <<'PY'
import json,sys
t=sys.argv[1]
o=json.load(open(f"/tmp/o_{t}.json")); pc=json.load(open(f"/tmp/d_{t}.json"))["detail"]["headline"]["per_call"]
def tot(kinds): return round(sum(r["avg_ms"]*r["launches_per_step"] for r in pc if r["kind"] in kinds),2)
print(f"{t or 'default':8} {o['value']:8.1f} {o['ms_per_step']:7.2f} | group {tot(('inter_group',))} ungroup {tot(('inter_ungroup',))} gemm {tot(('inter_gemm',))} dg {tot(('inter_gemm_dg',))} dw {tot(('inter_gemm_dw',))} intra {tot(('intra_gemm','intra_gemm_dw'))} basis {tot(('so3_basis',))} 1x1 {tot(('conv1x1_gemm','conv1x1_gemm_dw'))}")
PY
done
