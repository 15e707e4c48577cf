#!/bin/bash
# A/B of two library builds inside one box: tools/ab_builds.sh <tagged lib suffix> [bench args]
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
T=$1; shift
B="python bench.py --steps 20 --warmup 5 --no-extra-configs --no-cpu-baseline --no-native-line $@"
for i in 1 2; do
  for lib in "" "_$T"; do
    v=$(EPN_LIB=$R/epn_pointcloud_amd/libepn_so3conv$lib.so $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "lib${lib:-_default}: $v"
  done
done
