#!/bin/bash
# usage (GPU box): tools/step_timeline.sh <tag> [bench args]   -> gpurun_out/<tag>/timeline.csv + summary.txt
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-tl}; shift || true
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/raw -o s -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-native-line --no-extra-configs "$@" > $OUT/bench.log 2>&1
python $R/tools/step_timeline.py $(find $OUT/raw -name "*.db" | head -1) $OUT/timeline.csv | tee $OUT/summary.txt
rm -rf $OUT/raw
