#!/bin/bash
# repeated bench.py runs with the loss trace: first non-finite loss per run.  usage: tools/nan_hunt2.sh reps args...
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
REPS=$1; shift
for r in $(seq $REPS); do
  EPN_BENCH_TRACE_LOSS=1 python bench.py "$@" --no-cpu-baseline --no-native-line --no-extra-configs 2>&1 >/dev/null | grep "bench\] \(warm\|before\|probe\|untimed\|loss\)" | awk '{print $NF}' | tr "\n" " "
  echo
done
