#!/bin/bash
# Repeated bench.py runs with the loss trace (bench.py --trace-loss): one line of losses per run; how round 5 found the
# under-reported producer-side maxima (one non-finite data-parallel step in ten).  usage: tools/nan_hunt.sh reps bench-args...
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
REPS=$1; shift
for r in $(seq $REPS); do
  python bench.py --trace-loss "$@" --no-cpu-baseline --no-native-line --no-extra-configs 2>&1 >/dev/null | grep "bench\] \(warm-up\|untimed-step\)\? \?loss" | awk '{print $NF}' | tr "\n" " "
  echo
done
