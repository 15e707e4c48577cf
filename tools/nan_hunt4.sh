#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
N=${1:-12}; shift
for r in $(seq $N); do
  EPN_DEBUG_FLAGS=1 EPN_BENCH_TRACE_LOSS=3 python bench.py --dp-path --steps 1 --warmup 5 --no-cpu-baseline --no-native-line --no-extra-configs "$@" 2>&1 >/dev/null | grep "bench\] flag" | sed 's/\[bench\] flag //' | tr "\n" ";" | sed 's/warm-up //g; s/ finite//g'
  echo
done
