#!/bin/bash
# repeated default bench lines under a toggle: which embedded config fails?  usage: tools/nan_hunt.sh VAR "v1 v2 ..." [reps]
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
VAR=$1; VALS=$2; REPS=${3:-3}
for r in $(seq $REPS); do
  for v in $VALS; do
    env $VAR=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-native-line 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$VAR=$v', d['value'], {k:c.get('value',c.get('error')) for k,c in d['configs'].items()})"
  done
done
