#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
N=${1:-12}
count() { grep -c nan; }
echo "V1 dp graph:        $(bash tools/nan_hunt2.sh $N --dp-path --steps 2 --warmup 5 | count) of $N runs with nan"
echo "V2 + sync after capture: $(EPN_TMP_SYNC_AFTER_CAPTURE=1 bash tools/nan_hunt2.sh $N --dp-path --steps 2 --warmup 5 | count) of $N"
echo "V3 accumulate form: $(bash tools/nan_hunt2.sh $N --dp-path --dp-collect accumulate --steps 2 --warmup 5 | count) of $N"
echo "V4 plain graph:     $(bash tools/nan_hunt2.sh $N --steps 2 --warmup 5 | count) of $N"
