#!/bin/bash
# round 5, first GPU call: the default bench line (with configs.cls_dp_rank) + the rank program in both gather forms
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/r5a; mkdir -p $O; cd $R
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
B="python bench.py --steps 20 --warmup 5 --no-extra-configs --no-cpu-baseline --no-native-line"
$B > $O/plain.json 2>/dev/null
$B --dp-path --dp-collect accumulate > $O/dp_acc.json 2> $O/dp_acc.err
$B --dp-path --dp-collect pack > $O/dp_pack.json 2> $O/dp_pack.err
$B > $O/plain2.json 2>/dev/null
tail -c 1500 $O/bench.err | tail -3
for f in bench plain dp_acc dp_pack plain2; do python - <<PY
import json
try:
    d = json.loads(open("$O/$f.json").read().strip().splitlines()[-1])
    print("$f", d["value"], d["ms_per_step"], json.dumps(d.get("configs", {}).get("cls_dp_rank")), json.dumps(d["roofline"].get("step")))
except Exception as e:
    print("$f", "ERR", e)
PY
done
