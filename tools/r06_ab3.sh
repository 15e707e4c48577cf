#!/bin/bash
# round 6, one box: the cls step with the data gradient of the K = 16 layers on chip (default "auto") against the split pair
# everywhere (EPN_INTER_BWD_DATA=split) and the on-chip kernel everywhere (=onchip)
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
mkdir -p gpurun_out
B="python bench.py --steps 20 --warmup 3 --no-extra-configs --no-cpu-baseline --no-native-line"
for i in 1 2 3; do
  for v in auto split onchip; do
    val=$(EPN_INTER_BWD_DATA=$v EPN_BENCH_DETAIL=gpurun_out/ab3_cls_${v}_$i.json $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['hbm_peak_gb'], d.get('f16x2_overflow'))")
    echo "cls $v: $val" | tee -a gpurun_out/r06_ab_bwd_data_modes.txt
  done
done
for v in auto split; do val=$(EPN_INTER_BWD_DATA=$v $B --model reg --dtype f32 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"); echo "reg f32 $v: $val" | tee -a gpurun_out/r06_ab_bwd_data_modes.txt; done
