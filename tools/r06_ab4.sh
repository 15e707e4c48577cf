#!/bin/bash
# round 6, one box: the transposes of the grouping inside the training step -- "auto" (cloud-resident fixed-point form where it is
# faster) against "split" (LDS-pre-reduced atomic scatter everywhere) and "cloud" (wherever the kernel can), three networks
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
mkdir -p gpurun_out
B="python bench.py --steps 20 --warmup 3 --no-extra-configs --no-cpu-baseline --no-native-line"
run() { # name, mode, args...
  n=$1; v=$2; shift 2
  val=$(EPN_INTER_BWD_DATA=$v EPN_BENCH_DETAIL=gpurun_out/ab4_${n}_${v}.json $B "$@" 2>gpurun_out/ab4_err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['hbm_peak_gb'], d.get('f16x2_overflow'), d.get('fixed_point_range'))" 2>&1 | tail -1)
  echo "$n $v: $val" | tee -a gpurun_out/r06_ab_ungroup_cloud.txt
}
for i in 1 2; do
  for v in auto split cloud; do run cls $v; done
  for v in auto split; do run reg_bf16 $v --model reg --dtype bf16; done
  for v in auto split; do run inv_bf16 $v --model inv --dtype bf16; done
done
for v in auto split cloud; do run reg_f32 $v --model reg --dtype f32; done
