#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/r5h; mkdir -p $O; cd $R
B="python bench.py --steps 2 --warmup 1 --batch 4 --no-cpu-baseline --no-native-line --no-extra-configs"
for mode in f16x2 split; do
  for extra in "" "--no-graph" "--dp-path" "--dp-path --dp-collect accumulate"; do
    echo "== $mode $extra"; EPN_GEMM_FP32=$mode timeout 300 $B $extra 2>&1 | grep -E "non-finite|AssertionError|\"value\"" | cut -c1-120
  done
done
echo "== 2 ranks f16x2"; EPN_DP_SHARE_GPU=1 EPN_DP_BACKEND=gloo timeout 600 $B --gpus 2 2>&1 | grep -E "non-finite|\"value\"" | cut -c1-120
echo "== 2 ranks split"; EPN_GEMM_FP32=split EPN_DP_SHARE_GPU=1 EPN_DP_BACKEND=gloo timeout 600 $B --gpus 2 2>&1 | grep -E "non-finite|\"value\"" | cut -c1-120
