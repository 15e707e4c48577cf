#!/bin/bash
# round 6: NT tile configurations on the short-K (data-gradient) shapes, bf16 -- tuning library (tools/_tuning.py)
mkdir -p gpurun_out
for cfg in 2 7 8 10; do echo "== cfg $cfg"; timeout 200 python tools/gemm_bench.py --dtype bf16 --set dg --cfg $cfg 2>&1 | grep "^NT  [0-9]*x[0-9]*x[0-9]*: own"; done > gpurun_out/r06_bf16_dg_tile_sweep.txt 2>&1
cat gpurun_out/r06_bf16_dg_tile_sweep.txt
