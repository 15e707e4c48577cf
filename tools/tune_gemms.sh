#!/bin/bash
# Re-record epn_pointcloud_amd/gemm_tuning_gfx950.csv on an MI355X (about 10 minutes: every GEMM shape of the cls
# B=32 step is timed against all rocBLAS / hipBLASLt solutions during bench.py's warm-up):
#   gpurun --timeout 1500 -- 'bash tools/tune_gemms.sh'     -> gpurun_out/tunableop0.csv, copy it over the in-tree file
# CAUTION: every NEW GEMM shape is timed against every library solution; shapes with a ~1e6-long contraction take
# minutes each (a run that added the 1x1 skip-convolution weight gradients exceeded 30 minutes) -- always bound the call.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/gpurun_out
# seed with the in-tree table: only GEMM shapes that are not in it yet get tuned (delete the seed for a full re-tune)
[ -n "$EPN_TUNE_FROM_SCRATCH" ] || cp $R/epn_pointcloud_amd/gemm_tuning_gfx950.csv $R/gpurun_out/tunableop0.csv
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=$R/gpurun_out/tunableop.csv
export PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=100 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=10
python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline "$@"
ls -la $R/gpurun_out/tunableop*
