#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel stats + PMC passes of bench.py; raw output under gpurun_out/,
# summaries are produced by tools/rocprof_summary.py / tools/pmc_summary.py and committed under profiles/.
# PMC passes are separate runs with --kernel-trace only (gpurun refuses --pmc mixed with other trace domains).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-r01}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-native-line --no-extra-configs ${EPN_BENCH_ARGS:-}"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o s -- $BENCH > $OUT/stats.log 2>&1
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32" \
         "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  N=$(echo $C | cut -d" " -f1)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -o p -- $BENCH > $OUT/pmc_$N.log 2>&1
done
grep -v "^[WE]2026" $OUT/stats.log | tail -1 > $OUT/bench_under_rocprof.json
# summarise on the box and drop the raw traces (gpurun copies back at most 64 MiB)
python $R/tools/rocprof_summary.py $(find $OUT/stats -name "*.db" | head -1) $OUT/kernel_stats.csv
python $R/tools/pmc_summary.py $OUT $OUT/pmc_per_kernel.json
rm -rf $OUT/stats $OUT/pmc_*/ $OUT/pmc_*.log
ls $OUT
