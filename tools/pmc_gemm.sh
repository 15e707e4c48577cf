#!/bin/bash
# PMC passes over one GEMM problem (own kernel next to the BLAS library's): MFMA busy vs GPU-active cycles, stalls, LDS.
# usage (on the GPU box): tools/pmc_gemm.sh nt:245760,256,3072 out_dir
set -e
ONE=$1; OUT=${2:-gpurun_out/pmc_gemm}
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
cd /tmp
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" \
         "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C -d $R/$OUT/p$i --output-format csv -- python $R/tools/gemm_bench.py --one $ONE > $R/$OUT/p$i.log 2>&1 || true
done
cd $R
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/p*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k, v in agg.items():
            if "gemm" in k or "Cijk" in k:
                print(k, {c: round(x / n[(k, c)]) for c, x in v.items()})
PY
