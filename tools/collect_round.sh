#!/bin/bash
# Everything the round's profiles/ directory is built from, in ONE gpurun call (same box for all numbers):
#   rocprofv3 kernel stats + PMC passes of the cls bench (collect_profiles.sh), the same for the bf16 rotation network,
#   per-replay kernel times (replay_profile.sh), and the bench lines of every configuration.
# usage (GPU box): tools/collect_round.sh <tag>      -> gpurun_out/<tag>/...
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-r06}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
bash tools/collect_profiles.sh $TAG > $OUT/collect.log 2>&1
EPN_BENCH_ARGS="--model reg --dtype bf16" bash tools/collect_profiles.sh ${TAG}_reg > $OUT/collect_reg.log 2>&1
for f in kernel_stats.csv pmc_per_kernel.json bench_under_rocprof.json; do cp gpurun_out/${TAG}_reg/$f $OUT/reg_bf16_$f; done
EPN_BENCH_ARGS="--model inv --dtype bf16" bash tools/collect_profiles.sh ${TAG}_inv > $OUT/collect_inv.log 2>&1
for f in kernel_stats.csv pmc_per_kernel.json; do cp gpurun_out/${TAG}_inv/$f $OUT/inv_bf16_$f; done; rm -rf gpurun_out/${TAG}_inv
# the bf16 networks with the atomic scatter everywhere (EPN_INTER_BWD_DATA=split: the round-5 path), for the before / after of the LDS and L2-atomic counters
EPN_INTER_BWD_DATA=split EPN_BENCH_ARGS="--model reg --dtype bf16" bash tools/collect_profiles.sh ${TAG}_regsplit > $OUT/collect_regsplit.log 2>&1
for f in kernel_stats.csv pmc_per_kernel.json; do cp gpurun_out/${TAG}_regsplit/$f $OUT/reg_bf16_atomic_scatter_$f; done; rm -rf gpurun_out/${TAG}_regsplit
# the same step with the data gradient of every InterSO3Conv on chip (EPN_INTER_BWD_DATA=onchip: dG never written): kernel stats + PMC
EPN_INTER_BWD_DATA=onchip bash tools/collect_profiles.sh ${TAG}_onchip > $OUT/collect_onchip.log 2>&1
for f in kernel_stats.csv pmc_per_kernel.json bench_under_rocprof.json; do cp gpurun_out/${TAG}_onchip/$f $OUT/cls_bwd_onchip_$f; done; rm -rf gpurun_out/${TAG}_onchip
bash tools/replay_profile.sh ${TAG}_replay > $OUT/replay.log 2>&1
cp gpurun_out/${TAG}_replay/per_replay.csv $OUT/per_replay_cls.csv
# the driver's own command form first: ONE stdout line (< 3 KB) + the complete record in the detail file
EPN_BENCH_DETAIL=$OUT/bench_cls_detail.json python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_cls.json 2> $OUT/bench_cls.err
b() { n=$1; shift; EPN_BENCH_DETAIL=$OUT/bench_${n}_detail.json python bench.py "$@" > $OUT/bench_$n.json 2>/dev/null; }
EPN_INTER_BWD_DATA=onchip b cls_bwd_onchip --no-cpu-baseline --no-native-line --no-extra-configs --steps 20
b reg --model reg --dtype bf16 --no-cpu-baseline --steps 20
b inv --model inv --dtype bf16 --no-cpu-baseline --steps 20
b cls_fwd --forward-only --no-cpu-baseline --steps 20
b reg_f32 --model reg --dtype f32 --no-cpu-baseline
b cls_bf16 --dtype bf16 --no-cpu-baseline
EPN_GEMM_FP32=native b cls_native_fp32_mfma --no-cpu-baseline --no-native-line
(cd tools && python x3_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/x3_probe.txt; python x3_err.py 2>&1 | grep -v amdgpu.ids > $OUT/x3_err.txt)
python tools/gemm_bench.py > $OUT/gemm_bench_f32.txt 2>&1
python tools/gemm_bench.py --dtype bf16 > $OUT/gemm_bench_bf16.txt 2>&1
python tools/hbm_probe.py > $OUT/hbm_probe.txt 2>&1
python tools/tn_probe.py --dtype bf16 2>&1 | grep -v amdgpu.ids > $OUT/tn_probe.txt; python tools/tn_probe.py --dtype f32 2>&1 | grep -v amdgpu.ids >> $OUT/tn_probe.txt
(cd tools && python nt_shortk_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/nt_shortk_probe.txt; python nt_shortk_probe.py --bf16 2>&1 | grep -v amdgpu.ids >> $OUT/nt_shortk_probe.txt)
(cd tools && python c1_probe.py 2>&1 | grep -v "amdgpu.ids\|Warning\|run_backward" > $OUT/c1_probe.txt; EPN_AB=1 EPN_C1_MFMA=0 python c1_probe.py 2>&1 | grep -v "amdgpu.ids\|Warning\|run_backward" >> $OUT/c1_probe.txt)
# round 6: hardware probes + per-layer A/B of the on-chip data gradient + the transposes priced against the L2 atomic roof
hipcc --offload-arch=gfx950 -O3 tools/lds_tr_probe.hip -o $OUT/lds_tr_probe 2>/dev/null && $OUT/lds_tr_probe > $OUT/lds_tr_probe.txt 2>&1
hipcc --offload-arch=gfx950 -O3 tools/atomic_rate_probe.hip -o $OUT/atomic_rate_probe 2>/dev/null && $OUT/atomic_rate_probe > $OUT/atomic_rate_probe.txt 2>&1
hipcc --offload-arch=gfx950 -O3 -w tools/lds_atomic_probe.hip -o $OUT/lds_atomic_probe 2>/dev/null && $OUT/lds_atomic_probe > $OUT/lds_atomic_probe.txt 2>&1
hipcc --offload-arch=gfx950 -O3 -w tools/valu_rate_probe.hip -o $OUT/valu_rate_probe 2>/dev/null && $OUT/valu_rate_probe > $OUT/valu_rate_probe.txt 2>&1
rm -f $OUT/lds_tr_probe $OUT/atomic_rate_probe $OUT/lds_atomic_probe $OUT/valu_rate_probe
# the two transposes of the grouping per layer (atomic scatter | cloud-resident fixed point), and inside the step
for m in "reg bf16" "inv bf16" "cls f32" "reg f32"; do python tools/ungroup_cloud_probe.py $m 2>&1 | grep -v amdgpu.ids >> $OUT/ungroup_cloud_probe.txt; done
rm -f gpurun_out/r06_ab_ungroup_cloud.txt; bash tools/r06_ab4.sh > /dev/null 2>&1; cp gpurun_out/r06_ab_ungroup_cloud.txt $OUT/ab_ungroup_cloud.txt
python tools/bwd_onchip_probe.py cls 2>&1 | grep -v amdgpu.ids > $OUT/bwd_onchip_probe.txt
for m in "reg bf16" "inv bf16" "cls f32"; do python tools/ungroup_atomic_pricing.py $m 2>&1 | grep -v amdgpu.ids >> $OUT/ungroup_atomic_pricing.txt; done
EPN_BENCH_ARGS="--model reg --dtype bf16" bash tools/replay_profile.sh ${TAG}_replay_reg --model reg --dtype bf16 > $OUT/replay_reg.log 2>&1
cp gpurun_out/${TAG}_replay_reg/per_replay.csv $OUT/per_replay_reg_bf16.csv; rm -rf gpurun_out/${TAG}_replay_reg
rm -rf gpurun_out/${TAG}_reg gpurun_out/${TAG}_replay $OUT/pmc_*.log $OUT/stats.log
for f in $(ls $OUT/bench_*.json | grep -v _detail); do python - <<PY
import json
d = json.loads(open("$f").read().strip().splitlines()[-1])
r = d.get("roofline", {})
print("$(basename $f)", d["value"], d["ms_per_step"], d["dtype"], "|", r.get("kernel", "")[:60], r.get("bound"), r.get("achieved"), r.get("frac"))
PY
done
