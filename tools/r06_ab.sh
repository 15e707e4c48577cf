#!/bin/bash
# round 6, one box: hardware probes (LDS transposed reads, L2 atomics), the tests touched this round, and the A/B of the
# swizzled bf16 weight-gradient image + the permuted K = 64 transpose tile against the round-5 images (libepn_so3conv_noswz.so)
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 tools/lds_tr_probe.hip -o gpurun_out/lds_tr_probe 2>/dev/null && gpurun_out/lds_tr_probe > gpurun_out/r06_lds_tr_probe.txt 2>&1
hipcc --offload-arch=gfx950 -O3 tools/atomic_rate_probe.hip -o gpurun_out/atomic_rate_probe 2>/dev/null && gpurun_out/atomic_rate_probe > gpurun_out/r06_atomic_rate_probe.txt 2>&1
rm -f gpurun_out/lds_tr_probe gpurun_out/atomic_rate_probe
(time python -m pytest -q -p no:cacheprovider tests/test_gpu_f16x2_contract.py "tests/test_gpu_bf16.py::test_config_full_size_bf16" \
   "tests/test_gpu_bf16.py::test_gemm_tn_vs_fp64" "tests/test_gpu_bf16.py::test_gemm_tn_grouped" tests/test_gpu_models.py::test_full_width_cls_step_matches_oracle_loss \
   tests/test_gpu_models.py::test_pointnet_max_propagates_nan tests/test_gpu_fullsize.py tests/test_gpu_functional.py -m gpu -s 2>&1) > gpurun_out/r06_gputest_b.log 2>&1
tail -25 gpurun_out/r06_gputest_b.log
B="python bench.py --steps 20 --warmup 3 --no-extra-configs --no-cpu-baseline --no-native-line"
for i in 1 2; do
  for m in reg inv; do
    for lib in "" "_noswz"; do
      v=$(EPN_BENCH_DETAIL=gpurun_out/ab_${m}${lib}_$i.json EPN_LIB=$R/epn_pointcloud_amd/libepn_so3conv$lib.so $B --model $m 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
      echo "$m lib${lib:-_default}: $v" | tee -a gpurun_out/r06_ab_swz.txt
    done
  done
done
python - <<'PY' | tee -a gpurun_out/r06_ab_swz.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/ab_*_2.json")):
    d = json.load(open(f))
    pk = d["detail"]["headline"]["per_kernel"]
    rows = [(k, v) for k, v in pk.items() if "gemm_tn_bf16" in k or ("inter_ungroup_shared" in k and "bf16" in k)]
    for k, v in rows:
        print(f, k, {kk: v.get(kk) for kk in ("ms_per_step", "launches_per_step", "avg_launch_ms", "achieved", "frac", "bound")})
PY
