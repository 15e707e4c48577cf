#!/bin/bash
# round 6, one box: raw LDS barrier in the LDS-reduced scatters (default) against __syncthreads (libepn_so3conv_synbar.so),
# the on-chip data gradient per layer, and the step with EPN_INTER_BWD_DATA=onchip
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
mkdir -p gpurun_out
(timeout 900 python -m pytest -q -p no:cacheprovider tests/test_gpu_bwd_onchip.py tests/test_gpu_fullsize.py "tests/test_gpu_conv.py::test_group_ungroup_abi_vs_oracle" "tests/test_gpu_conv.py::test_ungroup_acc_adds_to_what_is_there" "tests/test_gpu_bf16.py::test_deterministic_data_gradient" tests/test_gpu_conv.py::test_shared_input_gradient_is_folded_into_the_data_gradient -m gpu 2>&1 | tail -5) | tee gpurun_out/r06_gputest_e.log
timeout 500 python tools/bwd_onchip_probe.py cls 2>&1 | grep -v amdgpu | tee gpurun_out/r06_bwd_onchip_probe.txt
EPN_LIB=$R/epn_pointcloud_amd/libepn_so3conv_synbar.so timeout 500 python tools/bwd_onchip_probe.py cls 2>&1 | grep -v amdgpu | tee gpurun_out/r06_bwd_onchip_probe_synbar.txt
B="python bench.py --steps 20 --warmup 3 --no-extra-configs --no-cpu-baseline --no-native-line"
for i in 1 2; do
  for cfg in "cls default" "cls synbar" "cls onchip" "reg default" "reg synbar" "inv default" "inv synbar"; do
    set -- $cfg; m=$1; v=$2
    lib=""; env=""
    [ "$v" = "synbar" ] && lib="_synbar"
    [ "$v" = "onchip" ] && env="EPN_INTER_BWD_DATA=onchip"
    val=$(env $env EPN_BENCH_DETAIL=gpurun_out/ab2_${m}_${v}_$i.json EPN_LIB=$R/epn_pointcloud_amd/libepn_so3conv$lib.so $B --model $m 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('f16x2_overflow'))")
    echo "$m $v: $val" | tee -a gpurun_out/r06_ab_rawbar_onchip.txt
  done
done
