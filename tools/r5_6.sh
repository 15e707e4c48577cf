#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/r5f; mkdir -p $O; cd $R
EPN_GEMM_FP32=f16x2 timeout 900 python bench.py --steps 10 --warmup 3 --no-extra-configs --no-cpu-baseline > $O/bench_f16x2.json 2> $O/bench_f16x2.err
tail -c 600 $O/bench_f16x2.json; echo
EPN_GEMM_FP32=f16x2 timeout 1800 python -m pytest tests/test_gpu_models.py tests/test_gpu_fullsize.py tests/test_gpu_conv.py tests/test_gpu_functional.py -q -m gpu 2>&1 | tail -25 | tee $O/tests_f16x2.txt
(cd tools && timeout 600 python x3_err.py 2>&1 | grep -v amdgpu.ids | tee $O/x3_err.txt)
