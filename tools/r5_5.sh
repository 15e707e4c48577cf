#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/r5e; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_bf16.py -x -q -m gpu -k "gemm" 2>&1 | tail -25 | tee $O/tests.txt
(cd tools && timeout 600 python x3_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/x3_probe.txt)
