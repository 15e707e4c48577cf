#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/r5c; mkdir -p $O; cd $R
(cd tools && timeout 900 python pp_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/pp_probe.txt)
timeout 1200 python -m pytest tests/test_gpu_models.py tests/test_gpu_dist.py -x -q -m gpu -k "default_bench or rccl or rehearsal" 2>&1 | tail -15 | tee $O/tests.txt
