#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/r5d; mkdir -p $O; cd $R
(cd tools && timeout 900 python pp2_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/pp2_probe.txt)
