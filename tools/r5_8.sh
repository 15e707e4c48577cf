#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/r5i; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -q -m gpu --deselect tests/test_gpu_models.py::test_bench_line_kernel_names_are_profiler_names --durations=8 2>&1 | tail -30 | tee $O/tests.txt
