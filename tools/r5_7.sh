#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/r5g; mkdir -p $O; cd $R
timeout 900 python bench.py --steps 10 --warmup 3 --no-extra-configs --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("native_fp32_mfma"), d.get("fp32_modes"))
txt=open("$O/bench.err").read()
i=txt.rindex("[bench] detail: ")
dd=json.loads(txt[i+16:].splitlines()[0])
pk=dd["detail"]["headline"]["per_kernel"]
for k,v in sorted(pk.items(), key=lambda kv:-kv[1]["ms_per_step"])[:16]:
    print(f"{k[:70]:70s} {v['ms_per_step']:7.3f} n={v['launches_per_step']}")
PY
timeout 2400 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_models.py::test_bench_line_kernel_names_are_profiler_names --durations=15 2>&1 | tail -40 | tee $O/tests.txt
