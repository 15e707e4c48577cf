#!/bin/bash
# round 5, second GPU call: stdout hygiene of the default bench run + the pre-split-planes NT probe
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/r5b; mkdir -p $O; cd $R
python bench.py --steps 10 --warmup 2 --cpu-clouds 1 --cpu-samples 1 > $O/bench.json 2> $O/bench.err
echo "stdout lines: $(wc -l < $O/bench.json)"; tail -c 400 $O/bench.json; echo
(cd tools && python pp_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/pp_probe.txt)
