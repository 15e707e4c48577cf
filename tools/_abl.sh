python -m pytest tests/test_gpu_bf16.py -x -q -m gpu -k gemm 2>&1 | tail -2
python tools/gemm_bench.py 2>&1 | grep -E "^grouped" | cut -c1-110
python bench.py --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-250
python bench.py --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-250
