#!/bin/bash
# round 3, call AB: attention output projection + dropout + residual + LayerNorm as one launch
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "add_ln" 2>&1 | tail -4
b() { timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))"; }
echo "warm-up run (ignore): $(b)"
for rep in 1 2; do
echo "default (projection + LayerNorm in one launch): $(b)"
echo "ASR_GEMM_LN=0: $(ASR_GEMM_LN=0 b)"
done
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_graph.py tests/test_gpu_baseline_shapes.py -m gpu -q -x 2>&1 | tail -3
