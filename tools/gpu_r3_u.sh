#!/bin/bash
# round 3, call U: conv.5 forward (64 -> 128 channels) as two passes of the register-weight kernel
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "conv" 2>&1 | tail -3
for v in "" "ASR_C64_SPLIT=0"; do
  echo "== ${v:-default (two passes of conv_c64)}"
  env $v timeout 600 python tools/microbench.py conv 2>&1 | grep "igemm (32, 80, 400, 64, 128)" | cut -c1-160
done
b() { timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))"; }
echo "warm-up run (ignore): $(b)"
for rep in 1 2; do
echo "default: $(b)"
echo "C64_SPLIT=0: $(ASR_C64_SPLIT=0 b)"
done
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_baseline_shapes.py -m gpu -q -x 2>&1 | tail -3
