#!/bin/bash
# round 3, call Q: grouped weight gradients: group size 16 / 32 / 24, on the second stream (overlapping the conv backward)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "grouped" 2>&1 | tail -3
b() { timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))"; }
echo "warm-up run (ignore): $(b)"
for rep in 1 2; do
echo "default (groups of 16, main stream): $(b)"
echo "groups of 32: $(ASR_WGRAD_GROUP=32 b)"
echo "groups of 24: $(ASR_WGRAD_GROUP=24 b)"
echo "side stream, groups of 16: $(ASR_WGRAD_SIDE=1 b)"
echo "side stream, groups of 32: $(ASR_WGRAD_SIDE=1 ASR_WGRAD_GROUP=32 b)"
done
