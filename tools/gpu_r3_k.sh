#!/bin/bash
# round 3, call K: whole-head attention kernels (tests, step A/B)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "attention" 2>&1 | tail -25 > gpurun_out/r3k_pytest.txt
tail -12 gpurun_out/r3k_pytest.txt | cut -c1-300
b() { timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; print('$1', round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))"; }
b "warm-up run (ignore):"
for rep in 1 2; do
b "default (whole-head fwd + bwd):"
ASR_ATTN_HEAD=0 ASR_ATTN_HEAD_BWD=0 b "tiled attention kernels (previous):"
ASR_ATTN_HEAD_BWD=0 b "whole-head forward only:"
done
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > gpurun_out/r3k_pytest_all.txt
tail -4 gpurun_out/r3k_pytest_all.txt | cut -c1-300
