#!/bin/bash
# round 3, call O: pooling with selection codes (op tests, step A/B, whole suite)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "pool" 2>&1 | tail -15 > gpurun_out/r3o_pytest.txt
tail -6 gpurun_out/r3o_pytest.txt | cut -c1-300
b() { timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))"; }
echo "warm-up run (ignore): $(b)"
for rep in 1 2; do
echo "default (selection codes): $(b)"
echo "ASR_POOL_CODES=0: $(ASR_POOL_CODES=0 b)"
done
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > gpurun_out/r3o_pytest_all.txt
tail -4 gpurun_out/r3o_pytest_all.txt | cut -c1-300
bash tools/gpu_profile.sh r3o_bench 13 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline
head -34 gpurun_out/r3o_bench_timeline.txt | cut -c1-130 | tail -30
