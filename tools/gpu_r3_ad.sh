#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "pool_tcf" 2>&1 | tail -2
b() { timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))"; }
echo "warm-up run (ignore): $(b)"
for rep in 1 2; do
echo "default (row tile fastest, one XCD): $(b)"
echo "ASR_CONV7_POOL=0: $(ASR_CONV7_POOL=0 b)"
done
bash tools/gpu_pmc_traffic.sh r03b > gpurun_out/r3ad_traffic.log 2>&1
grep -A1 "igemm_kernel<unsigned short, 128, 16, 1, 1, true>" gpurun_out/r03b_traffic_pmc.txt | cut -c1-120
