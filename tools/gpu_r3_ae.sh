#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "conv3x3" 2>&1 | tail -2
for v in "" "ASR_IGEMM_XCD=0"; do echo "== ${v:-default (consecutive tiles on one XCD)}"; env $v timeout 600 python tools/microbench.py conv 2>&1 | grep "igemm" | grep "(32, 80" | cut -c1-160; done
b() { timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))"; }
echo "warm-up run (ignore): $(b)"
for rep in 1 2; do
echo "default: $(b)"
echo "ASR_IGEMM_XCD=0: $(ASR_IGEMM_XCD=0 b)"
done
bash tools/gpu_pmc_traffic.sh r03c > gpurun_out/r3ae_traffic.log 2>&1
tail -4 gpurun_out/r3ae_traffic.log
