#!/bin/bash
# round 3, call R: implicit-GEMM convolution with the nine taps unrolled (operand reads = lane register + immediate)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "conv" 2>&1 | tail -5
timeout 600 python tools/microbench.py conv 2>&1 | tail -12 | cut -c1-160
ASR_IGEMM_UNROLL=0 timeout 600 python tools/microbench.py conv 2>&1 | tail -12 | cut -c1-160
b() { timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))"; }
echo "warm-up run (ignore): $(b)"
for rep in 1 2; do
echo "default (taps unrolled): $(b)"
echo "IGEMM_UNROLL=0: $(ASR_IGEMM_UNROLL=0 b)"
done
