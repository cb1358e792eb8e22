#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "conv1" 2>&1 | tail -3
for v in "" "ASR_CONV1_FWD_PX=0"; do echo "== ${v:-default (one pixel per thread, 1 KB stores)}"; env $v timeout 300 python tools/microbench.py misc 2>&1 | grep "conv1_fwd"; done
b() { timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))"; }
echo "warm-up run (ignore): $(b)"
for rep in 1 2; do
echo "default: $(b)"
echo "ASR_CONV1_FWD_PX=0: $(ASR_CONV1_FWD_PX=0 b)"
done
