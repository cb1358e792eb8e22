#!/bin/bash
export TMPDIR=/tmp
b() { timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))"; }
echo "warm-up run (ignore): $(b)"
for rep in 1 2; do
echo "default: $(b)"
echo "last grouped dW launch on the second stream (under the conv backward): $(ASR_WGRAD_SIDE=2 b)"
done
