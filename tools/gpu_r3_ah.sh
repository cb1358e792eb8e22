#!/bin/bash
export TMPDIR=/tmp
b() { timeout 600 python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))"; }
echo "warm-up (ignore): $(b --workload librispeech)"
for rep in 1 2; do
echo "librispeech, groups of 32 (default): $(b --workload librispeech)"
echo "librispeech, groups of 16: $(ASR_WGRAD_GROUP=16 b --workload librispeech)"
echo "librispeech, groups of 8: $(ASR_WGRAD_GROUP=8 b --workload librispeech)"
done
