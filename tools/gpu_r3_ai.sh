#!/bin/bash
export TMPDIR=/tmp
b() { timeout 600 python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))"; }
echo "warm-up (ignore): $(b)"
for rep in 1 2; do
echo "headline default (32): $(b)"
echo "headline 32 rev: $(ASR_WGRAD_REV=1 b)"
echo "headline 16 rev: $(ASR_WGRAD_REV=1 ASR_WGRAD_GROUP=16 b)"
echo "headline 300MB: $(ASR_WGRAD_MB=300 b)"
echo "headline 300MB rev: $(ASR_WGRAD_MB=300 ASR_WGRAD_REV=1 b)"
echo "headline 200MB rev: $(ASR_WGRAD_MB=200 ASR_WGRAD_REV=1 b)"
done
for rep in 1 2; do
echo "librispeech 16: $(ASR_WGRAD_GROUP=16 b --workload librispeech)"
echo "librispeech 16 rev: $(ASR_WGRAD_GROUP=16 ASR_WGRAD_REV=1 b --workload librispeech)"
echo "librispeech 12 rev: $(ASR_WGRAD_GROUP=12 ASR_WGRAD_REV=1 b --workload librispeech)"
echo "librispeech 32 rev: $(ASR_WGRAD_REV=1 b --workload librispeech)"
echo "librispeech 800MB: $(ASR_WGRAD_MB=800 b --workload librispeech)"
echo "librispeech 600MB rev: $(ASR_WGRAD_MB=600 ASR_WGRAD_REV=1 b --workload librispeech)"
done
