#!/bin/bash
export TMPDIR=/tmp
b() { timeout 600 python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))"; }
echo "warm-up (ignore): $(b)"
for rep in 1 2; do
echo "headline default (eight-wave NT from 48 blocks of 128 x 128): $(b)"
echo "headline ASR_GEMM_BIG_MIN=128: $(ASR_GEMM_BIG_MIN=128 b)"
echo "headline ASR_GEMM_BIG_MIN=128 ASR_NT_RING=1024: $(ASR_GEMM_BIG_MIN=128 ASR_NT_RING=1024 b)"
echo "headline ASR_GEMM_BIG_MIN=320 ASR_NT_RING=1600: $(ASR_GEMM_BIG_MIN=320 ASR_NT_RING=1600 b)"
done
echo "librispeech default: $(b --workload librispeech)"
echo "librispeech ASR_GEMM_BIG_MIN=128: $(ASR_GEMM_BIG_MIN=128 b --workload librispeech)"
