#!/bin/bash
export TMPDIR=/tmp
ASR_DEBUG_GROUP=1 timeout 600 python bench.py --workload librispeech --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | grep "gemm_tn" | sed 's/dy ([0-9, ]*)/dy/' | sort | uniq -c | sort -rn | head -20
