#!/bin/bash
export TMPDIR=/tmp
b() { timeout 600 python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))"; }
echo "warm-up (ignore): $(b --workload librispeech)"
for rep in 1 2; do
echo "librispeech default (equal pieces): $(b --workload librispeech)"
echo "librispeech per-slice 3200 rows: $(ASR_TN_GROUP_TILE=256 b --workload librispeech)"
echo "librispeech per-slice 2128 rows: $(ASR_TN_GROUP_TILE=256 ASR_TN_GROUP_MROWS=2128 b --workload librispeech)"
echo "librispeech per-slice 4256 rows: $(ASR_TN_GROUP_TILE=256 ASR_TN_GROUP_MROWS=4256 b --workload librispeech)"
echo "librispeech equal pieces, 512 workgroups: $(ASR_TN_GROUP_WGS=512 b --workload librispeech)"
done
