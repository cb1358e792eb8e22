#!/bin/bash
# round 3, call H: whole GPU suite; grouped cross-attention K|V A/B; weight gradients on the second stream A/B; launch sequence
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q -s 2>&1 | tail -40 > gpurun_out/r3h_pytest.txt
tail -12 gpurun_out/r3h_pytest.txt | cut -c1-300
b() { timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; print('$1', round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))"; }
b "default (grouped cross KV):"
ASR_GROUP_CROSS_KV=0 b "per-layer cross KV:"
ASR_WGRAD_SIDE=1 b "dW on side stream, groups of 16:"
ASR_WGRAD_SIDE=1 ASR_WGRAD_GROUP=8 b "dW on side stream, groups of 8:"
ASR_WGRAD_SIDE=1 ASR_WGRAD_GROUP=4 b "dW on side stream, groups of 4:"
ASR_WGRAD_GROUP=8 b "main stream, groups of 8:"
bash tools/gpu_profile.sh r3h_bench 13 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline
db=$(find /tmp/prof_r3h_bench -name "*.db" | head -1)
python tools/prof_sequence.py "$db" gpurun_out/r3h_sequence.txt
python tools/prof_families.py "$db" gpurun_out/r3h_replayed_families.json "python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline" > /dev/null 2>&1
head -30 gpurun_out/r3h_bench_timeline.txt | cut -c1-140
