#!/bin/bash
# round 3, call J: host-scheduled grouped dW; eight-wave NT selection; full suite
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > gpurun_out/r3j_pytest.txt
tail -6 gpurun_out/r3j_pytest.txt | cut -c1-300
b() { timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; print('$1', round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))"; }
b "warm-up run (ignore):"
for rep in 1 2; do
b "default:"
ASR_TN_GROUP_TILE=256 b "dW: one block per slice (previous):"
ASR_GEMM_BIG=0 b "four-wave NT only:"
ASR_GROUP_CROSS_KV=0 b "per-layer cross KV:"
done
ASR_TN_GROUP_STAGES=4 b "scheduled dW, 4 stages:"
ASR_TN_GROUP_WGS=512 b "scheduled dW, 512 pieces:"
bash tools/gpu_profile.sh r3j_bench 13 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline
db=$(find /tmp/prof_r3j_bench -name "*.db" | head -1)
python tools/prof_sequence.py "$db" gpurun_out/r3j_sequence.txt
grep -n "tn256\|gemm_big" gpurun_out/r3j_sequence.txt | cut -c1-140 | head -60
