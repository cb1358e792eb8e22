#!/bin/bash
# round 3, GPU call F: 256 x 256 grouped weight gradients (op test, A/B), attention forward final (A/B + PMC), replayed families
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short -p no:cacheprovider -k "grouped or attention" 2>&1 | tail -15 ) > gpurun_out/r3f_pytest_ops.txt
tail -8 gpurun_out/r3f_pytest_ops.txt
for v in "256 1600" "256 3200" "256 800" "256 0" "128 0"; do
  set -- $v
  ( ASR_TN_GROUP_TILE=$1 ASR_TN_GROUP_MROWS=$2 timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline ) > gpurun_out/r3f_bench_tn$1_m$2.txt 2>&1
  echo "TN_GROUP_TILE=$1 MROWS=$2: $(tail -1 gpurun_out/r3f_bench_tn$1_m$2.txt | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(round(d["ms_per_step"],3), "ms/step")' 2>&1 | tail -1)"
done
( ASR_TN_GROUP_STAGES=4 timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline ) > gpurun_out/r3f_bench_tn256_nst4.txt 2>&1
echo "TN256 NST=4: $(tail -1 gpurun_out/r3f_bench_tn256_nst4.txt | cut -c1-200)"
bash tools/gpu_profile.sh r3f_bench 13 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline
db=$(find /tmp/prof_r3f_bench -name "*.db" | head -1)
python tools/prof_families.py "$db" gpurun_out/r3f_replayed_families.json "python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline" > /dev/null 2>&1
head -42 gpurun_out/r3f_bench_timeline.txt | cut -c1-140
( timeout 600 python tools/ab/ab_attn_stagger.py ) > gpurun_out/r3f_attn_stagger.txt 2>&1
tail -5 gpurun_out/r3f_attn_stagger.txt | cut -c1-700
bash tools/gpu_attn_pmc.sh r3f_attnpmc > /dev/null 2>&1
grep -A9 "attn_fwd_pp" gpurun_out/r3f_attnpmc.txt | head -64
