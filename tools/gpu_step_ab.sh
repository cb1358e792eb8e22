#!/bin/bash
# Same-call A/B of the replayed headline step under environment settings (one gpurun call; boxes differ by up to 5 %, so only pairs
# taken inside ONE call compare).  usage: tools/gpu_step_ab.sh <tag> <reps> "<ENV1=a ENV2=b>" "<ENV1=c>" ...   ("" = defaults)
#   -> gpurun_out/<tag>_step_ab.txt: one line per run: the setting, ms per step, final loss
tag=$1; reps=$2; shift 2
cd ${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p gpurun_out; export TMPDIR=/tmp
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --soak-seconds 0 ${ASR_BENCH_FLAGS:-}"
out=gpurun_out/${tag}_step_ab.txt; : > $out
for rep in $(seq 1 $reps); do
  for cfg in "$@"; do
    echo -n "[${cfg:-defaults}] " | tee -a $out
    env $cfg timeout 300 $B 2>/dev/null | python -c "
import json, sys
l = sys.stdin.readlines()
d = json.loads(l[-1]) if l else {}
print(d.get('ms_per_step'), (d.get('config') or {}).get('final_loss'))" | tee -a $out
  done
done
