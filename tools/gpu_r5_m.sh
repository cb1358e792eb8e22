#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
rm -f gpurun_out/r5m_step_ab.txt
for i in 1 2; do
  for v in "ASR_TN_ROT=1" "ASR_WGRAD_GROUP=24" "ASR_TN_ROT_NST=4" "ASR_WGRAD_GROUP=32"; do
    echo "headline $v" | tee -a gpurun_out/r5m_step_ab.txt
    env $v timeout 300 python bench.py --steps 200 --warmup 20 --soak-seconds 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])" | tee -a gpurun_out/r5m_step_ab.txt
  done
done
