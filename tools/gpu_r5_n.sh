#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
rm -f gpurun_out/r5n_step_ab.txt
for v in "ASR_WGRAD_GROUP=48" "ASR_WGRAD_GROUP=21" "ASR_WGRAD_GROUP=16" "ASR_WGRAD_GROUP=27" "ASR_WGRAD_GROUP=32" "ASR_WGRAD_GROUP=48 ASR_TN_ROT=2" "ASR_WGRAD_GROUP=21 ASR_TN_ROT=2"; do
  echo "librispeech $v" | tee -a gpurun_out/r5n_step_ab.txt
  env $v timeout 300 python bench.py --workload librispeech --steps 40 --warmup 8 --soak-seconds 0 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])" | tee -a gpurun_out/r5n_step_ab.txt
done
