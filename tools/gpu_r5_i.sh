#!/bin/bash
# round 5 (i): grouped weight gradients (TN_ROT) and the delta epilogue (NN_ROWDOT) in the replayed step, both workloads
cd /root/repo; mkdir -p gpurun_out
rm -f gpurun_out/r5i_step_ab.txt
old="ASR_TN_ROT=0 ASR_WGRAD_GROUP=32 ASR_WGRAD_STAGES=38000"
for i in 1 2; do
  for v in "$old" "ASR_TN_ROT=1" "ASR_TN_ROT=1 ASR_NN_ROWDOT=0"; do
    echo "headline $v" | tee -a gpurun_out/r5i_step_ab.txt
    env $v timeout 300 python bench.py --steps 200 --warmup 20 --soak-seconds 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'])" | tee -a gpurun_out/r5i_step_ab.txt
  done
done
for v in "$old" "ASR_TN_ROT=1" "ASR_TN_ROT=1 ASR_WGRAD_GROUP=24"; do
  echo "librispeech $v" | tee -a gpurun_out/r5i_step_ab.txt
  env $v timeout 300 python bench.py --workload librispeech --steps 40 --warmup 8 --soak-seconds 0 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])" | tee -a gpurun_out/r5i_step_ab.txt
done
