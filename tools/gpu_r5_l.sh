#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 150 tools/bin/tn_grouped_test | tail -4 | tee gpurun_out/r5l_harness.txt
timeout 150 tools/bin/tn_grouped_test big | tail -4 | tee -a gpurun_out/r5l_harness.txt
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -k "tn or wgrad or grouped" 2>&1 | tail -2
rm -f gpurun_out/r5l_step_ab.txt
old="ASR_TN_ROT=0 ASR_WGRAD_GROUP=32 ASR_WGRAD_STAGES=38000"
for v in "$old" "ASR_TN_ROT=1" "ASR_TN_ROT=1 ASR_WGRAD_GROUP=32 ASR_WGRAD_STAGES=38000" "$old" "ASR_TN_ROT=1"; do
  echo "librispeech $v" | tee -a gpurun_out/r5l_step_ab.txt
  env $v timeout 300 python bench.py --workload librispeech --steps 40 --warmup 8 --soak-seconds 0 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])" | tee -a gpurun_out/r5l_step_ab.txt
done
