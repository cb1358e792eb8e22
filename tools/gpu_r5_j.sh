#!/bin/bash
# round 5 (j): grouped weight gradients with sliced over-long blocks: librispeech A/B, headline sequence profile
cd /root/repo; mkdir -p gpurun_out
rm -f gpurun_out/r5j_step_ab.txt
old="ASR_TN_ROT=0 ASR_WGRAD_GROUP=32 ASR_WGRAD_STAGES=38000"
timeout 150 tools/bin/tn_grouped_test | tail -4 | tee gpurun_out/r5j_harness.txt
for v in "$old" "ASR_TN_ROT=1" "ASR_TN_ROT=1 ASR_WGRAD_GROUP=24" "$old" "ASR_TN_ROT=1"; do
  echo "librispeech $v" | tee -a gpurun_out/r5j_step_ab.txt
  env $v timeout 300 python bench.py --workload librispeech --steps 40 --warmup 8 --soak-seconds 0 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])" | tee -a gpurun_out/r5j_step_ab.txt
done
bash tools/gpu_r5_seq.sh r5j --soak-seconds 0 | grep -E "tn256|gemm_tn" 
grep -E "tn256|gemm_tn|tn_" gpurun_out/r5j_step_sequence.txt | cut -c1-140
head -1 gpurun_out/r5j_step_sequence.txt
