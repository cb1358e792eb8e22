#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "attention" 2>&1 | tail -4
rm -f gpurun_out/r5o_step_ab.txt
for i in 1 2; do
  for v in "ASR_ATTN_BWD_FUSED=0" "ASR_ATTN_BWD_FUSED=1"; do
    echo "headline $v" | tee -a gpurun_out/r5o_step_ab.txt
    env $v timeout 300 python bench.py --steps 200 --warmup 20 --soak-seconds 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['config'].get('final_loss'))" | tee -a gpurun_out/r5o_step_ab.txt
  done
done
