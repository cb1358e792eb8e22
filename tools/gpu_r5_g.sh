#!/bin/bash
# round 5 (g): ReLU masks as bits -- harness parity + timing, pytest of the conv_ws file, step A/B
cd /root/repo; mkdir -p gpurun_out
timeout 300 tools/bin/conv_ws_test bits 2>&1 | tee gpurun_out/r5g_bits.txt
timeout 600 python -m pytest tests/test_gpu_conv_ws.py -x -q 2>&1 | tail -5 | tee gpurun_out/r5g_pytest.txt
for i in 1 2; do
  for v in 0 1; do
    echo "ASR_RELU_BITS=$v" | tee -a gpurun_out/r5g_step_ab.txt
    ASR_RELU_BITS=$v timeout 300 python bench.py --steps 200 --warmup 20 --soak-seconds 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'])" | tee -a gpurun_out/r5g_step_ab.txt
  done
done
