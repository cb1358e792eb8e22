#!/bin/bash
# round 5 (h): attention delta from the output projection's data-gradient epilogue -- op test, model tests, step A/B (both workloads)
cd /root/repo; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "rowdot or gemm_nn" 2>&1 | tail -5 | tee gpurun_out/r5h_pytest_ops.txt
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_graph.py -x -q 2>&1 | tail -5 | tee gpurun_out/r5h_pytest_model.txt
rm -f gpurun_out/r5h_step_ab.txt
for i in 1 2; do
  for v in 0 1; do
    echo "ASR_NN_ROWDOT=$v" | tee -a gpurun_out/r5h_step_ab.txt
    ASR_NN_ROWDOT=$v timeout 300 python bench.py --steps 200 --warmup 20 --soak-seconds 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'])" | tee -a gpurun_out/r5h_step_ab.txt
  done
done
for v in 0 1; do
  echo "librispeech ASR_NN_ROWDOT=$v" | tee -a gpurun_out/r5h_step_ab.txt
  ASR_NN_ROWDOT=$v timeout 300 python bench.py --workload librispeech --steps 40 --warmup 8 --soak-seconds 0 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])" | tee -a gpurun_out/r5h_step_ab.txt
done
