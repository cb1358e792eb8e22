#!/bin/bash
# In-kernel section timing of the grouped weight gradient (gemm_tn256g): asr_hip/libasr_hip_timing.so = the library with gemm.hip built
# -DTN_TIMING (by hand).  The bench's eager warm-up steps print the per-section totals of workgroup 0.  usage: tools/gpu_tn_timing.sh <tag>
tag=${1:-tnt}
cd ${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p gpurun_out; export TMPDIR=/tmp
L=end2end-asr-pytorch_amd/asr_hip
cp $L/libasr_hip.so /tmp/new.so; cp $L/libasr_hip_timing.so $L/libasr_hip.so
timeout 300 python bench.py --steps 2 --warmup 12 --eager --no-cpu-baseline --no-roofline --soak-seconds 0 2>&1 | grep "tn256g timing" > gpurun_out/${tag}_timing.txt
cp /tmp/new.so $L/libasr_hip.so
cat gpurun_out/${tag}_timing.txt
