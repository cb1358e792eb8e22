#!/bin/bash
# In-kernel section timing (s_memtime stamps of workgroup 0) of the level-0 forward kernel or of the grouped weight gradient.
# asr_hip/libasr_hip_timing.so = the library with csrc/conv_level0.hip built -DL0_TIMING, or csrc/gemm.hip built -DTN_TIMING (by hand:
# the other objects from asr_hip/_obj).  usage: tools/gpu_section_timing.sh <tag> level0|tn
tag=${1:-sect}; what=${2:-level0}
cd ${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p gpurun_out; export TMPDIR=/tmp
L=end2end-asr-pytorch_amd/asr_hip
cp $L/libasr_hip.so /tmp/new.so; cp $L/libasr_hip_timing.so $L/libasr_hip.so
if [ "$what" = tn ]; then      # the bench's eager warm-up steps print the totals
  timeout 300 python bench.py --steps 2 --warmup 12 --eager --no-cpu-baseline --no-roofline --soak-seconds 0 2>&1 | grep "tn256g timing" > gpurun_out/${tag}_timing.txt
else
  timeout 300 python tools/mb_level0.py 1 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_timing.txt
fi
cp /tmp/new.so $L/libasr_hip.so
cat gpurun_out/${tag}_timing.txt
