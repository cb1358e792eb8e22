#!/bin/bash
# In-kernel section timing of the level-0 forward kernel: asr_hip/libasr_hip_timing.so = the library with conv_level0.hip built -DL0_TIMING
# (by hand; see DESIGN.md / NOTEBOOK.md).  usage: tools/gpu_level0_timing.sh <tag>
tag=${1:-l0t}
cd ${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p gpurun_out; export TMPDIR=/tmp
L=end2end-asr-pytorch_amd/asr_hip
cp $L/libasr_hip.so /tmp/new.so; cp $L/libasr_hip_timing.so $L/libasr_hip.so
timeout 300 python tools/mb_level0.py 1 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_timing.txt
cp /tmp/new.so $L/libasr_hip.so
cat gpurun_out/${tag}_timing.txt
