#!/bin/bash
# same-box A/B of the built library against asr_hip/libasr_hip_prev.so (built by hand from an older source):  tools/gpu_ab_lib.sh <tag> <command...>
tag=$1; shift
mkdir -p gpurun_out
L=end2end-asr-pytorch_amd/asr_hip
cp $L/libasr_hip.so /tmp/new.so
{
  for rep in 1 2; do
    cp /tmp/new.so $L/libasr_hip.so; echo "== new"; "$@" 2>&1 | grep -v amdgpu.ids
    cp $L/libasr_hip_prev.so $L/libasr_hip.so; echo "== prev"; "$@" 2>&1 | grep -v amdgpu.ids
  done
  cp /tmp/new.so $L/libasr_hip.so
} > gpurun_out/${tag}_ab.txt 2>&1
cat gpurun_out/${tag}_ab.txt
