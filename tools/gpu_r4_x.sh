#!/bin/bash
mkdir -p gpurun_out
L=end2end-asr-pytorch_amd/asr_hip
cp $L/libasr_hip.so /tmp/new.so
cp $L/libasr_hip_abl.so $L/libasr_hip.so
{
  for a in 0 1 2 4 3 7; do
    echo "== ASR_IGEMM_ABLATE=$a (1 = no patch loads, 2 = no weight loads, 4 = no stores)"; ASR_IGEMM_ABLATE=$a python tools/microbench.py conv 2>&1 | grep 'igemm' | grep -v '161'
  done
} > gpurun_out/r4x_igemm_ablate.txt 2>&1
cp /tmp/new.so $L/libasr_hip.so
cat gpurun_out/r4x_igemm_ablate.txt
