#!/bin/bash
mkdir -p gpurun_out
L=end2end-asr-pytorch_amd/asr_hip
cp $L/libasr_hip.so /tmp/new.so
{
  for v in new abl1 abl2; do
    [ $v = new ] && cp /tmp/new.so $L/libasr_hip.so || cp $L/libasr_hip_$v.so $L/libasr_hip.so
    echo "== $v"; python tools/microbench.py wgrad 2>&1 | grep wgrad-NHWC
  done
  cp /tmp/new.so $L/libasr_hip.so
} > gpurun_out/r4w_wgrad_ablate.txt 2>&1
cat gpurun_out/r4w_wgrad_ablate.txt
