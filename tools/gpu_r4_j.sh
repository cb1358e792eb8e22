#!/bin/bash
# attention backward: what bounds it?  ablation builds of attention_fast.hip (libasr_hip_ablN.so, built by hand, -DATTN_ABL=N):
# 1 = no score math (exp / dropout / dS), 2 = no MFMAs, 3 = no tile prefetch, 4 = no barrier
mkdir -p gpurun_out
L=end2end-asr-pytorch_amd/asr_hip
cp $L/libasr_hip.so /tmp/new.so
{
  for v in new abl1 abl2 abl3 abl4 prev; do
    [ $v = new ] && cp /tmp/new.so $L/libasr_hip.so || cp $L/libasr_hip_$v.so $L/libasr_hip.so
    echo "== $v"
    python tools/mb_attn_bwd.py 32 8 800 800 0.1 2>&1 | grep 'attn bwd'
    python tools/mb_attn_bwd.py 32 8 800 800 0.0 2>&1 | grep 'attn bwd'
  done
  cp /tmp/new.so $L/libasr_hip.so
} > gpurun_out/r4j_attn_ablate.txt 2>&1
cat gpurun_out/r4j_attn_ablate.txt
