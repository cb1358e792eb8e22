#!/bin/bash
# A/B of the software-pipelined attention forward (ASR_ATTN_PIPE=1: 16-query waves, 2: 32-query waves) on the north-star shape.
for m in 0 2 1; do
  echo "== ASR_ATTN_PIPE=$m"
  ASR_ATTN_PIPE=$m ASR_ATTN_SHORT=0 python tools/microbench.py attn 2>&1 | grep "attn (32, 8, 800\|attn (16, 8, 795\|attn (32, 8, 200, 200" | sed 's/| bwd.*//'
done
