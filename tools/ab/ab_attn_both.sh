#!/bin/bash
# A/B on the headline bench: attention backward as ONE launch (default), as two launches on two streams (ASR_ATTN_SPLIT=1), on one stream (ASR_ATTN_BOTH=0).
for i in 1 2; do
for m in both split serial; do
  case $m in both) e="ASR_ATTN_SPLIT=0";; split) e="ASR_ATTN_SPLIT=1";; serial) e="ASR_ATTN_SPLIT=0 ASR_ATTN_BOTH=0";; esac
  echo "== $m"; env $e python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"ms_per_step": [0-9.]*'
done; done
