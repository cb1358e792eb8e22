#!/bin/bash
# A/B of the one-launch linear backward (ASR_NN_TN), its slice length (ASR_NNTN_STAGES), the fold riding in the next launch
# (ASR_TN_FOLD_NEXT) and the weight-size limit (ASR_NN_TN_MAX) on the headline bench.
for i in 1 2; do
for m in "ASR_NN_TN=0" "ASR_NN_TN=1" "ASR_TN_FOLD_NEXT=0" "ASR_NN_TN_MAX=100000000" "ASR_NNTN_STAGES=24" "ASR_NNTN_STAGES=12" "ASR_NN_TN_MAX=600000"; do
  echo "== $m"; env $m python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"ms_per_step": [0-9.]*'
done; done
