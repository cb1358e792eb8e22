#!/bin/bash
# A/B: data-gradient tile height (128 rows from ASR_NN_BIG 64x64 tiles on) now that dX shares its launch with dW.
for i in 1 2; do
for m in "ASR_NN_BIG=800" "ASR_NN_BIG=100000" "ASR_NN_BIG=400" "ASR_NN_BIG=1700"; do
  echo "== $m"; env $m python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"ms_per_step": [0-9.]*'
done; done
