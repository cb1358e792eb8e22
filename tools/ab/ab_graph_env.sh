#!/bin/bash
# A/B of HIP runtime switches that touch graph replay: AQL packet capture at instantiation (DEBUG_CLR_GRAPH_PACKET_CAPTURE).
for i in 1 2; do
for m in "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "X_DEFAULT=1"; do
  echo "== $m"; env $m python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"ms_per_step": [0-9.]*'
done; done
