#!/bin/bash
# A/B: kernel arguments of the replayed graph in device memory (HIP_FORCE_DEV_KERNARG=1) vs the runtime's default placement.
for i in 1 2; do
for m in "HIP_FORCE_DEV_KERNARG=0" "HIP_FORCE_DEV_KERNARG=1"; do
  echo "== $m"; env $m python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"ms_per_step": [0-9.]*'
done; done
echo "== decode"; for m in "HIP_FORCE_DEV_KERNARG=0" "HIP_FORCE_DEV_KERNARG=1"; do echo $m; env $m MICRO_DECODE_GRAPH_ONLY=1 python tools/microbench.py decode 2>&1 | grep "ms" | tail -2; done
