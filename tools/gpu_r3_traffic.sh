#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/gpu_pmc_traffic.sh r03 > gpurun_out/r3_traffic.log 2>&1
tail -12 gpurun_out/r3_traffic.log
