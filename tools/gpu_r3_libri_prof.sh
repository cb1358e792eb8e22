#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/gpu_profile.sh r03_librispeech 8 python bench.py --workload librispeech --steps 5 --warmup 3 --no-cpu-baseline --no-roofline
head -30 gpurun_out/r03_librispeech_timeline.txt | cut -c1-150
