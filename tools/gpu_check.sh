#!/bin/bash
# One gpurun call: GPU test suite, bench (1 GPU; 1 rank with the data-parallel path forced), kernel profile.
# usage: tools/gpu_check.sh [tag]       logs -> gpurun_out/<tag>_*
tag=${1:-r02}
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -q -m gpu --tb=short 2>&1 | tail -150 ) > gpurun_out/${tag}_pytest.log
tail -60 gpurun_out/${tag}_pytest.log
( timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | grep '^{"metric"' ) > gpurun_out/${tag}_bench.log
tail -c 3000 gpurun_out/${tag}_bench.log
( ASR_FORCE_DDP=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep '^{"metric"' ) > gpurun_out/${tag}_bench_ddp1.log
tail -c 1500 gpurun_out/${tag}_bench_ddp1.log
tools/gpu_profile.sh ${tag} 13 "python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline"
head -45 gpurun_out/${tag}_kernel_stats.txt
