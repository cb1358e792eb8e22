#!/bin/bash
# round 3, GPU call E: attention stagger / pipelining A/B, suite with the four-graph data-parallel step
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python tools/ab/ab_attn_stagger.py ) > gpurun_out/r3e_attn_stagger.txt 2>&1
cut -c1-420 gpurun_out/r3e_attn_stagger.txt | tail -14
( timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -40 ) > gpurun_out/r3e_pytest.txt
tail -12 gpurun_out/r3e_pytest.txt
( ASR_FORCE_DDP=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline ) > gpurun_out/r3e_bench_ddp1.txt 2>&1
tail -1 gpurun_out/r3e_bench_ddp1.txt | cut -c1-240
