#!/bin/bash
# round 3, GPU call A: blind diagnostics + A/B of the long-sequence attention forward, the whole GPU suite, PMC pass, bench line
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python tools/ab/ab_attn_pp.py all ) > gpurun_out/r3a_attn_pp.txt 2>&1
tail -60 gpurun_out/r3a_attn_pp.txt
( timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -80 ) > gpurun_out/r3a_pytest.txt
tail -40 gpurun_out/r3a_pytest.txt
bash tools/gpu_attn_pmc.sh r3a_attnpmc > /dev/null 2>&1
head -60 gpurun_out/r3a_attnpmc.txt
( timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline ) > gpurun_out/r3a_bench.txt 2>&1
tail -3 gpurun_out/r3a_bench.txt
