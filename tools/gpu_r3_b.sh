#!/bin/bash
# round 3, GPU call B: 8-wave ping-pong attention forward (diagnostics, A/B, PMC), full suite with the fused shadow write
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python tools/ab/ab_attn_pp.py all ) > gpurun_out/r3b_attn_pp.txt 2>&1
tail -70 gpurun_out/r3b_attn_pp.txt
( timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -60 ) > gpurun_out/r3b_pytest.txt
tail -30 gpurun_out/r3b_pytest.txt
ASR_ATTN_PP_WAVES=8 bash tools/gpu_attn_pmc.sh r3b_attnpmc8 > /dev/null 2>&1
grep -A9 "pp8" gpurun_out/r3b_attnpmc8.txt | head -70
( timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline ) > gpurun_out/r3b_bench.txt 2>&1
tail -1 gpurun_out/r3b_bench.txt | cut -c1-300
