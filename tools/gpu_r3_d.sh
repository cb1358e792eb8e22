#!/bin/bash
# round 3, GPU call D: GEMM tile x stages A/B, attention 4 vs 8 waves, suite (CTC under DDP, trainer graph path under a reducer)
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python tools/ab/ab_gemm_tiles.py ) > gpurun_out/r3d_gemm_tiles.txt 2>&1
cat gpurun_out/r3d_gemm_tiles.txt | cut -c1-330
( ASR_ATTN_PP_WAVES=4 timeout 300 python tools/ab/ab_attn_pp.py time ) > gpurun_out/r3d_attn_w4.txt 2>&1
( ASR_ATTN_PP_WAVES=8 timeout 300 python tools/ab/ab_attn_pp.py all ) > gpurun_out/r3d_attn_w8.txt 2>&1
grep "800,800\|795,795" gpurun_out/r3d_attn_w4.txt | cut -c1-200; tail -32 gpurun_out/r3d_attn_w8.txt | cut -c1-200
( timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -40 ) > gpurun_out/r3d_pytest.txt
tail -15 gpurun_out/r3d_pytest.txt
