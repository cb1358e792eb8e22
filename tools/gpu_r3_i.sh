#!/bin/bash
# round 3, call I: eight-wave GEMM blocks (op test + microbench + step), lifted decode limits, fixed tests of call H
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_decode.py tests/test_gpu_decode_fused.py tests/test_gpu_lowrank.py tests/test_gpu_model.py -m gpu -q -x 2>&1 | tail -25 > gpurun_out/r3i_pytest.txt
tail -8 gpurun_out/r3i_pytest.txt | cut -c1-300
timeout 600 python tools/ab/mb_gemm_big.py > gpurun_out/r3i_gemm_big.txt 2>&1
cut -c1-330 gpurun_out/r3i_gemm_big.txt
b() { timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; print('$1', round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))"; }
b "warm-up run (ignore):"
b "GEMM_BIG auto:"
ASR_GEMM_BIG=0 b "GEMM_BIG off:"
b "GEMM_BIG auto:"
ASR_GEMM_BIG=0 b "GEMM_BIG off:"
ASR_GROUP_CROSS_KV=0 b "auto, per-layer cross KV:"
ASR_GEMM_BIG_NS=2 b "auto, 128-blocks with 2 stages:"
