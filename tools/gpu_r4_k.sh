#!/bin/bash
mkdir -p gpurun_out
( timeout 1500 python -m pytest -q -m gpu --tb=short -x tests/test_gpu_ops.py tests/test_gpu_dropout_stats.py -k "attention or attn or dropout" 2>&1 | tail -4 ) > gpurun_out/r4k_pytest.log
cat gpurun_out/r4k_pytest.log
{
  echo "== N = 1 (default)"; python tools/mb_attn_bwd.py 32 8 800 800 0.1 2>&1 | grep 'attn bwd'; python tools/mb_attn_bwd.py 32 8 800 800 0.0 2>&1 | grep 'attn bwd'
  echo "== N = 2 (ASR_ATTN_SHORT_BWD=256)"; ASR_ATTN_SHORT_BWD=256 python tools/mb_attn_bwd.py 32 8 800 800 0.1 2>&1 | grep 'attn bwd'; ASR_ATTN_SHORT_BWD=256 python tools/mb_attn_bwd.py 32 8 800 800 0.0 2>&1 | grep 'attn bwd'
  python tools/mb_attn_bwd.py 32 8 200 200 0.1 2>&1 | grep 'attn bwd'
  python tools/mb_attn_bwd.py 16 8 795 795 0.1 2>&1 | grep 'attn bwd'
} > gpurun_out/r4k_attn.txt 2>&1
cat gpurun_out/r4k_attn.txt
