#!/bin/bash
# round 3, call T: implicit GEMM, taps unrolled + weights prefetched two taps ahead
export TMPDIR=/tmp
timeout 600 env ASR_IGEMM_UNROLL=1 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "conv3x3" 2>&1 | tail -3
for v in "" "ASR_IGEMM_UNROLL=1"; do
  echo "== ${v:-default (rolled, one tap ahead)}"
  env $v timeout 600 python tools/microbench.py conv 2>&1 | grep "igemm" | grep "(32, 80" | cut -c1-160
done
