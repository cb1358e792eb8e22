#!/bin/bash
# attention: op tests + micro-benchmark (new forward kernel vs the first generation)
tag=${1:-attn}
mkdir -p gpurun_out
( timeout 900 python -m pytest -q -m gpu --tb=short -k "attention" tests/test_gpu_ops.py 2>&1 | tail -15 ) > gpurun_out/${tag}_pytest.log
tail -15 gpurun_out/${tag}_pytest.log
( python tools/microbench.py attn; ASR_ATTN_FWD_V1=1 python tools/microbench.py attn ) > gpurun_out/${tag}_microbench.txt 2>&1
cat gpurun_out/${tag}_microbench.txt
