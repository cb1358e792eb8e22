#!/bin/bash
# Convolution front end after a kernel change: parity tests, then the replayed headline step against asr_hip/libasr_hip_prev.so (built by
# hand from older sources) in the same call.  usage: tools/gpu_conv_ab.sh <tag>
tag=${1:-conv}
cd ${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_level0.py tests/test_gpu_conv_ws.py tests/test_gpu_frontend_exact.py tests/test_gpu_pool_handover.py -x -q 2>&1 | tail -8 > gpurun_out/${tag}_tests.log
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k conv 2>&1 | tail -4 >> gpurun_out/${tag}_tests.log
cat gpurun_out/${tag}_tests.log
bash tools/gpu_ab_lib.sh ${tag}_step python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --soak-seconds 0 > /dev/null 2>&1
grep -o '"ms_per_step": [0-9.]*\|== [a-z]*\|"final_loss": [0-9.]*' gpurun_out/${tag}_step_ab.txt | paste - - -
