#!/bin/bash
mkdir -p gpurun_out
( timeout 1200 python -m pytest -q -m gpu --tb=short -x tests/test_gpu_ops.py tests/test_gpu_level0.py -k "conv or wgrad or vgg or level0" 2>&1 | tail -5 ) > gpurun_out/r4u_pytest.log
cat gpurun_out/r4u_pytest.log
bash tools/gpu_ab_lib.sh r4u python tools/microbench.py wgrad
bash tools/gpu_ab_lib.sh r4u2 python tools/mb_level0.py 2
