#!/bin/bash
# Linear-layer GEMMs after a kernel change: GEMM parity tests + the model-level gradient tests, then the headline step and configs[3] against
# asr_hip/libasr_hip_prev.so (built by hand from older sources) in the same call.  usage: tools/gpu_gemm_ab.sh <tag>
tag=${1:-gemm}
cd ${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_ops.py -x -q -k "gemm or grouped or colsum" 2>&1 | tail -4 > gpurun_out/${tag}_tests.log
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_graph.py -x -q 2>&1 | tail -4 >> gpurun_out/${tag}_tests.log
cat gpurun_out/${tag}_tests.log
bash tools/gpu_ab_lib.sh ${tag}_step python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --soak-seconds 0 > /dev/null 2>&1
grep -o '"ms_per_step": [0-9.]*\|== [a-z]*' gpurun_out/${tag}_step_ab.txt | paste - -
bash tools/gpu_ab_lib.sh ${tag}_libri python bench.py --workload librispeech --steps 40 --warmup 8 --no-cpu-baseline --no-roofline --soak-seconds 0 > /dev/null 2>&1
grep -o '"ms_per_step": [0-9.]*\|== [a-z]*' gpurun_out/${tag}_libri_ab.txt | paste - -
