#!/bin/bash
# Attention after a kernel change: dropout statistics + attention parity tests, then the attention microbenchmark, the headline step and
# configs[3] against asr_hip/libasr_hip_prev.so (built by hand from older sources) in the same call.  usage: tools/gpu_attn_ab.sh <tag>
tag=${1:-attn}
cd ${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_dropout_stats.py tests/test_gpu_ops.py -x -q -k "dropout or attention" 2>&1 | tail -6 > gpurun_out/${tag}_tests.log
cat gpurun_out/${tag}_tests.log
bash tools/gpu_ab_lib.sh ${tag}_mb python tools/microbench.py attn > /dev/null 2>&1
grep "==\|attn" gpurun_out/${tag}_mb_ab.txt
bash tools/gpu_ab_lib.sh ${tag}_step python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --soak-seconds 0 > /dev/null 2>&1
grep -o '"ms_per_step": [0-9.]*\|== [a-z]*' gpurun_out/${tag}_step_ab.txt | paste - -
bash tools/gpu_ab_lib.sh ${tag}_libri python bench.py --workload librispeech --steps 40 --warmup 8 --no-cpu-baseline --no-roofline --soak-seconds 0 > /dev/null 2>&1
grep -o '"ms_per_step": [0-9.]*\|== [a-z]*' gpurun_out/${tag}_libri_ab.txt | paste - -
