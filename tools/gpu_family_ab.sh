#!/bin/bash
# A kernel family after a change, one gpurun call: its parity tests, then the replayed headline step and configs[3] -- and for `attn` the
# attention microbenchmark -- against asr_hip/libasr_hip_prev.so (the library built by hand from older sources; tools/gpu_ab_lib.sh).
# usage: tools/gpu_family_ab.sh <tag> conv|gemm|attn
tag=${1:-fam}; fam=${2:-conv}
cd ${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p gpurun_out; export TMPDIR=/tmp
T="timeout 1500 python -m pytest -x -q"
case $fam in
  conv) $T tests/test_gpu_level0.py tests/test_gpu_conv_ws.py tests/test_gpu_frontend_exact.py tests/test_gpu_pool_handover.py 2>&1 | tail -6 > gpurun_out/${tag}_tests.log
        $T tests/test_gpu_ops.py -k conv 2>&1 | tail -4 >> gpurun_out/${tag}_tests.log ;;
  gemm) $T tests/test_gpu_ops.py -k "gemm or grouped or colsum" 2>&1 | tail -4 > gpurun_out/${tag}_tests.log
        $T tests/test_gpu_model.py tests/test_gpu_graph.py 2>&1 | tail -4 >> gpurun_out/${tag}_tests.log ;;
  attn) $T tests/test_gpu_dropout_stats.py tests/test_gpu_ops.py -k "dropout or attention" 2>&1 | tail -6 > gpurun_out/${tag}_tests.log ;;
esac
cat gpurun_out/${tag}_tests.log
if [ "$fam" = attn ]; then
  bash tools/gpu_ab_lib.sh ${tag}_mb python tools/microbench.py attn > /dev/null 2>&1
  grep "==\|attn" gpurun_out/${tag}_mb_ab.txt
fi
bash tools/gpu_ab_lib.sh ${tag}_step python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --soak-seconds 0 > /dev/null 2>&1
grep -o '"ms_per_step": [0-9.]*\|== [a-z]*' gpurun_out/${tag}_step_ab.txt | paste - -
bash tools/gpu_ab_lib.sh ${tag}_libri python bench.py --workload librispeech --steps 40 --warmup 8 --no-cpu-baseline --no-roofline --soak-seconds 0 > /dev/null 2>&1
grep -o '"ms_per_step": [0-9.]*\|== [a-z]*' gpurun_out/${tag}_libri_ab.txt | paste - -
