#!/bin/bash
# round 3, call M: eight-wave NN (data gradient) blocks: op tests, step A/B; which weight gradients of configs[3] leave the grouped launch
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "gemm" 2>&1 | tail -15 > gpurun_out/r3m_pytest.txt
tail -6 gpurun_out/r3m_pytest.txt | cut -c1-300
b() { timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))"; }
echo "warm-up run (ignore): $(b)"
for rep in 1 2; do
echo "default (eight-wave NN on): $(b)"
echo "GEMM_BIG_NN=0: $(ASR_GEMM_BIG_NN=0 b)"
done
ASR_DEBUG_GROUP=1 timeout 600 python bench.py --workload librispeech --steps 3 --warmup 2 --no-cpu-baseline --no-roofline 2>&1 | grep "gemm_tn_grouped" | sort | uniq -c | head -20
echo "librispeech default: $(b --workload librispeech)"
echo "librispeech GEMM_BIG_NN=0: $(ASR_GEMM_BIG_NN=0 b --workload librispeech)"
echo "lowrank bf16: $(b --workload lowrank)"
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > gpurun_out/r3m_pytest_all.txt
tail -4 gpurun_out/r3m_pytest_all.txt | cut -c1-300
