#!/bin/bash
# round 3, call L: configs[3] (librispeech shape) and configs[4] (low-rank) with the round's kernels; kernel stats of configs[3]
export TMPDIR=/tmp
mkdir -p gpurun_out
b() { timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), 'ms/step')"; }
echo "librispeech default: $(b --workload librispeech)"
echo "librispeech GEMM_BIG=0: $(ASR_GEMM_BIG=0 b --workload librispeech)"
echo "librispeech TN_GROUP_TILE=256: $(ASR_TN_GROUP_TILE=256 b --workload librispeech)"
echo "lowrank bf16: $(b --workload lowrank)"
echo "lowrank fp8: $(b --workload lowrank --precision fp8)"
bash tools/gpu_profile.sh r3l_libri 8 python bench.py --workload librispeech --steps 5 --warmup 3 --no-cpu-baseline --no-roofline
head -45 gpurun_out/r3l_libri_kernel_stats.txt | cut -c1-200
