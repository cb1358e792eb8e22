#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm_nt or gemm_nn" 2>&1 | tail -8
b() { timeout 600 python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))"; }
echo "warm-up (ignore): $(b)"
for rep in 1 2; do
echo "headline default (NT ring for <= 512 blocks): $(b)"
echo "headline ASR_NT_RING=0: $(ASR_NT_RING=0 b)"
echo "headline ASR_NT_RING=1024: $(ASR_NT_RING=1024 b)"
done
echo "librispeech default: $(b --workload librispeech)"
echo "librispeech ASR_NT_RING=0: $(ASR_NT_RING=0 b --workload librispeech)"
echo "lowrank default: $(b --workload lowrank)"
echo "lowrank ASR_NT_RING=0: $(ASR_NT_RING=0 b --workload lowrank)"
