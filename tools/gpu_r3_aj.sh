#!/bin/bash
export TMPDIR=/tmp
b() { timeout 600 python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))"; }
echo "warm-up (ignore): $(b)"
for rep in 1 2; do
echo "headline default (stage budget): $(b)"
echo "headline no budget: $(ASR_WGRAD_STAGES=0 b)"
echo "librispeech default (stage budget): $(b --workload librispeech)"
echo "librispeech no budget: $(ASR_WGRAD_STAGES=0 b --workload librispeech)"
echo "librispeech budget 30000: $(ASR_WGRAD_STAGES=30000 b --workload librispeech)"
done
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_train.py -m gpu -x -q 2>&1 | tail -3
