#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "window" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_baseline_shapes.py -m gpu -x -q 2>&1 | tail -4
b() { timeout 600 python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))"; }
echo "warm-up (ignore): $(b --workload librispeech)"
for rep in 1 2; do
echo "librispeech default (packet-of-rows dW for both convolutions): $(b --workload librispeech)"
echo "librispeech window-view dW: $(ASR_EMB_SHIFT_WGRAD=0 b --workload librispeech)"
done
