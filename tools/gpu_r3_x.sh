#!/bin/bash
# round 3, call X: the loss hands the vocabulary projection its gradient in bf16 (no fp32 dlogits, no cast / pad launch)
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_graph.py tests/test_gpu_baseline_shapes.py tests/test_gpu_ddp.py tests/test_gpu_fullsize_properties.py -m gpu -q -x 2>&1 | tail -4
b() { timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))"; }
echo "warm-up run (ignore): $(b)"
for rep in 1 2; do
echo "default (bf16 hand-over): $(b)"
echo "ASR_LOGIT_HANDOVER=0: $(ASR_LOGIT_HANDOVER=0 b)"
done
