#!/bin/bash
# round 3, call Z: the transformer's Adam update on the second stream under the conv backward
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_graph.py tests/test_gpu_train_cli.py tests/test_gpu_fullsize_properties.py -m gpu -q -x 2>&1 | tail -4
b() { timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))"; }
echo "warm-up run (ignore): $(b)"
for rep in 1 2; do
echo "default (early Adam): $(b)"
echo "ASR_EARLY_ADAM=0: $(ASR_EARLY_ADAM=0 b)"
done
echo "librispeech default: $(b --workload librispeech)"
echo "librispeech ASR_EARLY_ADAM=0: $(ASR_EARLY_ADAM=0 b --workload librispeech)"
