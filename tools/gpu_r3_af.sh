#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "wgrad" 2>&1 | tail -2
for v in "" "ASR_WGRAD_XCD=0"; do echo "== ${v:-default (sibling workgroups on one XCD)}"; env $v timeout 600 python tools/microbench.py wgrad 2>&1 | grep -i "conv\|wgrad" | head -8 | cut -c1-160; done
b() { timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))"; }
echo "warm-up run (ignore): $(b)"
for rep in 1 2; do
echo "default: $(b)"
echo "ASR_WGRAD_XCD=0: $(ASR_WGRAD_XCD=0 b)"
done
