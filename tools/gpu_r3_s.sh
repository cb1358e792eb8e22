#!/bin/bash
# round 3, call S: implicit-GEMM convolution at the 16-row tile: weight double buffer / two taps per step / taps unrolled
export TMPDIR=/tmp
for v in "" "ASR_IGEMM_WBUF=2" "ASR_IGEMM_TPS=2" "ASR_IGEMM_UNROLL=1"; do
  echo "== ${v:-default (one tap per step, one weight buffer, rolled)}"
  env $v timeout 600 python tools/microbench.py conv 2>&1 | grep "igemm (32, 80" | cut -c1-160
done
b() { timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))"; }
echo "warm-up run (ignore): $(b)"
for rep in 1 2; do
echo "default: $(b)"
echo "IGEMM_WBUF=2: $(ASR_IGEMM_WBUF=2 b)"
echo "IGEMM_TPS=2: $(ASR_IGEMM_TPS=2 b)"
done
