#!/bin/bash
# round 5: conv.7 forward's pooled epilogue on vertical tile pairs (ASR_WS_PAIR=1, default) against single tiles (0): harness parity + timing + same-box step A/B
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 tools/bin/conv_ws_test 2>&1 | tee gpurun_out/r5f_conv_ws_test.txt | grep -E "pooled=1|OK|FAILED"
( timeout 600 python -m pytest -q -m gpu --tb=short -x tests/test_gpu_ops.py -k "conv or pool or vgg" tests/test_gpu_level0.py 2>&1 | tail -3 )
B="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline --soak-seconds 0"
{
  for rep in 1 2 3; do
    for ws in 1 0; do echo "== ASR_WS_PAIR=$ws"; ASR_WS_PAIR=$ws $B 2>/dev/null | tail -1; done
  done
} > gpurun_out/r5f_step_ab.txt 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r5f_step_ab.txt"):
    l = l.strip()
    if l.startswith("=="): print(l, end="  ")
    elif l.startswith("{"): print("ms/step %.3f" % json.loads(l)["ms_per_step"])
PY
