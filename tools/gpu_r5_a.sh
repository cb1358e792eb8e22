#!/bin/bash
# round 5: conv tests on the weight-stationary kernel + same-box A/B of the step (ASR_WS128=0 restores the generic implicit GEMM)
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest -q -m gpu --tb=short tests/test_gpu_ops.py -k "conv or pool or vgg" tests/test_gpu_level0.py 2>&1 | tail -15 ) > gpurun_out/r5a_pytest.log
tail -4 gpurun_out/r5a_pytest.log
{
  for rep in 1 2; do
    for ws in 1 0; do
      echo "== ASR_WS128=$ws"; ASR_WS128=$ws python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1
    done
  done
} > gpurun_out/r5a_step_ab.txt 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r5a_step_ab.txt"):
    l = l.strip()
    if l.startswith("=="): print(l, end="  ")
    elif l.startswith("{"): print("ms/step %.3f" % json.loads(l)["ms_per_step"])
PY
