#!/bin/bash
# round 5: same-box A/B of the step over the WS128 selections (0 generic igemm, 1 conv_ws.hip everywhere, 2 conv_ws16.hip everywhere, unset = per-form default)
mkdir -p gpurun_out; export TMPDIR=/tmp
{
  for rep in 1 2; do
    for ws in default 1 2 0; do
      echo "== ASR_WS128=$ws"
      if [ $ws = default ]; then python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1
      else ASR_WS128=$ws python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1; fi
    done
  done
} > gpurun_out/r5b_step_ab.txt 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r5b_step_ab.txt"):
    l = l.strip()
    if l.startswith("=="): print(l, end="  ")
    elif l.startswith("{"): print("ms/step %.3f" % json.loads(l)["ms_per_step"])
PY
