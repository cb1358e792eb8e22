#!/bin/bash
# round 5, closing call on the final tree: the whole GPU suite, smoke, the headline bench line
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 2400 python -m pytest -q -m gpu --tb=short tests 2>&1 | tail -25 ) > gpurun_out/r05_pytest_gpu.log
tail -5 gpurun_out/r05_pytest_gpu.log
python __graft_entry__.py smoke 2>&1 | grep smoke > gpurun_out/r05_smoke.log; cat gpurun_out/r05_smoke.log
python bench.py > gpurun_out/r05_bench_line.json 2> gpurun_out/r05_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_bench_line.json").read().strip().splitlines()[-1]); r = d.get("roofline") or {}
print("ms/step %.3f value %.0f frac %s" % (d["ms_per_step"], d["value"], r.get("frac")))
PY
