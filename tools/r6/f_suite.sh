#!/bin/bash
# round 6, call F: whole GPU suite on the pruned tree + the two torch-free harnesses + a bench line
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r6f
( timeout 1800 python -m pytest -q -m gpu --tb=short tests 2>&1 | tail -40 ) > ${O}_pytest_gpu.log
tail -12 ${O}_pytest_gpu.log
timeout 300 tools/bin/tn_grouped_test 2>&1 | tail -8 | tee ${O}_tn_grouped_test.txt
timeout 300 tools/bin/conv_ws_test 2>&1 | grep -E "OK|FAILED" | tail -3 | tee ${O}_conv_ws_test.txt
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --soak-seconds 0 2>/dev/null | tail -1 > ${O}_bench_line.json
python -c "
import json; d=json.load(open('${O}_bench_line.json')); r=d['roofline']
print(d['ms_per_step'], r['frac'], r.get('frac_by_time_largest_family'), r.get('largest_family_by_time'))"
