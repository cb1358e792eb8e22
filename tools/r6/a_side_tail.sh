#!/bin/bash
# round 6, call A: (1) GPU suite on the refactored tree; (2) harness after the conv_ws dispatch edit; (3) the transformer tail on a second
# graph branch (ASR_SIDE_TAIL=0|1|2): graph tests + same-call step A/B + kernel sequence of the replayed step
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r6a
( timeout 1500 python -m pytest -q -m gpu --tb=short tests 2>&1 | tail -25 ) > ${O}_pytest_gpu.log
tail -3 ${O}_pytest_gpu.log
timeout 300 tools/bin/conv_ws_test 2>&1 | grep -E "OK|FAILED|time" | tail -20 > ${O}_conv_ws_test.txt
tail -3 ${O}_conv_ws_test.txt
for v in 1 2; do
  ( ASR_SIDE_TAIL=$v timeout 600 python -m pytest -q -m gpu --tb=short tests/test_gpu_graph.py 2>&1 | tail -5 ) > ${O}_graph_tests_tail$v.log
  tail -2 ${O}_graph_tests_tail$v.log
done
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --soak-seconds 0"
rm -f ${O}_step_ab.txt
for rep in 1 2 3; do
  for v in 0 1 2; do
    echo -n "ASR_SIDE_TAIL=$v " | tee -a ${O}_step_ab.txt
    ASR_SIDE_TAIL=$v timeout 300 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['config'].get('final_loss'))" | tee -a ${O}_step_ab.txt
  done
done
for v in 0 1 2; do
  out=/tmp/prof_r6a_$v; rm -rf $out
  ( ASR_SIDE_TAIL=$v timeout 600 rocprofv3 --kernel-trace --stats -d $out -o trace -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --soak-seconds 0 ) > ${O}_prof$v.log 2>&1
  db=$(find $out -name "*.db" | head -1)
  python tools/prof_sequence.py "$db" ${O}_step_sequence_tail$v.txt > /dev/null 2>&1
  python tools/prof_families.py "$db" ${O}_replayed_families_tail$v.json "bench.py ASR_SIDE_TAIL=$v" > /dev/null 2>&1
  head -1 ${O}_step_sequence_tail$v.txt
done
