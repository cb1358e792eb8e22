#!/bin/bash
# round 6, call B: side-tail variants (Adam behind the pooling backward = 4; pooling backward beside the grouped weight gradient = 16; both = 20),
# hyp_seq from the loss kernel; graph tests per mode; same-call step A/B; kernel sequence per mode
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r6b
for v in 0 4 16 20; do
  ( ASR_SIDE_TAIL=$v timeout 600 python -m pytest -q -m gpu --tb=short tests/test_gpu_graph.py 2>&1 | tail -5 ) > ${O}_graph_tests_tail$v.log
  echo "tail=$v: $(tail -1 ${O}_graph_tests_tail$v.log)"
done
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --soak-seconds 0"
rm -f ${O}_step_ab.txt
for rep in 1 2 3; do
  for v in 0 4 16 20; do
    echo -n "ASR_SIDE_TAIL=$v " | tee -a ${O}_step_ab.txt
    ASR_SIDE_TAIL=$v timeout 300 $B 2>/dev/null | python -c "
import json,sys
l=sys.stdin.readlines()
d=json.loads(l[-1]) if l else {}
print(d.get('ms_per_step'), (d.get('config') or {}).get('final_loss'))" | tee -a ${O}_step_ab.txt
  done
done
for v in 4 16 20; do
  out=/tmp/prof_r6b_$v; rm -rf $out
  ( ASR_SIDE_TAIL=$v timeout 600 rocprofv3 --kernel-trace --stats -d $out -o trace -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --soak-seconds 0 ) > ${O}_prof$v.log 2>&1
  db=$(find $out -name "*.db" | head -1)
  python tools/prof_sequence.py "$db" ${O}_step_sequence_tail$v.txt > /dev/null 2>&1
  head -1 ${O}_step_sequence_tail$v.txt
done
( timeout 900 python -m pytest -q -m gpu --tb=short tests/test_gpu_model.py tests/test_gpu_train_cli.py tests/test_gpu_ddp.py 2>&1 | tail -5 ) > ${O}_pytest_subset.log
tail -2 ${O}_pytest_subset.log
