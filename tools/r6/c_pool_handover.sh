#!/bin/bash
# round 6, call C: the pooling backward in the input projection's data-gradient epilogue (asr_gemm_nn_poolbwd) + the other new tests
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r6c
( timeout 900 python -m pytest -q -m gpu --tb=short -x tests/test_gpu_pool_handover.py tests/test_gpu_frontend_exact.py 2>&1 | tail -30 ) > ${O}_new_tests.log
tail -15 ${O}_new_tests.log
( timeout 900 python -m pytest -q -m gpu --tb=short tests/test_gpu_model.py tests/test_gpu_graph.py tests/test_gpu_conv_ws.py tests/test_gpu_level0.py 2>&1 | tail -15 ) > ${O}_subset.log
tail -6 ${O}_subset.log
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --soak-seconds 0"
rm -f ${O}_step_ab.txt
for rep in 1 2 3; do
  for v in 0 1; do
    echo -n "ASR_POOL_HANDOVER=$v " | tee -a ${O}_step_ab.txt
    ASR_POOL_HANDOVER=$v timeout 300 $B 2>/dev/null | python -c "
import json,sys
l=sys.stdin.readlines()
d=json.loads(l[-1]) if l else {}
print(d.get('ms_per_step'), (d.get('config') or {}).get('final_loss'))" | tee -a ${O}_step_ab.txt
  done
done
out=/tmp/prof_r6c; rm -rf $out
( timeout 600 rocprofv3 --kernel-trace --stats -d $out -o trace -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --soak-seconds 0 ) > ${O}_prof.log 2>&1
db=$(find $out -name "*.db" | head -1)
python tools/prof_sequence.py "$db" ${O}_step_sequence.txt > /dev/null 2>&1
python tools/prof_families.py "$db" ${O}_replayed_families.json "bench.py" > /dev/null 2>&1
head -1 ${O}_step_sequence.txt; grep -n "pool_bwd\|permute_cols\|gemm_big_nn_kernel<2>" ${O}_step_sequence.txt | tail -4 | cut -c1-150
