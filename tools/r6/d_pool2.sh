#!/bin/bash
# round 6, call D: pool hand-over with the contiguous-store epilogue; --ddp-graph auto verification numbers; A/B; sequence
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r6d
( timeout 900 python -m pytest -q -m gpu --tb=short tests/test_gpu_pool_handover.py tests/test_gpu_frontend_exact.py tests/test_gpu_graph.py 2>&1 | tail -40 ) > ${O}_new_tests.log
tail -25 ${O}_new_tests.log
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
out=/tmp/prof_r6d; rm -rf $out
( timeout 600 rocprofv3 --kernel-trace --stats -d $out -o trace -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --soak-seconds 0 ) > ${O}_prof.log 2>&1
db=$(find $out -name "*.db" | head -1)
python tools/prof_sequence.py "$db" ${O}_step_sequence.txt > /dev/null 2>&1
head -1 ${O}_step_sequence.txt; grep -n "pool_bwd\|permute_cols\|gemm_big_nn_kernel<2>" ${O}_step_sequence.txt | tail -3 | cut -c1-150
ASR_FORCE_DDP=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --soak-seconds 0 --ddp-graph auto 2>${O}_ddp_auto.err | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['launch_mode'], json.dumps(d['config']['ddp_graph']))" | tee ${O}_ddp_auto.txt
