#!/bin/bash
# grouped weight gradients, in-phase pieces (TN_GROUP_TILE=2) vs host-scheduled equal pieces (default)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "grouped" 2>&1 | tail -2
b() { timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))"; }
echo "warm-up run (ignore): $(b)"
for rep in 1 2; do
echo "default (equal pieces): $(b)"
echo "in-phase pieces of 25 stages: $(ASR_TN_GROUP_TILE=2 b)"
echo "in-phase pieces of 34 stages: $(ASR_TN_GROUP_TILE=2 ASR_TN_GROUP_PIECE=34 b)"
echo "in-phase pieces of 50 stages: $(ASR_TN_GROUP_TILE=2 ASR_TN_GROUP_PIECE=50 b)"
echo "in-phase pieces of 17 stages: $(ASR_TN_GROUP_TILE=2 ASR_TN_GROUP_PIECE=17 b)"
done
d=/tmp/pmc_l2p; rm -rf $d
ASR_TN_GROUP_TILE=2 timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE -d $d -o pmc -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > /tmp/l2_log.txt 2>&1
db=$(find $d -name "*.db" | head -1)
[ -n "$db" ] && python tools/pmc_summary.py "$db" gemm_tn256
