#!/bin/bash
# L2 hit rate of the grouped weight-gradient kernel: host-scheduled equal pieces (default) vs one workgroup per (block, slice) (in phase)
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/r03_tn_group_l2.txt
for v in "ASR_TN_GROUP_TILE=0" "ASR_TN_GROUP_TILE=256"; do
  d=/tmp/pmc_l2_${v##*=}; rm -rf $d
  env $v timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE -d $d -o pmc -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > /tmp/l2_log.txt 2>&1
  db=$(find $d -name "*.db" | head -1)
  echo "# $v: rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE -- python bench.py --steps 4 --warmup 2" >> gpurun_out/r03_tn_group_l2.txt
  [ -n "$db" ] && python tools/pmc_summary.py "$db" gemm_tn256 >> gpurun_out/r03_tn_group_l2.txt 2>&1 || tail -3 /tmp/l2_log.txt >> gpurun_out/r03_tn_group_l2.txt
done
cat gpurun_out/r03_tn_group_l2.txt
