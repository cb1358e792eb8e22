#!/bin/bash
# rocprofv3 --pmc passes (counters only: no trace domains) over the attention micro-benchmark at the north-star shape
tag=${1:-attnpmc}
export TMPDIR=/tmp
cat > /tmp/ab.py <<'PY'
import sys, os
sys.path.insert(0, "tools")
import microbench as M
M.attn([(32, 8, 800, 800, 64, False, 0.0), (32, 8, 800, 800, 64, False, 0.1)])
PY
: > gpurun_out/${tag}.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM"; do
  i=$((i+1)); out=/tmp/pmc_${tag}_$i; rm -rf $out
  ( cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --pmc $set -d $out -o pmc -- python /tmp/ab.py ) > gpurun_out/${tag}_log$i.txt 2>&1
  db=$(find $out -name "*.db" | head -1)
  [ -n "$db" ] && python tools/pmc_summary.py "$db" attn_ >> gpurun_out/${tag}.txt 2>&1 || tail -5 gpurun_out/${tag}_log$i.txt
done
cat gpurun_out/${tag}.txt
