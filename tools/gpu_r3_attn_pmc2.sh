#!/bin/bash
# rocprofv3 --pmc passes (counters only) over the attention micro-benchmark at the north-star shape, p = 0 and p = 0.1
export TMPDIR=/tmp
cat > /tmp/ab.py <<'PY'
import sys
sys.path.insert(0, "tools")
import microbench as M
M.attn([(32, 8, 800, 800, 64, False, 0.0), (32, 8, 800, 800, 64, False, 0.1)])
PY
tag=r03_attention_d64_pmc_final
: > gpurun_out/${tag}.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU"; do
  i=$((i+1)); out=/tmp/pmc_${tag}_$i; rm -rf $out
  ( cd $GRAFT_REPO_ROOT && timeout 300 rocprofv3 --pmc $set -d $out -o pmc -- python /tmp/ab.py ) > gpurun_out/${tag}_log$i.txt 2>&1
  db=$(find $out -name "*.db" | head -1)
  [ -n "$db" ] && python tools/pmc_summary.py "$db" attn_ >> gpurun_out/${tag}.txt 2>&1 || tail -5 gpurun_out/${tag}_log$i.txt
done
grep -v "^$" gpurun_out/${tag}.txt | head -60
