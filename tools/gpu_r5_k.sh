#!/bin/bash
# round 5 (k): PMC passes of the grouped weight-gradient harness (old: gemm_tn256s, new: gemm_tn256g<3, true>)
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
tag=r5k
: > gpurun_out/${tag}_pmc.txt
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_SALU" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1)); out=/tmp/pmc_${tag}_$i; rm -rf $out
  ( timeout 300 rocprofv3 --pmc $set -d $out -o pmc -- tools/bin/tn_grouped_test ) > gpurun_out/${tag}_pmc_log$i.txt 2>&1
  db=$(find $out -name "*.db" | head -1)
  echo "# rocprofv3 --pmc $set -- tools/bin/tn_grouped_test" >> gpurun_out/${tag}_pmc.txt
  if [ -n "$db" ]; then python tools/pmc_summary.py "$db" "tn256" >> gpurun_out/${tag}_pmc.txt 2>&1; else echo "no database" >> gpurun_out/${tag}_pmc.txt; tail -3 gpurun_out/${tag}_pmc_log$i.txt >> gpurun_out/${tag}_pmc.txt; fi
done
cat gpurun_out/${tag}_pmc.txt
