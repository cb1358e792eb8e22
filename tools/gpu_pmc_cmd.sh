#!/bin/bash
# PMC counters (counter-only passes) of an arbitrary command: tools/gpu_pmc_cmd.sh <tag> <kernel-name filter> <command...>
tag=$1; flt=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
: > $root/gpurun_out/${tag}_pmc.txt
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_SALU" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM"; do
  i=$((i+1)); out=/tmp/pmc_${tag}_$i; rm -rf $out
  ( cd $root && timeout 600 rocprofv3 --pmc $set -d $out -o pmc -- $* ) > $root/gpurun_out/${tag}_pmc_log$i.txt 2>&1
  db=$(find $out -name "*.db" | head -1)
  echo "# rocprofv3 --pmc $set -- $*" >> $root/gpurun_out/${tag}_pmc.txt
  if [ -n "$db" ]; then python $root/tools/pmc_summary.py "$db" "$flt" >> $root/gpurun_out/${tag}_pmc.txt 2>&1; else echo "no database" >> $root/gpurun_out/${tag}_pmc.txt; tail -5 $root/gpurun_out/${tag}_pmc_log$i.txt >> $root/gpurun_out/${tag}_pmc.txt; fi
done
cat $root/gpurun_out/${tag}_pmc.txt
