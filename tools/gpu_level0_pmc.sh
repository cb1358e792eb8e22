#!/bin/bash
# Level-0 + weight-gradient kernels: parity, microbenchmark (new / prev library), then the LDS / MFMA counters of the level-0 kernels at the
# benchmark shape (second and third counter set of tools/gpu_pmc_cmd.sh).  usage: tools/gpu_level0_pmc.sh <tag>
tag=${1:-l0pmc}
cd ${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/gpu_level0.sh ${tag} 3
bash tools/gpu_pmc_cmd.sh ${tag} "level0|wgrad_dma" python tools/mb_level0.py 1 > /dev/null 2>&1
grep -v "^#" gpurun_out/${tag}_pmc.txt | grep "level0\|wgrad\|LDS_BANK\|LDS_IDX\|MFMA_BUSY\|SQ_BUSY_CYCLES\|INSTS_VALU\|INSTS_LDS" | head -60
