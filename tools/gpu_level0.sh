#!/bin/bash
# Level-0 kernels: parity tests, then the microbenchmark at the benchmark shape against asr_hip/libasr_hip_prev.so when that exists
# (built by hand from an older conv_level0.hip) -- one gpurun call; `pmc`: also the LDS / MFMA counters of the level-0 and weight-gradient kernels
# (second and third counter set of tools/gpu_pmc_cmd.sh).  usage: tools/gpu_level0.sh <tag> [reps] [pmc]
tag=${1:-l0}; reps=${2:-3}; pmc=$3
cd ${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p gpurun_out; export TMPDIR=/tmp
L=end2end-asr-pytorch_amd/asr_hip
timeout 900 python -m pytest tests/test_gpu_level0.py tests/test_gpu_ops.py -x -q -k "level0 or conv" 2>&1 | tail -15 > gpurun_out/${tag}_tests.log
{
  echo "== new"; timeout 300 python tools/mb_level0.py $reps 2>&1 | grep -v amdgpu.ids
  if [ -f $L/libasr_hip_prev.so ]; then
    cp $L/libasr_hip.so /tmp/new.so; cp $L/libasr_hip_prev.so $L/libasr_hip.so
    echo "== prev"; timeout 300 python tools/mb_level0.py $reps 2>&1 | grep -v amdgpu.ids
    cp /tmp/new.so $L/libasr_hip.so
    echo "== new"; timeout 300 python tools/mb_level0.py $reps 2>&1 | grep -v amdgpu.ids
  fi
} > gpurun_out/${tag}_mb.txt 2>&1
tail -5 gpurun_out/${tag}_tests.log; cat gpurun_out/${tag}_mb.txt
if [ -n "$pmc" ]; then
  bash tools/gpu_pmc_cmd.sh ${tag} "level0|wgrad_dma" python tools/mb_level0.py 1 > /dev/null 2>&1
  grep -v "^#" gpurun_out/${tag}_pmc.txt | grep "level0\|wgrad\|LDS_BANK\|LDS_IDX\|MFMA_BUSY\|SQ_BUSY_CYCLES\|INSTS_VALU\|INSTS_LDS" | head -60
fi
