#!/bin/bash
# round 3, call P: HBM traffic of the conv implicit-GEMM family (PMC), MFMA-busy of every kernel of the step (PMC), replayed families
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/gpu_pmc_traffic.sh r03 > gpurun_out/r3p_traffic.log 2>&1
tail -8 gpurun_out/r3p_traffic.log
bash tools/gpu_pmc_mfma.sh r03_step > gpurun_out/r3p_mfma.log 2>&1
tail -25 gpurun_out/r3p_mfma.log | cut -c1-160
bash tools/gpu_profile.sh r3p_bench 13 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline
db=$(find /tmp/prof_r3p_bench -name "*.db" | head -1)
python tools/prof_families.py "$db" gpurun_out/r03_replayed_families.json "python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline" | head -50
python tools/prof_sequence.py "$db" gpurun_out/r03_step_sequence.txt
