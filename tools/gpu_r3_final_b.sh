#!/bin/bash
# round 3, final evidence (part B): configs[3] / configs[4] / forced-reducer lines, kernel trace of configs[3]
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python bench.py --workload librispeech > gpurun_out/r03_bench_line_librispeech.json 2>/dev/null; cut -c1-200 gpurun_out/r03_bench_line_librispeech.json
timeout 900 python bench.py --workload lowrank --no-cpu-baseline > gpurun_out/r03_bench_line_lowrank.json 2>/dev/null; cut -c1-200 gpurun_out/r03_bench_line_lowrank.json
ASR_FORCE_DDP=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 40 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > gpurun_out/r03_bench_line_ddp1_forced.json; cut -c1-200 gpurun_out/r03_bench_line_ddp1_forced.json
bash tools/gpu_profile.sh r03_librispeech 8 python bench.py --workload librispeech --steps 5 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1
head -16 gpurun_out/r03_librispeech_timeline.txt | cut -c1-150
