#!/bin/bash
# round 3, call V: what the driver runs -- smoke(), default bench (with roofline + cpu baseline), librispeech / lowrank lines
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 900 python bench.py > gpurun_out/r3v_bench.txt 2> gpurun_out/r3v_bench.err; tail -1 gpurun_out/r3v_bench.txt | cut -c1-600; tail -3 gpurun_out/r3v_bench.err
timeout 900 python bench.py --workload librispeech --no-cpu-baseline > gpurun_out/r3v_bench_librispeech.txt 2>/dev/null; tail -1 gpurun_out/r3v_bench_librispeech.txt | cut -c1-400
timeout 900 python bench.py --workload lowrank --no-cpu-baseline > gpurun_out/r3v_bench_lowrank.txt 2>/dev/null; tail -1 gpurun_out/r3v_bench_lowrank.txt | cut -c1-400
ASR_FORCE_DDP=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c1-500
