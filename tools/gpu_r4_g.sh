#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest -q -m gpu --tb=short -x tests/test_gpu_ddp.py tests/test_gpu_graph.py 2>&1 | tail -30 ) > gpurun_out/r4g_pytest.log
tail -12 gpurun_out/r4g_pytest.log
for w in fp32 bf16; do
ASR_FORCE_DDP=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 30 --warmup 6 --no-cpu-baseline --no-roofline --grad-wire $w 2> gpurun_out/r4g_bench_$w.err | python -c "import sys,json;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('ddp1 wire $w ms/step',round(d['ms_per_step'],4),d['config'].get('collective_library'),d['config'].get('gradient_allreduce'))"
done
