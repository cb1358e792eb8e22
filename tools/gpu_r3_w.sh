#!/bin/bash
# round 3, call W: deferred grouped weight gradients inside the four-graph data-parallel step
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ddp.py tests/test_gpu_graph.py -m gpu -q -x 2>&1 | tail -4
for v in "ASR_FORCE_DDP=1" "ASR_FORCE_DDP=1 ASR_DEFER_WGRAD=0"; do
echo "$v: $(env $v timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 40 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d.get('launch_mode'))")"
done
echo "single graph: $(timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],3))")"
