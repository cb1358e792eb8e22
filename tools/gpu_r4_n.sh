#!/bin/bash
mkdir -p gpurun_out
L=end2end-asr-pytorch_amd/asr_hip
( timeout 900 python -m pytest -q -m gpu --tb=short -x tests/test_gpu_level0.py 2>&1 | tail -4 ) > gpurun_out/r4n_pytest.log
cat gpurun_out/r4n_pytest.log
{
  echo "== new"; python tools/mb_level0.py 3 2>&1 | grep 'level 0'
  cp $L/libasr_hip.so /tmp/new.so; cp $L/libasr_hip_prev.so $L/libasr_hip.so
  echo "== prev"; python tools/mb_level0.py 3 2>&1 | grep 'level 0'
  for i in 1 2; do python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('headline prev', d['ms_per_step'])"; done
  cp /tmp/new.so $L/libasr_hip.so
  for i in 1 2; do python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('headline new', d['ms_per_step'])"; done
} > gpurun_out/r4n_ab.txt 2>&1
cat gpurun_out/r4n_ab.txt
