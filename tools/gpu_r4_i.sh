#!/bin/bash
# attention backward trims (hoisted staging / fragment offsets, running-sum dropout hash, rescale out of the loop): tests + A/B against
# the previous attention_fast.hip (asr_hip/libasr_hip_prev.so, built by hand from git HEAD), same box
mkdir -p gpurun_out
L=end2end-asr-pytorch_amd/asr_hip
( timeout 1500 python -m pytest -q -m gpu --tb=short -x tests/test_gpu_ops.py tests/test_gpu_dropout_stats.py -k "attention or attn or dropout" 2>&1 | tail -8 ) > gpurun_out/r4i_pytest.log
cat gpurun_out/r4i_pytest.log
cat > /tmp/ab.py <<'PY'
import sys
sys.path.insert(0, "tools")
import microbench as M
M.attn([(32, 8, 800, 800, 64, False, 0.0), (32, 8, 800, 800, 64, False, 0.1), (32, 8, 200, 200, 64, False, 0.1), (32, 8, 100, 200, 64, False, 0.1),
        (32, 8, 100, 100, 64, True, 0.1), (16, 8, 795, 795, 64, False, 0.1)])
PY
{
  echo "== new"; python /tmp/ab.py; python /tmp/ab.py
  cp $L/libasr_hip.so /tmp/new.so; cp $L/libasr_hip_prev.so $L/libasr_hip.so
  echo "== previous attention_fast.hip"; python /tmp/ab.py; python /tmp/ab.py
  cp /tmp/new.so $L/libasr_hip.so
  echo "== new again"; python /tmp/ab.py
} > gpurun_out/r4i_attn_ab.txt 2>&1
cat gpurun_out/r4i_attn_ab.txt
{
  for i in 1 2; do python bench.py --workload librispeech --steps 30 --warmup 6 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('librispeech new', d['ms_per_step'])"; done
  cp $L/libasr_hip_prev.so $L/libasr_hip.so
  for i in 1 2; do python bench.py --workload librispeech --steps 30 --warmup 6 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('librispeech prev', d['ms_per_step'])"; done
  cp /tmp/new.so $L/libasr_hip.so
  for i in 1 2; do python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('headline new', d['ms_per_step'])"; done
} > gpurun_out/r4i_step_ab.txt 2>&1
cat gpurun_out/r4i_step_ab.txt
