#!/bin/bash
mkdir -p gpurun_out
( timeout 2000 python -m pytest -q -m gpu --tb=short -x tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_graph.py tests/test_gpu_baseline_shapes.py -k "emb or win or bn or cfg3 or window" 2>&1 | tail -8 ) > gpurun_out/r4p_pytest.log
cat gpurun_out/r4p_pytest.log
for i in 1 2; do python bench.py --workload librispeech --steps 30 --warmup 6 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('librispeech', d['ms_per_step'])"; done | tee gpurun_out/r4p_ls.txt
