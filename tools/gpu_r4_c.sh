#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest -q -m gpu --tb=short -s tests/test_gpu_level0.py tests/test_gpu_model.py tests/test_gpu_graph.py tests/test_gpu_fullsize_properties.py 2>&1 | tail -60 ) > gpurun_out/r4c_pytest.log
tail -30 gpurun_out/r4c_pytest.log
( timeout 900 python -m pytest -q -m gpu --tb=short -s tests/test_gpu_baseline_shapes.py -k "bf16 and (cfg1_b32 or cfg0)" 2>&1 | tail -30 ) > gpurun_out/r4c_pytest2.log
tail -12 gpurun_out/r4c_pytest2.log
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-roofline > /dev/null 2>&1
for v in 0 1 0 1; do
ASR_LEVEL0=$v python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline 2> gpurun_out/r4c_bench_$v.err | python -c "import sys,json;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('LEVEL0=$v ms/step',round(d['ms_per_step'],4),'loss',d['config'].get('final_loss'))"
done
