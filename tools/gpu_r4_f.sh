#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest -q -m gpu --tb=short -x tests/test_gpu_graph.py tests/test_gpu_fullsize_properties.py tests/test_gpu_ddp.py "tests/test_gpu_baseline_shapes.py::test_graph_replay_equals_eager_at_dk64_bf16" 2>&1 | tail -30 ) > gpurun_out/r4f_pytest.log
tail -25 gpurun_out/r4f_pytest.log
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-roofline > /dev/null 2>&1
for v in 1 2 1 2; do
ASR_LANES=$v python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline 2> gpurun_out/r4f_bench_$v.err | python -c "import sys,json;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('LANES=$v ms/step',round(d['ms_per_step'],4),'loss',d['config'].get('final_loss'))"
done
tail -3 gpurun_out/r4f_bench_2.err
