#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest -q -m gpu --tb=short -x tests/test_gpu_level0.py 2>&1 | tail -15 ) > gpurun_out/r4d_pytest.log
tail -8 gpurun_out/r4d_pytest.log
bash tools/gpu_profile.sh r4d 11 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline
grep -E "level0|conv3x3|conv1|pool_bwd|total kernel" gpurun_out/r4d_kernel_stats.txt
for v in 0 1 1; do
ASR_LEVEL0=$v python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline 2> gpurun_out/r4c_bench_$v.err | python -c "import sys,json;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('LEVEL0=$v ms/step',round(d['ms_per_step'],4),'loss',d['config'].get('final_loss'))"
done
