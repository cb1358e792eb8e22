#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 1700 python -m pytest -q -m gpu --tb=short tests 2>&1 | tail -40 ) > gpurun_out/r4e_pytest.log
tail -15 gpurun_out/r4e_pytest.log
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-roofline > /dev/null 2>&1
for v in 0 1 1; do
ASR_LEVEL0=$v python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline 2> gpurun_out/r4e_bench_$v.err | python -c "import sys,json;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('LEVEL0=$v ms/step',round(d['ms_per_step'],4),'loss',d['config'].get('final_loss'))"
done
