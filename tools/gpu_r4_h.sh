#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 1700 python -m pytest -q -m gpu --tb=short tests 2>&1 | tail -25 ) > gpurun_out/r4h_pytest.log
tail -8 gpurun_out/r4h_pytest.log
python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('ms/step',round(d['ms_per_step'],4))"
python bench.py --steps 20 --warmup 5 --workload librispeech --no-cpu-baseline 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('librispeech ms/step',round(d['ms_per_step'],4))"
