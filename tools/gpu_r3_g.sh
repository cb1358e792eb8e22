#!/bin/bash
# round 3, call G: whole GPU suite (incl. dec_d128, grouped dW op test), default bench, bench with 4 stages
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -40 > gpurun_out/r3g_pytest.txt
tail -15 gpurun_out/r3g_pytest.txt
timeout 300 python bench.py > gpurun_out/r3g_bench.txt 2>gpurun_out/r3g_bench.err; cut -c1-400 gpurun_out/r3g_bench.txt
ASR_TN_GROUP_STAGES=4 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; print('NST4 mrows3200:', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; print('NST3 mrows3200:', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
