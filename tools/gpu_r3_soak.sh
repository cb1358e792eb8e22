#!/bin/bash
# round 3 soak: 600 replayed steps of the trainer's step body and of bench.py on the same synthetic batch: the loss must keep falling
export TMPDIR=/tmp
mkdir -p gpurun_out
RATE_BUCKETS=1 timeout 900 python tools/trainer_rate.py 600 2>&1 | tail -4 | cut -c1-300
timeout 900 python bench.py --steps 600 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench --steps 600:', round(d['ms_per_step'],3), 'ms/step; loss', d.get('final_loss', d.get('loss')))"
