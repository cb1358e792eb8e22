#!/bin/bash
# round 3, final evidence (part A): whole GPU suite, smoke, default bench line, kernel trace of the replayed step
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r3final_pytest.txt; tail -3 gpurun_out/r3final_pytest.txt | cut -c1-200
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r03_bench_line.json 2>/dev/null; cut -c1-300 gpurun_out/r03_bench_line.json
bash tools/gpu_profile.sh r03final_bench 13 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1
db=$(find /tmp/prof_r03final_bench -name "*.db" | head -1)
python tools/prof_families.py "$db" gpurun_out/r03_replayed_families.json "python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline" > /dev/null
python tools/prof_sequence.py "$db" gpurun_out/r03_step_sequence.txt
head -12 gpurun_out/r03final_bench_timeline.txt | cut -c1-150
