#!/bin/bash
# round 5: launch-by-launch sequence + kernel families of one replayed step on the current tree -> gpurun_out/<tag>_step_sequence.txt
tag=${1:-r5}; shift
mkdir -p gpurun_out; export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
cmd="python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline $*"
out=/tmp/prof_$tag; rm -rf $out
( cd $root && timeout 900 rocprofv3 --kernel-trace --stats -d $out -o trace -- $cmd ) > gpurun_out/${tag}_prof.log 2>&1
db=$(find $out -name "*.db" | head -1)
python tools/prof_sequence.py "$db" gpurun_out/${tag}_step_sequence.txt > /dev/null 2>&1
python tools/prof_families.py "$db" gpurun_out/${tag}_replayed_families.json "$cmd" > /dev/null 2>&1
grep -E "conv|pool|level0|wgrad" gpurun_out/${tag}_step_sequence.txt | cut -c1-150
head -1 gpurun_out/${tag}_step_sequence.txt
