#!/bin/bash
# round 4: the committed evidence files of the headline step on the current tree -> gpurun_out/r04_* (copied to profiles/ by hand)
mkdir -p gpurun_out; export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
cmd="python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline"
out=/tmp/prof_r04; rm -rf $out
( cd $root && timeout 900 rocprofv3 --kernel-trace --stats -d $out -o trace -- $cmd ) > gpurun_out/r04_prof.log 2>&1
db=$(find $out -name "*.db" | head -1)
python tools/prof_summary.py "$db" 11 "rocprofv3 --kernel-trace --stats -- $cmd (11 steps in the trace: eager warm-up + capture + replays)" > gpurun_out/r04_bench_kernel_stats.txt 2>&1
python tools/prof_timeline.py "$db" "timeline of the last 3 replayed steps: rocprofv3 --kernel-trace -- $cmd" > gpurun_out/r04_bench_timeline.txt 2>&1
python tools/prof_families.py "$db" gpurun_out/r04_replayed_families.json "$cmd" > /dev/null 2>&1
python tools/prof_sequence.py "$db" gpurun_out/r04_step_sequence.txt > /dev/null 2>&1
cmd2="python bench.py --workload librispeech --steps 5 --warmup 3 --no-cpu-baseline --no-roofline"
out2=/tmp/prof_r04_ls; rm -rf $out2
( cd $root && timeout 900 rocprofv3 --kernel-trace -d $out2 -o trace -- $cmd2 ) > gpurun_out/r04_prof_ls.log 2>&1
db2=$(find $out2 -name "*.db" | head -1)
python tools/prof_timeline.py "$db2" "timeline of the last 3 replayed steps: rocprofv3 --kernel-trace -- $cmd2" > gpurun_out/r04_librispeech_timeline.txt 2>&1
bash tools/gpu_pmc_traffic.sh r04 > /dev/null 2>&1
bash tools/gpu_pmc_mfma.sh r04_step > /dev/null 2>&1
head -30 gpurun_out/r04_bench_kernel_stats.txt; tail -12 gpurun_out/r04_step_mfma_pmc.txt; python -c "
import json;d=json.load(open('gpurun_out/r04_roofline_traffic.json'));print(d['hbm_bytes_per_step'], d['traffic_bytes_per_launch_avg']);[print(k,v.get('bound'),round(v['hbm_bytes_per_launch']/1e6)) for k,v in d['per_kernel'].items()]"
