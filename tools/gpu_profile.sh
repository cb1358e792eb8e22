#!/bin/bash
# rocprofv3 kernel trace of a command -> gpurun_out/<tag>_kernel_stats.txt (per-kernel table via tools/prof_summary.py).
# usage: tools/gpu_profile.sh <tag> <steps in the trace> "<command>"
tag=$1; steps=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
out=/tmp/prof_${tag}
rm -rf $out
( cd $root && timeout 900 rocprofv3 --kernel-trace --stats -d $out -o trace -- $* ) > $root/gpurun_out/${tag}_prof.log 2>&1
db=$(find $out -name "*.db" | head -1)
if [ -n "$db" ]; then
  python $root/tools/prof_summary.py "$db" $steps "rocprofv3 --kernel-trace --stats -- $* ($steps steps in the trace: eager warm-up + capture + replays)" > $root/gpurun_out/${tag}_kernel_stats.txt 2>&1
  python $root/tools/prof_timeline.py "$db" "timeline of the last 3 replayed steps: rocprofv3 --kernel-trace -- $*" > $root/gpurun_out/${tag}_timeline.txt 2>&1
else
  echo "no rocpd database produced" > $root/gpurun_out/${tag}_kernel_stats.txt; ls -R $out | head -30 >> $root/gpurun_out/${tag}_kernel_stats.txt
fi
