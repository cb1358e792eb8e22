#!/bin/bash
# rocprofv3 kernel trace of the attention micro-benchmark at the north-star shape -> per-kernel durations
tag=${1:-attnp}
export TMPDIR=/tmp
out=/tmp/prof_${tag}; rm -rf $out
cat > /tmp/ab.py <<'PY'
import sys, os
sys.path.insert(0, "tools")
import microbench as M
M.attn([(32, 8, 800, 800, 64, False, 0.0), (32, 8, 800, 800, 64, False, 0.1), (32, 8, 200, 200, 64, False, 0.1)])
PY
( cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --kernel-trace --stats -d $out -o trace -- python /tmp/ab.py ) > gpurun_out/${tag}_prof.log 2>&1
db=$(find $out -name "*.db" | head -1)
python tools/prof_summary.py "$db" 1 "rocprofv3 --kernel-trace: tools/microbench.py attn at (B=32,H=8,T=800,d=64) p=0 / p=0.1 and T=200" > gpurun_out/${tag}_kernel_stats.txt 2>&1
head -14 gpurun_out/${tag}_kernel_stats.txt | cut -c1-160
