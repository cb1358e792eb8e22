#!/bin/bash
# rocprofv3 kernel trace of the graph-replayed greedy decode (32 x 300 tokens) -> gpurun_out/<tag>_decode_trace.txt
tag=${1:-dec}
root=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
out=/tmp/prof_${tag}
rm -rf $out
( cd $root && MICRO_DECODE_GRAPH_ONLY=1 timeout 600 rocprofv3 --kernel-trace -d $out -o trace -- python tools/microbench.py decode ) > $root/gpurun_out/${tag}_decode_prof.log 2>&1
db=$(find $out -name "*.db" | head -1)
if [ -n "$db" ]; then
  python $root/tools/prof_decode.py "$db" "rocprofv3 --kernel-trace -- python tools/microbench.py decode (graph replay only)" > $root/gpurun_out/${tag}_decode_trace.txt 2>&1
fi
( cd $root && MICRO_DECODE_GRAPH_ONLY=1 python tools/microbench.py decode ) >> $root/gpurun_out/${tag}_decode_trace.txt 2>&1
