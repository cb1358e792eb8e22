#!/bin/bash
# HBM traffic of the conv implicit-GEMM family: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (no trace
# domains) over three eager steps of the headline workload -> gpurun_out/<tag>_traffic_pmc.txt + <tag>_roofline_traffic.json
tag=${1:-pmc}
root=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cmd="python bench.py --steps 2 --warmup 1 --eager --no-cpu-baseline --no-roofline --soak-seconds 0"
: > $root/gpurun_out/${tag}_traffic_pmc.txt
for ctr in FETCH_SIZE WRITE_SIZE; do
  out=/tmp/pmc_${tag}_${ctr}
  rm -rf $out
  ( cd $root && timeout 600 rocprofv3 --pmc $ctr -d $out -o pmc -- $cmd ) > $root/gpurun_out/${tag}_pmc_${ctr}.log 2>&1
  db=$(find $out -name "*.db" | head -1)
  echo "# rocprofv3 --pmc $ctr -- $cmd" >> $root/gpurun_out/${tag}_traffic_pmc.txt
  if [ -n "$db" ]; then python $root/tools/pmc_summary.py "$db" "conv3x3|vgg_level0" >> $root/gpurun_out/${tag}_traffic_pmc.txt 2>&1; else echo "no database" >> $root/gpurun_out/${tag}_traffic_pmc.txt; fi
done
python $root/tools/make_traffic_json.py $root/gpurun_out/${tag}_traffic_pmc.txt $root/gpurun_out/${tag}_replayed_families.json > $root/gpurun_out/${tag}_roofline_traffic.json
cat $root/gpurun_out/${tag}_roofline_traffic.json | tail -5
