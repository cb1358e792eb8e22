#!/bin/bash
# The evidence of a round on the current tree, one gpurun call -> gpurun_out/<rNN>_* (copy what is to be judged into profiles/):
#   whole GPU suite + smoke; kernel trace of the replayed step (per-kernel table, timeline, families, launch sequence); PMC traffic of the
#   conv family (separate FETCH_SIZE / WRITE_SIZE passes) and MFMA-busy counters of every kernel; the bench lines (headline with roofline +
#   cpu_baseline, configs[3], configs[4] bf16 and fp8, the forced one-rank data-parallel reducer in every --ddp-graph mode); train.py's own
#   step body rate.  usage: tools/gpu_evidence.sh r06 [quick]     (quick: skip the suite and the secondary bench lines)
r=${1:-r06}; quick=$2
cd ${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/$r
if [ -z "$quick" ]; then
  ( timeout 2400 python -m pytest -q -m gpu --tb=short tests 2>&1 | tail -25 ) > ${O}_pytest_gpu.log; tail -3 ${O}_pytest_gpu.log
  python __graft_entry__.py smoke 2>&1 | grep -i smoke > ${O}_smoke.log; cat ${O}_smoke.log
  [ -f gpurun_out/parity_r06.json ] && cp gpurun_out/parity_r06.json ${O}_parity_baseline_shapes.json
fi
cmd="python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --soak-seconds 0"
out=/tmp/prof_$r; rm -rf $out
( timeout 900 rocprofv3 --kernel-trace --stats -d $out -o trace -- $cmd ) > ${O}_prof.log 2>&1
db=$(find $out -name "*.db" | head -1)
python tools/prof_summary.py "$db" 11 "rocprofv3 --kernel-trace --stats -- $cmd (eager warm-up + capture + replays)" > ${O}_bench_kernel_stats.txt 2>&1
python tools/prof_timeline.py "$db" "timeline of the last 3 replayed steps: rocprofv3 --kernel-trace -- $cmd" > ${O}_bench_timeline.txt 2>&1
python tools/prof_families.py "$db" ${O}_replayed_families.json "$cmd" > /dev/null 2>&1
python tools/prof_sequence.py "$db" ${O}_step_sequence.txt > /dev/null 2>&1
head -1 ${O}_step_sequence.txt
bash tools/gpu_pmc_traffic.sh $r > /dev/null 2>&1
mkdir -p profiles; cp ${O}_replayed_families.json ${O}_roofline_traffic.json profiles/ 2>/dev/null      # bench.py reads them from profiles/
python bench.py > ${O}_bench_line.json 2> ${O}_bench.err
if [ -z "$quick" ]; then
  python bench.py --workload librispeech --steps 20 --warmup 5 --soak-seconds 0 > ${O}_bench_line_librispeech.json 2>> ${O}_bench.err
  python bench.py --workload lowrank --steps 20 --warmup 5 --soak-seconds 0 > ${O}_bench_line_lowrank.json 2>> ${O}_bench.err
  python bench.py --workload lowrank --precision fp8 --steps 20 --warmup 5 --soak-seconds 0 > ${O}_bench_line_lowrank_fp8.json 2>> ${O}_bench.err
  B="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline --soak-seconds 0"
  for m in four one auto; do ASR_FORCE_DDP=1 $B --ddp-graph $m > ${O}_bench_line_ddp1_forced_$m.json 2>> ${O}_bench.err; done
  ASR_FORCE_DDP=1 $B --grad-wire bf16 > ${O}_bench_line_ddp1_forced_bf16wire.json 2>> ${O}_bench.err
  bash tools/gpu_pmc_mfma.sh ${r}_step > /dev/null 2>&1
  { echo "# train.py's step body (trainer/asr/trainer.py:_run_batch through the prefetcher) on configs[1], B = 32, 300 steps"
    python tools/trainer_rate.py 300 2>&1 | grep -v amdgpu.ids | tail -4; } > ${O}_trainer_rate.txt 2>&1
fi
for f in ${O}_bench_line*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d.get("roofline") or {}
    print(sys.argv[1], "ms/step %.3f value %.0f frac %s by-time %s mode %s" % (d["ms_per_step"], d["value"], r.get("frac"), r.get("frac_by_time_largest_family"), (d["config"].get("ddp_graph") or {}).get("ran")))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
