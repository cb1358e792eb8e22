#!/bin/bash
# round 5, evidence on the current tree: the whole GPU suite, the bench lines (headline with roofline + cpu_baseline, librispeech, lowrank,
# forced one-rank data-parallel reducer: four graphs / one graph / bf16 wire), kernel trace + families + sequence of the replayed step,
# PMC traffic of the conv family, MFMA-busy counters of every kernel, trainer rate of train.py's default path.  -> gpurun_out/r05_*
mkdir -p gpurun_out; export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
( timeout 2400 python -m pytest -q -m gpu --tb=short tests 2>&1 | tail -25 ) > gpurun_out/r05_pytest_gpu.log
tail -5 gpurun_out/r05_pytest_gpu.log
# the kernel trace first: bench.py reads its families file (copied to profiles/ for the committed run)
cmd="python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --soak-seconds 0"
out=/tmp/prof_r05; rm -rf $out
( cd $root && timeout 900 rocprofv3 --kernel-trace --stats -d $out -o trace -- $cmd ) > gpurun_out/r05_prof.log 2>&1
db=$(find $out -name "*.db" | head -1)
python tools/prof_summary.py "$db" 11 "rocprofv3 --kernel-trace --stats -- $cmd (11 steps in the trace: eager warm-up + capture + replays)" > gpurun_out/r05_bench_kernel_stats.txt 2>&1
python tools/prof_timeline.py "$db" "timeline of the last 3 replayed steps: rocprofv3 --kernel-trace -- $cmd" > gpurun_out/r05_bench_timeline.txt 2>&1
python tools/prof_families.py "$db" gpurun_out/r05_replayed_families.json "$cmd" > /dev/null 2>&1
python tools/prof_sequence.py "$db" gpurun_out/r05_step_sequence.txt > /dev/null 2>&1
bash tools/gpu_pmc_traffic.sh r05 > /dev/null 2>&1
cp gpurun_out/r05_replayed_families.json gpurun_out/r05_roofline_traffic.json profiles/ 2>/dev/null
python bench.py > gpurun_out/r05_bench_line.json 2> gpurun_out/r05_bench.err
python bench.py --workload librispeech --steps 20 --warmup 5 --soak-seconds 0 > gpurun_out/r05_bench_line_librispeech.json 2>> gpurun_out/r05_bench.err
python bench.py --workload lowrank --steps 20 --warmup 5 --soak-seconds 0 > gpurun_out/r05_bench_line_lowrank.json 2>> gpurun_out/r05_bench.err
B="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline --soak-seconds 0"
ASR_FORCE_DDP=1 $B > gpurun_out/r05_bench_line_ddp1_forced.json 2>> gpurun_out/r05_bench.err
ASR_FORCE_DDP=1 ASR_DDP_ONE_GRAPH=1 $B > gpurun_out/r05_bench_line_ddp1_forced_one_graph.json 2>> gpurun_out/r05_bench.err
ASR_FORCE_DDP=1 $B --grad-wire bf16 > gpurun_out/r05_bench_line_ddp1_forced_bf16wire.json 2>> gpurun_out/r05_bench.err
for f in gpurun_out/r05_bench_line*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d.get("roofline") or {}
    print(sys.argv[1], "ms/step %.3f value %.0f frac %s soak %s" % (d["ms_per_step"], d["value"], r.get("frac"), (d["config"].get("soak") or {}).get("seconds")))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
bash tools/gpu_pmc_mfma.sh r05_step > /dev/null 2>&1
{
  echo "# train.py's step body (trainer/asr/trainer.py:_run_batch through the prefetcher) on configs[1], B = 32, 300 steps"
  python tools/trainer_rate.py 300 2>&1 | grep -v amdgpu.ids | tail -4
  echo "# the same with --graph-buckets 0 (eager launches)"
  RATE_BUCKETS=0 python tools/trainer_rate.py 300 2>&1 | grep -v amdgpu.ids | tail -4
} > gpurun_out/r05_trainer_rate.txt 2>&1
head -30 gpurun_out/r05_bench_kernel_stats.txt; tail -12 gpurun_out/r05_step_mfma_pmc.txt; cat gpurun_out/r05_trainer_rate.txt
python -c "
import json;d=json.load(open('gpurun_out/r05_roofline_traffic.json'));print(d['hbm_bytes_per_step'], d['traffic_bytes_per_launch_avg']);[print(k,v.get('measured_us_replayed_step'),round(v['hbm_bytes_per_launch']/1e6), v.get('x_of_mfma_bound'), v.get('x_of_hbm_bound')) for k,v in d['per_kernel'].items()]"
