#!/bin/bash
# round 4, final evidence on the final tree: the whole GPU suite, the bench lines (headline with roofline + cpu_baseline, librispeech, lowrank,
# forced data-parallel reducer with both wire types), then the profiles of tools/gpu_r4_profiles.sh.  Everything lands in gpurun_out/r04_*.
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 2400 python -m pytest -q -m gpu --tb=short tests 2>&1 | tail -25 ) > gpurun_out/r04_pytest_gpu.log
tail -5 gpurun_out/r04_pytest_gpu.log
python bench.py > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench.err
python bench.py --workload librispeech --steps 20 --warmup 5 > gpurun_out/r04_bench_line_librispeech.json 2>> gpurun_out/r04_bench.err
python bench.py --workload lowrank --steps 20 --warmup 5 > gpurun_out/r04_bench_line_lowrank.json 2>> gpurun_out/r04_bench.err
ASR_FORCE_DDP=1 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline > gpurun_out/r04_bench_line_ddp1_forced.json 2>> gpurun_out/r04_bench.err
ASR_FORCE_DDP=1 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline --grad-wire bf16 > gpurun_out/r04_bench_line_ddp1_forced_bf16wire.json 2>> gpurun_out/r04_bench.err
for f in gpurun_out/r04_bench_line*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d.get("roofline") or {}
    print(sys.argv[1], "ms/step %.3f value %.0f frac %s" % (d["ms_per_step"], d["value"], r.get("frac")))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
bash tools/gpu_r4_profiles.sh > gpurun_out/r04_profiles_script.log 2>&1
tail -3 gpurun_out/r04_profiles_script.log
