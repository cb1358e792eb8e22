#!/bin/bash
# round 5: the forced one-rank data-parallel step over nccl: four graphs with the collectives between them vs ONE graph with the collectives captured
mkdir -p gpurun_out; export TMPDIR=/tmp
B="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline --soak-seconds 0"
{
  for rep in 1 2; do
    echo "== single graph, no reducer"; timeout 300 $B 2>/dev/null | tail -1
    echo "== ASR_FORCE_DDP=1, four graphs"; ASR_FORCE_DDP=1 timeout 300 $B 2>/dev/null | tail -1
    echo "== ASR_FORCE_DDP=1 ASR_DDP_ONE_GRAPH=1"; ASR_FORCE_DDP=1 ASR_DDP_ONE_GRAPH=1 timeout 300 $B 2>gpurun_out/r5d_one_graph.err | tail -1
  done
} > gpurun_out/r5d_ddp_one_graph.txt 2>&1
grep -i "warn\|error\|fail" gpurun_out/r5d_one_graph.err | head -5
python - <<'PY'
import json
for l in open("gpurun_out/r5d_ddp_one_graph.txt"):
    l = l.strip()
    if l.startswith("=="): print(l, end="  ")
    elif l.startswith("{"):
        d = json.loads(l); print("ms/step %.3f  mode %s  version line %s" % (d["ms_per_step"], d.get("launch_mode"), d["config"].get("collective_version_line")))
    elif l: print(l[:200])
PY
