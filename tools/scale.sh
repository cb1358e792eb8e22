#!/bin/bash
# Scaling curve of the data-parallel training step on ONE node, in one command: bench.py at 1 / 2 / 4 / 8 GPUs, weak
# scaling (32 utterances per GPU) and the BASELINE configs[2] point (global batch 256 = 32 per GPU at 8 GPUs; for 2 / 4 GPUs also
# 128 / 64 per GPU so that the global batch stays 256), one JSON file with every line bench.py printed plus the derived ratios.
# Self-diagnosing: every point carries each rank's own ms/step (min / max / all: a straggler shows as a spread), the
# exposed part of the gradient exchange (the same steps with the collectives switched off, bench.py), the graph mode that actually
# ran (--ddp-graph: requested / ran) and RCCL's version line; a table of them is printed at the end.
# usage: tools/scale.sh [out.json] [steps] [warmup] [ddp-graph: four|one|auto]     (needs the GPUs visible; each point is a fresh torch.distributed.run)
out=${1:-gpurun_out/scale.json}
steps=${2:-20}
warmup=${3:-5}
ddpg=${4:-four}
root=$(cd "$(dirname "$0")/.." && pwd)
ngpu=$(python -c "import torch; print(torch.cuda.device_count())")
mkdir -p "$(dirname "$out")"
tmp=$(mktemp)
for n in 1 2 4 8; do
  [ "$n" -le "$ngpu" ] || { echo "{\"n_gpus\": $n, \"skipped\": \"only $ngpu GPU(s) visible\"}" >> "$tmp"; continue; }
  for batch in 32 $((256 / n)); do
    [ "$n" -eq 8 ] && [ "$batch" -ne 32 ] && continue
    [ "$n" -eq 1 ] && [ "$batch" -ne 32 ] && continue
    port=$((29600 + n * 10 + batch % 7))
    if [ "$n" -eq 1 ]; then
      line=$(cd "$root" && timeout 1200 python bench.py --gpus 1 --steps "$steps" --warmup "$warmup" --batch "$batch" --no-cpu-baseline --no-roofline 2>/dev/null | tail -1)
    else
      line=$(cd "$root" && HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 \
             --master-port "$port" bench.py --gpus "$n" --steps "$steps" --warmup "$warmup" --batch "$batch" --ddp-graph "$ddpg" --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{' | tail -1)
    fi
    [ -n "$line" ] && echo "$line" >> "$tmp" || echo "{\"n_gpus\": $n, \"batch_per_gpu\": $batch, \"failed\": true}" >> "$tmp"
  done
done
python - "$tmp" "$out" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip()]
base = next((r for r in rows if r.get("n_gpus") == 1 and "value" in r), None)
pts = []
for r in rows:
    if "value" not in r:
        pts.append(r)
        continue
    c = r["config"]
    ar = c.get("gradient_allreduce") or {}
    pts.append({"n_gpus": r["n_gpus"], "batch_per_gpu": c["global_batch"] // r["n_gpus"], "global_batch": c["global_batch"],
                "frames_per_s": r["value"], "ms_per_step": r["ms_per_step"], "rank0_ms_per_step": c.get("rank0_ms_per_step"),
                "collective_backend": c.get("collective_backend"), "collective_ranks": c.get("collective_ranks"),
                "exposed_allreduce_ms_per_step": ar.get("exposed_ms_per_step"), "ms_per_step_without_allreduce": ar.get("ms_per_step_without_allreduce"),
                "per_rank_ms_per_step": c.get("per_rank_ms_per_step"), "ddp_graph": c.get("ddp_graph"), "grad_wire": c.get("grad_wire"),
                "collective_library": c.get("collective_library"), "collective_version_line": c.get("collective_version_line"),
                "speedup_vs_1gpu": (r["value"] / base["value"]) if base else None,
                "efficiency": (r["value"] / base["value"] / r["n_gpus"]) if base else None, "launch_mode": r.get("launch_mode")})
json.dump({"points": pts, "lines": rows}, open(sys.argv[2], "w"), indent=1)
print(json.dumps(pts, indent=1))
print("%5s %6s %12s %9s %9s %9s %9s %8s %6s  %s" % ("gpus", "b/gpu", "frames/s", "ms/step", "rank min", "rank max", "exposed", "speedup", "eff", "graph mode"))
for p_ in pts:
    if "frames_per_s" not in p_:
        print("%5s  %s" % (p_.get("n_gpus"), {k: v for k, v in p_.items() if k != "n_gpus"}))
        continue
    pr = p_.get("per_rank_ms_per_step") or {}
    f = lambda v: ("%9.3f" % v) if isinstance(v, (int, float)) else "%9s" % "-"
    print("%5d %6d %12.0f %s %s %s %s %8s %6s  %s" % (p_["n_gpus"], p_["batch_per_gpu"], p_["frames_per_s"], f(p_["ms_per_step"]), f(pr.get("min")), f(pr.get("max")),
                                                  f(p_.get("exposed_allreduce_ms_per_step")), ("%.2f" % p_["speedup_vs_1gpu"]) if p_.get("speedup_vs_1gpu") else "-",
                                                  ("%.2f" % p_["efficiency"]) if p_.get("efficiency") else "-", (p_.get("ddp_graph") or {}).get("ran")))
PY
rm -f "$tmp"
