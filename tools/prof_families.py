#!/usr/bin/env python3
"""Per-family kernel time of the GRAPH-REPLAYED training step, from a rocprofv3 rocpd kernel trace of bench.py (the path bench.py
times; its own HIP-event pass can only bracket eager launches).  The last 3 complete replayed steps (delimited by
step_advance_kernel) are summed by kernel family and written as JSON for bench.py's `roofline.families_replayed`.

usage: python tools/prof_families.py <db> <out.json> "<command that produced the trace>" """
import json
import re
import sqlite3
import sys
from collections import defaultdict

FAMILIES = [
    ("conv3x3_igemm (3 fwd + 3 dgrad)", r"conv3x3_igemm_kernel|conv3x3_ws128_kernel|conv3x3_c64_kernel|vgg_level0_fwd_kernel|vgg_level0_dgrad_kernel"),
    ("conv3x3_wgrad", r"conv3x3_wgrad|wgrad_reduce_kernel|vgg_level0_wgrad_kernel"),
    ("conv1 + pooling (HBM-bound)", r"conv1_|pool_|vgg_level0_dw0_reduce"),
    ("linear GEMMs (fwd + dgrad + wgrad)", r"gemm_|tn_reduce|tn128_reduce"),
    ("attention fwd", r"attn_fwd"),
    ("attention bwd", r"attn_bwd|attn_delta"),
    ("LayerNorm + residual", r"add_ln|ln_partial|ln_reduce"),
    ("cross-entropy + arg-max", r"ce_fwd|ce_bwd|argmax_rows"),
    ("optimiser + casts", r"adam|cast_flat|pack_weight|sumsq|clip_coef|ratio_kernel"),
]


def main():
    db, out, cmd = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if "step_advance" in r[0]]
    if len(marks) < 4:
        raise SystemExit("fewer than 3 complete replayed steps in the trace")
    seg = rows[marks[-4]:marks[-1]]
    nsteps = 3
    fam = defaultdict(lambda: [0.0, 0])
    perk = defaultdict(lambda: [0.0, 0])
    for name, s, e in seg:
        if re.search(r"conv|vgg_level0|pool_", name):
            k = re.sub(r"\(anonymous namespace\)::", "", name)[:90]
            perk[k][0] += (e - s) / 1e3
            perk[k][1] += 1
        for label, pat in FAMILIES:
            if re.search(pat, name):
                break
        else:
            label = "other"
        fam[label][0] += (e - s) / 1e6 / nsteps
        fam[label][1] += 1
    wall = (max(r[2] for r in seg) - seg[0][1]) / 1e6 / nsteps
    res = {"source": "rocprofv3 --kernel-trace of: " + cmd, "steps_summed": nsteps, "wall_ms_per_step": wall,
           "kernel_time_ms_per_step": sum(v[0] for v in fam.values()), "launches_per_step": len(seg) // nsteps,
           "families": {k: {"ms_per_step": v[0], "launches_per_step": v[1] / nsteps} for k, v in sorted(fam.items(), key=lambda kv: -kv[1][0])},
           "conv_front_end_kernels_us": {k: {"avg_us": v[0] / v[1], "launches_per_step": v[1] / nsteps} for k, v in sorted(perk.items(), key=lambda kv: -kv[1][0])}}
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
