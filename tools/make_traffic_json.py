#!/usr/bin/env python3
"""profiles/rNN_roofline_traffic.json from the two-pass PMC summary written by tools/gpu_pmc_traffic.sh.
bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: both counters are in KB and FETCH_SIZE under-reports 16-byte-per-lane
reads by 2x on gfx950 (MI355X_MICROARCH.md, HBM section); launches per step of the headline workload: 64->64 forward (+pool) 1,
64->64 dgrad 1, 128-channel implicit GEMM 3, 128->64 dgrad 1."""
import json
import re
import sys

txt = open(sys.argv[1]).read().splitlines()
vals = {}
name = None
for ln in txt:
    if ln.startswith("#"):
        continue
    m = re.match(r"\s+(FETCH_SIZE|WRITE_SIZE)\s+n=\s*(\d+)\s+avg=\s*([0-9.]+)", ln)
    if m and name:
        vals.setdefault(name, {})[m.group(1)] = (int(m.group(2)), float(m.group(3)))
    elif ln.strip():
        name = ln.strip()
per_step = {"conv3x3_c64_kernel<16, 8, false, 3, true>": 1, "conv3x3_c64_kernel<16, 8, true, 2, false>": 1,
            "conv3x3_igemm_kernel<unsigned short, 128": 3, "conv3x3_igemm_kernel<unsigned short, 64": 1}
out = {"workload": "configs[1] B=32 bf16", "kernel_family": "asr_conv3x3_igemm / asr_conv3x3_relu_pool",
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 2 --warmup 1 --eager "
                 "--no-cpu-baseline --no-roofline`; bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (both counters in KB, FETCH_SIZE "
                 "under-reports 16-B/lane reads by 2x on gfx950)", "per_kernel": {}}
tot, n = 0.0, 0
for key, k in per_step.items():
    hit = [v for nm, v in vals.items() if key in nm and "FETCH_SIZE" in v and "WRITE_SIZE" in v]
    if not hit:
        continue
    f, w = hit[0]["FETCH_SIZE"][1], hit[0]["WRITE_SIZE"][1]
    b = (2 * f + w) * 1024
    out["per_kernel"][key] = {"launches_per_step": k, "FETCH_SIZE_KB_avg": f, "WRITE_SIZE_KB_avg": w, "hbm_bytes_per_launch": b}
    tot += b * k
    n += k
out["traffic_bytes_per_launch_avg"] = tot / n if n == 6 else None
print(json.dumps(out, indent=1))
