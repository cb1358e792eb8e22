#!/usr/bin/env python3
"""profiles/rNN_roofline_traffic.json from the two-pass PMC summary written by tools/gpu_pmc_traffic.sh.
bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: both counters are in KB and FETCH_SIZE under-reports coalesced
reads by 2x on gfx950 (MI355X_MICROARCH.md, HBM section; at EVERY access width from 1 to 16 bytes per lane and through the LDS-DMA:
profiles/r05_fetch_size_calibration.txt).  The family = every conv3x3_c64_kernel / conv3x3_igemm_kernel /
vgg_level0_fwd / vgg_level0_dgrad launch of a step; the average is over the SIX calls of a step (3 forward + 3 data-gradient
convolutions: what bench.py's roofline leg brackets) -- conv.5's forward is two kernel launches inside one call since round 3; since
round 4 conv.2's forward and data gradient are the full-resolution-level kernels of csrc/conv_level0.hip (conv.0, the first pool and the
first layer's weight gradient ride in them).  Per kernel the JSON also states which roof bounds it: algorithmic FLOPs / 2.5 PFLOP/s
against measured HBM bytes / 6.29 TB/s (the guide's achievable copy bandwidth)."""
import json
import re
import sys

STEPS = 3            # eager steps in the profiled command (--steps 2 --warmup 1)
PX1, PX2 = 32 * 161 * 800, 32 * 80 * 400


def algorithmic_flop(name):
    """2 * MAC per launch at the benchmark shape (B = 32): conv.2 (+ conv.0 / dW0 in the level-0 kernels), conv.5 per 64-channel pass,
    conv.7, and the two 128-channel data gradients."""
    if "vgg_level0_fwd" in name or "vgg_level0_dgrad" in name:
        return 2 * 9 * (64 * 64 + 64) * PX1
    if "conv3x3_c64_kernel" in name:
        return 2 * 9 * 64 * 64 * (PX1 if ", true>" in name or "true, 2" in name else PX2)      # pooled / masked forms = full resolution (ASR_LEVEL0=0)
    m = re.search(r"ws128_kernel<(\d+), (\d+), (\d+),", name)
    if m:                                     # template <CI input channels, TH tile rows, CO output channels, ...> (csrc/conv_ws.hip):
        return 2 * 9 * int(m.group(1)) * int(m.group(3)) * PX2      # conv.7's 128 -> 64 data gradient is HALF of the 128 -> 128 launches (VERDICT r5 #4a)
    if "igemm_kernel<unsigned short, 128" in name:
        return 2 * 9 * 128 * 128 * PX2
    if "igemm_kernel<unsigned short, 64" in name:
        return 2 * 9 * 128 * 64 * PX2
    return None
CALLS_PER_STEP = 6
_DUR = {}
if len(sys.argv) > 2:
    try:
        _DUR = json.load(open(sys.argv[2])).get("conv_front_end_kernels_us", {})
    except (OSError, ValueError):
        _DUR = {}


def duration_us(name):
    key = re.sub(r"\(anonymous namespace\)::", "", name)
    for k, v in _DUR.items():
        if k[:60] == key[:60]:
            return v["avg_us"]
    return None


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
out = {"workload": "configs[1] B=32 bf16", "kernel_family": "asr_conv3x3_igemm / asr_conv3x3_relu_pool_code / asr_conv3x3_relu_pool_tcf_code",
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 2 --warmup 1 --eager "
                 "--no-cpu-baseline --no-roofline`; bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (both counters in KB, FETCH_SIZE "
                 "under-reports 16-B/lane reads by 2x on gfx950); per step = sum over the family's kernels of bytes x launches / %d steps" % STEPS,
       "per_kernel": {}}
# steps in the profiled run = launches of a once-per-step kernel, per counter (the two passes are separate runs)
_once = [v for nm, v in vals.items() if "vgg_level0_fwd_kernel" in nm]
STEPS_F = _once[0]["FETCH_SIZE"][0] if _once and "FETCH_SIZE" in _once[0] else STEPS
tot = 0.0
for nm, v in sorted(vals.items()):
    if not re.search(r"conv3x3_c64_kernel|conv3x3_igemm_kernel|conv3x3_ws128_kernel|vgg_level0_fwd_kernel|vgg_level0_dgrad_kernel", nm) or "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        continue
    n, f = v["FETCH_SIZE"]
    w = v["WRITE_SIZE"][1]
    b = (2 * f + w) * 1024
    e = {"launches_per_step": n / STEPS_F, "FETCH_SIZE_KB_avg": f, "WRITE_SIZE_KB_avg": w, "hbm_bytes_per_launch": b}
    fl = algorithmic_flop(nm)
    if fl:
        e["algorithmic_gflop_per_launch"] = fl / 1e9
        e["t_mfma_us_at_2.5PF"] = fl / 2.5e15 * 1e6
        e["t_hbm_us_at_6.29TBs"] = b / 6.29e12 * 1e6
        # which roof is NEARER says little when a launch runs at 2x either of them: the measured duration of the
        # launch in the replayed step (second argument: tools/prof_families.py's JSON) next to both lower bounds
        us = duration_us(nm)
        if us:
            e["measured_us_replayed_step"] = us
            e["achieved_TFLOPs"] = fl / us / 1e6
            e["achieved_HBM_TBs"] = b / us / 1e6
            e["x_of_mfma_bound"] = us / e["t_mfma_us_at_2.5PF"]
            e["x_of_hbm_bound"] = us / e["t_hbm_us_at_6.29TBs"]
    out["per_kernel"][nm[:90]] = e
    tot += b * n / STEPS_F
out["hbm_bytes_per_step"] = tot
out["traffic_bytes_per_launch_avg"] = tot / CALLS_PER_STEP if out["per_kernel"] else None
print(json.dumps(out, indent=1))
