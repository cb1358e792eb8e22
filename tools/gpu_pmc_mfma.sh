#!/bin/bash
# MFMA-busy / wave-time counters of every kernel of the headline step: rocprofv3 --pmc in two counter-only passes (no trace domains)
# over eager steps -> gpurun_out/<tag>_mfma_pmc.txt (raw per-launch averages) + derived MfmaUtil per kernel.
tag=${1:-mfma}
root=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cmd="python bench.py --steps 2 --warmup 1 --eager --no-cpu-baseline --no-roofline --soak-seconds 0"
: > $root/gpurun_out/${tag}_mfma_pmc.txt
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_WAVES"; do
  i=$((i+1)); out=/tmp/pmc_${tag}_$i; rm -rf $out
  ( cd $root && timeout 600 rocprofv3 --pmc $set -d $out -o pmc -- $cmd ) > $root/gpurun_out/${tag}_mfma_log$i.txt 2>&1
  db=$(find $out -name "*.db" | head -1)
  echo "# rocprofv3 --pmc $set -- $cmd" >> $root/gpurun_out/${tag}_mfma_pmc.txt
  if [ -n "$db" ]; then python $root/tools/pmc_summary.py "$db" >> $root/gpurun_out/${tag}_mfma_pmc.txt 2>&1; else echo "no database" >> $root/gpurun_out/${tag}_mfma_pmc.txt; fi
done
python - "$root/gpurun_out/${tag}_mfma_pmc.txt" <<'PY' | tee -a "$root/gpurun_out/${tag}_mfma_pmc.txt"
import re, sys
cur = None; d = {}
for line in open(sys.argv[1]):
    if line.startswith("#"): continue
    if not line.startswith(" "):
        cur = line.strip(); d.setdefault(cur, {}); continue
    m = re.match(r"\s+(\S+)\s+n=\s*(\d+)\s+avg=\s*([\d.]+)", line)
    if m: d[cur][m.group(1)] = float(m.group(3))
print("# derived: MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); us = GRBM_GUI_ACTIVE / 8 / 2400")
rows = []
for k, v in d.items():
    if "GRBM_GUI_ACTIVE" in v and v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0:
        cyc = v["GRBM_GUI_ACTIVE"] / 8
        rows.append((v["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), cyc / 2400, v.get("SQ_INSTS_VALU", 0) / max(v.get("SQ_INSTS_MFMA", 1), 1), k))
for u, us, r, k in sorted(rows, key=lambda t: -t[1]):
    print("  MfmaUtil %5.1f %%  ~%7.1f us  VALU/MFMA %5.1f  %s" % (100 * u, us, r, k[:100]))
PY
