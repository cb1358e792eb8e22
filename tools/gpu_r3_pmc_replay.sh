#!/bin/bash
# MFMA-busy of the kernels of the GRAPH-REPLAYED step (the grouped weight-gradient kernel only exists there)
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/r03_replay_mfma_pmc.txt
: > $out
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_WAVES"; do
  i=$((i+1)); d=/tmp/pmc_replay_$i; rm -rf $d
  timeout 600 rocprofv3 --pmc $set -d $d -o pmc -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/r03_replay_pmc_log$i.txt 2>&1
  db=$(find $d -name "*.db" | head -1)
  echo "# rocprofv3 --pmc $set -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline" >> $out
  [ -n "$db" ] && python tools/pmc_summary.py "$db" >> $out 2>&1 || echo "no database" >> $out
done
python - $out <<'PY'
import re, sys
cur = None; d = {}
for line in open(sys.argv[1]):
    if line.startswith("#"): continue
    if not line.startswith(" "):
        cur = line.strip(); d.setdefault(cur, {}); continue
    m = re.match(r"\s+(\S+)\s+n=\s*(\d+)\s+avg=\s*([\d.]+)", line)
    if m: d[cur][m.group(1)] = float(m.group(3))
rows = []
for k, v in d.items():
    if "GRBM_GUI_ACTIVE" in v and v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0:
        cyc = v["GRBM_GUI_ACTIVE"] / 8
        rows.append((v["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), cyc / 2400, v.get("SQ_INSTS_VALU", 0) / max(v.get("SQ_INSTS_MFMA", 1), 1), k))
for u, us, r, k in sorted(rows, key=lambda t: -t[1])[:24]:
    print("  MfmaUtil %5.1f %%  ~%7.1f us  VALU/MFMA %5.1f  %s" % (100 * u, us, r, k[:100]))
PY
