#!/bin/bash
export TMPDIR=/tmp
cat > /tmp/ab.py <<'PY'
import sys
sys.path.insert(0, "tools")
import microbench as M
M.attn([(32, 8, 800, 800, 64, False, 0.1), (16, 8, 795, 795, 64, False, 0.1)])
PY
timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q -k "attention or emb or bf16_mode" 2>&1 | tail -3
timeout 200 python tools/ab/ab_attn_pp.py 2>&1 | grep -i "dropout" | head -4
python /tmp/ab.py 2>&1 | grep "attn ("
b() { timeout 600 python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))"; }
echo "librispeech: $(b --workload librispeech) / $(b --workload librispeech)"
