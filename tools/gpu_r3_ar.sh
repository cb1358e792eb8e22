#!/bin/bash
export TMPDIR=/tmp
cat > /tmp/ab.py <<'PY'
import sys
sys.path.insert(0, "tools")
import microbench as M
M.attn([(32, 8, 800, 800, 64, False, 0.1), (16, 8, 795, 795, 64, False, 0.1), (32, 8, 200, 200, 64, False, 0.1), (32, 8, 100, 200, 64, False, 0.1)])
PY
b() { timeout 600 python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))"; }
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "attention" 2>&1 | tail -3
S=end2end-asr-pytorch_amd/asr_hip
echo "== new (lane pairs share the dropout hashes in dK/dV)"; python /tmp/ab.py 2>&1 | grep "attn ("
echo "headline: $(b) / $(b)   librispeech: $(b --workload librispeech)"
cp $S/libasr_hip.so /tmp/new.so; cp $S/libasr_hip_prev.so $S/libasr_hip.so
echo "== previous build"; python /tmp/ab.py 2>&1 | grep "attn ("
echo "headline: $(b) / $(b)   librispeech: $(b --workload librispeech)"
cp /tmp/new.so $S/libasr_hip.so
