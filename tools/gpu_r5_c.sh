#!/bin/bash
# round 5: the new tests + the model-level suites on the current tree
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 1500 python -m pytest -q -m gpu --tb=short -x tests/test_gpu_model.py tests/test_gpu_train_cli.py tests/test_gpu_graph.py "tests/test_gpu_baseline_shapes.py::test_benched_conv_path_agrees_with_the_tapped_launch_chain" 2>&1 | tail -30 ) > gpurun_out/r5c_pytest.log
tail -30 gpurun_out/r5c_pytest.log
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/parity_r05.json"))
    k = "cfg1_b32/bf16/tap_off_vs_tap_on"
    print(k, {a: b for a, b in d[k].items() if a != "grad_rel_l2"})
    print(list(d[k]["grad_rel_l2"].items())[:6])
except Exception as e:
    print("no parity json", e)
PY
