#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 1500 python -m pytest -q -m gpu --tb=short -s \
   "tests/test_gpu_model.py::test_logit_handover_keeps_a_second_consumers_gradient" tests/test_gpu_ops.py -k "attention or handover" \
   tests/test_gpu_baseline_shapes.py -k "cfg3_b16 and fp32 or attention or handover" 2>&1 | tail -60 ) > gpurun_out/r4b_pytest.log
tail -40 gpurun_out/r4b_pytest.log
