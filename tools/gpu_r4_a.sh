#!/bin/bash
# round 4, call A: the new parity / statistics / data-parallel tests + a baseline bench line on this box
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 1500 python -m pytest -q -m gpu --tb=short -s tests/test_gpu_dropout_stats.py tests/test_gpu_ddp.py \
   "tests/test_gpu_model.py::test_logit_handover_keeps_a_second_consumers_gradient" \
   "tests/test_gpu_ops.py::test_device_side_step_is_cancelled_by_a_non_finite_guard" \
   tests/test_gpu_baseline_shapes.py -k "not cfg1_b2 and not cfg0" 2>&1 | tail -150 ) > gpurun_out/r4a_pytest.log
tail -60 gpurun_out/r4a_pytest.log



