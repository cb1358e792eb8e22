#!/bin/bash
# One gpurun call: a SUBSET of the GPU test suite, full output kept.
# usage: tools/gpu_tests.sh <tag> "<-k expression>" [files...]
tag=${1:-t}; kexpr=$2; shift 2
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest -q -m gpu --tb=short -s -k "$kexpr" ${@:-tests} 2>&1 | tail -400 ) > gpurun_out/${tag}_pytest.log
tail -80 gpurun_out/${tag}_pytest.log
