#!/bin/bash
# round 3, GPU call C: deferred grouped weight gradients (A/B + timeline), attention forward with exact score scaling, full suite
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline ) > gpurun_out/r3c_bench_defer1.txt 2>&1
tail -1 gpurun_out/r3c_bench_defer1.txt | cut -c1-260
( ASR_DEFER_WGRAD=0 timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline ) > gpurun_out/r3c_bench_defer0.txt 2>&1
tail -1 gpurun_out/r3c_bench_defer0.txt | cut -c1-260
( ASR_TN_GROUP_STAGES=4 timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline ) > gpurun_out/r3c_bench_defer1_nst4.txt 2>&1
tail -1 gpurun_out/r3c_bench_defer1_nst4.txt | cut -c1-260
( timeout 600 python bench.py --workload librispeech --steps 20 --warmup 5 --no-cpu-baseline --no-roofline ) > gpurun_out/r3c_bench_libri.txt 2>&1
tail -1 gpurun_out/r3c_bench_libri.txt | cut -c1-260
( ASR_DEFER_WGRAD=0 ASR_ATTN_PP=0 timeout 600 python bench.py --workload librispeech --steps 20 --warmup 5 --no-cpu-baseline --no-roofline ) > gpurun_out/r3c_bench_libri_old.txt 2>&1
tail -1 gpurun_out/r3c_bench_libri_old.txt | cut -c1-260
bash tools/gpu_profile.sh r3c_bench 13 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline
head -60 gpurun_out/r3c_bench_timeline.txt
( timeout 600 python tools/ab/ab_attn_pp.py all ) > gpurun_out/r3c_attn_pp.txt 2>&1
tail -34 gpurun_out/r3c_attn_pp.txt | cut -c1-200
( timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -40 ) > gpurun_out/r3c_pytest.txt
tail -15 gpurun_out/r3c_pytest.txt
