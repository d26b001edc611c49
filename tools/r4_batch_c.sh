#!/bin/bash
# round-4 batch C: PTI tuning step (token table, grouped context dX) tests, then everything, then the bench
mkdir -p gpurun_out && rm -f gpurun_out/.stage_dead
tools/gpu_stage.sh r4_tests_c1 900 python -m pytest tests/test_gpu_pti.py tests/test_gpu_dp.py -m gpu -x -q -s
tools/gpu_stage.sh r4_tests_c2 900 python -m pytest tests -m gpu -x -q
tools/gpu_stage.sh r4_bench_c 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline
