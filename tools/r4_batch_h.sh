#!/bin/bash
mkdir -p gpurun_out && rm -f gpurun_out/.stage_dead
tools/gpu_stage.sh r4_tests_h 1100 python -m pytest tests -m gpu -q --durations=15
tools/gpu_stage.sh r4_bench_h 900 python bench.py --steps 10 --warmup 3
