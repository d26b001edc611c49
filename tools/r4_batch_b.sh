#!/bin/bash
# round-4 batch B: wide (3r > 16) grouped q/k/v — kernel + trajectory tests, then the bench with the extra configs
mkdir -p gpurun_out && rm -f gpurun_out/.stage_dead
tools/gpu_stage.sh r4_tests_b1 600 python -m pytest tests/test_gpu_groups.py -m gpu -x -q -k "gemm_parts or ranks_of_configs or clip_attention or cfg5 or cfg3"
tools/gpu_stage.sh r4_tests_b2 800 python -m pytest tests -m gpu -x -q
tools/gpu_stage.sh r4_bench_b 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline
