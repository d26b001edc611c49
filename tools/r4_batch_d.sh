#!/bin/bash
mkdir -p gpurun_out && rm -f gpurun_out/.stage_dead
tools/gpu_stage.sh r4_tests_d1 900 python -m pytest tests/test_gpu_pti.py tests/test_gpu_dp.py -m gpu -x -q -s
tools/gpu_stage.sh r4_tests_d2 900 python -m pytest tests/test_gpu_groups.py -m gpu -x -q -s -k "full_size"
tools/gpu_stage.sh r4_overlap_ab 600 bash tools/ab_env.sh DFA_GRAD_OVERLAP 0 56
tools/gpu_stage.sh r4_overlap_ab2 600 bash tools/ab_env.sh DFA_GRAD_OVERLAP 28 112
