#!/bin/bash
mkdir -p gpurun_out && rm -f gpurun_out/.stage_dead
echo "== old library" > gpurun_out/r4_flash_ab.log
DFA_LIB_PATH=$PWD/build/liblora_old.so timeout -k 10 300 python tools/flash_check.py --time >> gpurun_out/r4_flash_ab.log 2>&1 || exit 1
echo "== new library" >> gpurun_out/r4_flash_ab.log
timeout -k 10 300 python tools/flash_check.py --time >> gpurun_out/r4_flash_ab.log 2>&1 || exit 1
grep -E "^==|T=|stock|FAIL|ALL" gpurun_out/r4_flash_ab.log
tools/gpu_stage.sh r4_tests_g 900 python -m pytest tests -m gpu -x -q
tools/gpu_stage.sh r4_bench_g 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra
