#!/bin/bash
# round-4 batch A: split-K slice→XCD affinity A/B (isolation, cold weights), GPU tests, quick bench
mkdir -p gpurun_out && rm -f gpurun_out/.stage_dead
export GB_ITERS=20
for aff in 0 1; do
  echo "=== LORA_SPLIT_AFFINITY=$aff (library plan)"
  LORA_SPLIT_AFFINITY=$aff GB_SHAPES=5,8,10,4,7 timeout -k 10 300 python tools/gemm_bench.py --cold-read || exit 1
  LORA_SPLIT_AFFINITY=$aff timeout -k 10 300 python tools/gemm_bench.py --cold-read --grouped || exit 1
done > gpurun_out/r4_split_aff.log 2>&1
for s in 2 4 8; do
  echo "=== LORA_SPLIT_AFFINITY=1 LORA_SPLITK=$s"
  LORA_SPLIT_AFFINITY=1 LORA_SPLITK=$s LORA_SPLIT_MINSTEPS=2 GB_SHAPES=5,8,10 timeout -k 10 300 python tools/gemm_bench.py --cold-read || exit 1
  LORA_SPLIT_AFFINITY=1 LORA_SPLITK=$s LORA_SPLIT_MINSTEPS=2 timeout -k 10 300 python tools/gemm_bench.py --cold-read --grouped || exit 1
done >> gpurun_out/r4_split_aff.log 2>&1
tools/gpu_stage.sh r4_tests_a 800 python -m pytest tests -m gpu -x -q
tools/gpu_stage.sh r4_bench_a 500 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra
