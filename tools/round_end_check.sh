#!/bin/bash
mkdir -p gpurun_out && rm -f gpurun_out/.stage_dead
for only in "" gB gA; do
  echo "== GB_GRADS_ONLY=$only"; GB_GRADS_ONLY=$only timeout -k 10 200 python tools/gemm_bench.py --grads || exit 1
done > gpurun_out/r4_grads_split.log 2>&1
cat gpurun_out/r4_grads_split.log | grep -E "^==|batched"
tools/gpu_stage.sh r4_tests_i 1100 python -m pytest tests -m gpu -q -x
tools/gpu_stage.sh r4_smoke_i 200 python __graft_entry__.py --smoke
bash tools/collect_profiles.sh ${1:-r04_final} > gpurun_out/r4_collect.log 2>&1
tail -15 gpurun_out/r4_collect.log
