#!/bin/bash
mkdir -p gpurun_out && rm -f gpurun_out/.stage_dead
tools/gpu_stage.sh r4_tests_f1 600 python -m pytest tests/test_gpu_groups.py -m gpu -x -q -s -k "unchanged_trainer or grouped_projections or clip_attention"
tools/gpu_stage.sh r4_tests_f2 900 python -m pytest tests -m gpu -x -q
for v in 0 1 0 1; do
  DFA_DROPIN_GROUPS=$v timeout -k 10 300 python bench.py --drop-in --steps 20 --warmup 5 > gpurun_out/r4_dropin_$v.log 2>&1 || exit 1
  echo "DFA_DROPIN_GROUPS=$v $(grep '^{' gpurun_out/r4_dropin_$v.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2), "img/s", round(d["ms_per_step"],2), "ms")')"
done
