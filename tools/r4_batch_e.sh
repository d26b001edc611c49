#!/bin/bash
mkdir -p gpurun_out && rm -f gpurun_out/.stage_dead
tools/gpu_stage.sh r4_tests_e1 900 python -m pytest tests/test_gpu_pti.py tests/test_gpu_dp.py -m gpu -x -q -s
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export PYTORCH_MIOPEN_SUGGEST_NHWC=1; fl=--channels-last; else unset PYTORCH_MIOPEN_SUGGEST_NHWC; fl=; fi
  tools/gpu_stage.sh r4_cl_$v 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra $fl
  python - <<P
import json
d=json.loads([l for l in open("gpurun_out/r4_cl_$v.log") if l.startswith("{")][-1])
print("channels_last=$v", round(d["value"],2), "img/s", round(d["ms_per_step"],3), "ms tail", d["config"].get("tail_ms_per_step"))
P
done
