#!/bin/bash
# In-model A/B of an environment knob on ONE box: tools/ab_env.sh VAR valueA valueB  → two alternating bench runs each
var=$1; a=$2; b=$3
for v in $a $b $a $b; do
env $var=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > gpurun_out/ab_env.log 2> gpurun_out/ab_env.err
python - <<P
import json
d=json.loads(open("gpurun_out/ab_env.log").read().strip().splitlines()[-1])
print("$var=$v", round(d["value"],2), "img/s", round(d["ms_per_step"],3), "ms  frac", round(d["roofline"]["frac"],4), " hot path", round(d["hot_path"]["kernel_ms_per_step"],3), "ms", {k[17:28]:(v["launches_per_step"], round(v["avg_us"],1)) for k,v in d["hot_path"]["kernels"].items() if "gemm" in k})
P
done
