#!/bin/bash
# In-model per-launch-class timings of the hot-path kernels: rocprofv3 kernel trace of a short HOST-LAUNCHED bench run
# (every dispatch is visible in order), reduced by tools/summarize_profile.py shapes.   usage: tools/trace_shapes.sh <tag>
tag=${1:-shapes}
out=gpurun_out/trace_$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d "$out/trace" -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-prof > "$out/run.log" 2>&1
t=$(ls "$out"/trace/*/*_kernel_trace.csv | head -1)
python3 tools/summarize_profile.py shapes "$t" 3 > "$out/shapes.txt"
rm -rf "$out/trace"
cat "$out/shapes.txt"
