#!/bin/bash
# Dev tool: the unchanged-trainer route of two trees on ONE box, alternating (host speed differs by ±20 % from box to box and drifts
# on a box: only interleaved runs compare).  usage: tools/dropin_ab.sh <out.log> <rounds> [steps]   — "before" = _ab_before/ (an
# export of the commit to compare against, with its own package and library), "after" = this tree.
set -e
out=$1; rounds=${2:-2}; steps=${3:-30}
root=$(pwd)
export BENCH_CONV_AUTOTUNE=0
: > "$out"
for r in $(seq 1 "$rounds"); do
  for tree in before after; do
    if [ "$tree" = before ]; then cd "$root/_ab_before"; else cd "$root"; fi
    line=$(python bench.py --drop-in --steps "$steps" --warmup 5 2>/dev/null | tail -1)
    echo "round $r $tree: $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.1f images/s  %.2f ms/step" % (d["value"], d["ms_per_step"]))')" | tee -a "$root/$out"
  done
done
cd "$root"
