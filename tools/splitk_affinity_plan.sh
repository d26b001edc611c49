#!/bin/bash
# Dev tool: the one split-K plan round 5 left untested — slice → XCD affinity with a slice count that divides 8
# AND keeps the launch in ONE round of 512 workgroup slots — against the library's plan, weights cold, alternating.
# usage (GPU box): bash tools/splitk_affinity_plan.sh > gpurun_out/<log>
export GB_ITERS=${GB_ITERS:-30}
one() {  # $1 label, $2 env string, $3.. gemm_bench args
  label=$1; envs=$2; shift 2
  echo "--- $label   [$envs]"
  env $envs timeout -k 10 200 python tools/gemm_bench.py --cold-read "$@" 2>&1 | grep -v amdgpu.ids | grep "bwd\|qkv   1024\|qkv    256"
}
for r in 1 2; do
  echo "=== round $r"
  # GEGLU proj backward 1024 x 10240 -> 1280  (shape 7), 80 tiles of 128 rows
  one "1024x10240->1280 library plan (6 slices of 128-row tiles, affinity off)" "GB_SHAPES=7"
  one "1024x10240->1280  S=4 x 80 tiles = 320, affinity ON" "GB_SHAPES=7 LORA_SPLIT_AFFINITY=1 LORA_SPLITK=4 LORA_SPLIT_BM=128 LORA_SPLIT_MINSTEPS=2"
  one "1024x10240->1280  S=4 x 80 tiles = 320, affinity off" "GB_SHAPES=7 LORA_SPLITK=4 LORA_SPLIT_BM=128 LORA_SPLIT_MINSTEPS=2"
  # 4096 x 5120 -> 640 (shape 4), 160 tiles of 128 rows
  one "4096x5120->640 library plan" "GB_SHAPES=4"
  one "4096x5120->640  S=2 x 160 tiles = 320, affinity ON" "GB_SHAPES=4 LORA_SPLIT_AFFINITY=1 LORA_SPLITK=2 LORA_SPLIT_BM=128 LORA_SPLIT_MINSTEPS=2"
  one "4096x5120->640  S=2 x 160 tiles = 320, affinity off" "GB_SHAPES=4 LORA_SPLITK=2 LORA_SPLIT_BM=128 LORA_SPLIT_MINSTEPS=2"
  # grouped q/k/v backward at 1024 rows: 160 tiles of 64 rows
  one "q/k/v backward library plan" "X=1" --grouped
  one "q/k/v backward  S=2 x 160 tiles (64 rows), affinity ON" "LORA_SPLIT_AFFINITY=1 LORA_SPLITK=2 LORA_SPLIT_BM=64 LORA_SPLIT_MINSTEPS=2" --grouped
  one "q/k/v backward  S=2 x 160 tiles (64 rows), affinity off" "LORA_SPLITK=2 LORA_SPLIT_BM=64 LORA_SPLIT_MINSTEPS=2" --grouped
done
