#!/bin/bash
# Dev tool: in-isolation sweep of the split-K plan (slice count × row-tile height) on the shapes it may apply to, weights cold.
# usage (GPU box): bash tools/splitk_sweep.sh > gpurun_out/splitk_sweep.log   ("-1 0" = the library's own plan)
export GB_ITERS=${GB_ITERS:-30}
for setting in ${SWEEP:-"-1:0 0:0 2:64 3:64 4:64 6:64 8:64 2:128 3:128 4:128 6:128"}; do
  s=${setting%%:*}; bm=${setting##*:}
  ms=0; [ "$s" -gt 0 ] && ms=2
  echo "=== LORA_SPLITK=$s LORA_SPLIT_BM=$bm"
  LORA_SPLITK=$s LORA_SPLIT_BM=$bm LORA_SPLIT_MINSTEPS=$ms GB_SHAPES=${GB_SHAPES:-3,6,9,7,8,4,5,10} timeout -k 10 300 python tools/gemm_bench.py --cold-read || exit 1
  LORA_SPLITK=$s LORA_SPLIT_BM=$bm LORA_SPLIT_MINSTEPS=$ms timeout -k 10 300 python tools/gemm_bench.py --cold-read --grouped || exit 1
done
