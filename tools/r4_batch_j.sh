#!/bin/bash
mkdir -p gpurun_out && rm -f gpurun_out/.stage_dead
echo "== old library (before the XCD remap)" > gpurun_out/r4_flash_xcd_ab.log
DFA_LIB_PATH=$PWD/build/liblora_old.so timeout -k 10 300 python tools/flash_check.py --time >> gpurun_out/r4_flash_xcd_ab.log 2>&1 || exit 1
echo "== new library" >> gpurun_out/r4_flash_xcd_ab.log
timeout -k 10 300 python tools/flash_check.py --time >> gpurun_out/r4_flash_xcd_ab.log 2>&1 || exit 1
grep -E "^==|T=|stock|FAIL|ALL" gpurun_out/r4_flash_xcd_ab.log
for v in old new old new; do
  if [ $v = old ]; then export DFA_LIB_PATH=$PWD/build/liblora_old.so; else unset DFA_LIB_PATH; fi
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > gpurun_out/r4_xcd_$v.log 2>&1 || exit 1
  python - <<P
import json
d=json.loads([l for l in open("gpurun_out/r4_xcd_$v.log") if l.startswith("{")][-1])
k=d["hot_path"]["kernels"]
print("$v", round(d["value"],2), "img/s", round(d["ms_per_step"],3), "ms; flash fwd/dq/dkdv us:", [round(v["avg_us"],1) for n,v in k.items() if "flash" in n])
P
done
