#!/bin/bash
# Dev tool (GPU box): correctness (tools/flash_check.py) and device time (tools/flash_ablate.py) of the long-context attention
# backward in several builds / switches, alternating in one call.
# usage: tools/flash_ab.sh <out.log> <rounds> [name[:ENV=VAL] ...]   (name = lib variant suffix, "" = the real library)
out=${1:-gpurun_out/flash_ab.log}; rounds=${2:-2}; shift 2
L=diffusion_finetuning_amd/lib
: > $out
run() {  # $1 = spec "variant:ENV=VAL", rest = command
  spec=$1; shift
  v=${spec%%:*}; e=""; [[ "$spec" == *:* ]] && e=${spec#*:}
  lib=$L/liblora_hip${v:+_$v}.so
  env DFA_LIB_PATH=$lib $e "$@"
}
for spec in "$@"; do
  [ "${spec%%:*}" = "r5" ] && continue
  echo "== correctness [$spec]" >> $out
  run "$spec" timeout -k 10 300 python tools/flash_check.py 2>&1 | grep -v amdgpu.ids >> $out || { echo "flash_check FAILED/timeout [$spec]" >> $out; tail -30 $out; exit 1; }
done
for r in $(seq $rounds); do
  for spec in "$@"; do
    echo "-- round $r [$spec]" >> $out
    run "$spec" timeout -k 10 120 python tools/flash_ablate.py 2>&1 | grep -v amdgpu.ids >> $out || exit 1
  done
done
grep -v "^OK" $out | tail -60
