#!/bin/bash
# usage: tools/gpu_stage.sh <name> <timeout_s> <cmd...>   — runs one GPU step under a timeout, logs to gpurun_out/<name>.log,
# and refuses to start if an earlier step of this call timed out or was killed (gpurun_out/.stage_dead).
name=$1; shift; tmo=$1; shift
mkdir -p gpurun_out
if [ -f gpurun_out/.stage_dead ]; then echo "[stage $name] skipped: an earlier step was killed"; exit 1; fi
echo "[stage $name] $(date +%H:%M:%S) start: $*"
timeout -k 10 "$tmo" "$@" > "gpurun_out/$name.log" 2>&1
rc=$?
echo "[stage $name] $(date +%H:%M:%S) rc=$rc"
if [ $rc -ge 124 ]; then touch gpurun_out/.stage_dead; fi
tail -n 6 "gpurun_out/$name.log"
exit 0
