#!/bin/bash
# Round-end evidence in ONE gpurun call: the full GPU suite, smoke(), then the profile collection (tools/collect_profiles.sh).
# usage (through gpurun): bash tools/round_end_check.sh <tag>      e.g. r05_final
tag=${1:-r05_final}
mkdir -p gpurun_out && rm -f gpurun_out/.stage_dead
tools/gpu_stage.sh ${tag}_tests 1150 python -m pytest tests -m gpu -q -x
tools/gpu_stage.sh ${tag}_smoke 200 python __graft_entry__.py --smoke
if [ -f gpurun_out/.stage_dead ]; then echo "a stage was killed: no profile collection"; exit 1; fi
bash tools/collect_profiles.sh $tag > gpurun_out/${tag}_collect.log 2>&1
tail -15 gpurun_out/${tag}_collect.log
