#!/bin/bash
# Collects the round's roofline evidence on the GPU box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the default bench command
#   2-4. three SEPARATE --pmc passes (FETCH_SIZE; WRITE_SIZE; MFMA-busy counters) of a short host-launched run
#   5. the plain default bench line (with cpu_baseline), run last so that it picks up the fresh PMC summary
# and reduces them to the small summaries that are committed under profiles/ (tools/summarize_profile.py).
# usage: tools/collect_profiles.sh <tag>      e.g. r02_final
set -u
tag=${1:-r02}
out=gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { name=$1; shift; echo "[prof $name] $(date +%H:%M:%S) $*"; timeout -k 10 700 "$@" > "$out/$name.log" 2>&1; echo "[prof $name] rc=$?"; }
run trace   rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra
run fetch   rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-prof --no-graph
run write   rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-prof --no-graph
run mfma    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$out/pmc_mfma" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-prof --no-graph
run sq1     rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d "$out/pmc_sq1" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-prof --no-graph
run sq2     rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$out/pmc_sq2" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-prof --no-graph
t=$(ls "$out"/trace/*/*_kernel_trace.csv | head -1)
f=$(ls "$out"/pmc_fetch/*/*_counter_collection.csv | head -1)
w=$(ls "$out"/pmc_write/*/*_counter_collection.csv | head -1)
m=$(ls "$out"/pmc_mfma/*/*_counter_collection.csv | head -1)
mkdir -p "$out/summary"
python3 tools/summarize_profile.py trace "$t" 9 "$out/summary/${tag}_timed_region_kernel_stats.csv" > /dev/null
cp "$(ls "$out"/trace/*/*_kernel_stats.csv | head -1)" "$out/summary/${tag}_rocprofv3_kernel_stats_full_process.csv"
python3 tools/summarize_profile.py shapes "$t" 9 > "$out/summary/${tag}_inmodel_launch_classes.txt"
python3 tools/summarize_profile.py pmc "$f" "$w" "$out/summary/${tag}_pmc_traffic.json" "$m" > /dev/null
python3 tools/summarize_profile.py sq $(ls "$out"/pmc_sq1/*/*_counter_collection.csv "$out"/pmc_sq2/*/*_counter_collection.csv 2>/dev/null) > "$out/summary/${tag}_sq_wait_counters.log" 2>&1
# the plain bench line LAST, with the fresh PMC summary in place, so that its roofline.traffic is this run's measurement
cp "$out/summary/${tag}_pmc_traffic.json" profiles/
run bench   python3 bench.py --steps 10 --warmup 3
grep '^{"metric"' "$out/bench.log" > "$out/summary/${tag}_bench.json"
# the raw counter CSVs are large: keep only the summaries (gpurun_out merges back <= 64 MiB)
rm -rf "$out/trace" "$out/pmc_fetch" "$out/pmc_write" "$out/pmc_mfma" "$out/pmc_sq1" "$out/pmc_sq2"
ls -la "$out/summary"
