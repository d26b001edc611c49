#!/bin/bash
# Pipe utilisation of the fused GEMM per launch class (tools/gemm_bench.py shapes): three --pmc passes (counters only, program
# directly after --), reduced by tools/pmc_gemm.py.   usage (through gpurun): bash tools/pmc_gemm.sh <tag>
tag=${1:-pmc}
out=gpurun_out/pmc_$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export GB_ITERS=4 GB_SHAPES=0,1,3,6,7   # 16384x320x320, 16384x320x2560, 4096x640x640, 1024x1280x1280, 1024x1280x10240
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$out/p1" -- python3 tools/gemm_bench.py > "$out/p1.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc TA_TA_BUSY_sum GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA --output-format csv -d "$out/p2" -- python3 tools/gemm_bench.py > "$out/p2.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d "$out/p3" -- python3 tools/gemm_bench.py > "$out/p3.log" 2>&1
python3 tools/pmc_gemm.py $(ls "$out"/p*/*/*_counter_collection.csv 2>/dev/null) > "$out/table.txt" 2>&1
rm -rf "$out/p1" "$out/p2" "$out/p3"
cat "$out/table.txt"
