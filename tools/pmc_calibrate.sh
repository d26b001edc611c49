cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export GB_ITERS=4 GB_SHAPES=0,3,7
rm -rf gpurun_out/calib; mkdir -p gpurun_out/calib
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/calib/f -- python3 tools/gemm_bench.py --cold-read > gpurun_out/calib/f.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/calib/w -- python3 tools/gemm_bench.py --cold-read > gpurun_out/calib/w.log 2>&1
python3 tools/pmc_gemm.py $(ls gpurun_out/calib/*/*/*_counter_collection.csv) > gpurun_out/calib/table.txt 2>&1
rm -rf gpurun_out/calib/f gpurun_out/calib/w
cat gpurun_out/calib/table.txt
