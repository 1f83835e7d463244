set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r3_base_tests.log 2>&1; tail -2 gpurun_out/r3_base_tests.log
timeout -k 10 300 python bench.py > gpurun_out/r3_base_bench.log 2>&1; tail -1 gpurun_out/r3_base_bench.log | cut -c1-400
bash tools/h125.sh > gpurun_out/r3_base_h125.txt 2>&1; cat gpurun_out/r3_base_h125.txt
bash tools/pmc_lds.sh r3lds; grep -A12 "trace_" gpurun_out/pmc_r3lds_summary.txt | head -60
