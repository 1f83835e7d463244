set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/final_gpu_tests.log 2>&1; tail -2 gpurun_out/final_gpu_tests.log
timeout -k 10 300 python bench.py > gpurun_out/final_bench.log 2>&1; tail -1 gpurun_out/final_bench.log > gpurun_out/final_bench.json; cut -c1-300 gpurun_out/final_bench.json
timeout -k 10 300 python tools/config_bench.py 2>/dev/null | grep "^{" > gpurun_out/final_config_bench.jsonl
timeout -k 10 300 python bench.py --heliostats 100 --rays 180 --n-cp 6 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/final_bench_config4.json
timeout -k 10 300 python bench.py --heliostats 125 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/final_bench_h125.json
timeout -k 10 300 python tools/blocking_bench.py 2>/dev/null | tail -1 > gpurun_out/final_blocking_bench.json
timeout -k 10 300 python tools/flux_bench.py 2>/dev/null | tail -1 > gpurun_out/final_flux_bench.json
bash tools/prof_pass.sh final > /dev/null 2>&1 || true
bash tools/pmc_hbm.sh final > /dev/null 2>&1 || true
echo done
