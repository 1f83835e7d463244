# GPU tests, verbose with per-test timeout (stops at the first failure or hang).  usage: bash tools/r4_tests.sh <tag> [-k expr]
cd $GRAFT_REPO_ROOT
tag=$1; shift
timeout -k 10 1000 python -m pytest tests -m gpu -x -v --tb=short --timeout 180 --durations=15 "$@" > gpurun_out/${tag}_gpu_tests.log 2>&1
grep -E "PASSED|FAILED|ERROR|Timeout|passed|failed|^E  " gpurun_out/${tag}_gpu_tests.log | cut -c1-300 | tail -40
