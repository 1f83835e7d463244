set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r3_c6_tests.log 2>&1 || { tail -60 gpurun_out/r3_c6_tests.log; exit 1; }
tail -1 gpurun_out/r3_c6_tests.log
timeout -k 10 400 python tools/blocking_bench.py 2>/dev/null | tail -1 > gpurun_out/r3_c6_blocking_bench.json; python -c "
import json; d=json.load(open('gpurun_out/r3_c6_blocking_bench.json')); print(json.dumps(d)[:1500])"
