set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_boundary.py -x -q > gpurun_out/r3_c2_boundary.log 2>&1 || { tail -40 gpurun_out/r3_c2_boundary.log; exit 1; }
tail -2 gpurun_out/r3_c2_boundary.log
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r3_c2_tests.log 2>&1 || { tail -40 gpurun_out/r3_c2_tests.log; exit 1; }
tail -2 gpurun_out/r3_c2_tests.log
bash tools/h125.sh > gpurun_out/r3_c2_h125.txt 2>&1; cat gpurun_out/r3_c2_h125.txt
