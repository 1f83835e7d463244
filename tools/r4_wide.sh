#!/bin/bash
# wide candidate lists: the tests that touch blocking, then the blocking bench A/B against the library before the change
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py -m gpu -x -q -s -k "candidate or blocking" > gpurun_out/wide_tests.log 2>&1
echo "tests rc=$?"; tail -15 gpurun_out/wide_tests.log
bash tools/r4_wide_ab2.sh
bash tools/r4_wide_prof.sh > gpurun_out/wide_prof.log 2>&1
