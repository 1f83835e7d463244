cd $GRAFT_REPO_ROOT
for rep in $(seq 10); do timeout -k 10 300 python -m pytest tests -m gpu -x -q > gpurun_out/r3_c14_rep.log 2>&1; tail -1 gpurun_out/r3_c14_rep.log | cut -c1-60; if grep -q "failed" gpurun_out/r3_c14_rep.log; then grep -n "AssertionError" gpurun_out/r3_c14_rep.log | cut -c1-1500; break; fi; done
