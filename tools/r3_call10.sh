cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4 5 6; do timeout -k 10 300 python -m pytest tests -m gpu -x -q > gpurun_out/r3_c10_rep$rep.log 2>&1; tail -1 gpurun_out/r3_c10_rep$rep.log; done
grep -l "failed" gpurun_out/r3_c10_rep*.log | head
