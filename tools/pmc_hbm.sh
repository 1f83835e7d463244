#!/bin/bash
# HBM traffic counters (separate passes: FETCH_SIZE and WRITE_SIZE do not fit one TCC pass on gfx950).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=$1; shift
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum"; do
  t2=$(echo $c | cut -d' ' -f1)
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/hbm_${tag}_$t2 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/hbm_${tag}_$t2.log 2>&1 || echo "fail $t2"
done
