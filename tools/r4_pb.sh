#!/bin/bash
export ARTIST_HIP_DEBUG=1
run() { timeout -k 10 200 python bench.py --heliostats 125 --steps 40 --warmup 10 --no-cpu-baseline --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'step', round(d['ms_per_step'],4), 'fwd', round(d['kernels']['trace_fwd_ms'],4), 'bwd', round(d['kernels']['trace_bwd_ms'],4))"; }
for rep in 1 2; do
  run base
  ARTIST_HIP_BWD_PBLOCK=1000 run bwd1000
  ARTIST_HIP_BWD_PBLOCK=834 run bwd834
  ARTIST_HIP_BWD_PBLOCK=625 run bwd625
  ARTIST_HIP_FWD_PBLOCK=1000 run fwd1000
  ARTIST_HIP_FWD_PBLOCK=834 run fwd834
  ARTIST_HIP_TAIL=2 run tail2
done
