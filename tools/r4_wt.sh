#!/bin/bash
export ARTIST_HIP_DEBUG=1

for rep in 1 2; do for wt in 0 1; do for h in 250 500; do
  ARTIST_HIP_WINDOW_TABLE=$wt timeout -k 10 200 python bench.py --heliostats $h --steps 30 --warmup 8 --no-cpu-baseline --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('table $wt H', d['config']['heliostats'], 'step', round(d['ms_per_step'],4), 'fwd', round(d['kernels']['trace_fwd_ms'],4), 'bwd', round(d['kernels']['trace_bwd_ms'],4))"
done; done; done
