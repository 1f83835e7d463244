#!/bin/bash
for rep in 1 2; do for v in main g16; do for h in 125 1000; do
  if [ $v = main ]; then unset ARTIST_HIP_LIB; else export ARTIST_HIP_LIB=$PWD/tools/bin/libw_$v.so; fi
  timeout -k 10 200 python bench.py --heliostats $h --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v H', d['config']['heliostats'], 'step', round(d['ms_per_step'],4), 'fwd', round(d['kernels']['trace_fwd_ms'],4), 'bwd', round(d['kernels']['trace_bwd_ms'],4), 'check', d['check']['flux_rel_l2'] if d.get('check') else None, d['check']['ray_counters_equal'] if d.get('check') else None)"
done; done; done
ARTIST_HIP_LIB=$PWD/tools/bin/libw_g16.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
