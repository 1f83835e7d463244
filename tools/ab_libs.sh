# same-box A/B of library builds: usage  bash tools/ab_libs.sh name...   (artist_amd/libablate_<name>.so; "main" = the product library)
run() { for h in 1000 125; do timeout -k 10 200 python bench.py --heliostats $h --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  H', d['config']['heliostats'], 'fwd', round(d['kernels']['trace_fwd_ms'],4), 'bwd', round(d['kernels']['trace_bwd_ms'],4), 'step', round(d['ms_per_step'],4))" || return 1; done
  timeout -k 10 300 python tools/config_bench.py 2>/dev/null | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    c=json.loads(l); print('  ', c['config'][:34], round(c['per_heliostat+segment_sum']['ms'],3), round(c['fused_per_target']['ms'],3))
"; }
for round in 1 2; do for v in "$@"; do echo "lib=$v"; if [ $v = main ]; then unset ARTIST_HIP_LIB; else export ARTIST_HIP_LIB=$PWD/artist_amd/libablate_$v.so; fi; run || exit 1; done; done
