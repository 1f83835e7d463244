# usage: bash tools/ablate_lib.sh NAME...  -> fwd/bwd kernel ms + step ms with artist_amd/libablate_NAME.so vs the product library
run() { for h in 1000 125; do timeout -k 10 200 python bench.py --heliostats $h --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['heliostats'], d['kernels']['trace_fwd_ms'], d['kernels']['trace_bwd_ms'], d['ms_per_step'])" || return 1; done; }
echo "baseline"; run || exit 1
for v in "$@"; do echo "lib=$v"; export ARTIST_HIP_LIB=$PWD/artist_amd/libablate_$v.so; run || exit 1; done
