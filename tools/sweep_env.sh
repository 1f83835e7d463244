# usage: bash tools/sweep_env.sh VAR v1 v2 ...   -> fwd/bwd kernel ms at H=1000 and H=125 for each value
var=$1; shift
for v in "$@"; do echo "$var=$v"; export $var=$v; for h in 1000 125; do timeout -k 10 200 python bench.py --heliostats $h --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['heliostats'], d['kernels']['trace_fwd_ms'], d['kernels']['trace_bwd_ms'], d['ms_per_step'])" || exit 1; done; done
