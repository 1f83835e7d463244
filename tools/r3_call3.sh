set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_flux_widening.py -x -q -s > gpurun_out/r3_c3_widening.log 2>&1 || { tail -60 gpurun_out/r3_c3_widening.log; exit 1; }
grep -E "rel L2|centre of mass|crop \+ KL|passed|failed" gpurun_out/r3_c3_widening.log
ARTIST_BENCH_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3_c3_bench_2rank_gloo.log 2> gpurun_out/r3_c3_bench_2rank_gloo.err || { tail -30 gpurun_out/r3_c3_bench_2rank_gloo.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_c3_bench_2rank_gloo.log').read().strip().splitlines()[-1])
s=d['sharding_check']; print('2-rank', d['n_gpus'], d['ms_per_step'], {k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk!='down16'}) for k,v in s.items()})
PY
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3_c3_bench_1rank.log 2>&1
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_c3_bench_1rank.log').read().strip().splitlines()[-1])
s=d['sharding_check']; print('1-rank', d['ms_per_step'], {kk:vv for kk,vv in s['reduced_flux'].items() if kk!='down16'}, d['check'])
PY
