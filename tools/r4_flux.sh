cd $GRAFT_REPO_ROOT
tag=$1
timeout -k 10 600 python -m pytest tests -m gpu -x -q --tb=short --timeout 240 -k "crop or loss or flux_epilogue or poison or loop_converges or boundary" > gpurun_out/${tag}_tests.log 2>&1
grep -E "FAILED|ERROR|Timeout| passed| failed|^E  " gpurun_out/${tag}_tests.log | cut -c1-300 | tail -12
for B in 1000 125; do timeout -k 10 200 python tools/flux_bench.py $B 2>/dev/null | tail -1 > gpurun_out/${tag}_flux_bench_$B.json; python -c "
import json,sys; d=json.load(open('gpurun_out/${tag}_flux_bench_$B.json')); print($B, {k:v['ms'] for k,v in d.items() if isinstance(v,dict)})"; done
timeout -k 10 300 python bench.py --heliostats 125 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_h125.json; cut -c1-330 gpurun_out/${tag}_bench_h125.json
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${tag}_bench.json; cut -c1-330 gpurun_out/${tag}_bench.json
