# One gpurun call: GPU tests, the bench line, one rank's share (125 heliostats), kernel stats of both.  usage: bash tools/r4_pass.sh <tag> [notests|"pytest -k expr"]
cd $GRAFT_REPO_ROOT
tag=$1
if [ "$2" != "notests" ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -x -v --tb=short --timeout 240 --durations=12 ${2:+-k "$2"} > gpurun_out/${tag}_gpu_tests.log 2>&1
  grep -E "FAILED|ERROR|Timeout| passed| failed|^E  " gpurun_out/${tag}_gpu_tests.log | cut -c1-300 | tail -15
fi
timeout -k 10 300 python tools/nurbs_bench.py 1000 125 2> gpurun_out/${tag}_nurbs_bench.err | tail -1 > gpurun_out/${tag}_nurbs_bench.json; cat gpurun_out/${tag}_nurbs_bench.json; tail -3 gpurun_out/${tag}_nurbs_bench.err
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_bench.log 2>&1; tail -1 gpurun_out/${tag}_bench.log > gpurun_out/${tag}_bench.json; cut -c1-600 gpurun_out/${tag}_bench.json
timeout -k 10 300 python bench.py --heliostats 125 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_h125.json; cut -c1-600 gpurun_out/${tag}_bench_h125.json
timeout -k 10 300 python tools/pipeline_probe.py 125 1 2 3 4 > gpurun_out/${tag}_pipeline_probe.txt 2>&1; tail -5 gpurun_out/${tag}_pipeline_probe.txt
bash tools/kstats.sh ${tag} > gpurun_out/${tag}_kstats.txt 2>&1 || true
bash tools/kstats.sh ${tag}125 --heliostats 125 > gpurun_out/${tag}_kstats_h125.txt 2>&1 || true
python tools/gap_report.py gpurun_out/ks_${tag}125 8 > gpurun_out/${tag}_gap_h125.txt 2>&1 || true
cat gpurun_out/${tag}_kstats.txt gpurun_out/${tag}_kstats_h125.txt gpurun_out/${tag}_gap_h125.txt
echo done
