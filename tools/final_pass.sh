# The round's closing measurements, one gpurun call (GPU box): tests, bench lines, profiles.  Output under gpurun_out/final_*.
# (kept going when a step fails: each step writes its own file; `cp`-ing the summaries into profiles/ is tools/collect_final.py)
cd $GRAFT_REPO_ROOT
# (the tests run in their default mode - tests/conftest.py switches the library's debug knobs on, some tests set them -; everything
#  measured below runs the product as shipped: knobs ignored)
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 300 > gpurun_out/final_gpu_tests.log 2>&1; tail -2 gpurun_out/final_gpu_tests.log
export ARTIST_HIP_DEBUG=0
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final_bench.log 2>&1; tail -1 gpurun_out/final_bench.log > gpurun_out/final_bench.json; cut -c1-300 gpurun_out/final_bench.json
timeout -k 10 300 python tools/config_bench.py 2>/dev/null | grep "^{" > gpurun_out/final_config_bench.jsonl
timeout -k 10 300 python bench.py --heliostats 100 --rays 180 --n-cp 6 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/final_bench_config4.json
timeout -k 10 300 python bench.py --heliostats 125 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/final_bench_h125.json; cut -c1-200 gpurun_out/final_bench_h125.json
ARTIST_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/final_bench_2rank_gloo.log 2> gpurun_out/final_bench_2rank_gloo.err || echo "2-rank rehearsal failed"
ARTIST_AMD_COLLECTIVES_AT_WORLD_1=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/final_bench_rccl_world1.log 2>&1 || echo "rccl world-1 rehearsal failed"
timeout -k 10 300 python tools/blocking_bench.py 2>/dev/null | tail -1 > gpurun_out/final_blocking_bench.json
timeout -k 10 400 python tools/blocking_bench.py --dense 2>/dev/null | tail -1 > gpurun_out/final_blocking_bench_dense.json
timeout -k 10 200 python tools/adam_bench.py 2>/dev/null | tail -1 > gpurun_out/final_adam_bench.json
timeout -k 10 200 python tools/per_target_bench.py 2>/dev/null | tail -1 > gpurun_out/final_per_target_bench.json
timeout -k 10 200 python tools/parity_margins.py 2>/dev/null | grep "hip-oracle" > gpurun_out/final_parity_margins.txt
for B in 1000 125; do timeout -k 10 300 python tools/flux_bench.py $B 2>/dev/null | tail -1 > gpurun_out/final_flux_bench_$B.json; done
timeout -k 10 300 python tools/flux_ab_r03.py 1000 > gpurun_out/final_flux_ab_r03.txt 2>/dev/null; timeout -k 10 300 python tools/flux_ab_r03.py 125 >> gpurun_out/final_flux_ab_r03.txt 2>/dev/null
timeout -k 10 300 python tools/cylinder_bench.py 2>/dev/null | tail -1 > gpurun_out/final_cylinder_bench.json
timeout -k 10 300 python tools/nurbs_bench.py 1000 125 2>/dev/null | tail -1 > gpurun_out/final_nurbs_bench.json
timeout -k 10 120 ./tools/bin/nurbs_mfma_bench > gpurun_out/final_nurbs_mfma_bench.json 2>&1
timeout -k 10 300 python tools/pipeline_probe.py 125 1 2 3 4 > gpurun_out/final_pipeline_probe.txt 2>/dev/null
bash tools/kstats.sh final > gpurun_out/final_kstats.txt 2>&1 || true
bash tools/kstats.sh final125 --heliostats 125 > gpurun_out/final_kstats_h125.txt 2>&1 || true
python tools/gap_report.py gpurun_out/ks_final125 8 > gpurun_out/final_gap_h125.txt 2>&1 || true
python tools/gap_report.py gpurun_out/ks_final 8 > gpurun_out/final_gap.txt 2>&1 || true
bash tools/prof_pass.sh final > /dev/null 2>&1 || true
bash tools/pmc_hbm.sh final > /dev/null 2>&1 || true
bash tools/pmc_lds.sh finallds > /dev/null 2>&1 || true
python tools/pmc_summary.py final > gpurun_out/final_pmc_summary.txt 2>&1 || true
echo done
