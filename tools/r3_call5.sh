set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_configs.py -x -q -k "field_groups" > gpurun_out/r3_c5_field.log 2>&1 || { tail -40 gpurun_out/r3_c5_field.log; exit 1; }
tail -1 gpurun_out/r3_c5_field.log
for g in 0 -1 8 16; do echo "FIELD_GROUP=$g"; ARTIST_HIP_FIELD_GROUP=$g ARTIST_HIP_PRINT_GEOMETRY=0 timeout -k 10 300 python tools/config_bench.py 2>/dev/null | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    c=json.loads(l); print('  ', c['config'][:40], round(c['per_heliostat+segment_sum']['ms'],3), round(c['fused_per_target']['ms'],3), c['fused_per_target_sum'])
"; done
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r3_c5_tests.log 2>&1 || { tail -40 gpurun_out/r3_c5_tests.log; exit 1; }
tail -1 gpurun_out/r3_c5_tests.log
bash tools/h125.sh
