set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r3_c4_tests.log 2>&1 || { tail -40 gpurun_out/r3_c4_tests.log; exit 1; }
tail -1 gpurun_out/r3_c4_tests.log
for rep in 1 2; do for ws in 0 1; do echo "WINDOW_SAMPLE=$ws"; ARTIST_HIP_WINDOW_SAMPLE=$ws bash tools/h125.sh; done; done
for ws in 0 1; do echo "WINDOW_SAMPLE=$ws"; ARTIST_HIP_WINDOW_SAMPLE=$ws timeout -k 10 300 python tools/config_bench.py 2>/dev/null | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    c=json.loads(l); print('  ', c['config'][:40], round(c['per_heliostat+segment_sum']['ms'],3), round(c['fused_per_target']['ms'],3))
"; done
