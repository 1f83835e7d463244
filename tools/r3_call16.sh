cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_boundary.py -x -q -k "lds_held" 2>&1 | tail -3
timeout -k 10 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for t in 0 1; do echo "TAIL=$t"; ARTIST_HIP_TAIL=$t bash tools/h125.sh; done
for t in 0 1; do echo "TAIL=$t"; ARTIST_HIP_TAIL=$t bash tools/h125.sh; done
