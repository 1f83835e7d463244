cd $GRAFT_REPO_ROOT
echo "== without the fix (diagnostic build): expected to FAIL"; ARTIST_HIP_LIB=$PWD/tools/bin/libvar_nofix.so timeout -k 10 300 python -m pytest tests/test_gpu_boundary.py -x -q -k "lds_held" 2>&1 | tail -4 | cut -c1-300
echo "== shipped"; timeout -k 10 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
bash tools/h125.sh
