cd $GRAFT_REPO_ROOT
echo "== five gpu files x8"; for rep in $(seq 8); do timeout -k 10 300 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_configs.py tests/test_gpu_flux_widening.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -x -q 2>&1 | tail -1; done | cut -c1-40 | sort | uniq -c
echo "== tests -m gpu x4"; for rep in $(seq 4); do timeout -k 10 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -1; done | cut -c1-40 | sort | uniq -c
