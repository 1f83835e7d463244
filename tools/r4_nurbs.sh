cd $GRAFT_REPO_ROOT
tag=$1
timeout -k 10 600 python -m pytest tests -m gpu -x -q --tb=short --timeout 240 -k "nurbs or control_point or facet_sized" > gpurun_out/${tag}_tests.log 2>&1
grep -E "FAILED|ERROR|Timeout| passed| failed|^E  " gpurun_out/${tag}_tests.log | cut -c1-300 | tail -12
timeout -k 10 300 python tools/nurbs_bench.py 1000 125 2> gpurun_out/${tag}_nurbs_bench.err | tail -1 > gpurun_out/${tag}_nurbs_bench.json
python - <<PY
import json
d=json.load(open("gpurun_out/${tag}_nurbs_bench.json"))
for c in d["cases"]:
    print(c["heliostats"], c["alignment"], "TP", c["tensor_product"], "SC", c["scattered"], c["points_bit_equal"], c["normals_bit_equal"], "%.1e"%c["grad_rel_l2"])
PY
