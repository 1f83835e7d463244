#!/bin/bash
# the GPU tests, then the bench at the metric size and at one rank's share
set -o pipefail
mkdir -p gpurun_out
bash tools/r4_tests.sh r4x || exit 1
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r4x_bench.json
timeout -k 10 300 python bench.py --heliostats 125 --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r4x_bench_h125.json
python - <<'PY'
import json
for f in ("gpurun_out/r4x_bench.json","gpurun_out/r4x_bench_h125.json"):
    d=json.loads(open(f).read())
    print(f, "ms/step", round(d["ms_per_step"],4), "value", d["value"], {k:(round(v,4) if isinstance(v,float) else v) for k,v in d.get("kernels",{}).items() if k.endswith("_ms")}, d.get("step_events"))
PY
