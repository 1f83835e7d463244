#!/bin/bash
# Per-kernel average durations of a short bench run (rocprofv3 --kernel-trace --stats), top 16.  GPU box only.
# usage: bash tools/kstats.sh <tag> [bench args]
R=$GRAFT_REPO_ROOT; tag=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_$tag -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-check "$@" > $R/gpurun_out/ks_$tag.log 2>&1 || echo "rocprof failed"
f=$(find $R/gpurun_out/ks_$tag -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f}")
PY
