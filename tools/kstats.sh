#!/bin/bash
# Per-kernel average durations of a short bench run (rocprofv3 --kernel-trace --stats), top 16.  GPU box only.
# usage: bash tools/kstats.sh <tag> [bench args]
R=$GRAFT_REPO_ROOT; tag=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_$tag -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-check "$@" > $R/gpurun_out/ks_$tag.log 2>&1 || echo "rocprof failed"
f=$(find $R/gpurun_out/ks_$tag -name "*kernel_stats.csv" | head -1)
t=$(find $R/gpurun_out/ks_$tag -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$t" "$R/gpurun_out/ks_$tag.log" <<'PY'
import csv,json,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f}")
# The dominant kernel's launches INSIDE bench.py's timed region (its backward launches number warmup .. warmup + steps - 1, in
# start order) against the figure the same run's JSON line carries (HIP events around the same launches: roofline.ms_per_launch)
line = [l for l in open(sys.argv[3]) if l.startswith("{")]
if line and sys.argv[2]:
    d = json.loads(line[-1]); W, K = d["warmup"], d["steps"]
    bwd = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(sys.argv[2]))
                 if "trace_bwd_lds_kernel" in r["Kernel_Name"])
    timed = [(e - s) / 1e3 for s, e in bwd[W:W + K]]
    print(f"trace_bwd_lds_kernel, the {len(timed)} launches of the timed region (kernel trace): mean {sum(timed) / len(timed):.1f} us "
          f"(min {min(timed):.1f}, max {max(timed):.1f}); the run's own JSON line, HIP events around the same launches: "
          f"{d['roofline']['ms_per_launch'] * 1e3:.1f} us; ms_per_step {d['ms_per_step']:.3f}")
PY
