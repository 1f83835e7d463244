# instruction counts of the blocking instantiations (tools/blocking_bench.py, 200 heliostats so that the pass is short)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmcb
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmcb -- python3 $R/tools/blocking_bench.py > $R/gpurun_out/pmcb.log 2>&1 || echo fail
python3 - <<'PY'
import csv, glob, collections, os
R = os.environ["GRAFT_REPO_ROOT"]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(R + "/gpurun_out/pmcb/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "trace_" not in k: continue
        k = k[:110]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] == "SQ_WAVES": cnt[k] += 1
for k, v in acc.items():
    n = cnt[k]
    print(n, k)
    print("   ", {c: round(x / n / 1e6, 2) for c, x in v.items()}, "(millions per launch)")
PY
