#!/bin/bash
# HBM-side bytes of the flux epilogue kernels (tools/flux_bench.py, 1000 bitmaps): FETCH_SIZE / WRITE_SIZE in separate passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  t2=$(echo $c | cut -d' ' -f1)
  rm -rf $R/gpurun_out/fluxpmc_$t2
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/fluxpmc_$t2 -- python3 $R/tools/flux_bench.py 1000 > $R/gpurun_out/fluxpmc_$t2.log 2>&1 || echo "fail $t2"
done
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
for t2 in ("FETCH_SIZE","WRITE_SIZE","TCC_HIT_sum"):
    files=glob.glob(f"{R}/gpurun_out/fluxpmc_{t2}/**/*counter_collection.csv", recursive=True)
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            if "art::" in k and ("crop" in k or "loss" in k): acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        print(t2, k, {c: round(sum(x)/len(x),1) for c,x in v.items()}, "n", len(next(iter(v.values()))))
PY
