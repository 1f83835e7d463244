#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R && timeout -k 10 200 python -m pytest tests/test_gpu_optim.py -m gpu -x -q 2>&1 | tail -2
cd /tmp
rm -rf $R/gpurun_out/profa
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/profa -- python3 $R/tools/adam_bench.py > $R/gpurun_out/adam_bench.log 2>&1 || echo fail
f=$(find $R/gpurun_out/profa -name "*kernel_stats.csv" | head -1)
grep -i "adam\|multi_tensor" $f | cut -c1-200
t=$(find $R/gpurun_out/profa -name "*kernel_trace.csv" | head -1)
python3 - $t <<'PY'
import csv,sys,collections
d=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "adam_step" in r["Kernel_Name"]:
        d[r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X","?")].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in d.items(): print("grid",k,"n",len(v),"median us", sorted(v)[len(v)//2])
PY
tail -1 $R/gpurun_out/adam_bench.log
rm -rf $R/gpurun_out/profa
