# kernel durations of tools/blocking_bench.py (exact mode + reference-tree mode)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/profb
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/profb -- python3 $R/tools/blocking_bench.py > $R/gpurun_out/profb.log 2>&1 || echo fail
f=$(find $R/gpurun_out/profb -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/blocking_kernel_stats.csv
find $R/gpurun_out/profb -name "*kernel_trace.csv" -exec cp {} $R/gpurun_out/blocking_kernel_trace.csv \;
rm -rf $R/gpurun_out/profb
cut -c1-200 $R/gpurun_out/blocking_kernel_stats.csv | head -30
