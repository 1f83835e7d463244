#!/bin/bash
# per-kernel durations of the blocking bench: the library before the wide-list change vs now
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in head new; do
  rm -rf $R/gpurun_out/profw
  if [ $v = head ]; then export ARTIST_HIP_LIB=$R/tools/bin/libhead_r4.so; else unset ARTIST_HIP_LIB; fi
  ART_BLOCKING_CANDIDATES=32 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/profw -- python3 $R/tools/blocking_bench.py > $R/gpurun_out/profw_$v.log 2>&1 || echo fail
  f=$(find $R/gpurun_out/profw -name "*kernel_stats.csv" | head -1)
  cp $f $R/gpurun_out/wide_kernel_stats_$v.csv
  rm -rf $R/gpurun_out/profw
  echo "== $v"; cut -c1-160 $R/gpurun_out/wide_kernel_stats_$v.csv | head -14
done
