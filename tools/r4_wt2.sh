#!/bin/bash
export ARTIST_HIP_DEBUG=1
for wt in 0 1; do
  export ARTIST_HIP_WINDOW_TABLE=$wt
  echo "== table $wt"; bash tools/kstats.sh wt$wt 2>&1 | grep "trace_fwd_lds\|trace_bwd_lds\|window_table\|ms_per_step"
done
