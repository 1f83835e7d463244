#!/bin/bash
for v in head inl new head inl new; do
  if [ $v = head ]; then export ARTIST_HIP_LIB=$PWD/tools/bin/libhead_r4.so; elif [ $v = new ]; then unset ARTIST_HIP_LIB; else export ARTIST_HIP_LIB=$PWD/tools/bin/libw_$v.so; fi
  echo "== $v"; ART_BLOCKING_CANDIDATES=32 timeout -k 10 200 python tools/blocking_bench.py 2>&1 | tail -1 | grep -o '"blocking_exact": {"fwd_ms": [0-9.]*, "fwd_bwd_ms": [0-9.]*\|"filtered": [0-9]*'
done
