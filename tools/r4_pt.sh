#!/bin/bash
for rep in 1 2; do for v in main pt32 pt64; do
  if [ $v = main ]; then unset ARTIST_HIP_LIB; else export ARTIST_HIP_LIB=$PWD/tools/bin/libw_$v.so; fi
  echo "== $v $(timeout -k 10 100 python tools/per_target_bench.py 2>/dev/null | tail -1)"
done; done
