#!/bin/bash
for rep in 1 2; do for v in main "$@"; do
  if [ $v = main ]; then unset ARTIST_HIP_LIB; else export ARTIST_HIP_LIB=$PWD/tools/bin/libw_$v.so; fi
  echo "== $v"; timeout -k 10 200 python tools/flux_bench.py 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print({k:v['ms'] for k,v in d.items() if isinstance(v,dict) and ('crop' in k)})"
done; done
