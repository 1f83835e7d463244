#!/bin/bash
export ARTIST_HIP_DEBUG=1
for B in 125 64; do for parts in 1 2 4; do
  echo "B $B parts $parts: $(ARTIST_HIP_LOSS_PARTS=$parts timeout -k 10 200 python tools/flux_bench.py $B 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print({k:v['ms'] for k,v in d.items() if isinstance(v,dict) and ('crop_pixel' in k)})")"
done; done
