#!/bin/bash
export ARTIST_HIP_DEBUG=1   # the library reads its ARTIST_HIP_* knobs only in debug mode
# Workgroup size / point-block sweep of the windowed trace kernels on the metric field (same box, back to back).
# usage (GPU box): bash tools/sweep_block.sh > gpurun_out/sweep_block.txt
cd "$(dirname "$0")/.."
run() {  # label, env assignments...
  label=$1; shift
  out=$(env "$@" timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1)
  echo "$label $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); k=d["kernels"]; print("fwd %.3f ms  bwd %.3f ms  step %.3f ms" % (k["trace_fwd_ms"], k["trace_bwd_ms"], d["ms_per_step"]))')"
}
run "block=1024 (default)        " ARTIST_X=0
run "block=768  pb=768/1536      " ARTIST_HIP_FWD_BLOCK=768 ARTIST_HIP_FWD_PBLOCK=768 ARTIST_HIP_BWD_PBLOCK=1536
run "block=768  pb=715/1429 exact" ARTIST_HIP_PBLOCK_EXACT=1 ARTIST_HIP_FWD_BLOCK=768 ARTIST_HIP_FWD_PBLOCK=715 ARTIST_HIP_BWD_PBLOCK=1429
run "block=768  pb=1536/2304     " ARTIST_HIP_FWD_BLOCK=768 ARTIST_HIP_FWD_PBLOCK=1536 ARTIST_HIP_BWD_PBLOCK=2304
run "block=512  pb=1024/2048     " ARTIST_HIP_FWD_BLOCK=512 ARTIST_HIP_FWD_PBLOCK=1024 ARTIST_HIP_BWD_PBLOCK=2048
run "block=1024 (default, again) " ARTIST_X=0
