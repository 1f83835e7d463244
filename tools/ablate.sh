#!/bin/bash
# Ablation builds of the trace kernels: what does each part of the ray body cost?  (diagnostic libraries only: results are wrong)
# usage: bash tools/ablate.sh build   (anywhere)   |   bash tools/ablate.sh run   (GPU box; prints fwd/bwd ms per variant)
set -e
cd "$(dirname "$0")/.."
VARIANTS="base:-DART_X=0 noatomics:-DART_ABLATE_NO_LDS_ATOMICS nostrays:-DART_ABLATE_NO_STRAYS noloads:-DART_ABLATE_NO_LOADS noflush:-DART_ABLATE_NO_FLUSH aluonly:-DART_ABLATE_NO_LDS_ATOMICS,-DART_ABLATE_NO_STRAYS,-DART_ABLATE_NO_LOADS"
if [ "$1" = build ]; then
  mkdir -p tools/bin
  for v in $VARIANTS; do
    name=${v%%:*}; defs=$(echo ${v#*:} | tr ',' ' ')
    ( cd artist_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fno-slp-vectorize \
        $defs -I. -c ../../tools/diag/trace_kernels_diag.hip -o /tmp/abl_$name.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/libabl_$name.so /tmp/abl_$name.o blocking_kernels.o flux_kernels.o nurbs_kernels.o align_kernels.o kinematics_kernels.o optim_kernels.o capi.o ) &
  done
  wait
  exit 0
fi
for v in $VARIANTS; do
  name=${v%%:*}
  out=$(ARTIST_HIP_LIB=$PWD/tools/bin/libabl_$name.so timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-check "${@:2}" 2>/dev/null | tail -1)
  echo "$name $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); k=d["kernels"]; print("fwd %.3f ms  bwd %.3f ms" % (k["trace_fwd_ms"], k["trace_bwd_ms"]))')"
done
