#!/bin/bash
export ARTIST_HIP_DEBUG=1   # the library reads its ARTIST_HIP_* knobs only in debug mode
# Where does a forward workgroup's time go?  Builds a diagnostic library (-DART_DEBUG_TIMELINE: every forward workgroup
# stamps its phases with the 100 MHz real-time counter), runs one forward trace of the metric field on the GPU box and
# prints per-phase medians + the gap a CU leaves between two workgroups.
# usage: bash tools/timeline.sh build   (anywhere: hipcc cross-compiles)   then, on the GPU box: bash tools/timeline.sh run [H]
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
( cd artist_amd/csrc && for f in trace_kernels blocking_kernels flux_kernels nurbs_kernels align_kernels kinematics_kernels optim_kernels capi; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fno-slp-vectorize \
      -DART_DEBUG_TIMELINE $ART_EXTRA_DEFS -I. -c $( [ $f = trace_kernels ] && echo ../../tools/diag/trace_kernels_diag.hip || echo $f.hip ) -o /tmp/tl_$f.o; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libtimeline.so /tmp/tl_*.o )
exit 0
fi
H=${2:-1000}
ARTIST_HIP_LIB=$PWD/artist_amd/libtimeline.so ART_TIMELINE_OUT=/tmp/timeline.bin ART_TIMELINE_OUT_BWD=/tmp/timeline_bwd.bin \
  timeout -k 10 300 python bench.py --heliostats $H --steps 2 --warmup 1 --no-cpu-baseline --no-check > /dev/null
echo "forward:"; if [ "${ARTIST_HIP_LEAN:-1}" = 1 ]; then python tools/timeline_lean_report.py /tmp/timeline.bin; else python tools/timeline_report.py /tmp/timeline.bin; fi
echo "backward (phases: window, staging of dL/dflux, -, trace, -):"; python tools/timeline_report.py /tmp/timeline_bwd.bin
