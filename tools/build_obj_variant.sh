#!/bin/bash
# usage: bash tools/build_obj_variant.sh FILE NAME [-DFLAG ...]  ->  tools/bin/lib<FILE>_<NAME>.so (artist_amd/csrc/<FILE>.hip rebuilt with the
# flags, the other objects as built by `make`; FILE = trace_kernels builds the INSTRUMENTED copy tools/diag/trace_kernels_diag.hip - the
# shipped source has no build-time variants)
set -e
cd "$(dirname "$0")/../artist_amd/csrc"
file=$1; name=$2; shift; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fno-slp-vectorize "$@" -I. -c $( [ $file = trace_kernels ] && echo ../../tools/diag/trace_kernels_diag.hip || echo $file.hip ) -o /tmp/ovar_${file}_$name.o
objs=""
for f in trace_kernels blocking_kernels flux_kernels nurbs_kernels align_kernels kinematics_kernels optim_kernels capi; do
  if [ $f = $file ]; then objs="$objs /tmp/ovar_${file}_$name.o"; else objs="$objs $f.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/lib${file}_$name.so $objs
