#!/bin/bash
# usage: bash tools/build_variant.sh NAME [-DFLAG ...]  ->  tools/bin/libvar_NAME.so (the instrumented copy tools/diag/trace_kernels_diag.hip built with the flags)
set -e
cd "$(dirname "$0")/../artist_amd/csrc"
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fno-slp-vectorize "$@" -I. -c ../../tools/diag/trace_kernels_diag.hip -o /tmp/var_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/libvar_$name.so /tmp/var_$name.o blocking_kernels.o flux_kernels.o nurbs_kernels.o align_kernels.o kinematics_kernels.o optim_kernels.o capi.o
