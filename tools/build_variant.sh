#!/bin/bash
# usage: bash tools/build_variant.sh NAME [-DFLAG ...]  ->  tools/bin/libvar_NAME.so (trace_kernels.hip rebuilt with the flags)
set -e
cd "$(dirname "$0")/../artist_amd/csrc"
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fno-slp-vectorize "$@" -c trace_kernels.hip -o /tmp/var_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/libvar_$name.so /tmp/var_$name.o blocking_kernels.o flux_kernels.o nurbs_kernels.o align_kernels.o kinematics_kernels.o capi.o
