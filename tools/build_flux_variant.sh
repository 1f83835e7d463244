#!/bin/bash
# usage: bash tools/build_flux_variant.sh NAME [-DFLAG ...]  ->  tools/bin/libflux_NAME.so (flux_kernels.hip rebuilt with the flags)
set -e
cd "$(dirname "$0")/../artist_amd/csrc"
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fno-slp-vectorize "$@" -c flux_kernels.hip -o /tmp/fvar_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/libflux_$name.so /tmp/fvar_$name.o trace_kernels.o blocking_kernels.o nurbs_kernels.o align_kernels.o kinematics_kernels.o capi.o
