// sqrt_check.hip - TEST INFRASTRUCTURE (tests/test_gpu_parity.py::test_scale_free_square_root_is_the_ieee_square_root).
// Counts the floats in [lo_bits, hi_bits) for which art::sqrt_noscale differs from sqrtf in any bit.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC -o tests/bin/libsqrt_check.so tests/sqrt_check.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../artist_amd/csrc/ray_math.hpp"

__global__ void sqrt_check_kernel(unsigned lo_bits, unsigned long long n, unsigned long long* __restrict__ mismatches)
{
    unsigned long long bad = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = __builtin_bit_cast(float, (unsigned)(lo_bits + i));
        const float a = art::sqrt_noscale(x), b = sqrtf(x);
        bad += __builtin_bit_cast(unsigned, a) != __builtin_bit_cast(unsigned, b);
    }
    if (bad) atomicAdd(mismatches, bad);
}

extern "C" int sqrt_check(unsigned lo_bits, unsigned long long n, unsigned long long* mismatches, void* stream_)
{
    hipLaunchKernelGGL(sqrt_check_kernel, dim3(4096), dim3(256), 0, static_cast<hipStream_t>(stream_), lo_bits, n, mismatches);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
