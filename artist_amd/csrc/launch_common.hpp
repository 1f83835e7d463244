// launch_common.hpp - error plumbing + small wave-level helpers shared by the kernel files.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/artist_hip.h"

namespace art {

// hipError_t of the last failing HIP call on this thread (art_last_hip_error()).
extern thread_local int g_last_hip_error;

#define ART_HIP(expr)                                   \
    do {                                                \
        hipError_t err__ = (expr);                      \
        if (err__ != hipSuccess) {                      \
            art::g_last_hip_error = (int)err__;         \
            return ART_ELAUNCH;                         \
        }                                               \
    } while (0)

// 64-lane wave sum (gfx950 wavefront = 64); result valid in lane 0.
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__device__ __forceinline__ float wave_sum_f32(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

}  // namespace art
