// launch_common.hpp - error plumbing + small wave-level helpers shared by the kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>

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

// Diagnostic knobs (launch geometry, A/B paths; results are the same bits or within the documented rounding for every value):
// the library reads ARTIST_HIP_* variables ONLY when ARTIST_HIP_DEBUG=1 is set - a caller's environment cannot change what the
// product launches (tests and tools/ set both).
static inline const char* debug_env_str(const char* name)       // nullptr unless ARTIST_HIP_DEBUG=1 and the variable is set
{
    const char* dbg = getenv("ARTIST_HIP_DEBUG");
    if (dbg == nullptr || dbg[0] != '1') return nullptr;
    return getenv(name);
}
static inline int debug_env_int(const char* name, int dflt)
{
    const char* v = debug_env_str(name);
    return (v && *v) ? atoi(v) : dflt;
}

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

// out_j = sum_k x_k M[j][k], k sequential (row vector times M^T)
__device__ __forceinline__ float4 apply_mt(const float4 x, const float* __restrict__ M)
{
    float4 o;
    o.x = ((x.x * M[0] + x.y * M[1]) + x.z * M[2]) + x.w * M[3];
    o.y = ((x.x * M[4] + x.y * M[5]) + x.z * M[6]) + x.w * M[7];
    o.z = ((x.x * M[8] + x.y * M[9]) + x.z * M[10]) + x.w * M[11];
    o.w = ((x.x * M[12] + x.y * M[13]) + x.z * M[14]) + x.w * M[15];
    return o;
}

// g_x_k = sum_j g_j M[j][k]
__device__ __forceinline__ float4 apply_m(const float4 g, const float* __restrict__ M)
{
    float4 o;
    o.x = ((g.x * M[0] + g.y * M[4]) + g.z * M[8]) + g.w * M[12];
    o.y = ((g.x * M[1] + g.y * M[5]) + g.z * M[9]) + g.w * M[13];
    o.z = ((g.x * M[2] + g.y * M[6]) + g.z * M[10]) + g.w * M[14];
    o.w = ((g.x * M[3] + g.y * M[7]) + g.z * M[11]) + g.w * M[15];
    return o;
}

}  // namespace art
