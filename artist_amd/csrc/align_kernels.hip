// align_kernels.hip - rigid alignment of surface points and normals, forward + backward (gfx950).
//
// Replaces the two batched 4x4 matmuls of HeliostatGroupRigidBody.align_surfaces_with_* :
//     active_surface_points  = active_surface_points  @ orientations.transpose(1, 2)
//     active_surface_normals = active_surface_normals @ orientations.transpose(1, 2)
// (artist/field/heliostat_group_rigid_body.py:217-222, 265-270).  As a BLAS call each of them streams
// 32 B/point at ~0.5 TB/s (the GEMM kernel is built for large K); here both tensors go through ONE
// HBM-bound pass (64 B in + 64 B out per point, float4 accesses) with the 4x4 matrix in SGPRs, and the
// backward produces dL/dpoints, dL/dnormals and - for kinematics optimisation - dL/dorientation.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "launch_common.hpp"

namespace art {

constexpr int kAlignBlock = 256;

__global__ __launch_bounds__(kAlignBlock) void align_fwd_kernel(const float4* __restrict__ points,
                                                                 const float4* __restrict__ normals,
                                                                 const float* __restrict__ orientation, int P, int n_tiles,
                                                                 float4* __restrict__ out_points,
                                                                 float4* __restrict__ out_normals)
{
    const int h = blockIdx.x / n_tiles;
    const int p = (blockIdx.x % n_tiles) * kAlignBlock + threadIdx.x;
    if (p >= P) return;
    const float* M = orientation + (int64_t)h * 16;       // wave-uniform -> scalar loads
    const int64_t i = (int64_t)h * P + p;
    out_points[i] = apply_mt(points[i], M);
    out_normals[i] = apply_mt(normals[i], M);
}

// g_points = g_out_points @ M, g_normals = g_out_normals @ M (the gradient of the matrix itself: align_gm_kernel below; the
// variant of this kernel that added it up with float atomics was dead code since round 2 and is gone)
__global__ __launch_bounds__(kAlignBlock) void align_bwd_kernel(const float* __restrict__ orientation,
                                                                 const float4* __restrict__ g_out_points,
                                                                 const float4* __restrict__ g_out_normals, int P, int n_tiles,
                                                                 float4* __restrict__ g_points,
                                                                 float4* __restrict__ g_normals)
{
    const int h = blockIdx.x / n_tiles;
    const int p = (blockIdx.x % n_tiles) * kAlignBlock + threadIdx.x;
    if (p >= P) return;
    const float* M = orientation + (int64_t)h * 16;
    const int64_t i = (int64_t)h * P + p;
    g_points[i] = apply_m(g_out_points[i], M);
    g_normals[i] = apply_m(g_out_normals[i], M);
}

// dL/dM[j][k] = sum_p (gP_j xP_k + gN_j xN_k) of ONE heliostat per workgroup, summed in a fixed order (thread-strided
// partial sums, DPP-free shuffle tree, waves in index order): bit-reproducible, no atomics and nothing to pre-zero.
// The kinematics optimisers sit on this gradient (heliostat_group_rigid_body.py:217-222 in the autograd graph), and a
// 250-step Adam run amplifies a last-bit difference into a different path.
constexpr int kGmBlock = 1024;
__global__ __launch_bounds__(kGmBlock) void align_gm_kernel(const float4* __restrict__ points, const float4* __restrict__ normals,
                                                            const float4* __restrict__ g_out_points,
                                                            const float4* __restrict__ g_out_normals, int P,
                                                            float* __restrict__ g_orientation)
{
    __shared__ float s_part[kGmBlock / 64][16];
    const int h = blockIdx.x;
    float gm[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) gm[k] = 0.0f;
    for (int p = threadIdx.x; p < P; p += kGmBlock) {
        const int64_t i = (int64_t)h * P + p;
        const float4 gp = g_out_points[i], gn = g_out_normals[i], xp = points[i], xn = normals[i];
        const float g4[4] = {gp.x, gp.y, gp.z, gp.w}, h4[4] = {gn.x, gn.y, gn.z, gn.w};
        const float x4[4] = {xp.x, xp.y, xp.z, xp.w}, y4[4] = {xn.x, xn.y, xn.z, xn.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) gm[4 * j + k] += g4[j] * x4[k] + h4[j] * y4[k];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const float v = wave_sum_f32(gm[k]);
        if (lane == 0) s_part[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < kGmBlock / 64; ++w) v += s_part[w][threadIdx.x];
        g_orientation[(int64_t)h * 16 + threadIdx.x] = v;
    }
}

}  // namespace art

namespace art {
// geometry.reflect (artist/raytracing/geometry.py:11-41) for every surface point of H heliostats: out = i - 2 (i.n) n over
// all four components, in the reference's operation order (the file is compiled with -ffp-contract=off).  The trace
// kernels reflect in registers; this kernel only serves callers that want the directions as a tensor
// (`heliostat_group.preferred_reflection_directions`, heliostat_ray_tracer.py:285-290): one pass instead of torch's five.
__global__ __launch_bounds__(kAlignBlock) void reflect_kernel(const float4* __restrict__ incident, const float4* __restrict__ normals,
                                                              int64_t P, int64_t total, float4* __restrict__ out)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= total) return;
    const float4 i = incident[k / P], n = normals[k];
    const float s = ((i.x * n.x + i.y * n.y) + i.z * n.z) + i.w * n.w;          // torch.sum over the last dimension
    const float s2 = 2.0f * s;
    out[k] = make_float4(i.x - s2 * n.x, i.y - s2 * n.y, i.z - s2 * n.z, i.w - s2 * n.w);
}
}  // namespace art

using namespace art;

extern "C" int art_reflect(const float* incident, const float* normals, int64_t H, int64_t P, float* out, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (H < 0 || P < 0) return ART_EINVAL;
    if (H == 0 || P == 0) return ART_OK;
    const int64_t total = H * P;
    if (!incident || !normals || !out || (total + kAlignBlock - 1) / kAlignBlock > 2147483647LL) return ART_EINVAL;
    hipLaunchKernelGGL(reflect_kernel, dim3((unsigned)((total + kAlignBlock - 1) / kAlignBlock)), dim3(kAlignBlock), 0, stream,
                       reinterpret_cast<const float4*>(incident), reinterpret_cast<const float4*>(normals), P, total,
                       reinterpret_cast<float4*>(out));
    ART_HIP(hipGetLastError());
    return ART_OK;
}

extern "C" int art_align_fwd(const float* points, const float* normals, const float* orientation, int64_t H,
                             int64_t P, float* out_points, float* out_normals, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (H == 0) return ART_OK;
    if (!points || !normals || !orientation || !out_points || !out_normals || H < 0 || P <= 0 || P > 2147483647LL ||
        H * ((P + kAlignBlock - 1) / kAlignBlock) > 2147483647LL)
        return ART_EINVAL;
    if (H == 0) return ART_OK;
    const int n_tiles = (int)((P + kAlignBlock - 1) / kAlignBlock);
    hipLaunchKernelGGL(align_fwd_kernel, dim3((unsigned)(H * n_tiles)), dim3(kAlignBlock), 0, stream,
                       reinterpret_cast<const float4*>(points), reinterpret_cast<const float4*>(normals), orientation,
                       (int)P, n_tiles,
                       reinterpret_cast<float4*>(out_points), reinterpret_cast<float4*>(out_normals));
    ART_HIP(hipGetLastError());
    return ART_OK;
}

extern "C" int art_align_bwd(const float* points, const float* normals, const float* orientation,
                             const float* grad_out_points, const float* grad_out_normals, int64_t H, int64_t P,
                             float* grad_points, float* grad_normals, float* grad_orientation, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (H == 0) return ART_OK;
    if (!orientation || !grad_out_points || !grad_out_normals || !grad_points || !grad_normals || H < 0 || P <= 0 ||
        P > 2147483647LL || H * ((P + kAlignBlock - 1) / kAlignBlock) > 2147483647LL ||
        (grad_orientation && (!points || !normals)))
        return ART_EINVAL;
    if (H == 0) return ART_OK;
    const int n_tiles = (int)((P + kAlignBlock - 1) / kAlignBlock);
    const dim3 grid((unsigned)(H * n_tiles));
    hipLaunchKernelGGL(align_bwd_kernel, grid, dim3(kAlignBlock), 0, stream, orientation,
                       reinterpret_cast<const float4*>(grad_out_points), reinterpret_cast<const float4*>(grad_out_normals),
                       (int)P, n_tiles, reinterpret_cast<float4*>(grad_points), reinterpret_cast<float4*>(grad_normals));
    if (grad_orientation)
        hipLaunchKernelGGL(align_gm_kernel, dim3((unsigned)H), dim3(kGmBlock), 0, stream,
                           reinterpret_cast<const float4*>(points), reinterpret_cast<const float4*>(normals),
                           reinterpret_cast<const float4*>(grad_out_points), reinterpret_cast<const float4*>(grad_out_normals),
                           (int)P, grad_orientation);
    ART_HIP(hipGetLastError());
    return ART_OK;
}
