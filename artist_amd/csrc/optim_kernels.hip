// optim_kernels.hip - the optimiser step of the reconstruction epoch (gfx950).
//
// ARTIST's reconstructors step a torch.optim.Adam over ONE tensor per heliostat group - the NURBS control points
// [H,F,nu,nv,3] (artist/optim/surface_reconstructor.py:452-455, step at :779) or the kinematics deviation tables
// (artist/optim/kinematics_reconstructor.py).  torch's implementations of that step are launch-bound at these sizes: the
// `foreach` path is five multi-tensor launches, the `fused` one a single launch that walks 64 K elements per workgroup (three
// workgroups for one rank's 150 000 control-point coordinates: 41-48 us whatever the size, profiles/r03_kernel_stats_top*.txt).
// Here: one elementwise kernel, one thread per 4 elements, the update rule of torch.optim.Adam / _single_tensor_adam
// (torch/optim/adam.py) in fp32 with the bias corrections computed on the host in double from the host-side step count -
// no device-side step tensor, no synchronisation.  Optional edge lock: the reference zeroes the gradient of the outer-edge
// control points before the step (surface_reconstructor.py:779 via lock_control_points_on_outer_edges: their first two
// components - the net keeps its outline, z stays free); with nu, nv > 0 the kernel treats those gradient components of the
// first / last row and column of every [nu,nv,3] net as zero (the moments still decay).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>

#include "launch_common.hpp"

namespace art {

struct AdamArgs {
    float* param; const float* grad; float* exp_avg; float* exp_avg_sq;
    int64_t n;
    float beta1, beta2, one_minus_beta1, one_minus_beta2, eps, step_size, inv_bc2_sqrt, weight_decay, grad_sign;
    int nu, nv;          // > 0: lock the outer edge of every [nu,nv,3] net
};

// one element of the update (torch/optim/adam.py, _single_tensor_adam)
__device__ __forceinline__ void adam_element(const AdamArgs& a, float& p, float g, float& m, float& v, bool locked)
{
    float gr = locked ? 0.0f : a.grad_sign * g;
    if (a.weight_decay != 0.0f) gr = gr + a.weight_decay * p;
    m = m + (gr - m) * a.one_minus_beta1;                       // exp_avg.lerp_(grad, 1 - beta1)
    v = v * a.beta2 + (a.one_minus_beta2 * gr) * gr;            // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    const float denom = sqrtf(v) * a.inv_bc2_sqrt + a.eps;      // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
    p = p - a.step_size * (m / denom);                          // param.addcdiv_(exp_avg, denom, value=-step_size)
}

// is element i a u or v component of an edge control point of its [nu,nv,3] net?  (surface_reconstructor.py:1212-1222: the edge
// control points keep their u and v - components 0, 1 -; z stays free)
__device__ __forceinline__ bool adam_locked(const AdamArgs& a, unsigned i)
{
    if (a.nu <= 0) return false;
    const unsigned cell = i / 3u, comp = i - 3u * cell;
    const unsigned line = cell / (unsigned)a.nv, col = cell - line * (unsigned)a.nv, row = line % (unsigned)a.nu;
    return comp < 2u && (row == 0u || row == (unsigned)a.nu - 1u || col == 0u || col == (unsigned)a.nv - 1u);
}

// One thread per four elements, everything in registers: 16-byte loads and stores where the four tensors are 16-byte aligned
// and the group is whole, element by element otherwise (the last thread of an odd size; unaligned views).
// (Round 4's first version kept the four elements in arrays that the element-wise path indexed in a loop: the compiler moved
//  them to LDS - 16 KB per workgroup - and the kernel took 21 us at any size.)
__global__ __launch_bounds__(256) void adam_step_kernel(AdamArgs a)
{
    const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i0 >= a.n) return;
    const bool vec = i0 + 4 <= a.n && (((uintptr_t)a.param | (uintptr_t)a.grad | (uintptr_t)a.exp_avg | (uintptr_t)a.exp_avg_sq) & 15) == 0;
    if (vec) {
        float4 p = *reinterpret_cast<const float4*>(a.param + i0);
        const float4 g = *reinterpret_cast<const float4*>(a.grad + i0);
        float4 m = *reinterpret_cast<const float4*>(a.exp_avg + i0);
        float4 v = *reinterpret_cast<const float4*>(a.exp_avg_sq + i0);
        const unsigned i = (unsigned)i0;              // (a lock implies n < 2^31: checked by the caller)
        adam_element(a, p.x, g.x, m.x, v.x, adam_locked(a, i));
        adam_element(a, p.y, g.y, m.y, v.y, adam_locked(a, i + 1u));
        adam_element(a, p.z, g.z, m.z, v.z, adam_locked(a, i + 2u));
        adam_element(a, p.w, g.w, m.w, v.w, adam_locked(a, i + 3u));
        *reinterpret_cast<float4*>(a.param + i0) = p;
        *reinterpret_cast<float4*>(a.exp_avg + i0) = m;
        *reinterpret_cast<float4*>(a.exp_avg_sq + i0) = v;
        return;
    }
    for (int64_t i = i0; i < a.n && i < i0 + 4; ++i) {
        float p = a.param[i], m = a.exp_avg[i], v = a.exp_avg_sq[i];
        adam_element(a, p, a.grad[i], m, v, adam_locked(a, (unsigned)i));
        a.param[i] = p; a.exp_avg[i] = m; a.exp_avg_sq[i] = v;
    }
}

}  // namespace art

using namespace art;

extern "C" int art_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr,
                             double beta1, double beta2, double eps, double weight_decay, int64_t step, int maximize,
                             int64_t lock_nu, int64_t lock_nv, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n == 0) return ART_OK;
    if (!param || !grad || !exp_avg || !exp_avg_sq || n < 0 || step < 1 || !(beta1 >= 0.0 && beta1 < 1.0) ||
        !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0) || lock_nu < 0 || lock_nv < 0 || (lock_nu > 0) != (lock_nv > 0) ||
        lock_nu > 4096 || lock_nv > 4096 || (lock_nu > 0 && (n % (lock_nu * lock_nv * 3) != 0 || n > 2147483647LL)))
        return ART_EINVAL;
    AdamArgs a;
    a.param = param; a.grad = grad; a.exp_avg = exp_avg; a.exp_avg_sq = exp_avg_sq; a.n = n;
    const double bc1 = 1.0 - std::pow(beta1, (double)step), bc2 = 1.0 - std::pow(beta2, (double)step);
    a.beta1 = (float)beta1; a.beta2 = (float)beta2;
    a.one_minus_beta1 = (float)(1.0 - beta1); a.one_minus_beta2 = (float)(1.0 - beta2);
    a.eps = (float)eps;
    a.step_size = (float)(lr / bc1);
    a.inv_bc2_sqrt = (float)(1.0 / std::sqrt(bc2));
    a.weight_decay = (float)weight_decay;
    a.grad_sign = maximize ? -1.0f : 1.0f;
    a.nu = (int)lock_nu; a.nv = (int)lock_nv;
    const int64_t threads = (n + 3) / 4;
    const int64_t blocks = (threads + 255) / 256;
    if (blocks > 2147483647LL) return ART_EINVAL;
    hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    ART_HIP(hipGetLastError());
    return ART_OK;
}
