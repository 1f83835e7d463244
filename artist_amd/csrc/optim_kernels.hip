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

__global__ __launch_bounds__(256) void adam_step_kernel(AdamArgs a)
{
    const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i0 >= a.n) return;
    const bool vec = i0 + 4 <= a.n && (((uintptr_t)a.param | (uintptr_t)a.grad | (uintptr_t)a.exp_avg | (uintptr_t)a.exp_avg_sq) & 15) == 0;
    float p[4], g[4], m[4], v[4];
    const int cnt = (int)(a.n - i0 < 4 ? a.n - i0 : 4);
    if (vec) {
        *reinterpret_cast<float4*>(p) = *reinterpret_cast<const float4*>(a.param + i0);
        *reinterpret_cast<float4*>(g) = *reinterpret_cast<const float4*>(a.grad + i0);
        *reinterpret_cast<float4*>(m) = *reinterpret_cast<const float4*>(a.exp_avg + i0);
        *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(a.exp_avg_sq + i0);
    } else {
        for (int k = 0; k < cnt; ++k) { p[k] = a.param[i0 + k]; g[k] = a.grad[i0 + k]; m[k] = a.exp_avg[i0 + k]; v[k] = a.exp_avg_sq[i0 + k]; }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= cnt) break;
        float gr = a.grad_sign * g[k];
        if (a.nu > 0) {
            const int64_t cell = (i0 + k) / 3;
            const int comp = (int)((i0 + k) - 3 * cell);
            const int c = (int)(cell % a.nv), r = (int)((cell / a.nv) % a.nu);
            // surface_reconstructor.py:1212-1222: the edge control points keep their u and v (components 0, 1); z stays free
            if (comp < 2 && (r == 0 || r == a.nu - 1 || c == 0 || c == a.nv - 1)) gr = 0.0f;
        }
        if (a.weight_decay != 0.0f) gr = gr + a.weight_decay * p[k];
        m[k] = m[k] + (gr - m[k]) * a.one_minus_beta1;                    // exp_avg.lerp_(grad, 1 - beta1)
        v[k] = v[k] * a.beta2 + (a.one_minus_beta2 * gr) * gr;            // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
        const float denom = sqrtf(v[k]) * a.inv_bc2_sqrt + a.eps;         // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
        p[k] = p[k] - a.step_size * (m[k] / denom);                       // param.addcdiv_(exp_avg, denom, value=-step_size)
    }
    if (vec) {
        *reinterpret_cast<float4*>(a.param + i0) = *reinterpret_cast<const float4*>(p);
        *reinterpret_cast<float4*>(a.exp_avg + i0) = *reinterpret_cast<const float4*>(m);
        *reinterpret_cast<float4*>(a.exp_avg_sq + i0) = *reinterpret_cast<const float4*>(v);
    } else {
        for (int k = 0; k < cnt; ++k) { a.param[i0 + k] = p[k]; a.exp_avg[i0 + k] = m[k]; a.exp_avg_sq[i0 + k] = v[k]; }
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
        lock_nu > 4096 || lock_nv > 4096 || (lock_nu > 0 && n % (lock_nu * lock_nv * 3) != 0))
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
