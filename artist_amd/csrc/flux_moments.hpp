// flux_moments.hpp - the centre-of-mass sums of a bitmap, shared by the flux epilogue (flux_kernels.hip) and by the trace's
// conversion pass (trace_kernels.hip): the pass that turns the pixel accumulators into the fp32 bitmap streams every pixel anyway and
// can leave these sums behind, so that the crop which follows in a reconstruction epoch does not read the bitmaps once more.
// A bitmap's rows are summed in kLossParts parts: fixed per-thread order, fixed block tree - whoever forms them.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace art {

constexpr int kLossParts = 4;
constexpr int kMomentsBlock = 1024;   // threads of every kernel that forms these sums (the tree depends on it)

// torch.linspace(-1, 1, n)[k]: the first half is filled from the start, the second half from the end
__device__ __forceinline__ float lin11(int k, int n)
{
    if (n == 1) return -1.0f;
    const float step = 2.0f / (float)(n - 1);
    return k < n / 2 ? -1.0f + step * (float)k : 1.0f - step * (float)(n - 1 - k);
}

// N sums at once, each with block_sum's tree (xor-shuffle inside a wave, then the waves in order): one pair of barriers for
// all of them.  s_red: 16 * N doubles.
template <int N>
__device__ __forceinline__ void block_sum_n(double (&v)[N], double* s_red)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int k = 0; k < N; ++k)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_xor(v[k], off, 64);
    __syncthreads();
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < N; ++k) s_red[wave * N + k] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) {
        double r = 0.0;
        for (int w = 0; w < nw; ++w) r += s_red[w * N + k];
        v[k] = r;
    }
}

__device__ __forceinline__ void part_rows(int Hh, int v, int& r0, int& r1)
{
    r0 = (int)(((int64_t)Hh * v) / kLossParts);
    r1 = (int)(((int64_t)Hh * (v + 1)) / kLossParts);
}

// (sum f, sum x f, sum y f) over the rows of part v of one bitmap, x / y in normalised coordinates; block-wide result.
// `load4(k)` / `load1(k)`: the k-th float4 / float of the bitmap - flux_com_parts_kernel reads them from the bitmap, the trace's
// conversion pass makes them from its pixel accumulators (and writes the bitmap on the way): the SAME sums, bit for bit.
template <typename Load4, typename Load1>
__device__ __forceinline__ void com_part_sums_from(Load4&& load4, Load1&& load1, int Hh, int W, int v, double* s_red, double& s, double& xs,
                                                   double& ys)
{
    int r0, r1;
    part_rows(Hh, v, r0, r1);
    double a[3] = {0.0, 0.0, 0.0};
    if ((W & 3) == 0) {
        const int W4 = W >> 2;
        const int64_t base4 = (int64_t)r0 * W4;
        const int n = (r1 - r0) * W4;
        int x4 = threadIdx.x % W4, y = r0 + threadIdx.x / W4;              // (no division in the loop)
        const int dx = blockDim.x % W4, dy = blockDim.x / W4;
#pragma unroll 4
        for (int k = threadIdx.x; k < n; k += blockDim.x) {
            const float4 q = load4(base4 + k);
            const int x = 4 * x4;
            a[0] += (double)((q.x + q.y) + (q.z + q.w));
            a[1] += (double)((lin11(x, W) * q.x + lin11(x + 1, W) * q.y) + (lin11(x + 2, W) * q.z + lin11(x + 3, W) * q.w));
            a[2] += (double)(lin11(y, Hh) * ((q.x + q.y) + (q.z + q.w)));
            x4 += dx; y += dy;
            if (x4 >= W4) { x4 -= W4; ++y; }
        }
    } else {
        const int n = (r1 - r0) * W;
        int x = threadIdx.x % W, y = r0 + threadIdx.x / W;
        const int dx = blockDim.x % W, dy = blockDim.x / W;
        for (int k = threadIdx.x; k < n; k += blockDim.x) {
            const float q = load1((int64_t)r0 * W + k);
            a[0] += (double)q; a[1] += (double)(lin11(x, W) * q); a[2] += (double)(lin11(y, Hh) * q);
            x += dx; y += dy;
            if (x >= W) { x -= W; ++y; }
        }
    }
    block_sum_n<3>(a, s_red);
    s = a[0]; xs = a[1]; ys = a[2];
}

__device__ __forceinline__ void com_part_sums(const float* __restrict__ f, int Hh, int W, int v, double* s_red, double& s, double& xs, double& ys)
{
    com_part_sums_from([&](int64_t k) { return reinterpret_cast<const float4*>(f)[k]; }, [&](int64_t k) { return f[k]; }, Hh, W, v, s_red,
                       s, xs, ys);
}

}  // namespace art
