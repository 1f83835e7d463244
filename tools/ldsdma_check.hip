// What does global_load_lds_dword do on gfx950?  The backward trace items stage their gradient window with it
// (artist_amd/csrc/trace_kernels.hip, stage_grad_window): lane l's dword must land at LDS address M0 + 4 l, lanes switched off by
// EXEC must write nothing, and the data must be in LDS after s_waitcnt vmcnt(0) + a barrier.  This program stages a 175 x 229
// window of a 256 x 256 bitmap exactly like the kernel does (12 waves, every twelfth group of 64 cells) and compares per-thread sums
// of the staged cells with the host's.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/ldsdma_check tools/ldsdma_check.hip ;
// run on the GPU box: prints "lds-direct staging: 0 mismatches of 49152".
#include <hip/hip_runtime.h>
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(1))) const float glb_f32;
__global__ void k(const float* __restrict__ G, float* out, int tw, int pth, int W)
{
    extern __shared__ __attribute__((aligned(16))) float gtile[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int n_cells = tw * pth;
    const int step = nwaves * 64;
    const int sq = step / tw, sr = step - sq * tw;
    int i = wave * 64 + lane;
    int row = i / tw, col = i - row * tw;
    for (int base = wave * 64; base < n_cells; base += step) {
        if (base + lane < n_cells)
            __builtin_amdgcn_global_load_lds((glb_f32*)(G + (int64_t)row * W + col), (lds_f32*)(gtile + base), 4, 0, 0);
        col += sr; row += sq;
        if (col >= tw) { col -= tw; ++row; }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    float s = 0;
    for (int c = tid; c < n_cells; c += blockDim.x) s += gtile[c];
    out[blockIdx.x * blockDim.x + tid] = s;
}
#include <cstdio>
#include <vector>
int main()
{
    const int W = 256, Hh = 256, tw = 175, pth = 229, threads = 768, blocks = 64;
    std::vector<float> h(W * Hh);
    for (int i = 0; i < W * Hh; ++i) h[i] = (float)((i * 7919) % 1013);
    float *G, *out;
    hipMalloc(&G, h.size() * 4); hipMalloc(&out, blocks * threads * 4);
    hipMemcpy(G, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 158 * 1024, 0, G + 3 * W + 17, out, tw, pth, W);
    std::vector<float> o(blocks * threads);
    hipMemcpy(o.data(), out, o.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < blocks; ++b)
        for (int t = 0; t < threads; ++t) {
            float s = 0;
            for (int c = t; c < tw * pth; c += threads) s += h[(3 + c / tw) * W + 17 + c % tw];
            if (s != o[b * threads + t]) ++bad;
        }
    printf("lds-direct staging: %d mismatches of %d\n", bad, blocks * threads);
    return bad != 0;
}
