// What does the memory system deliver to the ACCESS PATTERN of the trace kernels' distortion stream?
// The lean forward item reads, per sample, one 8-byte (u, e) pair per point of its block: 2500 points = 20 KB contiguous, the next
// sample 80 KB further on (layout [H, R, P, 2]: the reference's).  One persistent 1024-thread workgroup per CU (the window fills
// the LDS), a ring of DEPTH non-temporal loads per lane, items from a work counter - as in trace_fwd_item_lean.  This file times
// that pattern without the ray arithmetic (and with a dummy arithmetic load), against the same bytes laid out so that an item's
// samples are contiguous ("blocked": [H, blocks, R, block points, 2]), for ring depths 4 / 8 / 16.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/stream_pattern tools/stream_pattern.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

constexpr int kThreads = 1024;
typedef float f2 __attribute__((ext_vector_type(2)));

template <int DEPTH, int WORK>
__global__ __launch_bounds__(kThreads) void stream_kernel(const f2* __restrict__ dist, int H, int R, int P, int pblock, int n_pblocks,
                                                          long long sh, long long sr, long long sb, unsigned* counter,
                                                          float* sink)
{
    extern __shared__ float lds[];
    __shared__ int s_next;
    const int n_items = H * n_pblocks;
    int item = blockIdx.x;
    float acc = 0.0f;
    while (item < n_items) {
        const int h = item / n_pblocks, b = item % n_pblocks;
        const int p0 = b * pblock, p1 = min(p0 + pblock, P);
        if (threadIdx.x == 0) s_next = (int)(gridDim.x + atomicAdd(counter, 1u));
        // sb = 0: the reference's layout, a point's offset is p; sb != 0: blocked, offset = b * sb + (p - p0)
        const f2* base = dist + (long long)h * sh + (sb ? (long long)b * sb : (long long)p0);
        for (int p = p0 + threadIdx.x; p < p1; p += kThreads) {
            const f2* q = base + (p - p0);
            f2 ring[DEPTH];
#pragma unroll
            for (int k = 0; k < DEPTH; ++k) ring[k] = __builtin_nontemporal_load(q + (long long)min(k, R - 1) * sr);
            for (int r = 0; r < R; r += DEPTH) {
#pragma unroll
                for (int k = 0; k < DEPTH; ++k) {
                    f2 v = ring[k];
                    asm volatile("" : "+v"(v.x), "+v"(v.y));
                    ring[k] = __builtin_nontemporal_load(q + (long long)min(r + k + DEPTH, R - 1) * sr);
                    float x = v.x, y = v.y;
#pragma unroll
                    for (int w = 0; w < WORK; ++w) { x = x * 1.0001f + y; y = y * 0.9999f + x; }      // 4 VALU per round
                    acc += x + y;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        __syncthreads();
        item = s_next;
        __syncthreads();
    }
    if (acc == 123.456f) sink[0] = acc + lds[0];
}

template <int DEPTH, int WORK>
static double run(const f2* dist, int H, int R, int P, int pblock, bool blocked, unsigned* counter, float* sink)
{
    const int n_pblocks = (P + pblock - 1) / pblock;
    const long long sh = (long long)R * P;
    const long long sr = blocked ? pblock : P;
    const long long sb = blocked ? (long long)R * pblock : 0;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_kernel<DEPTH, WORK>), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    std::vector<float> ms;
    for (int rep = 0; rep < 7; ++rep) {
        (void)hipMemsetAsync(counter, 0, 4);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((stream_kernel<DEPTH, WORK>), dim3(256), dim3(kThreads), 158 * 1024, 0, dist, H, R, P, pblock, n_pblocks, sh, sr, sb,
                           counter, sink);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float t; (void)hipEventElapsedTime(&t, e0, e1);
        if (rep >= 2) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

int main(int argc, char** argv)
{
    const int H = argc > 1 ? atoi(argv[1]) : 1000, R = 100, P = 10000, pblock = 2500;
    const size_t n = (size_t)H * R * P;
    f2* dist; unsigned* counter; float* sink;
    if (hipMalloc(&dist, n * sizeof(f2)) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(dist, 0, n * sizeof(f2));
    (void)hipMalloc(&counter, 4); (void)hipMalloc(&sink, 4);
    const double gb = (double)n * 8 / 1e9;
    printf("{\"heliostats\": %d, \"bytes\": %.3e", H, (double)n * 8);
#define CASE(D, W, BL, NAME) { const double t = run<D, W>(dist, H, R, P, pblock, BL, counter, sink); \
        printf(", \"%s\": {\"ms\": %.3f, \"GBps\": %.0f}", NAME, t, gb / t * 1e3); fflush(stdout); }
    CASE(8, 0, false, "reference_layout_ring8")
    CASE(8, 0, true, "blocked_layout_ring8")
    CASE(4, 0, false, "reference_layout_ring4")
    CASE(16, 0, false, "reference_layout_ring16")
    CASE(16, 0, true, "blocked_layout_ring16")
    CASE(8, 24, false, "reference_layout_ring8_96valu")
    CASE(8, 24, true, "blocked_layout_ring8_96valu")
    CASE(16, 24, false, "reference_layout_ring16_96valu")
    printf("}\n");
    return 0;
}
