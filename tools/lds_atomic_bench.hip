// Micro-benchmark: LDS atomic throughput on gfx950 by data type and address pattern.
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o tools/lds_atomic_bench tools/lds_atomic_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr int TILE = 32768;   // floats (128 KB)

template <int MODE, int PATTERN>
__global__ __launch_bounds__(1024) void k(const unsigned* __restrict__ idx, int iters, float* out)
{
    extern __shared__ float tile[];
    for (int i = threadIdx.x; i < TILE; i += blockDim.x) tile[i] = 0.f;
    __syncthreads();
    unsigned a = idx[blockIdx.x * blockDim.x + threadIdx.x];
    unsigned lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
        unsigned addr;
        if (PATTERN == 0) addr = (a + it * 977u) % TILE;                 // random
        else if (PATTERN == 1) addr = ((threadIdx.x >> 6) * 2048 + lane + it * 64) % TILE;  // conflict-free, consecutive
        else if (PATTERN == 2) addr = (it * 31) % TILE;                  // same address for all lanes
        else addr = ((a % 64) * 33 + (a >> 6) % 16 + it * 7) % TILE;     // random within a small 2-D neighbourhood
        if (MODE == 0) atomicAdd(&tile[addr], 1.0f);
        else if (MODE == 1) atomicAdd(reinterpret_cast<unsigned*>(tile) + addr, 1u);
        else if (MODE == 2) atomicAdd(reinterpret_cast<unsigned long long*>(tile) + (addr >> 1), 1ull);
        else if (MODE == 3) { float v = tile[addr]; tile[addr] = v + 1.0f; }   // non-atomic RMW (wrong, for rate only)
        else if (MODE == 4) { tile[addr] = 1.0f; }                            // plain store
        else if (MODE == 5) atomicAdd(reinterpret_cast<double*>(tile) + (addr >> 1), 1.0);
        else if (MODE == 7) { unsigned o = atomicAdd(reinterpret_cast<unsigned*>(tile) + addr, 0x00400000u); if (o > 0xFFC00000u) out[1 + (addr & 1023)] = 1.0f; }   // returning u32 + carry check
        else if (MODE == 6) { float o = atomicAdd(&tile[addr], 1.0f); a += (o == 7.5f); }   // returning form
        a = a * 1664525u + 1013904223u;
    }
    __syncthreads();
    float s = 0;
    for (int i = threadIdx.x; i < TILE; i += blockDim.x) s += tile[i];
    if (s == 12345.678f) out[0] = s;
}

template <int MODE, int PATTERN>
void run(const char* name, const unsigned* d_idx, float* d_out)
{
    const int blocks = 256 * 4, iters = 2000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE, PATTERN>), hipFuncAttributeMaxDynamicSharedMemorySize, TILE * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, PATTERN><<<blocks, 1024, TILE * 4>>>(d_idx, 10, d_out);
    hipEventRecord(e0);
    k<MODE, PATTERN><<<blocks, 1024, TILE * 4>>>(d_idx, iters, d_out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double ops = (double)blocks * 1024 * iters;
    // per CU: 4 blocks sequentially (1 resident at 128 KB LDS)
    double cyc_per_wave_instr = ms * 1e-3 * 2.4e9 / ((double)blocks / 256 * 16 * iters);
    printf("%-40s %8.3f ms  %8.2f Glane-ops/s  ~%6.1f cycles per wave-instr per CU\n", name, ms, ops / ms / 1e6, cyc_per_wave_instr);
}

int main()
{
    const int n = 256 * 4 * 1024;
    std::vector<unsigned> h(n);
    uint32_t s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s >> 4; }
    unsigned* d_idx; float* d_out;
    hipMalloc(&d_idx, n * 4); hipMalloc(&d_out, 4 * 2048);
    hipMemcpy(d_idx, h.data(), n * 4, hipMemcpyHostToDevice);
    run<0, 0>("ds_add_f32  random", d_idx, d_out);
    run<0, 1>("ds_add_f32  conflict-free", d_idx, d_out);
    run<0, 2>("ds_add_f32  same-address", d_idx, d_out);
    run<0, 3>("ds_add_f32  neighbourhood", d_idx, d_out);
    run<1, 0>("ds_add_u32  random", d_idx, d_out);
    run<1, 1>("ds_add_u32  conflict-free", d_idx, d_out);
    run<1, 2>("ds_add_u32  same-address", d_idx, d_out);
    run<1, 3>("ds_add_u32  neighbourhood", d_idx, d_out);
    run<2, 0>("ds_add_u64  random", d_idx, d_out);
    run<2, 1>("ds_add_u64  conflict-free", d_idx, d_out);
    run<5, 0>("ds_add_f64  random", d_idx, d_out);
    run<5, 1>("ds_add_f64  conflict-free", d_idx, d_out);
    run<6, 0>("ds_add_rtn_f32 random", d_idx, d_out);
    run<7, 0>("ds_add_rtn_u32 + carry check, random", d_idx, d_out);
    run<7, 3>("ds_add_rtn_u32 + carry check, neighbourhood", d_idx, d_out);
    run<3, 0>("read+write  random (non-atomic)", d_idx, d_out);
    run<4, 0>("ds_write_b32 random", d_idx, d_out);
    run<4, 1>("ds_write_b32 conflict-free", d_idx, d_out);
    return 0;
}
