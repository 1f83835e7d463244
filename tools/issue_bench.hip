// Micro-benchmark: instruction issue rates on gfx950 for VALU / SALU mixes at 1, 2, 4, 8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/issue_bench tools/issue_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>

// MODE 0: 64 independent-ish v_fma chains (4 accumulators)      -> pure VALU
// MODE 1: same + one s_add per VALU                               -> VALU + SALU interleaved
// MODE 2: dependent chain (1 accumulator)                         -> VALU latency
// MODE 3: v_pk_fma_f32 (2 accumulators of float2)                 -> packed
// MODE 4: v_fma + v_cndmask + v_cmp mix
template <int MODE>
__global__ void k(float* out, int iters, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 * 1.1f, a2 = a0 * 1.2f, a3 = a0 * 1.3f;
    const float b = 1.000001f, c = 1e-7f;
    int s = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (MODE == 0 || MODE == 1 || MODE == 5 || MODE == 6) {
                asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
                if (MODE == 1) asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1" : "+s"(s) : : "scc");
                if (MODE == 5) asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0");
                if (MODE == 6) asm volatile("s_and_b64 vcc, exec, vcc\n s_or_b64 vcc, vcc, exec\n s_and_b64 vcc, exec, vcc\n s_or_b64 vcc, vcc, exec" ::: "vcc", "scc");
            } else if (MODE == 2) {
                asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2"
                             : "+v"(a0) : "v"(b), "v"(c));
            } else if (MODE == 3) {
                typedef float v2f __attribute__((ext_vector_type(2)));
                v2f x = {a0, a1}, y = {a2, a3}, bb = {b, b}, cc = {c, c};
                asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3"
                             : "+v"(x), "+v"(y) : "v"(bb), "v"(cc));
                a0 = x.x; a1 = x.y; a2 = y.x; a3 = y.y;
            } else if (MODE == 4) {
                asm volatile("v_fma_f32 %0, %0, %4, %5\n v_cmp_lt_f32 vcc, %1, %0\n v_cndmask_b32 %2, %2, %3, vcc\n v_fma_f32 %3, %3, %4, %5"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (float)s;
}

template <int MODE>
void run(const char* name, float* d_out, int instr_per_iter)
{
    const int iters = 4000;
    for (int wps : {1, 2, 4, 8}) {           // waves per SIMD: block = 64*4*wps threads, 1 block per CU
        const int threads = 64 * 4 * wps;
        if (threads > 1024) {                // 8 waves/SIMD = 2 blocks of 1024 per CU
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            k<MODE><<<512, 1024>>>(d_out, 10, 1.0f);
            (void)hipEventRecord(e0);
            k<MODE><<<512, 1024>>>(d_out, iters, 1.0f);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            double cyc = ms * 1e-3 * 2.4e9;
            printf("%-34s waves/SIMD=%d  %.2f cycles per wave-instruction per SIMD\n", name, wps, cyc / ((double)iters * instr_per_iter * 8));
            continue;
        }
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        k<MODE><<<256, threads>>>(d_out, 10, 1.0f);
        (void)hipEventRecord(e0);
        k<MODE><<<256, threads>>>(d_out, iters, 1.0f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        double cyc = ms * 1e-3 * 2.4e9;
        printf("%-34s waves/SIMD=%d  %.2f cycles per wave-instruction per SIMD\n", name, wps, cyc / ((double)iters * instr_per_iter * wps));
    }
}

int main()
{
    float* d_out; (void)hipMalloc(&d_out, 512 * 1024 * 4);
    run<0>("v_fma x4 independent", d_out, 64);
    run<1>("v_fma x4 + s_add x4 (count all)", d_out, 128);
    run<5>("v_fma x4 + s_nop x4 (count all)", d_out, 128);
    run<6>("v_fma x4 + s_and/or_b64 x4 (all)", d_out, 128);
    run<2>("v_fma dependent chain", d_out, 64);
    run<3>("v_pk_fma_f32 (2 chains)", d_out, 64);
    run<4>("fma/cmp/cndmask/fma", d_out, 64);
    return 0;
}
