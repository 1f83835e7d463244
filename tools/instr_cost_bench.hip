// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD, 4 waves/SIMD) of the instruction
// kinds that make up the trace kernels' hot loop on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -w -o tools/instr_cost_bench tools/instr_cost_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define KERNEL(NAME, ASM4, ...)                                                                   \
    __global__ void NAME(float* out, int iters, float seed)                                       \
    {                                                                                             \
        float a0 = seed + threadIdx.x, a1 = a0 * 1.1f, a2 = a0 * 1.2f, a3 = a0 * 1.3f;            \
        float b = 1.000001f, c = 1e-7f;                                                           \
        for (int it = 0; it < iters; ++it) {                                                      \
            _Pragma("unroll") for (int j = 0; j < 16; ++j)                                        \
                asm volatile(ASM4 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : __VA_ARGS__); \
        }                                                                                         \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;                           \
    }

KERNEL(k_fma, "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5", "vcc")
KERNEL(k_mul, "v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4", "vcc")
KERNEL(k_cvt, "v_cvt_i32_f32 %0, %0\n v_cvt_f32_i32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_cvt_f32_i32 %1, %1", "vcc")
KERNEL(k_rpi, "v_cvt_rpi_i32_f32 %0, %4\n v_cvt_rpi_i32_f32 %1, %4\n v_cvt_rpi_i32_f32 %2, %4\n v_cvt_rpi_i32_f32 %3, %4", "vcc")
KERNEL(k_trunc, "v_trunc_f32 %0, %0\n v_trunc_f32 %1, %1\n v_trunc_f32 %2, %2\n v_trunc_f32 %3, %3", "vcc")
KERNEL(k_rcp, "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3", "vcc")
KERNEL(k_divscale, "v_div_scale_f32 %0, vcc, %0, %4, %0\n v_div_scale_f32 %1, vcc, %1, %4, %1\n v_div_scale_f32 %2, vcc, %2, %4, %2\n v_div_scale_f32 %3, vcc, %3, %4, %3", "vcc")
KERNEL(k_divfmas, "v_div_fmas_f32 %0, %0, %4, %5\n v_div_fmas_f32 %1, %1, %4, %5\n v_div_fmas_f32 %2, %2, %4, %5\n v_div_fmas_f32 %3, %3, %4, %5", "vcc")
KERNEL(k_divfixup, "v_div_fixup_f32 %0, %0, %4, %5\n v_div_fixup_f32 %1, %1, %4, %5\n v_div_fixup_f32 %2, %2, %4, %5\n v_div_fixup_f32 %3, %3, %4, %5", "vcc")
KERNEL(k_cmp_e64, "v_cmp_lt_f32_e64 s[10:11], %0, %4\n v_cmp_lt_f32_e64 s[12:13], %1, %4\n v_cmp_lt_f32_e64 s[10:11], %2, %4\n v_cmp_lt_f32_e64 s[12:13], %3, %4", "s10", "s11", "s12", "s13")
KERNEL(k_cmp_cnd, "v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %5, vcc\n v_cmp_lt_f32 vcc, %2, %4\n v_cndmask_b32 %3, %3, %5, vcc", "vcc")
KERNEL(k_mad24, "v_mad_u32_u24 %0, %0, %4, %5\n v_mad_u32_u24 %1, %1, %4, %5\n v_mad_u32_u24 %2, %2, %4, %5\n v_mad_u32_u24 %3, %3, %4, %5", "vcc")
KERNEL(k_mad64, "v_mad_u64_u32 v[20:21], vcc, %0, %4, v[20:21]\n v_mad_u64_u32 v[22:23], vcc, %1, %4, v[22:23]\n v_mad_u64_u32 v[20:21], vcc, %2, %4, v[20:21]\n v_mad_u64_u32 v[22:23], vcc, %3, %4, v[22:23]", "vcc", "v20", "v21", "v22", "v23")
KERNEL(k_addco, "v_add_co_u32 %0, vcc, %0, %4\n v_add_co_u32 %1, vcc, %1, %4\n v_add_co_u32 %2, vcc, %2, %4\n v_add_co_u32 %3, vcc, %3, %4", "vcc")
KERNEL(k_pkmul, "v_pk_mul_f32 v[20:21], v[20:21], v[22:23]\n v_pk_mul_f32 v[24:25], v[24:25], v[22:23]\n v_pk_mul_f32 v[20:21], v[20:21], v[22:23]\n v_pk_mul_f32 v[24:25], v[24:25], v[22:23]", "vcc", "v20", "v21", "v22", "v23", "v24", "v25")
KERNEL(k_bcnt, "v_cmp_lt_f32 vcc, %0, %4\n s_bcnt1_i32_b64 s10, vcc\n s_add_u32 s11, s11, s10\n v_fma_f32 %1, %1, %4, %5", "vcc", "s10", "s11", "scc")
KERNEL(k_branch, "v_fma_f32 %0, %0, %4, %5\n s_cbranch_vccz 1f\n 1:\n v_fma_f32 %1, %1, %4, %5\n s_cbranch_execz 2f\n 2:", "vcc")
KERNEL(k_readlane, "v_readfirstlane_b32 s10, %0\n v_fma_f32 %1, %1, %4, %5\n v_readfirstlane_b32 s11, %2\n v_fma_f32 %3, %3, %4, %5", "s10", "s11")

template <typename K>
void run(const char* name, K kern, float* d_out)
{
    const int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    kern<<<256, 1024>>>(d_out, 10, 1.0f);
    (void)hipEventRecord(e0);
    kern<<<256, 1024>>>(d_out, iters, 1.0f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    if (hipGetLastError() != hipSuccess) printf("launch error\n");
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-14s %.2f cycles per wave-instruction per SIMD (4 waves/SIMD, 2.4 GHz nominal)\n", name,
           ms * 1e-3 * 2.4e9 / ((double)iters * 64 * 4));
}

int main()
{
    float* d_out; (void)hipMalloc(&d_out, 256 * 1024 * 4);
    run("v_fma_f32", k_fma, d_out); run("v_mul_f32", k_mul, d_out); run("v_cvt i<->f", k_cvt, d_out);
    run("v_cvt_rpi", k_rpi, d_out); run("v_trunc_f32", k_trunc, d_out); run("v_rcp_f32", k_rcp, d_out);
    run("v_div_scale", k_divscale, d_out); run("v_div_fmas", k_divfmas, d_out); run("v_div_fixup", k_divfixup, d_out);
    run("v_cmp_e64", k_cmp_e64, d_out); run("cmp+cndmask", k_cmp_cnd, d_out); run("v_mad_u32_u24", k_mad24, d_out);
    run("v_mad_u64_u32", k_mad64, d_out); run("v_add_co_u32", k_addco, d_out); run("v_pk_mul_f32", k_pkmul, d_out);
    run("cmp/bcnt/sadd/fma", k_bcnt, d_out); run("fma+branch x2", k_branch, d_out); run("readfirstlane", k_readlane, d_out);
    return 0;
}
