// Micro-benchmark: what one wave64 instruction costs a gfx950 SIMD, in TRUE shader cycles.
//
// Every wave stamps s_memtime (shader clock) and s_memrealtime (100 MHz) around its loop, so the result does not
// depend on an assumed clock: cycles per instruction per SIMD = median wave's (t1 - t0) / (instructions per wave x waves
// per SIMD), and the clock the chip held is reported next to it.  One workgroup per CU, 1 / 2 / 4 waves per SIMD.
// Replaces the round-1 version that converted hipEvent milliseconds with a hard-coded 2.4 GHz.
//
// Build: hipcc --offload-arch=gfx950 -O3 -w -o /tmp/issue_bench tools/issue_bench.hip     Run: /tmp/issue_bench [json]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned long long memtime()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
__device__ __forceinline__ unsigned long long memrealtime()
{
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

constexpr int kUnroll = 8;

// ASM = `n_instr` instructions on the operands below; repeated kUnroll x iters times per wave.
//   %0-%3  float accumulators a0..a3        %4,%5  v2f accumulators p0,p1      %6,%7  float inputs b,c
//   %8     v2f input bb                     %9     SGPR float sb               %10    int LDS byte address (per lane)
//   %11    unsigned u0 (int accumulator)
#define BENCH(NAME, ASM, ...)                                                                                       \
    __global__ __launch_bounds__(1024) void NAME(float* out, unsigned long long* stamps, int iters, float seed)     \
    {                                                                                                               \
        __shared__ unsigned lds[16384];                                                                             \
        for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 0u;                                          \
        __syncthreads();                                                                                            \
        float a0 = seed + threadIdx.x, a1 = a0 * 1.1f, a2 = a0 * 1.2f, a3 = a0 * 1.3f;                              \
        v2f p0 = {a0 * 0.9f, a1 * 0.8f}, p1 = {a2 * 0.7f, a3 * 0.6f};                                               \
        const float b = 1.000001f, c = 1e-7f;                                                                       \
        const v2f bb = {1.000001f, 0.999999f};                                                                      \
        const float sb = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, seed)));  \
        /* pseudo-random cell per lane (what the splat's window sees: neighbouring lanes land within a few cells) */ \
        int addr = (int)(((threadIdx.x * 2654435761u) >> 18) & 4095u) * 8;                                         \
        unsigned u0 = threadIdx.x;                                                                                  \
        const unsigned long long r0 = memrealtime(), t0 = memtime();                                                \
        for (int it = 0; it < iters; ++it) {                                                                        \
            _Pragma("unroll") for (int j = 0; j < kUnroll; ++j)                                                     \
                asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(p0), "+v"(p1)                       \
                             : "v"(b), "v"(c), "v"(bb), "s"(sb), "v"(addr), "v"(u0)                                 \
                             : "memory", __VA_ARGS__);                                                              \
        }                                                                                                           \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                          \
        const unsigned long long t1 = memtime(), r1 = memrealtime();                                                \
        if ((threadIdx.x & 63) == 0) {                                                                              \
            unsigned long long* s = stamps + 4 * (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));             \
            s[0] = t0; s[1] = t1; s[2] = r0; s[3] = r1;                                                             \
        }                                                                                                           \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + p0.x + p0.y + p1.x + p1.y + lds[threadIdx.x]; \
    }

BENCH(k_fma, "v_fma_f32 %0, %0, %6, %7\n v_fma_f32 %1, %1, %6, %7\n v_fma_f32 %2, %2, %6, %7\n v_fma_f32 %3, %3, %6, %7", "vcc")
BENCH(k_mul, "v_mul_f32 %0, %0, %6\n v_mul_f32 %1, %1, %6\n v_mul_f32 %2, %2, %6\n v_mul_f32 %3, %3, %6", "vcc")
BENCH(k_add, "v_add_f32 %0, %0, %7\n v_add_f32 %1, %1, %7\n v_add_f32 %2, %2, %7\n v_add_f32 %3, %3, %7", "vcc")
BENCH(k_mul_s, "v_mul_f32 %0, %9, %0\n v_mul_f32 %1, %9, %1\n v_mul_f32 %2, %9, %2\n v_mul_f32 %3, %9, %3", "vcc")
BENCH(k_fma_dep, "v_fma_f32 %0, %0, %6, %7\n v_fma_f32 %0, %0, %6, %7\n v_fma_f32 %0, %0, %6, %7\n v_fma_f32 %0, %0, %6, %7", "vcc")
BENCH(k_pkfma, "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8", "vcc")
BENCH(k_pkmul, "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8", "vcc")
BENCH(k_pkadd, "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8", "vcc")
BENCH(k_pk_fma_mix, "v_pk_mul_f32 %4, %4, %8\n v_fma_f32 %0, %0, %6, %7\n v_pk_mul_f32 %5, %5, %8\n v_fma_f32 %1, %1, %6, %7", "vcc")
BENCH(k_trunc, "v_trunc_f32 %0, %0\n v_trunc_f32 %1, %1\n v_trunc_f32 %2, %2\n v_trunc_f32 %3, %3", "vcc")
BENCH(k_rpi, "v_cvt_rpi_i32_f32 %0, %6\n v_cvt_rpi_i32_f32 %1, %6\n v_cvt_rpi_i32_f32 %2, %6\n v_cvt_rpi_i32_f32 %3, %6", "vcc")
BENCH(k_cvt, "v_cvt_i32_f32 %0, %6\n v_cvt_i32_f32 %1, %6\n v_cvt_i32_f32 %2, %6\n v_cvt_i32_f32 %3, %6", "vcc")
BENCH(k_rcp, "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3", "vcc")
BENCH(k_med3, "v_med3_f32 %0, %0, %6, %7\n v_med3_f32 %1, %1, %6, %7\n v_med3_f32 %2, %2, %6, %7\n v_med3_f32 %3, %3, %6, %7", "vcc")
BENCH(k_cmp_vcc, "v_cmp_lt_f32 vcc, %0, %6\n v_cmp_lt_f32 vcc, %1, %6\n v_cmp_lt_f32 vcc, %2, %6\n v_cmp_lt_f32 vcc, %3, %6", "vcc")
BENCH(k_cmp_e64, "v_cmp_lt_f32_e64 s[10:11], %0, %6\n v_cmp_lt_f32_e64 s[12:13], %1, %6\n v_cmp_lt_f32_e64 s[10:11], %2, %6\n v_cmp_lt_f32_e64 s[12:13], %3, %6", "s10", "s11", "s12", "s13")
BENCH(k_cndmask, "v_cndmask_b32 %0, %0, %6, vcc\n v_cndmask_b32 %1, %1, %6, vcc\n v_cndmask_b32 %2, %2, %6, vcc\n v_cndmask_b32 %3, %3, %6, vcc", "s10")
BENCH(k_mad24, "v_mad_u32_u24 %0, %0, %6, %7\n v_mad_u32_u24 %1, %1, %6, %7\n v_mad_u32_u24 %2, %2, %6, %7\n v_mad_u32_u24 %3, %3, %6, %7", "vcc")
BENCH(k_iadd, "v_add_u32 %0, %0, %6\n v_add_u32 %1, %1, %6\n v_add_u32 %2, %2, %6\n v_add_u32 %3, %3, %6", "vcc")
BENCH(k_lshl_add, "v_lshl_add_u32 %0, %0, 2, %6\n v_lshl_add_u32 %1, %1, 2, %6\n v_lshl_add_u32 %2, %2, 2, %6\n v_lshl_add_u32 %3, %3, 2, %6", "vcc")
BENCH(k_mov, "v_mov_b32 %0, %6\n v_mov_b32 %1, %6\n v_mov_b32 %2, %6\n v_mov_b32 %3, %6", "vcc")
BENCH(k_dpp, "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "vcc")
// operand kinds: inline constants, 32-bit literals, source modifiers (VOP3 encodings), SGPR operands
BENCH(k_add_inline, "v_add_f32 %0, 1.0, %0\n v_add_f32 %1, 1.0, %1\n v_add_f32 %2, 1.0, %2\n v_add_f32 %3, 1.0, %3", "vcc")
BENCH(k_fma_inline, "v_fma_f32 %0, %0, %6, -0.5\n v_fma_f32 %1, %1, %6, -0.5\n v_fma_f32 %2, %2, %6, -0.5\n v_fma_f32 %3, %3, %6, -0.5", "vcc")
BENCH(k_fmamk_lit, "v_fmamk_f32 %0, %0, 0xb9500d01, %7\n v_fmamk_f32 %1, %1, 0xb9500d01, %7\n v_fmamk_f32 %2, %2, 0xb9500d01, %7\n v_fmamk_f32 %3, %3, 0xb9500d01, %7", "vcc")
BENCH(k_fmaak_lit, "v_fmaak_f32 %0, %0, %6, 0xbe2aaaab\n v_fmaak_f32 %1, %1, %6, 0xbe2aaaab\n v_fmaak_f32 %2, %2, %6, 0xbe2aaaab\n v_fmaak_f32 %3, %3, %6, 0xbe2aaaab", "vcc")
BENCH(k_fmac, "v_fmac_f32 %0, %6, %7\n v_fmac_f32 %1, %6, %7\n v_fmac_f32 %2, %6, %7\n v_fmac_f32 %3, %6, %7", "vcc")
BENCH(k_fma_neg, "v_fma_f32 %0, -%0, %6, %7\n v_fma_f32 %1, -%1, %6, %7\n v_fma_f32 %2, -%2, %6, %7\n v_fma_f32 %3, -%3, %6, %7", "vcc")
BENCH(k_mul_abs, "v_mul_f32_e64 %0, %6, |%0|\n v_mul_f32_e64 %1, %6, |%1|\n v_mul_f32_e64 %2, %6, |%2|\n v_mul_f32_e64 %3, %6, |%3|", "vcc")
BENCH(k_fma_s, "v_fma_f32 %0, %0, %9, %7\n v_fma_f32 %1, %1, %9, %7\n v_fma_f32 %2, %2, %9, %7\n v_fma_f32 %3, %3, %9, %7", "vcc")
BENCH(k_sub, "v_sub_f32 %0, %0, %7\n v_sub_f32 %1, %1, %7\n v_sub_f32 %2, %2, %7\n v_sub_f32 %3, %3, %7", "vcc")
BENCH(k_max, "v_max_f32 %0, %0, %7\n v_max_f32 %1, %1, %7\n v_max_f32 %2, %2, %7\n v_max_f32 %3, %3, %7", "vcc")
BENCH(k_min_u32, "v_min_u32 %0, %0, %7\n v_min_u32 %1, %1, %7\n v_min_u32 %2, %2, %7\n v_min_u32 %3, %3, %7", "vcc")
BENCH(k_and, "v_and_b32 %0, %0, %7\n v_and_b32 %1, %1, %7\n v_and_b32 %2, %2, %7\n v_and_b32 %3, %3, %7", "vcc")
BENCH(k_or3, "v_or3_b32 %0, %0, %6, %7\n v_or3_b32 %1, %1, %6, %7\n v_or3_b32 %2, %2, %6, %7\n v_or3_b32 %3, %3, %6, %7", "vcc")
BENCH(k_lshl, "v_lshlrev_b32 %0, 2, %0\n v_lshlrev_b32 %1, 2, %1\n v_lshlrev_b32 %2, 2, %2\n v_lshlrev_b32 %3, 2, %3", "vcc")
BENCH(k_sub_u32, "v_sub_u32 %0, %0, %6\n v_sub_u32 %1, %1, %6\n v_sub_u32 %2, %2, %6\n v_sub_u32 %3, %3, %6", "vcc")
BENCH(k_cvt_u32, "v_cvt_u32_f32 %0, %6\n v_cvt_u32_f32 %1, %6\n v_cvt_u32_f32 %2, %6\n v_cvt_u32_f32 %3, %6", "vcc")
BENCH(k_fract, "v_fract_f32 %0, %0\n v_fract_f32 %1, %1\n v_fract_f32 %2, %2\n v_fract_f32 %3, %3", "vcc")
BENCH(k_floor, "v_floor_f32 %0, %0\n v_floor_f32 %1, %1\n v_floor_f32 %2, %2\n v_floor_f32 %3, %3", "vcc")
// v_cndmask: VOP2 with implicit VCC, VOP3 with an SGPR pair, with inline constants, and behind the v_cmp that feeds it
BENCH(k_cndmask_e64, "v_cndmask_b32_e64 %0, %0, %6, s[10:11]\n v_cndmask_b32_e64 %1, %1, %6, s[10:11]\n v_cndmask_b32_e64 %2, %2, %6, s[10:11]\n v_cndmask_b32_e64 %3, %3, %6, s[10:11]", "s10", "s11")
BENCH(k_cndmask_c, "v_cndmask_b32 %0, 0, %6, vcc\n v_cndmask_b32 %1, 0, %6, vcc\n v_cndmask_b32 %2, 0, %6, vcc\n v_cndmask_b32 %3, 0, %6, vcc", "s10")
BENCH(k_cndmask_indep, "v_cndmask_b32 %0, %6, %7, vcc\n v_cndmask_b32 %1, %6, %7, vcc\n v_cndmask_b32 %2, %6, %7, vcc\n v_cndmask_b32 %3, %6, %7, vcc", "s10")
BENCH(k_cmp_cnd, "v_cmp_lt_f32 vcc, %0, %6\n v_cndmask_b32 %1, %1, %7, vcc\n v_cmp_lt_f32 vcc, %2, %6\n v_cndmask_b32 %3, %3, %7, vcc", "vcc")
BENCH(k_cmp_fma_cnd, "v_cmp_lt_f32 vcc, %0, %6\n v_fma_f32 %2, %2, %6, %7\n v_fma_f32 %3, %3, %6, %7\n v_cndmask_b32 %1, %1, %7, vcc", "vcc")
// EXEC masking instead of a select: is writing EXEC cheap?
BENCH(k_saveexec, "s_mov_b64 s[12:13], exec\n s_and_saveexec_b64 s[10:11], s[12:13]\n v_fma_f32 %0, %0, %6, %7\n s_mov_b64 exec, s[10:11]\n v_fma_f32 %1, %1, %6, %7", "s10", "s11", "s12", "s13", "scc")
BENCH(k_cmpx, "v_cmpx_lt_f32 %6, %7\n v_fma_f32 %0, %0, %6, %7\n s_mov_b64 exec, -1\n v_fma_f32 %1, %1, %6, %7", "vcc")
// Do the costs add?  one slow-class instruction among three fast ones (additive: (slow + 3 x 1.95) / 4)
#define FMA3 "v_fma_f32 %1, %1, %6, %7\n v_fma_f32 %2, %2, %6, %7\n v_fma_f32 %3, %3, %6, %7"
BENCH(k_mix_trunc, "v_trunc_f32 %0, %0\n " FMA3, "vcc")
BENCH(k_mix_cvt, "v_cvt_i32_f32 %0, %6\n " FMA3, "vcc")
BENCH(k_mix_med3, "v_med3_f32 %0, %0, %6, %7\n " FMA3, "vcc")
BENCH(k_mix_max, "v_max_f32 %0, %0, %7\n " FMA3, "vcc")
BENCH(k_mix_cmp, "v_cmp_lt_f32_e64 s[10:11], %0, %6\n " FMA3, "s10", "s11")
BENCH(k_mix_cnd64, "v_cndmask_b32_e64 %0, %0, %6, s[10:11]\n " FMA3, "s10", "s11")
BENCH(k_mix_rcp, "v_rcp_f32 %0, %0\n " FMA3, "vcc")
BENCH(k_mix_muls, "v_mul_f32 %0, %9, %0\n " FMA3, "vcc")
BENCH(k_mix_mad24, "v_mad_u32_u24 %0, %0, %6, %7\n " FMA3, "vcc")
BENCH(k_mix_pk, "v_pk_mul_f32 %4, %4, %8\n " FMA3, "vcc")
BENCH(k_mix2_trunc, "v_trunc_f32 %0, %0\n v_fma_f32 %1, %1, %6, %7\n v_trunc_f32 %2, %2\n v_fma_f32 %3, %3, %6, %7", "vcc")
// a select whose mask the SCALAR unit wrote last (the splat's inwin mask is an s_and of ballots)
BENCH(k_sand_cnd, "s_and_b64 vcc, s[10:11], exec\n v_cndmask_b32 %0, %0, %6, vcc\n v_fma_f32 %1, %1, %6, %7\n v_fma_f32 %2, %2, %6, %7", "vcc", "s10", "s11", "scc")
BENCH(k_sand_cnd3, "s_and_b64 vcc, s[10:11], exec\n v_cndmask_b32 %0, %0, %6, vcc\n v_cndmask_b32 %1, %1, %6, vcc\n v_cndmask_b32 %2, %2, %6, vcc", "vcc", "s10", "s11", "scc")
BENCH(k_cmp_cnd3, "v_cmp_lt_f32 vcc, %3, %6\n v_cndmask_b32 %0, %0, %6, vcc\n v_cndmask_b32 %1, %1, %6, vcc\n v_cndmask_b32 %2, %2, %6, vcc", "vcc")
// the ray body's rhythm: 12 fast, 1 slow, ... as one long block (24 fast + 6 slow + 2 SALU)
#define F4 "v_fma_f32 %0, %0, %6, %7\n v_mul_f32 %1, %1, %6\n v_add_f32 %2, %2, %7\n v_fma_f32 %3, %3, %6, %7\n"
BENCH(k_body_like, F4 "v_trunc_f32 %0, %0\n" F4 "v_cmp_lt_f32_e64 s[10:11], %1, %6\n" F4 "v_cvt_i32_f32 %2, %6\n" F4 "v_med3_f32 %3, %3, %6, %7\n s_and_b64 s[12:13], s[10:11], exec\n" F4 "v_cndmask_b32_e64 %0, %0, %6, s[10:11]\n" F4 "v_mad_u32_u24 %1, %1, %6, %7\n s_bcnt1_i32_b64 s14, s[12:13]", "s10", "s11", "s12", "s13", "s14", "scc")
// VALU beside scalar work
BENCH(k_fma_sadd, "v_fma_f32 %0, %0, %6, %7\n s_add_u32 s10, s10, 1\n v_fma_f32 %1, %1, %6, %7\n s_add_u32 s11, s11, 1", "s10", "s11", "scc")
BENCH(k_fma_sand64, "v_fma_f32 %0, %0, %6, %7\n s_and_b64 s[10:11], s[10:11], exec\n v_fma_f32 %1, %1, %6, %7\n s_or_b64 s[12:13], s[12:13], exec", "s10", "s11", "s12", "s13", "scc")
BENCH(k_cmp_bcnt, "v_cmp_lt_f32 vcc, %0, %6\n s_bcnt1_i32_b64 s10, vcc\n s_add_u32 s11, s11, s10\n v_fma_f32 %1, %1, %6, %7", "vcc", "s10", "s11", "scc")
// LDS atomics alone and beside VALU: are they hidden behind vector issue?
BENCH(k_dsadd_rtn, "ds_add_rtn_u32 %0, %10, %11\n ds_add_rtn_u32 %1, %10, %11 offset:4\n ds_add_rtn_u32 %2, %10, %11 offset:1024\n ds_add_rtn_u32 %3, %10, %11 offset:1028\n s_waitcnt lgkmcnt(0)", "vcc")
BENCH(k_dsadd, "ds_add_u32 %10, %11\n ds_add_u32 %10, %11 offset:4\n ds_add_u32 %10, %11 offset:1024\n ds_add_u32 %10, %11 offset:1028", "vcc")
BENCH(k_dsadd_u64, "ds_add_u64 %10, %4\n ds_add_u64 %10, %5 offset:8\n ds_add_u64 %10, %4 offset:1024\n ds_add_u64 %10, %5 offset:1032", "vcc")
// 32 fma + 4 LDS atomics (the splat's ratio is ~135 VALU : 4 atomics; 32 : 4 is the harsher mix)
#define FMA8 "v_fma_f32 %0, %0, %6, %7\n v_fma_f32 %1, %1, %6, %7\n v_fma_f32 %2, %2, %6, %7\n v_fma_f32 %3, %3, %6, %7\n v_fma_f32 %0, %0, %6, %7\n v_fma_f32 %1, %1, %6, %7\n v_fma_f32 %2, %2, %6, %7\n v_fma_f32 %3, %3, %6, %7\n"
BENCH(k_fma32, FMA8 FMA8 FMA8 FMA8 "s_nop 0", "vcc")
BENCH(k_fma32_dsadd, FMA8 FMA8 FMA8 FMA8 "ds_add_u32 %10, %11\n ds_add_u32 %10, %11 offset:4\n ds_add_u32 %10, %11 offset:1024\n ds_add_u32 %10, %11 offset:1028", "vcc")
BENCH(k_fma32_dsrtn, FMA8 FMA8 FMA8 FMA8 "ds_add_rtn_u32 v20, %10, %11\n ds_add_rtn_u32 v21, %10, %11 offset:4\n ds_add_rtn_u32 v22, %10, %11 offset:1024\n ds_add_rtn_u32 v23, %10, %11 offset:1028", "vcc", "v20", "v21", "v22", "v23")
BENCH(k_fma32_dsread, FMA8 FMA8 FMA8 FMA8 "ds_read_b32 v20, %10\n ds_read_b32 v21, %10 offset:4\n ds_read_b32 v22, %10 offset:1024\n ds_read_b32 v23, %10 offset:1028", "vcc", "v20", "v21", "v22", "v23")
BENCH(k_fma32_dsread2, FMA8 FMA8 FMA8 FMA8 "ds_read2_b32 v[20:21], %10 offset1:1\n ds_read2_b32 v[22:23], %10 offset0:64 offset1:65", "vcc", "v20", "v21", "v22", "v23")

struct Result { const char* name; int wps; double cyc; double ghz; };

template <typename K>
static void run(const char* name, K kern, int n_instr, float* d_out, unsigned long long* d_stamps, std::vector<Result>& res)
{
    const int iters = 1500;
    for (int wps : {1, 2, 3, 4}) {
        const int threads = 256 * wps, blocks = 256, waves = blocks * threads / 64;
        for (int rep = 0; rep < 3; ++rep) kern<<<blocks, threads>>>(d_out, d_stamps, iters, 1.0f);     // warm the clock governor
        const hipError_t err = hipDeviceSynchronize();
        const hipError_t err2 = hipGetLastError();
        if (err != hipSuccess || err2 != hipSuccess) { printf("%s: error %s / %s\n", name, hipGetErrorString(err), hipGetErrorString(err2)); return; }
        std::vector<unsigned long long> s(4 * (size_t)waves);
        (void)hipMemcpy(s.data(), d_stamps, s.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> dt(waves), f(waves);
        for (int w = 0; w < waves; ++w) {
            dt[w] = (double)(s[4 * w + 1] - s[4 * w]);
            f[w] = dt[w] / (double)(s[4 * w + 3] - s[4 * w + 2]) * 0.1;    // GHz: shader ticks per 10 ns tick
        }
        std::nth_element(dt.begin(), dt.begin() + waves / 2, dt.end());
        std::nth_element(f.begin(), f.begin() + waves / 2, f.end());
        const double cyc = dt[waves / 2] / ((double)iters * kUnroll * n_instr * wps);
        res.push_back({name, wps, cyc, f[waves / 2]});
        printf("%-22s waves/SIMD=%d  %6.2f cycles per instruction per SIMD   clock %.2f GHz\n", name, wps, cyc, f[waves / 2]);
    }
}

int main(int argc, char** argv)
{
    setvbuf(stdout, nullptr, _IOLBF, 0);
    float* d_out; (void)hipMalloc(&d_out, 256 * 1024 * 4);
    unsigned long long* d_stamps; (void)hipMalloc(&d_stamps, 256 * 16 * 4 * 8);
    std::vector<Result> res;
#define RUN(K, N) run(#K, K, N, d_out, d_stamps, res)
    RUN(k_fma, 4); RUN(k_mul, 4); RUN(k_add, 4); RUN(k_mul_s, 4); RUN(k_fma_dep, 4);
    RUN(k_pkfma, 4); RUN(k_pkmul, 4); RUN(k_pkadd, 4); RUN(k_pk_fma_mix, 4);
    RUN(k_trunc, 4); RUN(k_rpi, 4); RUN(k_cvt, 4); RUN(k_rcp, 4); RUN(k_med3, 4);
    RUN(k_cmp_vcc, 4); RUN(k_cmp_e64, 4); RUN(k_cndmask, 4); RUN(k_mad24, 4); RUN(k_iadd, 4); RUN(k_lshl_add, 4);
    RUN(k_mov, 4); RUN(k_dpp, 4);
    RUN(k_add_inline, 4); RUN(k_fma_inline, 4); RUN(k_fmamk_lit, 4); RUN(k_fmaak_lit, 4); RUN(k_fmac, 4); RUN(k_fma_neg, 4);
    RUN(k_mul_abs, 4); RUN(k_fma_s, 4); RUN(k_sub, 4); RUN(k_max, 4); RUN(k_min_u32, 4); RUN(k_and, 4); RUN(k_or3, 4);
    RUN(k_lshl, 4); RUN(k_sub_u32, 4); RUN(k_cvt_u32, 4); RUN(k_fract, 4); RUN(k_floor, 4);
    RUN(k_cndmask_e64, 4); RUN(k_cndmask_c, 4); RUN(k_cndmask_indep, 4); RUN(k_cmp_cnd, 4); RUN(k_cmp_fma_cnd, 4);
    RUN(k_saveexec, 4); RUN(k_cmpx, 4);
    RUN(k_mix_trunc, 4); RUN(k_mix_cvt, 4); RUN(k_mix_med3, 4); RUN(k_mix_max, 4); RUN(k_mix_cmp, 4); RUN(k_mix_cnd64, 4);
    RUN(k_mix_rcp, 4); RUN(k_mix_muls, 4); RUN(k_mix_mad24, 4); RUN(k_mix_pk, 4); RUN(k_mix2_trunc, 4);
    RUN(k_sand_cnd, 4); RUN(k_sand_cnd3, 4); RUN(k_cmp_cnd3, 4); RUN(k_body_like, 32);
    RUN(k_fma_sadd, 4); RUN(k_fma_sand64, 4); RUN(k_cmp_bcnt, 4);
    RUN(k_dsadd_rtn, 4); RUN(k_dsadd, 4); RUN(k_dsadd_u64, 4);
    RUN(k_fma32, 32); RUN(k_fma32_dsadd, 32); RUN(k_fma32_dsrtn, 32); RUN(k_fma32_dsread, 32); RUN(k_fma32_dsread2, 32);
    if (argc > 1) {
        if (FILE* fo = fopen(argv[1], "w")) {
            fprintf(fo, "{\"unit\": \"shader cycles per wave64 instruction per SIMD (s_memtime), one workgroup per CU\", "
                        "\"note\": \"k_fma32_* rows count the 32 v_fma only: the excess over k_fma32 is what the 4 LDS operations cost\", \"rows\": [\n");
            for (size_t i = 0; i < res.size(); ++i)
                fprintf(fo, "  {\"kernel\": \"%s\", \"waves_per_simd\": %d, \"cycles\": %.3f, \"clock_ghz\": %.3f}%s\n", res[i].name,
                        res[i].wps, res[i].cyc, res[i].ghz, i + 1 < res.size() ? "," : "");
            fprintf(fo, "]}\n");
            fclose(fo);
        }
    }
    return 0;
}
