// trace_kernels.hip - fused heliostat ray trace (forward + backward) for gfx950 / MI355X.
//
// One thread owns one surface point (= ray origin) of one heliostat and walks a chunk of the R
// distortion samples of that point in registers: reflect once per point, then per ray
// scatter -> plane hit -> bilinear splat.  No per-ray intermediate ever reaches HBM: the only
// per-ray traffic is the 8 B (u,e) distortion pair, read coalesced (consecutive lanes =
// consecutive points of the same sample r).
//
// Replaces (ARTIST v2.0.0): heliostat_ray_tracer.py:285-290 (reflect), :328-335 + :510-561
// (scatter_rays), :390-409 (line_plane_intersections), :411-433 + geometry.py:207-445 (cylindrical
// receivers, template parameter CYL), :435-480 + blocking.py:212-354 (soft blocking mask over the
// rectangles art_blocking_filter selected, template parameter BLOCKING), :482-487 (intensities),
// :489-494 + :610-778 (bilinear_splatting), :498-506 (factors), :563-608 (per-target sums, mode 1).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <functional>
#include <cstdio>
#include <mutex>
#include <vector>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "trace_common.hpp"
#include "flux_moments.hpp"

namespace art {

// round-half-up float -> int32 in one instruction (floor(x + 0.5)); 0 <= x < 2^22 by construction.
__device__ __forceinline__ unsigned cvt_nearest_u32(float x)
{
    int q;
    asm("v_cvt_rpi_i32_f32_e32 %0, %1" : "=v"(q) : "v"(x));
    return (unsigned)q;
}

// A contribution that bypasses the window (a stray ray): |v| in accumulator units, rounded to nearest.
__device__ __forceinline__ unsigned long long to_accum(float v, float scale_g) { return (unsigned long long)cvt_nearest_u32(fabsf(v) * scale_g); }
// The windows' cell unit in accumulator units (log2), and a weighted intensity in cell units handed to an accumulator: w * Is
// with Is = |I| * (cell scale), the very product a window ray adds to its LDS cell.
constexpr int kCellShift = 7;
__device__ __forceinline__ unsigned long long cell_to_accum(float w_times_Is) { return (unsigned long long)cvt_nearest_u32(w_times_Is) << kCellShift; }

// --------------------------------------------------------------------------------------------
// Forward, global-atomic splat (ARTIST_HIP_FWD=global: the plain formulation, kept as an independent cross-check of the
// windowed kernels; same integer accumulators, so also bit-reproducible).
// grid.x = H * n_rchunks * n_ptiles ; block = 256.
// counts: uint32 [2,H] (aliases rows 0,1 of the factors output until finalize_factors runs).
// --------------------------------------------------------------------------------------------
template <bool INTERLEAVED>
__global__ __launch_bounds__(kBlock) void trace_fwd_kernel(TraceArgs a, float* __restrict__ flux,
                                                           unsigned int* __restrict__ counts)
{
    const int bid = blockIdx.x;
    const int ptile = bid % a.n_ptiles;
    const int rchunk = (bid / a.n_ptiles) % a.n_rchunks;
    const int h = bid / (a.n_ptiles * a.n_rchunks);
    const int p = ptile * kBlock + threadIdx.x;
    const bool active = p < a.P;

    const int t = a.target_idx[h];
    if (!target_in_range(a, t) || t >= a.T) return;   // planar receivers only (host refuses cylinders for this variant)
    const Plane pl = load_plane(a.centers, a.pnormals, a.dims, t, a.W, a.Hh, a.mag, a.k_ext, a.k_refl);
    unsigned long long* __restrict__ bitmap = a.accum + (int64_t)(a.mode == 0 ? h : t) * a.Hh * a.W;   // integer accumulators

    unsigned n_on = 0, n_int = 0;
    if (active) {
        const float4 o = a.origins[(int64_t)h * a.P + p];
        const float4 n = a.normals[(int64_t)h * a.P + p];
        const float4 inc = a.incident[h];
        float4 d; float s;
        reflect(inc, n, d, s);
        const float numer = plane_numer(pl, o);

        const int r0 = rchunk * a.r_chunk;
        const int r1 = min(r0 + a.r_chunk, a.R);
        int64_t off = (int64_t)h * a.sh + (int64_t)r0 * a.sr + (int64_t)p * a.sp;
        for (int r = r0; r < r1; ++r, off += a.sr) {
            float u, e;
            load_dist<INTERLEAVED>(a, off, u, e);
            const Rot m = make_rot(e, u);
            float rx, ry, rz;
            scatter(m, d, rx, ry, rz);
            const Hit hit = intersect(pl, o, numer, rx, ry, rz);
            const float I = ((hit.I0 * 1.0f) * pl.k_ext) * pl.k_refl;   // (1 - blocked) == 1
            n_on += hit.I0 > 0.0f;
            n_int += I > 0.0f;
            const Splat sp = splat_weights(hit.be, hit.bu, a.W, a.Hh);
            if (sp.on) {
                // flat row k is output row Hh-1-k (flip, heliostat_ray_tracer.py:778)
                unsigned long long* row_hi = bitmap + (int64_t)(a.Hh - 2 - sp.iu) * a.W + sp.ie;   // flat row iu+1
                unsigned long long* row_lo = row_hi + a.W;                                          // flat row iu
                // rounded to the windowed kernels' cell unit, product by product as in trace_fwd_item: the two formulations
                // then give the same bits
                const float Is = fabsf(I) * (a.scale_g * (1.0f / (float)(1 << kCellShift)));
                atomicAdd(row_hi, cell_to_accum(sp.cle * sp.chu * Is));        // pixel 1
                atomicAdd(row_hi + 1, cell_to_accum(sp.che * sp.chu * Is));    // pixel 2
                atomicAdd(row_lo + 1, cell_to_accum(sp.che * sp.clu * Is));    // pixel 3
                atomicAdd(row_lo, cell_to_accum(sp.cle * sp.clu * Is));        // pixel 4
            }
        }
    }
    // factors: block-reduce the two counters, one atomic pair per block.
    __shared__ unsigned s_cnt[2];
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    n_on = wave_sum_u32(n_on);
    n_int = wave_sum_u32(n_int);
    if ((threadIdx.x & 63) == 0) { atomicAdd(&s_cnt[0], n_int); atomicAdd(&s_cnt[1], n_on); }
    __syncthreads();
    if (threadIdx.x < 2 && s_cnt[threadIdx.x]) atomicAdd(&counts[threadIdx.x * a.H + h], s_cnt[threadIdx.x]);
}

// --------------------------------------------------------------------------------------------
// Forward, LDS-privatised splat (the production kernel).
//
// A workgroup owns a block of `p_block` consecutive surface points of one heliostat and a chunk
// of its distortion samples.  Rays of neighbouring mirror points land within a few pixels of each
// other (plus the sun-shape blur), so the workgroup's footprint on the receiver is a small window
// of the bitmap: it is accumulated in LDS and flushed ONCE with row-contiguous global atomics (one
// per touched pixel instead of four per ray - global float atomics execute at the memory side at
// ~2e10 scattered lanes/s and bound the kernel otherwise).
//
// The LDS accumulator is 32-bit FIXED POINT with carry-out, not float: measured on gfx950
// (tools/lds_atomic_bench.hip) ds_add_f32 retires one lane every ~3 cycles (193 cycles per wave
// instruction, for any address pattern) whereas ds_add_rtn_u32 takes ~12 cycles per wave instruction
// on random addresses.  All contributions of one launch have the sign of mag*k_ext*k_refl, so |v| is
// scaled by a power of two S chosen per workgroup such that |v| S < 2^22, rounded to an integer
// (quantum 2^-22 of the largest possible contribution; the rounding errors are independent and
// average out over the rays that share a pixel) and added exactly.  When a 32-bit cell wraps, the
// returning atomic shows it and 2^32/S goes straight to the global pixel; the cell residue follows
// at the flush.  The window sum is therefore independent of the order of the adds.  4-byte cells
// give a 36 864-pixel window in 144 KB, which the oblique near-field heliostats need.
//
//   phase 1  window:  chief rays (no scatter) of the block's points -> bounding box in pixels,
//                     padded by the largest scatter angle seen in the block's first sample x the
//                     longest path (a guess that only affects speed: rays that fall outside the
//                     window take the global-atomic path, so the result never depends on it).
//   phase 2  trace:   thread <-> point, loop over the chunk's samples, 4 LDS adds per ray.
//   phase 3  flush:   window rows -> global bitmap (up-down flipped).
//
// grid.x = H * n_pblocks * n_rchunks ; block = any multiple of 64 ; dynamic LDS = 4 B * tile_cap.
// --------------------------------------------------------------------------------------------
struct Window {
    int e0, u0, tw, th;   // union window: origin (un-flipped flat coordinates) and size
    int ths, npass;       // rows per pass (tw*ths <= tile_cap) and number of passes; pass k covers flat rows
                          // [u0 + k(ths-1), u0 + k(ths-1) + ths): consecutive passes share one row because a
                          // ray splats rows iu and iu+1 and belongs to the pass that holds row iu
    float scale;          // S  (power of two): window cells count units of 1 / S
    int shift;            // log2(global accumulator units per cell unit) = (28 - ex_g) - log2 S  (>= 0)
    int pad_e, pad_u;     // how far (pixels) a point's scattered rays land from its chief ray (~4.2 sigma of the sun shape)
};

// Wave reductions on the DPP network (VALU ops, a few cycles each) instead of __shfl_xor (ds_bpermute through the LDS
// crossbar, ~100 cycles each): the window phase reduces 13 quantities per wave and was 2 us of shuffles.
enum { kMin, kMax, kSum };
template <int OP> __device__ __forceinline__ float red_op(float x, float y)
{
    if constexpr (OP == kMin) return fminf(x, y);
    else if constexpr (OP == kMax) return fmaxf(x, y);
    else return x + y;
}
template <int OP> __device__ __forceinline__ float red_identity() { return OP == kMin ? 3.0e38f : (OP == kMax ? -3.0e38f : 0.0f); }
template <int CTRL, int ROW_MASK> __device__ __forceinline__ float dpp_f32(float old, float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL,
                                                                 ROW_MASK, 0xF, false));
}
// reduction over the 16 lanes of each DPP row; every lane of a row ends with its row's result
template <int OP> __device__ __forceinline__ float row_reduce(float v)
{
    const float id = red_identity<OP>();
    v = red_op<OP>(v, dpp_f32<0xB1, 0xF>(id, v));     // quad_perm [1,0,3,2]
    v = red_op<OP>(v, dpp_f32<0x4E, 0xF>(id, v));     // quad_perm [2,3,0,1]
    v = red_op<OP>(v, dpp_f32<0x141, 0xF>(id, v));    // row_half_mirror
    v = red_op<OP>(v, dpp_f32<0x140, 0xF>(id, v));    // row_mirror
    return v;
}
// reduction over the wave; the (wave-uniform) result is returned to every lane
template <int OP> __device__ __forceinline__ float wave_reduce(float v)
{
    const float id = red_identity<OP>();
    v = row_reduce<OP>(v);
    v = red_op<OP>(v, dpp_f32<0x142, 0xA>(id, v));    // row_bcast15: rows 1, 3 take in rows 0, 2
    v = red_op<OP>(v, dpp_f32<0x143, 0xC>(id, v));    // row_bcast31: rows 2, 3 take in lane 31 (rows 0 + 1)
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ float wave_min_f32(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float wave_sum_all_f32(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max_f32(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// q = round(|v| S) as an unsigned fixed-point increment.
__device__ __forceinline__ unsigned to_fixed(float v, float scale) { return cvt_nearest_u32(fabsf(v) * scale); }

// Four returning LDS adds of one ray whose results are examined one ray LATER (software pipelining:
// the returns have long arrived by then, so no wave ever waits on the LDS round trip).
struct PendingSplat {
    unsigned o1, o2, o3, o4;   // cell values before the add
    unsigned q1, q2, q3, q4;   // increments (0 = nothing pending)
    int ie, iu;                // low pixel of the pending ray (pointer math only if a carry happened)
};

// A cell wrapped iff old + q < old (unsigned).  Rare: a cell holds ~2^10 full-size contributions.  The 2^32 cell units
// go straight to the pixel's accumulator.
__device__ __forceinline__ void resolve_carries(const PendingSplat& ps, unsigned long long* __restrict__ acc, int W, int Hh,
                                                int shift)
{
    const bool c1 = ps.o1 + ps.q1 < ps.o1, c2 = ps.o2 + ps.q2 < ps.o2;
    const bool c3 = ps.o3 + ps.q3 < ps.o3, c4 = ps.o4 + ps.q4 < ps.o4;
    if (__builtin_expect(wave_any(c1 | c2 | c3 | c4), 0)) {
        const unsigned long long carry = 1ull << (32 + shift);
        unsigned long long* row_hi = acc + (int64_t)(Hh - 2 - ps.iu) * W + ps.ie;   // flat row iu + 1 of the flipped bitmap
        unsigned long long* row_lo = row_hi + W;
        if (c1) atomicAdd(row_hi, carry);
        if (c2) atomicAdd(row_hi + 1, carry);
        if (c3) atomicAdd(row_lo + 1, carry);
        if (c4) atomicAdd(row_lo, carry);
    }
}

// Phase 1 of the windowed kernels: bounding box (in un-flipped bitmap coordinates) of where the
// workgroup's rays can land, clipped to `tile_cap` pixels, plus the fixed-point scale of the forward
// accumulator.  Result is left in *s_win after a __syncthreads().
// A thread's first surface point and its first distortion sample, loaded by the caller ahead of phase 1.
struct FirstPoint { float4 o, n; float u, e; };

// The point a thread looks at in a SAMPLED window phase (more points than threads): evenly spaced over the block, first
// point included, in mirror order.
__device__ __forceinline__ int window_sample_point(int p0, int p1)
{
    return p0 + (int)(((unsigned)threadIdx.x * (unsigned)(p1 - p0)) / blockDim.x);
}

// What the window phase collects per thread, wave and workgroup: the bounding box of the chief-ray hits, their first and
// second moments, the hit-point travel per radian of scatter (ke / ku: metres along world E / U, |t| sqrt(1 + (r_E/a)^2) -
// the footprint of an oblique beam is stretched by the obliquity along the projection of the ray only), the largest scatter
// angle of the sample looked at and the largest squared length of a reflected direction.
struct WindowStats {
    float emin = 3.0e38f, emax = -3.0e38f, umin = 3.0e38f, umax = -3.0e38f, ke = 0.0f, ku = 0.0f, angmax = 0.0f;
    float dmax2 = 0.0f, esum = 0.0f, usum = 0.0f, cnt = 0.0f, esq = 0.0f, usq = 0.0f;
};

// One surface point's chief ray (no scatter) and its first distortion sample (u, e) -> the thread's statistics.
template <bool CYL>
__device__ __forceinline__ void window_consume(WindowStats& w, const Plane& pl, const Cyl& cy, const float4 inc, const float4 o,
                                               const float4 n, const float u, const float e)
{
    float4 d; float s;
    reflect(inc, n, d, s);
    w.dmax2 = fmaxf(w.dmax2, d.x * d.x + d.y * d.y + d.z * d.z);
    bool valid; float hbe, hbu, hke, hku;
    if constexpr (CYL) {
        // chief ray on the cylinder; hit-point travel per radian ~ t / cos(incidence) along both axes
        const CylPoint cp = cyl_point(cy, o);
        const CylHit ch = cyl_hit(cy, cp, d.x, d.y, d.z);
        valid = ch.ok; hbe = ch.be; hbu = ch.bu;
        hke = hku = ch.t / fmaxf(ch.abi, 0.1f);
    } else {
        const float numer = plane_numer(pl, o);
        const Hit hit = intersect(pl, o, numer, d.x, d.y, d.z);
        valid = hit.valid; hbe = hit.be; hbu = hit.bu;
        const float ia = 1.0f / hit.a, qe = d.x * ia, qu = d.z * ia;
        hke = hit.t * sqrtf(1.0f + qe * qe);
        hku = hit.t * sqrtf(1.0f + qu * qu);
    }
    if (valid) {
        w.emin = fminf(w.emin, hbe); w.emax = fmaxf(w.emax, hbe);
        w.umin = fminf(w.umin, hbu); w.umax = fmaxf(w.umax, hbu);
        w.esum += hbe; w.usum += hbu; w.cnt += 1.0f;
        w.esq += hbe * hbe; w.usq += hbu * hbu;
        w.ke = fmaxf(w.ke, hke);
        w.ku = fmaxf(w.ku, hku);
    }
    w.angmax = fmaxf(w.angmax, fmaxf(fabsf(u), fabsf(e)));
}

// Threads' statistics -> the workgroup's, valid in thread 0 (one __syncthreads inside; s_red is free for reuse after the
// caller's next barrier).
__device__ __forceinline__ void window_reduce(WindowStats& w, float (*s_red)[16])
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    w.emin = wave_reduce<kMin>(w.emin); w.emax = wave_reduce<kMax>(w.emax); w.umin = wave_reduce<kMin>(w.umin); w.umax = wave_reduce<kMax>(w.umax);
    w.ke = wave_reduce<kMax>(w.ke); w.ku = wave_reduce<kMax>(w.ku); w.angmax = wave_reduce<kMax>(w.angmax); w.dmax2 = wave_reduce<kMax>(w.dmax2);
    w.esum = wave_reduce<kSum>(w.esum); w.usum = wave_reduce<kSum>(w.usum); w.cnt = wave_reduce<kSum>(w.cnt);
    w.esq = wave_reduce<kSum>(w.esq); w.usq = wave_reduce<kSum>(w.usq);
    if (lane == 0) {
        s_red[8][wave] = w.esum; s_red[9][wave] = w.usum; s_red[10][wave] = w.cnt; s_red[11][wave] = w.esq; s_red[12][wave] = w.usq;
        s_red[0][wave] = w.emin; s_red[1][wave] = w.emax; s_red[2][wave] = w.umin; s_red[3][wave] = w.umax;
        s_red[4][wave] = w.ke; s_red[5][wave] = w.ku; s_red[6][wave] = w.angmax; s_red[7][wave] = w.dmax2;
    }
    __syncthreads();
    if (wave == 0) {
        // the (at most 16) wave partials sit in the lanes of DPP row 0: one row reduction per quantity
        const bool has = lane < nwaves;
        const int k = has ? lane : 0;
        w.emin = row_reduce<kMin>(has ? s_red[0][k] : 3.0e38f); w.emax = row_reduce<kMax>(has ? s_red[1][k] : -3.0e38f);
        w.umin = row_reduce<kMin>(has ? s_red[2][k] : 3.0e38f); w.umax = row_reduce<kMax>(has ? s_red[3][k] : -3.0e38f);
        w.ke = row_reduce<kMax>(has ? s_red[4][k] : -3.0e38f); w.ku = row_reduce<kMax>(has ? s_red[5][k] : -3.0e38f);
        w.angmax = row_reduce<kMax>(has ? s_red[6][k] : -3.0e38f); w.dmax2 = row_reduce<kMax>(has ? s_red[7][k] : -3.0e38f);
        w.esum = row_reduce<kSum>(has ? s_red[8][k] : 0.0f); w.usum = row_reduce<kSum>(has ? s_red[9][k] : 0.0f);
        w.cnt = row_reduce<kSum>(has ? s_red[10][k] : 0.0f); w.esq = row_reduce<kSum>(has ? s_red[11][k] : 0.0f);
        w.usq = row_reduce<kSum>(has ? s_red[12][k] : 0.0f);
    }
}

// The workgroup's statistics -> its window (thread 0).  ppm_e / ppm_u: pixels per metre on the receiver surface; pad_k: how
// much of the largest scatter angle SEEN is kept as the pad (the sample looked at holds the extreme of a few thousand
// draws, a little below what is worth keeping in the window: ~4.2 sigma).
__device__ __forceinline__ Window window_decide(const TraceArgs& a, const WindowStats& w, const float ppm_e, const float ppm_u,
                                                const float pad_k)
{
    // The cell unit is the SAME for every item of a launch - 2^(kCellShift) accumulator units, i.e. 2^-21 of 2^ex_g
    // > |mag k_ext k_refl| - and a stray ray is rounded to it like a window ray (to_cell): a ray contributes the same
    // integer whichever workgroup traces it and whether or not it meets a window, so the bitmap does not depend on
    // the launch geometry.  |contribution| <= |k| |d| |m| (the scatter matrix is a rotation; cylinder: |d| ||R_xy||_F
    // = sqrt 2 |d|), i.e. < 2^22 cell units for unit normals; a returning add copes with up to 2^31.
    // (scale and shift are set even for an EMPTY window: its rays are all strays, and strays are rounded to the cell unit)
    Window win = {0, 0, 0, 0, 0, 1, a.scale_g * (1.0f / (float)(1 << kCellShift)), kCellShift, 0, 0};
    if (w.emax >= w.emin) {
        // + 2 px for the bilinear footprint and rounding
        const float pad_e = fminf(pad_k * w.angmax * w.ke * ppm_e + 2.0f, 32768.0f);
        const float pad_u = fminf(pad_k * w.angmax * w.ku * ppm_u + 2.0f, 32768.0f);
        int e0 = max((int)w.emin - (int)pad_e, 0), e1 = min((int)w.emax + 1 + (int)pad_e, a.W - 1);
        int u0 = max((int)w.umin - (int)pad_u, 0), u1 = min((int)w.umax + 1 + (int)pad_u, a.Hh - 1);
        int tw = e1 - e0 + 1, th = u1 - u0 + 1;
        if ((int64_t)tw * th > a.tile_cap && (int64_t)tw * th <= (int64_t)a.tile_cap * a.multipass_ratio) {
            // Too large, but by less than multipass_ratio: keep the densest part, centred on the mean
            // chief-ray hit, and let the tails take the global-memory path.  A stray ray costs more than a
            // window ray (scattered global atomics) while a second pass costs every ray 2x, so trimming
            // wins as long as the tails hold a few percent of the rays - which a peaked (Gaussian-like) footprint does.
            // Aspect ratio from the second moments of the chief-ray hits widened by the scatter pad
            // (pad ~ 4.25 sigma of the sun shape): the window spans the same number of standard deviations
            // along E and U, which minimises the stray fraction of a Gaussian-like footprint.
            const float n1 = fmaxf(w.cnt, 1.0f);
            const float me = w.esum / n1, mu = w.usum / n1;
            const float se = sqrtf(fmaxf(w.esq / n1 - me * me, 0.0f) + (pad_e * pad_e) * (1.0f / 18.0f)) + 0.5f;
            const float su = sqrtf(fmaxf(w.usq / n1 - mu * mu, 0.0f) + (pad_u * pad_u) * (1.0f / 18.0f)) + 0.5f;
            const float kk = sqrtf((float)a.tile_cap / (se * su));
            int tw2 = max(2, min(tw, (int)(kk * se)));
            int th2 = max(2, min(th, a.tile_cap / tw2));
            tw2 = max(2, min(tw, a.tile_cap / th2));             // hand back what the clamp on th freed
            const int ce = (int)me, cu = (int)mu;
            e0 = min(max(ce - tw2 / 2, e0), e0 + tw - tw2);
            u0 = min(max(cu - th2 / 2, u0), u0 + th - th2);
            tw = tw2; th = th2;
        }
        if (tw > a.tile_cap / 2) { e0 += (tw - a.tile_cap / 2) / 2; tw = a.tile_cap / 2; }   // absurdly wide bitmaps
        win.e0 = e0; win.u0 = u0; win.tw = tw; win.th = th;
        win.pad_e = (int)pad_e; win.pad_u = (int)pad_u;
        // larger footprints (near, oblique heliostats) are swept in several passes over row bands
        win.ths = min(th, a.tile_cap / tw);
        win.npass = win.ths >= th ? 1 : (th - 1 + win.ths - 2) / (win.ths - 1);
    }
    return win;
}

template <bool INTERLEAVED, bool CYL>
__device__ __forceinline__ void compute_window(const TraceArgs& a, const Plane& pl, const Cyl& cy, const float4 inc,
                                               const float4* __restrict__ org, const float4* __restrict__ nrm,
                                               int p0, int p1, int64_t dbase, float (*s_red)[16], Window* s_win,
                                               const FirstPoint* first = nullptr)
{
    const int tid = threadIdx.x;
    WindowStats w;
    // One point per trip, the NEXT point's loads issued before the current one is worked on: a workgroup that owns
    // many points per thread (one sun sample per point at field scale: 10 trips) is otherwise a chain of exposed
    // HBM round trips (30 us of a 77 us workgroup, tools/timeline.sh).
    // a.win_sample: the window is a guess that only affects speed, so a block with more points than threads takes ONE
    // evenly spaced point per thread (window_sample_point) instead of walking all of them in dependent trips
    const bool sampled = a.win_sample != 0 && p1 - p0 > (int)blockDim.x;
    {
        int p = sampled ? window_sample_point(p0, p1) : p0 + tid;
        bool have = p < p1;
        float4 o = {0.0f, 0.0f, 0.0f, 1.0f}, n = {0.0f, 0.0f, 1.0f, 0.0f};
        float u = 0.0f, e = 0.0f;
        if (have) {
            if (first != nullptr) { o = first->o; n = first->n; u = first->u; e = first->e; }      // came with the caller
            else { o = org[p]; n = nrm[p]; load_dist<INTERLEAVED>(a, dbase + (int64_t)p * a.sp, u, e); }
        }
        while (have) {
            const int pn = p + (int)blockDim.x;
            const bool have_n = !sampled && pn < p1;
            float4 o2 = o, n2 = n;
            float u2 = 0.0f, e2 = 0.0f;
            if (have_n) { o2 = org[pn]; n2 = nrm[pn]; load_dist<INTERLEAVED>(a, dbase + (int64_t)pn * a.sp, u2, e2); }
            window_consume<CYL>(w, pl, cy, inc, o, n, u, e);
            o = o2; n = n2; u = u2; e = e2; p = pn; have = have_n;
        }
    }
    window_reduce(w, s_red);
    if (tid == 0) {
        float ppm_e, ppm_u;   // pixels per metre on the receiver surface
        if constexpr (CYL) { ppm_e = cy.wm1 / fabsf(cy.opening * sqrtf(cy.r2)); ppm_u = cy.hm1 / fabsf(cy.height); }
        else { ppm_e = pl.wm1 / fabsf(pl.w); ppm_u = pl.hm1 / fabsf(pl.h); }
        // x1.15: the first sample's extreme (~3.7 sigma over 2 p_block draws) is a little below what is worth keeping in
        // the window (~4.2 sigma); a sampled window phase sees the extreme of fewer draws (1024 of 2500 points: ~3.65 sigma)
        *s_win = window_decide(a, w, ppm_e, ppm_u, sampled ? 1.15f * 1.06f : 1.15f);
    }
    __syncthreads();
}

// Blocking rectangles of one heliostat in LDS (empty when the kernel is instantiated without blocking).
// (grad is DOUBLE: on gfx950 ds_add_f64 retires a wave instruction in ~25 cycles, ds_add_f32 in ~193 - tools/lds_atomic_bench.hip)
// (GRAD = false: the forward items - no gradient sums, 3 KB less static LDS: room for 4 KB more window)
// (n_wide / xmask: a heliostat with more candidates than the tables hold - see "wide heliostats" below)
constexpr int kWideFirst = kMaxCand - 1;     // rectangles of such a heliostat that live in the tables
template <bool BLOCKING, bool GRAD = true> struct PrimTable {
    Prim prim[kMaxCand]; PrimAux aux[kMaxCand]; int id[kMaxCand]; double grad[GRAD ? kMaxCand * 12 : 1];
    int n_wide; unsigned long long xmask[16];
};
template <bool GRAD> struct PrimTable<false, GRAD> { Prim prim[1]; PrimAux aux[1]; int id[1]; double grad[1]; };

// candidates of heliostat h -> LDS; returns their number (workgroup-uniform).  Ends with a barrier.
template <bool BLOCKING, bool GRAD = true>
__device__ __forceinline__ int load_prims(const TraceArgs& a, int h, PrimTable<BLOCKING, GRAD>& tab)
{
    if constexpr (!BLOCKING) return 0;
    else {
    const int listed = min(a.cand_count[h], a.Cmax);
    // a list longer than the tables: kWideFirst rectangles in the tables, the others are read from the list ("wide heliostats")
    const int n = listed > kMaxCand ? kWideFirst : listed;
    if (threadIdx.x == 0) tab.n_wide = listed > kMaxCand ? listed : 0;
    for (int c = threadIdx.x; c < n; c += blockDim.x) {
        const int k = a.cand[(int64_t)h * a.Cmax + c];
        tab.id[c] = k;
        tab.prim[c] = make_prim(a.prim_corners, a.prim_spans, a.prim_normals, k);
        tab.aux[c] = make_prim_aux(tab.prim[c]);
    }
    if constexpr (GRAD) for (int c = threadIdx.x; c < n * 12; c += blockDim.x) tab.grad[c] = 0.0;
    __syncthreads();
    return n;
    }
}

// ---- "wide" heliostats: more than kMaxCand candidate rectangles ---------------------------------------------------------
// The tables above and the lanes' 32-bit masks serve a heliostat's first kMaxCand candidates - every heliostat of every field
// met so far.  A heliostat with a longer list (a dense row field under a low sun) is not refused: the soft mask is
// exp(-alpha sum sigma), so the candidates beyond the tables add their sigmas to the same sum.  Such a heliostat keeps
// kWideFirst = kMaxCand - 1 rectangles in the tables (load_prims; PrimTable::n_wide = the length of its list), and bit 31 of
// its masks stands for all the others: set in a wave's mask when a ray of the wave's points can reach one of them (so the
// kernels' one test "wmask != 0" still decides whether a ray looks at rectangles at all, and their hot loops carry nothing
// new), set in a ray's `near` mask when the ray entered the soft edge of one of them.  The others are read from the caller's
// tables (wave-uniform addresses), their constants rebuilt per use (make_prim), and every lane evaluates them with the same
// functions as the tabled ones - soft_plane, soft_uv, soft_sigma: a ray outside a rectangle's soft edge fails their own tests,
// which is all the lane masks ever anticipated.  Per point a wave keeps (in LDS, PrimTable::xmask) a 64-bit mask of the first 64
// such candidates its rays can reach at all (bounding sphere against the cone of the scattered rays, as in cone_mask);
// candidates beyond those 64 are evaluated for every ray.  Everything here runs behind "bit 31 is set": cold code.
struct WideTabs {
    const float* corners; const float* spans; const float* normals;
    const int32_t* row;        // this heliostat's candidate list
    double* grad_row;          // backward: its [n - kWideFirst, 12] gradient sums (NULL: forward)
    float cone_cos, cone_sin;
    int n;                     // length of the list (> kMaxCand)
};
// rows of an item's slab of rectangle gradients: the candidates of the tables (the others' sums: TraceArgs::wide_grad)
__host__ __device__ __forceinline__ int slab_rows(const TraceArgs& a) { return a.Cmax < kMaxCand ? a.Cmax : kMaxCand; }
__device__ __forceinline__ WideTabs wide_tabs(const TraceArgs& a, int h, int n)
{
    WideTabs w;
    w.corners = a.prim_corners; w.spans = a.prim_spans; w.normals = a.prim_normals;
    w.row = a.cand + (int64_t)h * a.Cmax;
    w.grad_row = a.wide_grad != nullptr ? a.wide_grad + ((int64_t)h * (a.Cmax - kWideFirst)) * 12 : nullptr;
    w.cone_cos = a.cone_cos; w.cone_sin = a.cone_sin;
    w.n = n;
    return w;
}
__device__ __forceinline__ Prim wide_prim(const WideTabs& w, int c)
{
    const int k = __builtin_amdgcn_readfirstlane(w.row[c]);
    return make_prim(w.corners, w.spans, w.normals, k);
}
// One point per lane of a wide heliostat (o, chief direction d - not normalised): which of the candidates kWideFirst ...
// kWideFirst + 63 can a ray of this wave's points touch -> the wave's word of tab.xmask; returns bit 31 if any can, or if the
// list goes on beyond them.  0 for a heliostat whose list fits the tables.
template <bool GRAD>
__device__ __forceinline__ unsigned wide_point(const TraceArgs& a, int h, PrimTable<true, GRAD>& tab, float ox, float oy, float oz,
                                               float dx, float dy, float dz)
{
    const int n_list = __builtin_amdgcn_readfirstlane(tab.n_wide);
    if (__builtin_expect(n_list == 0, 1)) return 0u;
    const WideTabs w = wide_tabs(a, h, n_list);
    const float il = rsqrtf(fmaxf(dx * dx + dy * dy + dz * dz, 1e-30f));
    dx *= il; dy *= il; dz *= il;
    unsigned long long bits = 0ull;
    const int n = min(w.n, kWideFirst + 64);
    for (int c = kWideFirst; c < n; ++c) {
        const Prim q = wide_prim(w, c);
        const float wx = q.cx - ox, wy = q.cy - oy, wz = q.cz - oz;
        const float l2 = wx * wx + wy * wy + wz * wz;
        const float t = wx * dx + wy * dy + wz * dz;
        const float perp = sqrtf(fmaxf(l2 - t * t, 0.0f));
        if (wave_any(perp * w.cone_cos - t * w.cone_sin <= q.rho || l2 <= q.rho * q.rho)) bits |= 1ull << (c - kWideFirst);
    }
    if ((threadIdx.x & 63) == 0) tab.xmask[threadIdx.x >> 6] = bits;       // (read back by this wave only: no barrier)
    return bits != 0ull || w.n > kWideFirst + 64 ? 0x80000000u : 0u;
}
template <bool GRAD> __device__ __forceinline__ unsigned wide_point(const TraceArgs&, int, PrimTable<false, GRAD>&, float, float, float, float, float, float) { return 0u; }

// One ray per lane, the wave's mask has bit 31: if the heliostat is wide (otherwise bit 31 is its 32nd tabled rectangle and
// nothing happens), the bit is cleared in `wm` - the mask of TABLED rectangles for soft_transmittance and the adjoint - and
// the sigmas of the listed candidates come back (sum; near: this lane's ray entered a soft edge among them with a gradient).
struct WideSum { float sum; int near; };
template <typename ONE> __device__ __forceinline__ void wide_for_each(const WideTabs& w, unsigned long long bits, ONE&& one)
{
    for (int c = kWideFirst; c < w.n; ++c) {
        if (c < kWideFirst + 64 && ((bits >> (c - kWideFirst)) & 1ull) == 0ull) continue;       // (wave-uniform)
        one(c);
    }
}
// (out of line, plain arguments: one copy per kernel instead of one per inlined ray body - the lean backward item has nine, and
//  with the listed candidates' code inlined its kernel grew from 57 to 76 KB, past the instruction cache: +4.6 % on a field
//  without a single wide heliostat)
__device__ __attribute__((noinline)) WideSum wide_sum_listed(const WideTabs w, unsigned long long bits, float ox, float oy, float oz,
                                                             float rx, float ry, float rz)
{
    float sum = 0.0f;
    bool near = false;
    wide_for_each(w, bits, [&](int c) {
        const Prim q = wide_prim(w, c);
        SoftHit s;
        const bool in_front = soft_plane(q, ox, oy, oz, rx, ry, rz, s);
        if (!wave_any(in_front)) return;
        soft_uv(q, ox, oy, oz, rx, ry, rz, in_front, s);
        if (!wave_any(s.near)) return;
        SoftSig g;
        const float sg = soft_sigma(s, g);
        sum += s.near ? sg : 0.0f;
        near |= s.near && g.sigma_raw != 1.0f;
    });
    WideSum out = {sum, near ? 1 : 0};
    return out;
}
template <bool GRAD>
__device__ __forceinline__ WideSum wide_ray(const TraceArgs& a, int h, PrimTable<true, GRAD>& tab, unsigned& wm, float ox, float oy,
                                            float oz, float rx, float ry, float rz)
{
    WideSum out = {0.0f, 0};
    const int n_list = __builtin_amdgcn_readfirstlane(tab.n_wide);
    if (n_list == 0) return out;
    wm &= 0x7FFFFFFFu;
    unsigned long long bits = tab.xmask[threadIdx.x >> 6];
    bits = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(bits >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)bits);
    return wide_sum_listed(wide_tabs(a, h, n_list), bits, ox, oy, oz, rx, ry, rz);
}
template <bool GRAD> __device__ __forceinline__ WideSum wide_ray(const TraceArgs&, int, PrimTable<false, GRAD>&, unsigned&, float, float, float, float, float, float) { WideSum z = {0.0f, 0}; return z; }

// Work items of the windowed kernels.  A CU holds one workgroup (the window fills its LDS), so the grid is one
// PERSISTENT workgroup per CU that pulls (heliostat, point block, sample chunk) items from a counter in global memory:
//   * no dispatcher between two items (measured with tools/timeline.sh: 5 us mean, 30 % of the hand-overs > 5 us),
//   * the eight XCDs share one queue (the hardware deals every eighth workgroup to an XCD, whatever their length),
//     (Dealing the last items as two halves of their sample range, so that the kernel ends within half an item, measured
//     neutral at 1000 and 125 heliostats in round 1 and needed float atomics in the backward kernel: removed.)
struct WorkItem { int h, pblock, rchunk, r0, r1; bool tail; };     // tail: an item of the queue's finer-grained end (see TraceArgs::tail_h)
// Blocking on, but a heliostat none of whose rays can meet a rectangle (an empty candidate list - with the reference's tree
// that is almost every heliostat) is traced by the LEAN kernels: the call then makes two launches that share the field,
// a.split = 1 (lean: the unblocked heliostats) and 2 (blocking instantiation: the others).  Workgroup-uniform.
__device__ __forceinline__ bool other_launch_owns(const TraceArgs& a, int h)
{
    if (a.split == 0 || a.split == 3) return false;     // (3: the cylinder launch of a mixed tower - the receiver type decides)
    const bool blocked = a.cand_count[h] > 0;
    return a.split == 1 ? blocked : !blocked;
}
// Points [p0, p1) of point block `pblock`: blocks never straddle two facets (a.facet_points consecutive points - the
// caller's facet size, a whole multiple of it when blocks are larger than a facet, or P when no facet structure is
// known) - two facets' images are two blobs and one window serves them badly.
__device__ __forceinline__ void block_range(const TraceArgs& a, int pblock, int& p0, int& p1, bool tail = false)
{
    const int bpf = tail ? a.tail_bpf : a.blocks_per_facet, pb = tail ? a.tail_pblock : a.p_block;
    const int f = pblock / bpf, i = pblock - f * bpf;
    p0 = f * a.facet_points + i * pb;
    p1 = min(p0 + pb, min((f + 1) * a.facet_points, a.P));
}
// (a.h_group > 1: an item is a GROUP of consecutive heliostats - trace_fwd_item_field - and the queue holds a.n_groups rows)
// (a.tail_h > 0: the last a.tail_h heliostats of the queue are cut into a.tail_npb point blocks each instead of a.n_pblocks -
//  the queue ends with smaller items, so that the CUs finish closer together; n_rchunks == 1 then)
__device__ __forceinline__ int work_item_count(const TraceArgs& a)
{
    if (a.tail_h > 0) return (a.H - a.tail_h) * a.n_pblocks + a.tail_h * a.tail_npb;
    return (a.h_group > 1 ? a.n_groups : a.H) * a.n_pblocks * a.n_rchunks;
}
// Longest items first: an item's cost grows with the distance between heliostat and target (wider image, more rays
// beyond the window: 83 -> 115 us from the nearest to the farthest tenth of the metric field), and a queue that ends
// with the long items ends with idle CUs.  Fields are usually listed row by row, so the cheap test is which END of the
// list is farther from its target.  (Forward kernel only: -3 % at 1000 heliostats, -10 % at 125; the backward
// kernel measured 0.3-2 % slower in that order and keeps the list order.)
__device__ __forceinline__ bool farther_end_is_last(const TraceArgs& a)
{
    if (a.reverse_items >= 0) return a.reverse_items != 0;
    auto distance2 = [&](int h) {
        const float4 o = a.origins[(int64_t)h * a.P];
        const int t = a.target_idx[h];
        if ((unsigned)t >= (unsigned)(a.T + a.Tc)) return 0.0f;          // reported by the item that owns the heliostat
        const float* c = t < a.T ? a.centers + 4 * t : a.cyl_centers + 4 * (t - a.T);
        const float dx = o.x - c[0], dy = o.y - c[1], dz = o.z - c[2];
        return dx * dx + dy * dy + dz * dz;
    };
    return distance2(a.H - 1) > distance2(0);
}

__device__ __forceinline__ WorkItem decode_work_item(const TraceArgs& a, int item, bool reverse = false)
{
    if (a.tail_h > 0) {
        // queue order: the heliostats in list order (or, `reverse`, from the last to the first), whole samples per item; the
        // queue's last a.tail_h heliostats in finer blocks
        const int head_items = (a.H - a.tail_h) * a.n_pblocks;
        WorkItem w;
        int hq;
        if (item < head_items) { hq = item / a.n_pblocks; w.pblock = item - hq * a.n_pblocks; w.tail = false; }
        else { const int j = item - head_items; const int k = j / a.tail_npb; hq = a.H - a.tail_h + k; w.pblock = j - k * a.tail_npb; w.tail = true; }
        w.h = reverse ? a.H - 1 - hq : hq;
        w.rchunk = 0; w.r0 = 0; w.r1 = a.R;
        return w;
    }
    if (reverse) item = work_item_count(a) - 1 - item;
    const int base = item;
    WorkItem w;
    const int rchunk = base % a.n_rchunks;
    w.pblock = (base / a.n_rchunks) % a.n_pblocks;
    w.h = base / (a.n_rchunks * a.n_pblocks);
    w.rchunk = rchunk;
    w.r0 = rchunk * a.r_chunk;
    w.r1 = min(w.r0 + a.r_chunk, a.R);
    w.tail = false;
    return w;
}

// ---- the windows of a launch's items, made ahead of the launch ------------------------------------------------------------
// A CU holds ONE trace workgroup, so the window phase of an item - a load round trip, thirteen block reductions, two barriers:
// ~5 us - is 5 us of an idle CU, sixteen times per CU at the metric size (and what keeps small items from paying).
// The lean planar launches therefore get their windows from a table, filled by this kernel before they start: one small
// workgroup per item, thousands of them at once, the same routine (compute_window) on a sample of 256 of the item's points.
// The window is a guess that only decides which rays go through LDS: results do not depend on who made it.
// `order`: 0 / 1 the queue runs first-to-last / last-to-first, -1 decided here as the forward kernel decides it.
template <bool INTERLEAVED>
__global__ __launch_bounds__(256) void window_table_kernel(TraceArgs a, Window* __restrict__ table, int order)
{
    __shared__ float s_red[13][16];
    __shared__ Window s_win;
    const int item = blockIdx.x;
    const bool reverse = order < 0 ? farther_end_is_last(a) : order != 0;
    const WorkItem w = decode_work_item(a, item, reverse);
    const int t = a.target_idx[w.h];
    Window empty = {0, 0, 0, 0, 0, 1, a.scale_g * (1.0f / (float)(1 << kCellShift)), kCellShift, 0, 0};
    if ((unsigned)t >= (unsigned)a.T || w.r1 <= w.r0) {        // (skipped by the trace kernel as well; cylinders: not served)
        if (threadIdx.x == 0) table[item] = empty;
        return;
    }
    const Plane pl = load_plane(a.centers, a.pnormals, a.dims, t, a.W, a.Hh, a.mag, a.k_ext, a.k_refl);
    const Cyl cy = {};
    int p0, p1;
    block_range(a, w.pblock, p0, p1, w.tail);
    compute_window<INTERLEAVED, false>(a, pl, cy, a.incident[w.h], a.origins + (int64_t)w.h * a.P, a.normals + (int64_t)w.h * a.P, p0, p1,
                                       (int64_t)w.h * a.sh + (int64_t)w.r0 * a.sr, s_red, &s_win);
    if (threadIdx.x == 0) table[item] = s_win;
}

// The k-th heliostat (k = 0, 1, ...) with a non-empty candidate list, or -1.  Workgroup-wide (ballots + a scan of the wave
// totals in LDS); s_scan holds 18 ints.  The blocking launch of a split call has one workgroup per item: mapping the
// workgroups to the BLOCKED heliostats in order puts the (with the reference's tree: few dozen) workgroups that have work
// at the head of the grid, where they start beside the lean launch instead of behind it - the others exit here.
// (CYLINDERS = true: the k-th heliostat that aims at a cylinder instead - the cylinder launch of a tower with both receiver
//  types, a.split == 3, has the same problem beside the lean launch of the planes.)
template <bool CYLINDERS = false>
__device__ __forceinline__ int kth_blocked_heliostat(const TraceArgs& a, int k, int* s_scan)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    if (tid == 0) s_scan[17] = -1;
    __syncthreads();
    int running = 0;
    for (int base = 0; base < a.H; base += blockDim.x) {
        const int h = base + tid;
        const bool flag = h < a.H && (CYLINDERS ? a.target_idx[h] >= a.T : a.cand_count[h] > 0);
        const unsigned long long m = __builtin_amdgcn_ballot_w64(flag);
        if (lane == 0) s_scan[wave] = __popcll(m);
        __syncthreads();
        int before = running, total = 0;
        for (int w = 0; w < nwaves; ++w) {
            const int c = s_scan[w];
            if (w < wave) before += c;
            total += c;
        }
        if (flag && before + __popcll(m & ((1ull << lane) - 1ull)) == k) s_scan[17] = h;
        running += total;
        __syncthreads();
        if (running > k) break;                      // workgroup-uniform
    }
    return s_scan[17];
}

// One fetch from a launch's work counter.  A launch of n_items items makes EXACTLY n_items fetches - one per item it
// processes or skips - which return 0 .. n_items - 1, each once: whoever gets n_items - 1 has made the launch's last access to
// the counter and puts it back to zero.  The counter (one per stream and role, stream_work_counters) is therefore zero
// whenever a launch starts, with no memset per launch and no ring of slots that a long queue of launches could wrap around.
// work_counter == nullptr: a launch with one workgroup per item (nothing to fetch).
__device__ __forceinline__ unsigned fetch_work_item(unsigned int* __restrict__ work_counter, const TraceArgs& a)
{
    if (work_counter == nullptr) return 0u;
    const unsigned v = atomicAdd(work_counter, 1u);
    if (v + 1u == (unsigned)work_item_count(a)) __hip_atomic_store(work_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;
}

template <bool INTERLEAVED, bool CYL, bool BLOCKING>
__device__ __forceinline__ void trace_fwd_item(const TraceArgs& a, float* __restrict__ flux, unsigned int* __restrict__ counts,
                                               const int bid, const WorkItem item, unsigned int* __restrict__ work_counter,
                                               int* s_next)
{
    extern __shared__ __attribute__((aligned(16))) unsigned tile[];
    __shared__ float s_red[13][16];    // per-wave partials: emin, emax, umin, umax, ke, ku, angmax, dmax2, sums
    __shared__ Window s_win;
    __shared__ unsigned s_cnt[3];
    __shared__ PrimTable<BLOCKING, false> s_tab;

    const int pblock = item.pblock;
    const int h = item.h;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;

    const int t = a.target_idx[h];
    if (!target_in_range(a, t) || (t >= a.T) != CYL || item.r1 <= item.r0 || other_launch_owns(a, h)) {   // workgroup-uniform: a bad index, or another launch owns this heliostat
        if (tid == 0) *s_next = (int)(gridDim.x + fetch_work_item(work_counter, a));
        return;
    }
    Plane pl; Cyl cy;
    if constexpr (CYL) cy = load_cyl(a.cyl_centers, a.cyl_normals, a.cyl_axes, a.cyl_radii, a.cyl_heights, a.cyl_opening,
                                     t - a.T, a.W, a.Hh, a.mag, a.k_ext, a.k_refl);
    else pl = load_plane(a.centers, a.pnormals, a.dims, t, a.W, a.Hh, a.mag, a.k_ext, a.k_refl);
    const float k_ext = a.k_ext, k_refl = a.k_refl;
    unsigned long long* __restrict__ acc = a.accum + (int64_t)(a.mode == 0 ? h : t) * a.Hh * a.W;   // this bitmap's accumulators
    const float4 inc = a.incident[h];
    int p0, p1;
    block_range(a, pblock, p0, p1);
    const int r0 = item.r0;
    const int r1 = item.r1;
    const float4* __restrict__ org = a.origins + (int64_t)h * a.P;
    const float4* __restrict__ nrm = a.normals + (int64_t)h * a.P;
    const int64_t dbase = (int64_t)h * a.sh + (int64_t)r0 * a.sr;

    // ---- phase 1: window ---------------------------------------------------------------------
    if (tid < 3) s_cnt[tid] = 0;
    // This thread's first point and its first distortion sample are requested before anything else, and the tile is
    // cleared while they are in flight: a CU holds ONE workgroup (the window fills its LDS), so every microsecond of
    // latency in this prologue is a microsecond of idle VALUs (tools/timeline.sh).
    const int pf = (a.win_sample != 0 && p1 - p0 > (int)blockDim.x) ? window_sample_point(p0, p1) : p0 + tid;
    FirstPoint fp = {{0.0f, 0.0f, 0.0f, 1.0f}, {0.0f, 0.0f, 1.0f, 0.0f}, 0.0f, 0.0f};
    if (pf < p1) {
        fp.o = org[pf]; fp.n = nrm[pf];
        load_dist_row<INTERLEAVED>(a.dist_u + dbase, a.dist_e + dbase, pf * (int)a.sp, fp.u, fp.e);
    }
    {   // 16-byte stores (tile_cap is a multiple of 256 cells) + the two spare cells behind the window
        uint4* t4 = reinterpret_cast<uint4*>(tile);
        for (int i = tid; i < a.tile_cap / 4; i += blockDim.x) t4[i] = make_uint4(0u, 0u, 0u, 0u);
        if (tid < 2) tile[a.tile_cap + tid] = 0u;
    }
    const int n_prims = load_prims<BLOCKING, false>(a, h, s_tab);
    compute_window<INTERLEAVED, CYL>(a, pl, cy, inc, org, nrm, p0, p1, dbase, s_red, &s_win, &fp);
    const Window win = s_win;
    // (an empty window - no chief ray reaches the target - must hold no ray: bounds of 0, not of (unsigned)-1)
    const unsigned twm1 = (unsigned)max(win.tw - 1, 0), uthm1 = (unsigned)max(win.th - 1, 0);
    const int dummy = a.tile_cap;                    // two spare cells: [tile_cap], [tile_cap + 1]
    const float Wf = (float)a.W, Hf = (float)a.Hh;
    // lean mode needs I0 > 0 and I > 0 to be implied by `valid`: positive, sanely scaled intensity factors
    // (a cylinder's Lambert term is clamped at 0, so a valid ray can carry no intensity: never lean)
    // (blocking: an intensity can be attenuated to 0 as well)
    const bool lean = !CYL && !BLOCKING && a.mag >= 1e-6f && k_ext >= 1e-6f && k_refl >= 1e-6f && a.mag <= 1e6f && k_ext <= 1e6f &&
                      k_refl <= 1e6f;
    const unsigned long long lean_all = lean ? ~0ull : 0ull;
  for (int pass = 0; pass < win.npass; ++pass) {
    const int pu0 = win.u0 + pass * (win.ths - 1);                       // first flat row of this pass
    const int pth = min(win.ths, win.u0 + win.th - pu0);                 // rows held in LDS in this pass
    const unsigned thm1 = (unsigned)max(pth - 1, 0);
    const bool first = pass == 0;
    unsigned n_valid = 0, n_int = 0;     // rays with I0 > 0 / I > 0 among the valid ones (wave totals)
    unsigned n_free = 0;                 // rays with blocked < 1e-3 (heliostat_ray_tracer.py:501-503)
    if (!first) {                          // pass 0 starts on the tile cleared in the prologue
        const int npx = win.tw * pth;
        for (int i = tid; i < npx; i += blockDim.x) tile[i] = 0u;
        __syncthreads();
    }

    // ---- phase 2: trace ----------------------------------------------------------------------
    // Every ray issues its four LDS adds unconditionally: rays that are off the bitmap or outside the
    // window add 0 to a dummy cell behind the window (no divergence in the common path); the rare
    // in-bitmap-but-outside-window ray goes to global memory.
    PendingSplat ps = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int p = p0 + tid; p < p1; p += blockDim.x) {
        const float4 o = org[p];
        const float4 n = nrm[p];
        float4 d; float s;
        reflect(inc, n, d, s);
        float numer = 0.0f; CylPoint cp;
        if constexpr (CYL) cp = cyl_point(cy, o); else numer = plane_numer(pl, o);
        unsigned pmask = 0u, wmask = 0u;       // rectangles this point's / this wave's rays can touch
        if constexpr (BLOCKING) {
            if (n_prims > 0) {
                const float il = rsqrtf(fmaxf(d.x * d.x + d.y * d.y + d.z * d.z, 1e-30f));
                pmask = cone_mask(s_tab.prim, s_tab.aux, n_prims, o.x, o.y, o.z, d.x * il, d.y * il, d.z * il, a.cone_cos, a.cone_sin,
                                  a.slab_cull != 0);
                wmask = wave_or_mask(pmask, n_prims);
                wmask |= wide_point(a, h, s_tab, o.x, o.y, o.z, d.x, d.y, d.z);     // (bit 31: see there)
            }
        }
        // One ray: scatter -> hit -> weights -> 4 pipelined LDS adds.  Ray arithmetic is the reference's
        // (ray_math.hpp); the masks are reduced to the one question the LDS path asks ("does this ray land inside
        // this pass's window?"), and everything rare - a scatter angle beyond the small-angle kernel, a valid ray
        // outside the window, a cell that may have wrapped - sits behind wave-uniform branches.
        auto trace_one = [&](auto small_angles, const float u, const float e) {
            const Rot m = make_rot_t<decltype(small_angles)::value>(e, u);
            float rx, ry, rz;
            scatter(m, d, rx, ry, rz);
            float be, bu, I0; bool valid;
            // lane masks as wave-uniform 64-bit scalars (ballots of the raw compares: no round trip through a VGPR):
            // the ray counters and the rare-branch tests then run on the scalar unit
            unsigned long long m_valid;
            if constexpr (CYL) {
                const CylHit ch = cyl_hit(cy, cp, rx, ry, rz);                     // geometry.py:287-445
                valid = ch.ok; be = ch.be; bu = ch.bu; I0 = ch.I0;
                m_valid = __builtin_amdgcn_ballot_w64(valid);
            } else {
                const float ah = (rx * pl.mx + ry * pl.my) + rz * pl.mz;           // geometry.py:116-118
                const bool front = ah < 0.0f;
                const float tt = div_noscale(numer, front ? ah : 1.0f);            // :130-131
                const float hx = o.x + rx * tt, hz = o.z + rz * tt;                // :133-136
                const float be0 = div_const((hx + pl.half_w) - pl.cx, pl.w, pl.inv_w) * pl.wm1;   // :148-169
                bu = div_const((hz + pl.half_h) - pl.cz, pl.h, pl.inv_h) * pl.hm1;                // :154-174
                // :178-184.  0 <= x <= hi  <=>  x == med3(x, 0, hi)   (NaN compares false)
                const bool e_ok = be0 == __builtin_amdgcn_fmed3f(be0, 0.0f, pl.wm1);
                const bool u_ok = bu == __builtin_amdgcn_fmed3f(bu, 0.0f, pl.hm1);
                valid = front && e_ok && u_ok;
                m_valid = __builtin_amdgcn_ballot_w64(front) & __builtin_amdgcn_ballot_w64(e_ok) &
                          __builtin_amdgcn_ballot_w64(u_ok);
                be = pl.wm1 - be0;                                                 // :195-197
                I0 = pl.mag * (-ah);                                               // :139
            }

            const float tbe = truncf(be), tbu = truncf(bu);                    // heliostat_ray_tracer.py:674-675
            const float cle = (tbe + 1.0f) - be, clu = (tbu + 1.0f) - bu, che = be - tbe, chu = bu - tbu;   // :694-700
            const int ie = (int)tbe, iu = (int)tbu;
            const int le = ie - win.e0, lu = iu - pu0;
            const bool inwin = valid && (unsigned)le < twm1 && (unsigned)lu < thm1;     // implies ie+1 < W, iu+1 < Hh
            const unsigned long long m_inwin = m_valid & __builtin_amdgcn_ballot_w64((unsigned)le < twm1) &
                                               __builtin_amdgcn_ballot_w64((unsigned)lu < thm1);
            const int cell_lo = inwin ? (int)__umul24(lu, win.tw) + le : dummy;        // flat row iu
            const int cell_hi = inwin ? cell_lo + win.tw : dummy;                       // flat row iu + 1
            float keep = 1.0f;                                                          // 1 - blocked
            if constexpr (BLOCKING) {
                // soft mask over this heliostat's rectangles, for every ray - also those that miss the target
                // (blocking.py:212-354; heliostat_ray_tracer.py:462-480)
                float blocked = 0.0f;
                if (wmask != 0u) {
                    unsigned near, wm = wmask;
                    float tail = 0.0f;           // the sigmas of the candidates beyond the tables (a wide heliostat)
                    if (__builtin_expect((wmask >> 31) != 0u, 0)) tail = wide_ray(a, h, s_tab, wm, o.x, o.y, o.z, rx, ry, rz).sum;
                    blocked = 1.0f - soft_transmittance(s_tab.prim, wm, pmask, o.x, o.y, o.z, rx, ry, rz, near, 0.0f, 0.0f, tail);   // :364-365
                    keep = 1.0f - blocked;       // exactly 0 once the transmittance drops below 2^-25, as in the reference
                }
                n_free += __popcll(__builtin_amdgcn_ballot_w64(blocked < 1e-3f));
            }
            const float I = ((I0 * keep) * k_ext) * k_refl;                             // heliostat_ray_tracer.py:482-487
            // S is a power of two, so (w I) S == w (I S) bit for bit: scale the intensity once
            const float Is = inwin ? fabsf(I) * win.scale : 0.0f;
            // ray counters live in SGPRs (v_cmp + s_bcnt1).  With positive, sanely scaled intensity factors
            // (`lean`) I0 > 0 and I > 0 follow from `valid` and one count serves both.
            // (`lean_all` = all ones when a valid ray is known to carry intensity: no data-dependent control flow here)
            n_valid += __popcll(m_valid & (__builtin_amdgcn_ballot_w64(I0 > 0.0f) | lean_all));
            n_int += __popcll(m_valid & __builtin_amdgcn_ballot_w64(I > 0.0f) & ~lean_all);
            // the previous ray's adds have landed; a carry needs a cell that was already above 2^31 (q < 2^22)
            if (__builtin_expect(wave_any(((ps.o1 | ps.o2 | ps.o3 | ps.o4) >> 31) != 0u), 0))
                resolve_carries(ps, acc, a.W, a.Hh, win.shift);
            ps.q1 = cvt_nearest_u32(cle * chu * Is); ps.q2 = cvt_nearest_u32(che * chu * Is);
            ps.q3 = cvt_nearest_u32(che * clu * Is); ps.q4 = cvt_nearest_u32(cle * clu * Is);
            ps.ie = ie; ps.iu = iu;
            ps.o1 = atomicAdd(tile + cell_hi, ps.q1);
            ps.o2 = atomicAdd(tile + cell_hi + 1, ps.q2);
            ps.o3 = atomicAdd(tile + cell_lo + 1, ps.q3);
            ps.o4 = atomicAdd(tile + cell_lo, ps.q4);
            if (__builtin_expect((m_valid & ~m_inwin) != 0ull, 0)) {
                // valid but not in this pass's band: another band's ray, the last pixel row/column
                // (heliostat_ray_tracer.py:723-728), or a stray of the union window -> global atomics, once
                const bool on = (tbe + 1.0f < Wf) && (tbu + 1.0f < Hf);
                const bool in_union = (unsigned)le < twm1 && (unsigned)(iu - win.u0) < uthm1;
                if (first && valid && on && !in_union) {
                    unsigned long long* row_hi = acc + (int64_t)(a.Hh - 2 - iu) * a.W + ie;
                    unsigned long long* row_lo = row_hi + a.W;
                    const float Iss = fabsf(I) * win.scale;              // as for a window ray
                    atomicAdd(row_hi, cell_to_accum(cle * chu * Iss)); atomicAdd(row_hi + 1, cell_to_accum(che * chu * Iss));
                    atomicAdd(row_lo + 1, cell_to_accum(che * clu * Iss)); atomicAdd(row_lo, cell_to_accum(cle * clu * Iss));
                }
            }
        };
        // Distortion stream, software-prefetched in groups of four samples: the loads of group g+1 are issued
        // before group g is traced, so their HBM latency hides behind ~600 instructions.  (The loads are
        // consumed in the iteration after the one that issues them; hipcc drains vmcnt at the loop back-edge,
        // by which time they have landed.)
        const int lane_off = p * (int)a.sp;
        const int nr = r1 - r0;
        const float* __restrict__ bu_ = a.dist_u + dbase;     // wave-uniform
        const float* __restrict__ be_ = a.dist_e + dbase;
        float cu0, ce0, cu1, ce1, cu2, ce2, cu3, ce3;
        load_dist_row<INTERLEAVED>(bu_, be_, lane_off, cu0, ce0);
        load_dist_row<INTERLEAVED>(bu_ + (int64_t)min(1, nr - 1) * a.sr, be_ + (int64_t)min(1, nr - 1) * a.sr, lane_off, cu1, ce1);
        load_dist_row<INTERLEAVED>(bu_ + (int64_t)min(2, nr - 1) * a.sr, be_ + (int64_t)min(2, nr - 1) * a.sr, lane_off, cu2, ce2);
        load_dist_row<INTERLEAVED>(bu_ + (int64_t)min(3, nr - 1) * a.sr, be_ + (int64_t)min(3, nr - 1) * a.sr, lane_off, cu3, ce3);
        for (int k = 0; k < nr; k += 4) {
            float nu0, ne0, nu1, ne1, nu2, ne2, nu3, ne3;
            const int64_t o4 = (int64_t)min(k + 4, nr - 1) * a.sr, o5 = (int64_t)min(k + 5, nr - 1) * a.sr;
            const int64_t o6 = (int64_t)min(k + 6, nr - 1) * a.sr, o7 = (int64_t)min(k + 7, nr - 1) * a.sr;
            load_dist_row<INTERLEAVED>(bu_ + o4, be_ + o4, lane_off, nu0, ne0);
            load_dist_row<INTERLEAVED>(bu_ + o5, be_ + o5, lane_off, nu1, ne1);
            load_dist_row<INTERLEAVED>(bu_ + o6, be_ + o6, lane_off, nu2, ne2);
            load_dist_row<INTERLEAVED>(bu_ + o7, be_ + o7, lane_off, nu3, ne3);
            // one range test for the group's eight angles: sun-shape angles are milliradians, so the Taylor kernels
            // serve every lane almost always and the full-range sin/cos code stays out of the hot loop
            const float amax = fmaxf(fmaxf(fmaxf(fabsf(cu0), fabsf(ce0)), fmaxf(fabsf(cu1), fabsf(ce1))),
                                     fmaxf(fmaxf(fabsf(cu2), fabsf(ce2)), fmaxf(fabsf(cu3), fabsf(ce3))));
            if (__builtin_expect(wave_any(!(amax <= kSmallAngle)), 0)) {
                trace_one(std::false_type{}, cu0, ce0);
                if (k + 1 < nr) trace_one(std::false_type{}, cu1, ce1);
                if (k + 2 < nr) trace_one(std::false_type{}, cu2, ce2);
                if (k + 3 < nr) trace_one(std::false_type{}, cu3, ce3);
            } else {
                trace_one(std::true_type{}, cu0, ce0);
                if (k + 1 < nr) trace_one(std::true_type{}, cu1, ce1);
                if (k + 2 < nr) trace_one(std::true_type{}, cu2, ce2);
                if (k + 3 < nr) trace_one(std::true_type{}, cu3, ce3);
            }
            cu0 = nu0; ce0 = ne0; cu1 = nu1; ce1 = ne1; cu2 = nu2; ce2 = ne2; cu3 = nu3; ce3 = ne3;
        }
    }
    resolve_carries(ps, acc, a.W, a.Hh, win.shift);
    if (first && lane == 0) {                                                           // wave totals
        atomicAdd(&s_cnt[0], lean ? n_valid : n_int); atomicAdd(&s_cnt[1], n_valid);
        if constexpr (BLOCKING) atomicAdd(&s_cnt[2], n_free);
    }
    __syncthreads();
    // the next work item is requested now and published after the flush: the counter's round trip hides behind it
    unsigned next_item = 0u;
    if (tid == 0 && pass == win.npass - 1) next_item = fetch_work_item(work_counter, a);

    // ---- phase 3: flush (one wave per window row; lanes along e -> contiguous global atomics) ----
    for (int row = wave; row < pth; row += nwaves) {
        unsigned long long* g = acc + (int64_t)(a.Hh - 1 - (pu0 + row)) * a.W + win.e0;
        const unsigned* trow = tile + row * win.tw;
        for (int c = lane; c < win.tw; c += 64) {
            const unsigned q = trow[c];
            if (q != 0u) atomicAdd(g + c, (unsigned long long)q << win.shift);
        }
    }
    if (tid == 0 && pass == win.npass - 1) *s_next = (int)(gridDim.x + next_item);
    __syncthreads();   // the band is flushed before the next pass re-zeroes the tile
  }
    if (tid < 3 && s_cnt[tid]) atomicAdd(&counts[tid * a.H + h], s_cnt[tid]);
}


// --------------------------------------------------------------------------------------------
// The planar, non-blocking forward item with the LEAN ray body (the production path of the metric workload).
//
// What tools/issue_bench.hip measured on gfx950 (true shader cycles, profiles/r02_issue_bench.json) shapes this body:
//   * v_fma / v_mul / v_add / v_sub f32, v_add_u32, v_mov on VGPRs issue every ~2 cycles per SIMD; everything else
//     (v_cvt, v_trunc, v_med3, v_cmp, v_max, v_mad_u32_u24, v_lshl_add, DPP, ops with an SGPR operand, v_pk_*_f32)
//     needs ~3.4 back to back but hides behind fast instructions of the other waves when interleaved with them;
//   * v_cndmask_b32 in its VOP2 form (implicit VCC) costs ~13 CYCLES EACH when the scalar unit wrote VCC last
//     (s_and_b64 vcc, ... ; v_cndmask x3 = 41 cycles) - the round-1 body did exactly that three times per ray (10 % of
//     the kernel).  The VOP3 form with an explicit SGPR pair costs a normal slot.
// So: no select on the division's denominator (a back-facing ray's garbage is masked by `front` anyway), the window
// test and the cell address are formed in floating point from the truncated pixel coordinates (exact: small
// integers) and clamped with one v_med3 instead of selected, masks are compared on the float BIT patterns (one
// unsigned compare = a two-sided range test, NaN and negative values fail), the single remaining select (the ray's
// scaled intensity, zero outside the window) is a VOP3 v_cndmask on the ballot mask, the intensity scalars are folded
// into one per-workgroup factor, and cle = 1 - che replaces (trunc + 1) - be (identical: both are exact).
// Per ray: ~95 vector instructions instead of 135.  Ray POSITIONS (scatter, hit, pixel coordinates) keep the
// reference's operation order bit for bit; the ray's intensity differs from the reference's three roundings by
// at most 2 ULP, far below the fixed-point quantum of the window.
// Requires positive, sanely scaled intensity factors (`lean` of the generic item), decided on the host.
// --------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) unsigned lds_u32;

__device__ __forceinline__ float select_or_zero(unsigned long long mask, float x)
{
    float r;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(x), "s"(mask));
    return r;
}
// mask ? a : b with the lane mask in an SGPR pair
__device__ __forceinline__ float select_mask(unsigned long long mask, float a, float b)
{
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(mask));
    return r;
}
__device__ __forceinline__ unsigned f32_bits(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ unsigned long long ballot64(bool c) { return __builtin_amdgcn_ballot_w64(c); }

constexpr int kLeanFwdPoints = 2560;   // points per item of the lean forward kernel when the facet structure is known
constexpr int kPackTrips = 4;          // a block whose edge points are packed holds at most this many trips of points
constexpr int kPackPoints = 2560;      // ... and this many points (room for the permutation)

// Stable partition of a block's points - interior points in mirror order, then (or, see below, just before the partial
// last trip) the EDGE points: those whose chief ray lands within a.pack_edge / 64 scatter pads of the window's border, or
// outside it.  perm[j] = index within the block of the point that slot j serves (thread j % blockDim, trip j / blockDim).
// s_edge: kPackTrips * 16 + 1 ints of LDS.  Ends with a barrier.
__device__ __forceinline__ void pack_edge_points(const TraceArgs& a, const Plane& pl, const float4 inc, const float4* __restrict__ org,
                                                 const float4* __restrict__ nrm, const int p0, const int n_pts, const Window& win,
                                                 const int pack_edge, int* s_edge, unsigned short* perm)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const float Wf = (float)a.W, Hf = (float)a.Hh;
    {
        const int me = (win.pad_e * pack_edge) >> 6, mu = (win.pad_u * pack_edge) >> 6;      // margin = pad * pack_edge / 64
        unsigned long long flags[kPackTrips];
#pragma unroll
        for (int k = 0; k < kPackTrips; ++k) {
            const int i = k * (int)blockDim.x + tid;
            bool edge = false;
            if (i < n_pts) {
                const float4 o = org[p0 + i];
                float4 d; float s_;
                reflect(inc, nrm[p0 + i], d, s_);
                const RaySplat c = hit_and_weights(pl, o, plane_numer(pl, o), d.x, d.y, d.z, Wf, Hf);
                const int le = c.ie - win.e0, lu = c.iu - win.u0;
                edge = !c.valid || le < me || le > win.tw - 2 - me || lu < mu || lu > win.th - 2 - mu;
            }
            flags[k] = ballot64(edge);
            if (lane == 0) s_edge[k * 16 + wave] = __popcll(flags[k]);
        }
        __syncthreads();
        if (tid == 0) {
            int run = 0;
            for (int k = 0; k < kPackTrips; ++k)
                for (int w = 0; w < nwaves; ++w) { const int c = s_edge[k * 16 + w]; s_edge[k * 16 + w] = run; run += c; }
            s_edge[kPackTrips * 16] = run;
        }
        __syncthreads();
        const int n_edge = s_edge[kPackTrips * 16];
        // Where the edge points go.  A small partial trip at the end of the block is the best place: its few waves have the
        // CU almost to themselves and their stalls cost little (2500 points on 768 threads: 3.71 ms with the edge points there,
        // 3.84 ms with them at the end of the last full trip).  A large partial trip is a bad one: its waves walk one trip more
        // than the others and should not be the slow ones as well (1250 points: 0.548 ms at the end, 0.527 ms at the end of
        // the full trip, 125 heliostats).
        const int full = (n_pts / (int)blockDim.x) * (int)blockDim.x;
        const bool at_trip_end = full > 0 && n_edge <= full && 2 * (n_pts - full) >= (int)blockDim.x;
        const int e_first = at_trip_end ? full - n_edge : n_pts - n_edge;
#pragma unroll
        for (int k = 0; k < kPackTrips; ++k) {
            const int i = k * (int)blockDim.x + tid;
            if (i < n_pts) {
                const int before = s_edge[k * 16 + wave] + __popcll(flags[k] & ((1ull << lane) - 1ull));   // edge points ahead of i
                const bool edge = (flags[k] >> lane) & 1ull;
                const int q = i - before;                                                                  // interior points ahead of i
                perm[edge ? e_first + before : (q < e_first ? q : q + n_edge)] = (unsigned short)i;
            }
        }
        __syncthreads();
    }
}

// ring depth = 8: forward: 2 / 4 / 6 / 8 slots (4 costs the forward 4 %)
// ring depth bwd = 2: backward: same-box sweep 8 / 6 / 4 / 2 slots = 3.76 / 3.71 / 3.67 / 3.65 ms (125 heliostats: 0.517 -> 0.489):
// with twelve waves of 0.7 us ray steps the stream's latency is covered anyway, and a stray's
// vmcnt(0) has less to wait for
// BLOCKING: the same item with the soft blocking mask (blocking.py:212-354) of the heliostat's candidate rectangles in every
// ray - the launch of a split call that owns the heliostats WITH candidates.  What changes against the plain lean item: the
// rectangle tables in LDS, a bitmask per point of the rectangles its scatter cone can touch, `keep` = 1 - blocked in the ray's
// intensity, and three ray counters instead of one (a valid ray can now carry no light: I > 0 needs keep > 0; the blocking
// factor counts ALL rays with blocked < 1e-3, heliostat_ray_tracer.py:501-503).
template <bool INTERLEAVED, bool BLOCKING = false, bool CYL = false>
__device__ __forceinline__ void trace_fwd_item_lean(const TraceArgs& a, float* __restrict__ flux, unsigned int* __restrict__ counts,
                                                    const int bid, const WorkItem item, unsigned int* __restrict__ work_counter,
                                                    int* s_next)
{
    extern __shared__ __attribute__((aligned(16))) unsigned tile[];
    __shared__ float s_red[13][16];
    __shared__ Window s_win;
    __shared__ unsigned s_cnt[3];
    __shared__ PrimTable<BLOCKING, false> s_tab;
    const int pblock = item.pblock;
    const int h = item.h;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;

    const int t = a.target_idx[h];
    static_assert(!(BLOCKING && CYL), "the lean body serves blocking on planes or cylinders without blocking");
    if (!target_in_range(a, t) || (t >= a.T) != CYL || item.r1 <= item.r0 || other_launch_owns(a, h)) {   // a bad index, another launch's heliostat (other receiver type, blocked) or an empty item
        if (tid == 0) *s_next = (int)(gridDim.x + fetch_work_item(work_counter, a));
        return;
    }
    Plane pl = {}; Cyl cy = {};
    if constexpr (CYL) cy = load_cyl(a.cyl_centers, a.cyl_normals, a.cyl_axes, a.cyl_radii, a.cyl_heights, a.cyl_opening, t - a.T, a.W, a.Hh,
                                     a.mag, a.k_ext, a.k_refl);
    else pl = load_plane(a.centers, a.pnormals, a.dims, t, a.W, a.Hh, a.mag, a.k_ext, a.k_refl);
    unsigned long long* __restrict__ acc = a.accum + (int64_t)(a.mode == 0 ? h : t) * a.Hh * a.W;   // this bitmap's accumulators
    const float4 inc = a.incident[h];
    int p0, p1;
    block_range(a, pblock, p0, p1, item.tail);
    const int r0 = item.r0;
    const int r1 = item.r1;
    const float4* __restrict__ org = a.origins + (int64_t)h * a.P;
    const float4* __restrict__ nrm = a.normals + (int64_t)h * a.P;
    const int64_t dbase = (int64_t)h * a.sh + (int64_t)r0 * a.sr;

    // ---- phase 1: window (as in the generic item) - or, when the launch came with its windows (window_table_kernel), only the
    //      clearing of the tile -----------------------------------------------------------------------------------------------
    if (tid < 3) s_cnt[tid] = 0;
    const Window* const win_table = static_cast<const Window*>(a.win_table);
    const int pf = (a.win_sample != 0 && p1 - p0 > (int)blockDim.x) ? window_sample_point(p0, p1) : p0 + tid;
    FirstPoint fp = {{0.0f, 0.0f, 0.0f, 1.0f}, {0.0f, 0.0f, 1.0f, 0.0f}, 0.0f, 0.0f};
    if (win_table == nullptr && pf < p1) {
        fp.o = org[pf]; fp.n = nrm[pf];
        load_dist_row<INTERLEAVED>(a.dist_u + dbase, a.dist_e + dbase, pf * (int)a.sp, fp.u, fp.e);
    }
    if (win_table != nullptr && tid == 0) s_win = win_table[bid];
    {
        uint4* t4 = reinterpret_cast<uint4*>(tile);
        for (int i = tid; i < a.tile_cap / 4; i += blockDim.x) t4[i] = make_uint4(0u, 0u, 0u, 0u);
        if (tid < 2) tile[a.tile_cap + tid] = 0u;
    }
    const int n_prims = load_prims<BLOCKING, false>(a, h, s_tab);
    if (win_table != nullptr) __syncthreads();
    else compute_window<INTERLEAVED, CYL>(a, pl, cy, inc, org, nrm, p0, p1, dbase, s_red, &s_win, &fp);
    const Window win = s_win;
    const float Wf = (float)a.W, Hf = (float)a.Hh;
    // (Packing the edge points into the block's last waves, which takes 8 % off the backward kernel, was measured here too:
    //  3.25 -> 3.32 ... 3.43 ms for margins of 1/4 ... 3/4 of the scatter pad - a stray costs this kernel its atomics, which
    //  packing does not remove, and the partition is paid by every item.)
    [[maybe_unused]] const unsigned wm1_bits = f32_bits(pl.wm1), hm1_bits = f32_bits(pl.hm1);
    // |I| S = |r.m| * kS for a front-facing ray:  I = ((mag (-a)) k_ext) k_refl  (heliostat_ray_tracer.py:482-487, geometry.py:139)
    // (cylinders: the parked / splatted number is the ray's intensity I itself, in the generic item's product order, and kS = S)
    const float kI = (pl.mag * pl.k_ext) * pl.k_refl;
    const float kS = CYL ? win.scale : kI * win.scale;
    const float lds_base = (float)(unsigned)(size_t)(lds_u32*)tile;         // byte address of cell 0 (a link-time constant)
    const float e0f = (float)win.e0, tw4f = (float)(4 * win.tw);
    const unsigned twm2_bits = f32_bits((float)(win.tw - 2));               // low pixel column le in [0, tw - 2]
    const unsigned tw4 = 4u * (unsigned)win.tw;
    const float u0f = (float)win.u0;
    const unsigned uthm2_bits = f32_bits((float)(win.th - 2));           // low pixel row of the UNION window in [0, th - 2]
  for (int pass = 0; pass < win.npass; ++pass) {
    const int pu0 = win.u0 + pass * (win.ths - 1);
    const int pth = min(win.ths, win.u0 + win.th - pu0);
    const bool first = pass == 0;
    const float pu0f = (float)pu0;
    const unsigned thm2_bits = f32_bits((float)(pth - 2));
    // an empty window (no chief ray of the block reaches the target) holds no ray: every valid ray is then a stray
    const unsigned long long win_ok = (win.tw >= 2 && pth >= 2) ? ~0ull : 0ull;
    // largest byte address a ray of this pass can produce for its LOW row; rays outside the window are clamped into
    // [cell 0, that] and add zero there (an empty window: onto cell 0 - the tile is cleared, so that is a legal place too)
    [[maybe_unused]] const float addr_hi_f = win_ok != 0ull ? lds_base + (float)(4 * (win.tw * (pth - 2) + win.tw - 2)) : lds_base;
    // A ray outside the window (a stray, or one that misses the target) still issues its four adds - of zero.  Clamped onto the
    // window's first / last cell (round 2) every such lane of a wave hit the SAME four cells, and atomics on one address are
    // served lane after lane; each lane now adds its zeros to cells of its own (2 lane, 2 lane + 1 and the two above them:
    // inside every window's allocation, which is never smaller than 1024 cells).  Worth ~1 % where a tenth of the rays miss
    // (the far heliostats of the metric field; same-box A/B 0.546 -> 0.541 ms at 125 heliostats, 3.315 -> 3.294 at 1000).
    [[maybe_unused]] const float own_cell_f = lds_base + (float)(8 * lane);
    const unsigned tw4p = win_ok != 0ull ? tw4 : 0u;           // (a degenerate window may be one long row: its masked rays stay in cells 0 .. 127)
    unsigned n_valid = 0;
    [[maybe_unused]] unsigned n_int = 0, n_free = 0;      // (blocking, cylinders) rays with I > 0 / (blocking) with blocked < 1e-3
    [[maybe_unused]] unsigned n_pos = 0;                  // (cylinders) rays with I_angle > 0
    if (!first) {
        const int npx = win.tw * pth;
        for (int i = tid; i < npx; i += blockDim.x) tile[i] = 0u;
        __syncthreads();
    }

    // ---- phase 2: trace ------------------------------------------------------------------------
    unsigned long long m_parked = 0ull;               // lanes holding a parked stray ray
    float pk_be = 0.0f, pk_bu = 0.0f, pk_ah = 0.0f;   // its bitmap coordinates (geometry.py:186-197) and direction cosine (x keep)
    auto unpark = [&]() {                             // the parked rays' four weights -> their pixels' accumulators
        if ((m_parked >> lane) & 1ull) {
            const float tbe = truncf(pk_be), tbu = truncf(pk_bu);                              // heliostat_ray_tracer.py:674-675
            if ((tbe + 1.0f < Wf) && (tbu + 1.0f < Hf)) {                                      // :723-728
                const float che = pk_be - tbe, chu = pk_bu - tbu, cle = 1.0f - che, clu = 1.0f - chu;      // :694-700, as in trace_one
                const float Is = fabsf(pk_ah) * kS;                                            // the products of a window ray:
                const float wa = chu * Is, wb = clu * Is;                                      // the same integers reach the bitmap
                unsigned long long* row_hi = acc + (int64_t)(a.Hh - 2 - (int)tbu) * a.W + (int)tbe;   // flat row iu + 1, flipped
                unsigned long long* row_lo = row_hi + a.W;
                atomicAdd(row_hi, cell_to_accum(cle * wa)); atomicAdd(row_hi + 1, cell_to_accum(che * wa));
                atomicAdd(row_lo + 1, cell_to_accum(che * wb)); atomicAdd(row_lo, cell_to_accum(cle * wb));
            }
        }
        m_parked = 0ull;
    };
    // (Round 3 measured an alternative to un-parking whenever a lane that holds a parked ray strays again - every ~3 strays,
    //  the strays come from the same few lanes: the parked ray MOVED to a lane of the wave that held none (v_readlane + a
    //  select), so that a wave un-parked once per ~64 strays.  590 -> 30 un-park events per item of a far heliostat, and the
    //  item took 147 instead of 135 us: the un-park events were never the cost - tools/timeline.sh)
    unsigned po1 = 0u, po2 = 0u, po3 = 0u, po4 = 0u, pq1 = 0u, pq2 = 0u, pq3 = 0u, pq4 = 0u;   // the previous ray's adds
    float ptbe = 0.0f, ptbu = 0.0f;
    for (int p = p0 + tid; p < p1; p += blockDim.x) {
        const float4 o = org[p];
        const float4 n = nrm[p];
        float4 d; float s;
        reflect(inc, n, d, s);
        [[maybe_unused]] float numer = 0.0f;
        [[maybe_unused]] CylPoint cp = {};
        if constexpr (CYL) cp = cyl_point(cy, o); else numer = plane_numer(pl, o);
        [[maybe_unused]] unsigned pmask = 0u, wmask = 0u;       // rectangles this point's / this wave's rays can touch
        if constexpr (BLOCKING) {
            if (n_prims > 0) {
                const float il = rsqrtf(fmaxf(d.x * d.x + d.y * d.y + d.z * d.z, 1e-30f));
                pmask = cone_mask(s_tab.prim, s_tab.aux, n_prims, o.x, o.y, o.z, d.x * il, d.y * il, d.z * il, a.cone_cos, a.cone_sin,
                                  a.slab_cull != 0);
                wmask = wave_or_mask(pmask, n_prims);
            }
        }
        [[maybe_unused]] float bnum0 = 0.0f, bnum1 = 0.0f;      // (c0 - o).n of the wave mask's first two rectangles: per point, not per ray
        if constexpr (BLOCKING) {
            if (wmask != 0u) {
                bnum0 = soft_plane_num(s_tab.prim[__builtin_ctz(wmask)], o.x, o.y, o.z);
                const unsigned rest = wmask & (wmask - 1u);
                if (rest != 0u) bnum1 = soft_plane_num(s_tab.prim[__builtin_ctz(rest)], o.x, o.y, o.z);
            }
        }
        if constexpr (BLOCKING) {          // (after the two numerators: they belong to tabled rectangles)
            if (n_prims > 0) wmask |= wide_point(a, h, s_tab, o.x, o.y, o.z, d.x, d.y, d.z);     // (bit 31: see there)
        }
        auto carries = [&]() {                       // cold: a cell of the previous ray wrapped (see resolve_carries)
            PendingSplat ps = {po1, po2, po3, po4, pq1, pq2, pq3, pq4, (int)ptbe, (int)ptbu};
            resolve_carries(ps, acc, a.W, a.Hh, win.shift);
        };
        // `live`: all ones, or zero for the padding rays of the last ring round (they then fail every mask)
        auto trace_one = [&](const float u, const float e, const unsigned long long live) {
            // sun-shape angles are milliradians: the Taylor kernels serve every lane almost always; only the rotation's
            // sines and cosines sit behind the (wave-uniform) branch - two copies of the whole ray body made the compiler
            // merge their tails and spill what crossed the join
            // (the Taylor values are computed first and REPLACED on the rare path: as two arms of a branch the common arm ended
            //  in register copies; |u| + |e| bounds the larger angle with one instruction instead of three)
            Rot m = make_rot_t<true>(e, u);
            if (__builtin_expect(wave_any(!(fabsf(u) + fabsf(e) <= kSmallAngle)), 0)) m = make_rot(e, u);
            float rx, ry, rz;
            scatter(m, d, rx, ry, rz);
            float ah, be, bu;
            unsigned long long m_front, m_valid;
            if constexpr (CYL) {
                const CylHit ch = cyl_hit(cy, cp, rx, ry, rz);                 // geometry.py:287-445
                be = ch.be; bu = ch.bu;
                ah = (ch.I0 * a.k_ext) * a.k_refl;                             // the ray's intensity (heliostat_ray_tracer.py:482-487)
                m_valid = ballot64(ch.ok) & live;
                m_front = m_valid;
                n_pos += __popcll(m_valid & ballot64(ch.I0 > 0.0f));           // :498-506: rays with I_angle > 0 ...
                n_int += __popcll(m_valid & ballot64(ah > 0.0f));              // ... and with I > 0
            } else {
            ah = (rx * pl.mx + ry * pl.my) + rz * pl.mz;                       // geometry.py:116-118
            m_front = ballot64(ah < 0.0f) & live;
            const float tt = div_noscale(numer, ah);                           // :130-131 (back-facing: masked below)
            const float hx = o.x + rx * tt, hz = o.z + rz * tt;                // :133-136
            const float be0 = div_const((hx + pl.half_w) - pl.cx, pl.w, pl.inv_w) * pl.wm1;   // :148-169
            bu = div_const((hz + pl.half_h) - pl.cz, pl.h, pl.inv_h) * pl.hm1;    // :154-174
            be = pl.wm1 - be0;                                                 // :195-197
            // :178-184 on the bit patterns: 0 <= x <= hi  <=>  bits(x) <= bits(hi) unsigned (negative, NaN: larger)
            m_valid = m_front & ballot64(f32_bits(be0) <= wm1_bits) & ballot64(f32_bits(bu) <= hm1_bits);
            }
            const float tbe = truncf(be), tbu = truncf(bu);                    // heliostat_ray_tracer.py:674-675
            const float che = be - tbe, chu = bu - tbu;                        // :694-700 (exact)
            const float cle = 1.0f - che, clu = 1.0f - chu;                    // == (tbe + 1) - be: both exact
            const float lef = tbe - e0f, luf = tbu - pu0f;                     // window coordinates of the low pixel
            const unsigned long long m_in = m_front & win_ok & ballot64(f32_bits(lef) <= twm2_bits) & ballot64(f32_bits(luf) <= thm2_bits);
            if constexpr (!CYL) n_valid += __popcll(m_valid);
            const float af = select_mask(m_in, fmaf(luf, tw4f, fmaf(lef, 4.0f, lds_base)), own_cell_f);
            const unsigned addr_lo = (unsigned)af;
            const unsigned addr_up = addr_lo + tw4p;                           // flat row iu + 1
            float ahk = ah;                                                    // the direction cosine, attenuated by the blocking mask
            if constexpr (BLOCKING) {
                // soft mask over this heliostat's rectangles, for every ray - also those that miss the target
                // (blocking.py:212-354; heliostat_ray_tracer.py:462-480); keep is exactly 0 once the transmittance drops below
                // 2^-25, as in the reference
                float blocked = 0.0f, keep = 1.0f;
                if (wmask != 0u) {
                    unsigned near, wm = wmask;
                    float tail = 0.0f;           // the sigmas of the candidates beyond the tables (a wide heliostat)
                    if (__builtin_expect((wmask >> 31) != 0u, 0)) tail = wide_ray(a, h, s_tab, wm, o.x, o.y, o.z, rx, ry, rz).sum;
                    blocked = 1.0f - soft_transmittance<true>(s_tab.prim, wm, pmask, o.x, o.y, o.z, rx, ry, rz, near, bnum0, bnum1, tail);
                    keep = 1.0f - blocked;
                }
                n_free += __popcll(ballot64(blocked < 1e-3f) & live);
                n_int += __popcll(m_valid & ballot64(keep > 0.0f));
                ahk = ah * keep;
            }
            const float Is = select_or_zero(m_in, fabsf(ahk) * kS);
            const float wa = chu * Is, wb = clu * Is;
            // the previous ray's adds have landed; a carry needs a cell that was already above 2^31 (q < 2^22)
            if (__builtin_expect(wave_any(((po1 | po2 | po3 | po4) >> 31) != 0u), 0)) carries();
            pq1 = cvt_nearest_u32(cle * wa); pq2 = cvt_nearest_u32(che * wa);
            pq3 = cvt_nearest_u32(che * wb); pq4 = cvt_nearest_u32(cle * wb);
            ptbe = tbe; ptbu = tbu;
            lds_u32* up = (lds_u32*)(size_t)addr_up;
            lds_u32* lo = (lds_u32*)(size_t)addr_lo;
            po1 = __hip_atomic_fetch_add(up, pq1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            po2 = __hip_atomic_fetch_add(up + 1, pq2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            po3 = __hip_atomic_fetch_add(lo + 1, pq3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            po4 = __hip_atomic_fetch_add(lo, pq4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            // Valid rays outside this pass's window: strays of the union window (or, when the footprint is swept in
            // bands, rays of another band; or rays on the last pixel row / column, heliostat_ray_tracer.py:723-728).
            // A stray lane PARKS its ray - pixel coordinates and direction cosine, three registers - and the wave moves
            // on: one stray lane used to drag its whole wave through ~50 instructions of address arithmetic and four
            // atomics in 45 % of the ray steps.  The parked rays go to their pixels' accumulators when a lane that is
            // already holding one strays again (every ~10 strays of a wave) and at the end of the item.
            unsigned long long m_out = m_valid & ~m_in;
            if (__builtin_expect(m_out != 0ull, 0)) {
                if (!first) m_out = 0ull;
                else if (win.npass > 1)                         // banded sweep: only what no band holds
                    m_out &= ~(win_ok & ballot64(f32_bits(lef) <= twm2_bits) & ballot64(f32_bits(tbu - u0f) <= uthm2_bits));
                if (m_out & m_parked) unpark();
                pk_be = select_mask(m_out, be, pk_be); pk_bu = select_mask(m_out, bu, pk_bu); pk_ah = select_mask(m_out, ahk, pk_ah);
                m_parked |= m_out;
            }
        };
        // The distortion stream: a ring of 8 (8) samples per thread, each slot re-requested the moment its value
        // has been read, so that seven loads (3.5 KB per wave, 56 KB per CU) are in flight at every instant.  The ablation
        // builds (tools/ablate.sh) showed what round 1 missed: with the loads of a group of four issued only one group
        // ahead, the waves waited for this stream 40 % of the forward and 60 % of the backward kernel's time.
        const int lane_off = p * (int)a.sp;
        const int nr = r1 - r0;
        // the stream's addresses: a wave-uniform row pointer that walks from sample to sample on the scalar unit (and stops
        // at the last row: the padding requests of the last ring round re-read it) + the lane's 32-bit byte offset
        const float* __restrict__ bu_ = a.dist_u + dbase;     // wave-uniform
        const float* __restrict__ be_ = a.dist_e + dbase;
        int next_r = 0;
        auto request = [&](int, float& u, float& e) {         // requests are made in sample order
            load_dist_stream<INTERLEAVED>(bu_, be_, lane_off, u, e);
            const int64_t step = next_r + 1 < nr ? a.sr : 0;
            bu_ += step; be_ += step; ++next_r;
        };
        if (nr >= 8) {
            // Every step of a round runs unconditionally (no control flow re-defines a slot: a conditional step made the
            // compiler copy freshly requested slots around and wait for them at once); the rays that pad the last round
            // re-read sample nr - 1 and are masked out.
            [[maybe_unused]] float su0, se0, su1, se1, su2, se2, su3, se3, su4, se4, su5, se5, su6, se6, su7, se7;
            request(0, su0, se0); request(1, su1, se1);
            request(2, su2, se2); request(3, su3, se3);
            request(4, su4, se4); request(5, su5, se5);
            request(6, su6, se6); request(7, su7, se7);
#define ART_RING_STEP(j)                                                            \
            {                                                                       \
                /* the slot's value moves to registers of its own first, so that the new request can land in the */ \
                /* SAME registers: otherwise the slots rotate and the loop's back-edge has to copy (= wait for) all of them */ \
                float u, e;                                                         \
                asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=v"(u), "=v"(e) : "v"(su##j), "v"(se##j) : "memory"); \
                request(k + j + 8, su##j, se##j);                      \
                trace_one(u, e, k + j < nr ? ~0ull : 0ull);                          \
                /* a ray's arithmetic stays inside its step: hoisting the next step's head above it made the */ \
                /* compiler spill the ray's live values around the hoisted code */   \
                __builtin_amdgcn_sched_barrier(0);                                  \
            }
            for (int k = 0; k < nr; k += 8) {
                ART_RING_STEP(0) ART_RING_STEP(1)
                ART_RING_STEP(2) ART_RING_STEP(3)
                ART_RING_STEP(4) ART_RING_STEP(5)
                ART_RING_STEP(6) ART_RING_STEP(7)
            }
#undef ART_RING_STEP
        } else {
            for (int r = 0; r < nr; ++r) {           // few samples per point (field-scale prediction): nothing to pipeline
                float u, e;
                request(r, u, e);
                trace_one(u, e, ~0ull);
            }
        }
    }
    {
        PendingSplat ps = {po1, po2, po3, po4, pq1, pq2, pq3, pq4, (int)ptbe, (int)ptbu};
        resolve_carries(ps, acc, a.W, a.Hh, win.shift);
    }
    if (m_parked != 0ull) unpark();
    if (first && lane == 0) {
        if constexpr (BLOCKING) { atomicAdd(&s_cnt[0], n_int); atomicAdd(&s_cnt[1], n_valid); atomicAdd(&s_cnt[2], n_free); }
        else if constexpr (CYL) { atomicAdd(&s_cnt[0], n_int); atomicAdd(&s_cnt[1], n_pos); }
        else { atomicAdd(&s_cnt[0], n_valid); atomicAdd(&s_cnt[1], n_valid); }
    }
    __syncthreads();
    unsigned next_item = 0u;
    if (tid == 0 && pass == win.npass - 1) next_item = fetch_work_item(work_counter, a);

    // ---- phase 3: flush ------------------------------------------------------------------------
    for (int row = wave; row < pth; row += nwaves) {
        unsigned long long* g = acc + (int64_t)(a.Hh - 1 - (pu0 + row)) * a.W + win.e0;
        const unsigned* trow = tile + row * win.tw;
        for (int c = lane; c < win.tw; c += 64) {
            const unsigned q = trow[c];
            if (q != 0u) atomicAdd(g + c, (unsigned long long)q << win.shift);
        }
    }
    if (tid == 0 && pass == win.npass - 1) *s_next = (int)(gridDim.x + next_item);
    __syncthreads();
  }
       // (diagnostic build: the flush is not stamped apart from the trace here)
    if (tid < (BLOCKING ? 3 : 2) && s_cnt[tid]) atomicAdd(&counts[tid * a.H + h], s_cnt[tid]);
}

// --------------------------------------------------------------------------------------------
// Field-scale prediction (BASELINE config 5: 10 000 heliostats x 10 000 points x ONE sun sample, all bitmaps summed per
// target): with a handful of samples per point an item of one heliostat is 1e4 rays between a window build, a 158 KB clear
// and a flush of ~36 000 non-zero cells - and in per-target mode every one of those flushes adds 64-bit integers onto the
// SAME bitmap (3.6e8 atomics per launch at the memory side: 1.1 of the launch's 2.45 ms).  Here an item is a GROUP of
// a.h_group consecutive heliostats: runs of heliostats aimed at the same planar target share ONE window - placed on the
// union of their (sampled) chief-ray images - and flush it once.  Nothing else changes: the same ray body as
// trace_fwd_item_lean, the same integers into the same accumulators (a ray contributes the same cell-unit integer whichever
// window it meets or misses), so the bitmap is bit-identical to the ungrouped launch's
// (tests/test_gpu_configs.py::test_field_groups_change_speed_only).  Mode 1, planar, no blocking, R < 8 samples per point,
// whole heliostats per item (n_pblocks == n_rchunks == 1).  A footprint larger than the window is trimmed, never swept.
// Each thread's NEXT point (origin, normal, first sample) is requested before the current one is traced: with one ray per
// point the loop is otherwise a chain of exposed HBM round trips.
// --------------------------------------------------------------------------------------------
template <bool INTERLEAVED>
__device__ __forceinline__ void trace_fwd_item_field(const TraceArgs& a, unsigned int* __restrict__ counts, const int group,
                                                     unsigned int* __restrict__ work_counter, int* s_next)
{
    extern __shared__ __attribute__((aligned(16))) unsigned tile[];
    __shared__ float s_red[13][16];
    __shared__ Window s_win;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int g0 = group * a.h_group, g1 = min(g0 + a.h_group, a.H);
    const int nr = a.R;
    const float Wf = (float)a.W, Hf = (float)a.Hh;
    const Cyl cy = {};
    unsigned next_item = 0u;
    bool fetched = false;
    for (int hs = g0; hs < g1;) {
        const int t = a.target_idx[hs];
        // workgroup-uniform: a bad index (reported), a cylinder's or a blocked heliostat (another launch's)
        if (!target_in_range(a, t) || t >= a.T || other_launch_owns(a, hs)) { ++hs; continue; }
        int he = hs + 1;
        while (he < g1 && a.target_idx[he] == t && !other_launch_owns(a, he)) ++he;      // the run that shares a window
        const Plane pl = load_plane(a.centers, a.pnormals, a.dims, t, a.W, a.Hh, a.mag, a.k_ext, a.k_refl);
        unsigned long long* __restrict__ acc = a.accum + (int64_t)t * a.Hh * a.W;        // (mode 1: the target's accumulators)
        // ---- phase 1: clear, then the window of the run (one evenly spaced point per thread and heliostat) ---------------
        {
            uint4* t4 = reinterpret_cast<uint4*>(tile);
            for (int i = tid; i < a.tile_cap / 4; i += blockDim.x) t4[i] = make_uint4(0u, 0u, 0u, 0u);
            if (tid < 2) tile[a.tile_cap + tid] = 0u;
        }
        {
            WindowStats w;
            // (a quarter of the threads look: 256 evenly spaced points per heliostat are plenty for the union of a run's
            //  images, and every sampled point is 40 B of extra HBM traffic on a path that is bound by exactly that)
            const int ps = (tid & 3) != 0 ? a.P : (a.P > (int)blockDim.x ? window_sample_point(0, a.P) : tid);
            for (int hh = hs; hh < he; ++hh) {
                if (ps < a.P) {
                    float u, e;
                    load_dist_row<INTERLEAVED>(a.dist_u + (int64_t)hh * a.sh, a.dist_e + (int64_t)hh * a.sh, ps * (int)a.sp, u, e);
                    window_consume<false>(w, pl, cy, a.incident[hh], a.origins[(int64_t)hh * a.P + ps], a.normals[(int64_t)hh * a.P + ps], u, e);
                }
            }
            window_reduce(w, s_red);
            if (tid == 0) {
                TraceArgs trimmed_only = a;                      // never swept in bands: trimmed to the window's capacity
                trimmed_only.multipass_ratio = 1 << 20;
                // the sample holds (he - hs) x blockDim draws: about the extreme a full block of the other kernels sees
                s_win = window_decide(trimmed_only, w, pl.wm1 / fabsf(pl.w), pl.hm1 / fabsf(pl.h), 1.15f);
                if (s_win.npass > 1) { s_win.th = s_win.ths; s_win.npass = 1; }     // (cannot happen with that ratio)
            }
            __syncthreads();
        }
        const Window win = s_win;
        const unsigned wm1_bits = f32_bits(pl.wm1), hm1_bits = f32_bits(pl.hm1);
        const float kI = (pl.mag * pl.k_ext) * pl.k_refl;
        const float kS = kI * win.scale;
        const float lds_base = (float)(unsigned)(size_t)(lds_u32*)tile;
        const float e0f = (float)win.e0, tw4f = (float)(4 * win.tw), u0f = (float)win.u0;
        const unsigned twm2_bits = f32_bits((float)(win.tw - 2)), thm2_bits = f32_bits((float)(win.th - 2));
        const unsigned long long win_ok = (win.tw >= 2 && win.th >= 2) ? ~0ull : 0ull;
        const unsigned tw4 = win_ok != 0ull ? 4u * (unsigned)win.tw : 0u;
        [[maybe_unused]] const float addr_hi_f = win_ok != 0ull ? lds_base + (float)(4 * (win.tw * (win.th - 2) + win.tw - 2)) : lds_base;
        [[maybe_unused]] const float own_cell_f = lds_base + (float)(8 * lane);      // masked rays add their zeros here (see trace_fwd_item_lean)
        // ---- phase 2: trace every heliostat of the run (the ray body of trace_fwd_item_lean) ---------------------------------
        unsigned long long m_parked = 0ull;
        float pk_be = 0.0f, pk_bu = 0.0f, pk_ah = 0.0f;
        auto unpark = [&]() {
            if ((m_parked >> lane) & 1ull) {
                const float tbe = truncf(pk_be), tbu = truncf(pk_bu);
                if ((tbe + 1.0f < Wf) && (tbu + 1.0f < Hf)) {
                    const float che = pk_be - tbe, chu = pk_bu - tbu, cle = 1.0f - che, clu = 1.0f - chu;
                    const float Is = fabsf(pk_ah) * kS;
                    const float wa = chu * Is, wb = clu * Is;
                    unsigned long long* row_hi = acc + (int64_t)(a.Hh - 2 - (int)tbu) * a.W + (int)tbe;
                    unsigned long long* row_lo = row_hi + a.W;
                    atomicAdd(row_hi, cell_to_accum(cle * wa)); atomicAdd(row_hi + 1, cell_to_accum(che * wa));
                    atomicAdd(row_lo + 1, cell_to_accum(che * wb)); atomicAdd(row_lo, cell_to_accum(cle * wb));
                }
            }
            m_parked = 0ull;
        };
        unsigned po1 = 0u, po2 = 0u, po3 = 0u, po4 = 0u, pq1 = 0u, pq2 = 0u, pq3 = 0u, pq4 = 0u;
        float ptbe = 0.0f, ptbu = 0.0f;
        for (int hh = hs; hh < he; ++hh) {
            const float4 inc = a.incident[hh];
            const float4* __restrict__ org = a.origins + (int64_t)hh * a.P;
            const float4* __restrict__ nrm = a.normals + (int64_t)hh * a.P;
            const float* __restrict__ row_u = a.dist_u + (int64_t)hh * a.sh;       // sample 0 of this heliostat (wave-uniform)
            const float* __restrict__ row_e = a.dist_e + (int64_t)hh * a.sh;
            unsigned n_valid = 0;
            int p = tid;
            float4 o = {0.0f, 0.0f, 0.0f, 1.0f}, n = {0.0f, 0.0f, 1.0f, 0.0f};
            float u0 = 0.0f, e0 = 0.0f;
            if (p < a.P) { o = org[p]; n = nrm[p]; load_dist_stream<INTERLEAVED>(row_u, row_e, p * (int)a.sp, u0, e0); }
            while (p < a.P) {
                const int pn = p + (int)blockDim.x;
                float4 o2 = o, n2 = n;
                float u2 = 0.0f, e2 = 0.0f;
                if (pn < a.P) { o2 = org[pn]; n2 = nrm[pn]; load_dist_stream<INTERLEAVED>(row_u, row_e, pn * (int)a.sp, u2, e2); }
                float4 d; float s;
                reflect(inc, n, d, s);
                const float numer = plane_numer(pl, o);
                for (int r = 0; r < nr; ++r) {
                    float u = u0, e = e0;
                    if (r > 0) load_dist_stream<INTERLEAVED>(row_u + (int64_t)r * a.sr, row_e + (int64_t)r * a.sr, p * (int)a.sp, u, e);
                    Rot m = make_rot_t<true>(e, u);
                    if (__builtin_expect(wave_any(!(fabsf(u) + fabsf(e) <= kSmallAngle)), 0)) m = make_rot(e, u);
                    float rx, ry, rz;
                    scatter(m, d, rx, ry, rz);
                    const float ah = (rx * pl.mx + ry * pl.my) + rz * pl.mz;           // geometry.py:116-118
                    const unsigned long long m_front = ballot64(ah < 0.0f);
                    const float tt = div_noscale(numer, ah);                           // :130-131 (back-facing: masked below)
                    const float hx = o.x + rx * tt, hz = o.z + rz * tt;                // :133-136
                    const float be0 = div_const((hx + pl.half_w) - pl.cx, pl.w, pl.inv_w) * pl.wm1;   // :148-169
                    const float bu = div_const((hz + pl.half_h) - pl.cz, pl.h, pl.inv_h) * pl.hm1;    // :154-174
                    const float be = pl.wm1 - be0;                                     // :195-197
                    const float tbe = truncf(be), tbu = truncf(bu);                    // heliostat_ray_tracer.py:674-675
                    const float che = be - tbe, chu = bu - tbu;                        // :694-700 (exact)
                    const float cle = 1.0f - che, clu = 1.0f - chu;
                    const float lef = tbe - e0f, luf = tbu - u0f;
                    const unsigned long long m_valid = m_front & ballot64(f32_bits(be0) <= wm1_bits) & ballot64(f32_bits(bu) <= hm1_bits);
                    const unsigned long long m_in = m_front & win_ok & ballot64(f32_bits(lef) <= twm2_bits) & ballot64(f32_bits(luf) <= thm2_bits);
                    n_valid += __popcll(m_valid);
                    const float af = select_mask(m_in, fmaf(luf, tw4f, fmaf(lef, 4.0f, lds_base)), own_cell_f);
                    const unsigned addr_lo = (unsigned)af;
                    const unsigned addr_up = addr_lo + tw4;
                    const float Is = select_or_zero(m_in, fabsf(ah) * kS);
                    const float wa = chu * Is, wb = clu * Is;
                    if (__builtin_expect(wave_any(((po1 | po2 | po3 | po4) >> 31) != 0u), 0)) {
                        PendingSplat ps = {po1, po2, po3, po4, pq1, pq2, pq3, pq4, (int)ptbe, (int)ptbu};
                        resolve_carries(ps, acc, a.W, a.Hh, win.shift);
                    }
                    pq1 = cvt_nearest_u32(cle * wa); pq2 = cvt_nearest_u32(che * wa);
                    pq3 = cvt_nearest_u32(che * wb); pq4 = cvt_nearest_u32(cle * wb);
                    ptbe = tbe; ptbu = tbu;
                    lds_u32* up = (lds_u32*)(size_t)addr_up;
                    lds_u32* lo = (lds_u32*)(size_t)addr_lo;
                    po1 = __hip_atomic_fetch_add(up, pq1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    po2 = __hip_atomic_fetch_add(up + 1, pq2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    po3 = __hip_atomic_fetch_add(lo + 1, pq3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    po4 = __hip_atomic_fetch_add(lo, pq4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const unsigned long long m_out = m_valid & ~m_in;                  // strays park (see trace_fwd_item_lean)
                    if (__builtin_expect(m_out != 0ull, 0)) {
                        if (m_out & m_parked) unpark();
                        pk_be = select_mask(m_out, be, pk_be); pk_bu = select_mask(m_out, bu, pk_bu); pk_ah = select_mask(m_out, ah, pk_ah);
                        m_parked |= m_out;
                    }
                }
                o = o2; n = n2; u0 = u2; e0 = e2; p = pn;
            }
            // this heliostat's ray counters (a valid ray carries intensity: one count serves both factors)
            if (lane == 0 && n_valid != 0u) { atomicAdd(&counts[hh], n_valid); atomicAdd(&counts[a.H + hh], n_valid); }
        }
        {
            PendingSplat ps = {po1, po2, po3, po4, pq1, pq2, pq3, pq4, (int)ptbe, (int)ptbu};
            resolve_carries(ps, acc, a.W, a.Hh, win.shift);
        }
        if (m_parked != 0ull) unpark();
        __syncthreads();
        // the next work item is requested before the group's last flush and published after it
        if (he >= g1 && tid == 0) { next_item = fetch_work_item(work_counter, a); fetched = true; }
        // ---- phase 3: flush ------------------------------------------------------------------------
        for (int row = wave; row < win.th; row += nwaves) {
            unsigned long long* g = acc + (int64_t)(a.Hh - 1 - (win.u0 + row)) * a.W + win.e0;
            const unsigned* trow = tile + row * win.tw;
            for (int c = lane; c < win.tw; c += 64) {
                const unsigned q = trow[c];
                if (q != 0u) atomicAdd(g + c, (unsigned long long)q << win.shift);
            }
        }
        __syncthreads();                 // flushed before the next run clears the tile
        hs = he;
    }
    if (tid == 0) {
        if (!fetched) next_item = fetch_work_item(work_counter, a);     // (the group ended with skipped heliostats)
        *s_next = (int)(gridDim.x + next_item);
    }
}

// Everything the forward kernel is launched with, as ONE argument: the persistent loop below re-reads what an item
// needs from the kernarg segment instead of carrying it in registers.
struct FwdLaunch { TraceArgs a; float* flux; unsigned int* counts; unsigned int* work_counter; };
// The blocking instantiations: one item per workgroup in the forward kernel (inside the persistent loop it spills enough
// to lose 8 % in exact mode), the persistent loop in the backward kernel (same-box, tools/blocking_bench.py: backward of
// exact mode 12.6 -> 11.8 ms, of reference-tree mode 5.3 -> 5.0 ms - its few dozen items no longer wait behind 5000
// workgroups that exit at once but need a CU's LDS to do so).
constexpr bool kBlockingPersistentFwd = false, kBlockingPersistentBwd = true;
constexpr bool kCylPersistentBwd = false;       // (measured neutral: 18.57 vs 18.43 ms)

constexpr int kLeanFwdThreads = 1024;
constexpr int kLeanBlockFwdThreads = 768;   // the lean body + the soft mask: 168 registers instead of 128
// the lean body + the cylinder hit: four waves per SIMD with 88 B of spills beat three without (same box, tools/cylinder_bench.py,
// forward / forward + backward: 640 threads 6.86 / 14.2 ms, 768 5.61 / 12.6, 896 5.52 / 12.5, 1024 5.40 / 12.3; the blocking body:
// 8.33, 7.26, 7.44, 7.18 ms forward - within the noise of the box from 768 on)
constexpr int kLeanCylFwdThreads = 1024;
constexpr int kCylFwdThreads = 1024;             // (768: within the noise of the box, 512: 17 % slower)
// LEAN: 0 the generic item, 1 trace_fwd_item_lean, 2 trace_fwd_item_field (groups of heliostats, see there)
template <bool INTERLEAVED, bool CYL, bool BLOCKING, int LEAN = 0>
__global__ __launch_bounds__(LEAN ? (BLOCKING ? kLeanBlockFwdThreads : (CYL ? kLeanCylFwdThreads : kLeanFwdThreads)) : (CYL ? kCylFwdThreads : 1024)) void trace_fwd_lds_kernel(FwdLaunch launch)
{
    static_assert(!LEAN || ((LEAN == 1 || (!BLOCKING && !CYL)) && !(BLOCKING && CYL)),
                  "the lean body serves planes, planes with blocking, or cylinders; the field item is planar without blocking");
    __shared__ int s_next, s_reverse;
    // The cylinder and blocking instantiations keep more values alive per ray; inside the persistent loop they spill
    // enough to lose 5-9 % (tools/blocking_bench.py, same-box A/B), so they take one item per workgroup.
    constexpr bool single_item = (CYL && LEAN == 0) || (BLOCKING && LEAN == 0 && !kBlockingPersistentFwd);
    if constexpr (single_item) {
        int item = (int)blockIdx.x;
        if (item >= work_item_count(launch.a)) return;
        if constexpr (BLOCKING) {
            if (launch.a.split == 2) {               // the blocked heliostats first (see kth_blocked_heliostat)
                __shared__ int s_scan[18];
                const int per = launch.a.n_pblocks * launch.a.n_rchunks;
                const int hk = kth_blocked_heliostat(launch.a, item / per, s_scan);
                if (hk < 0) return;                  // workgroup-uniform: fewer blocked heliostats than that
                item = hk * per + item % per;
            }
        }
        if constexpr (CYL && !BLOCKING) {
            if (launch.a.split == 3) {               // the heliostats that aim at a cylinder first
                __shared__ int s_scan[18];
                const int per = launch.a.n_pblocks * launch.a.n_rchunks;
                const int hk = kth_blocked_heliostat<true>(launch.a, item / per, s_scan);
                if (hk < 0) return;
                item = hk * per + item % per;
            }
        }
        trace_fwd_item<INTERLEAVED, CYL, BLOCKING>(launch.a, launch.flux, launch.counts, item, decode_work_item(launch.a, item),
                                                   nullptr, &s_next);
        return;
    }
    if (threadIdx.x == 0) {
        s_next = (int)blockIdx.x;                             // the first gridDim.x items need no counter
        s_reverse = farther_end_is_last(launch.a) ? 1 : 0;
    }
    __syncthreads();
    for (;;) {
        // The trace loop of an item fills the register file (100 SGPRs, 114 VGPRs).  Anything carried across items - the
        // ~80 argument loads the compiler would hoist out of this loop, even the three pointers - overflows the SGPRs
        // into VGPR lanes and makes the hot loop spill (measured: 258 scratch accesses per ray group).  So an item starts
        // from nothing but the kernarg segment pointer, whose origin the empty asm hides from the optimiser: every
        // field is re-read with scalar loads (cached) where it is needed, and the item number comes from LDS.
        typedef const FwdLaunch __attribute__((address_space(4))) * KernargPtr;
        KernargPtr lp = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(lp));
        const FwdLaunch& L = *(const FwdLaunch*)lp;
        const int item = __builtin_amdgcn_readfirstlane(s_next);
        __syncthreads();                                     // everybody has read s_next before this item overwrites it
        if (item >= work_item_count(L.a)) break;             // workgroup-uniform
        if constexpr (LEAN == 2)
            trace_fwd_item_field<INTERLEAVED>(L.a, L.counts, s_reverse != 0 ? work_item_count(L.a) - 1 - item : item, L.work_counter, &s_next);
        else if constexpr (LEAN == 1)
            trace_fwd_item_lean<INTERLEAVED, BLOCKING, CYL>(L.a, L.flux, L.counts, item, decode_work_item(L.a, item, s_reverse != 0),
                                                       L.work_counter, &s_next);
        else
            trace_fwd_item<INTERLEAVED, CYL, BLOCKING>(L.a, L.flux, L.counts, item, decode_work_item(L.a, item, s_reverse != 0),
                                                       L.work_counter, &s_next);
        __syncthreads();
    }
}

constexpr int kCountersPerStream = 8;      // work counters per (device, stream): see stream_work_counters
// What a forward call zeroes before its kernels start - the three ray counters per heliostat (they alias `factors`) and the
// work counters of its (at most two) launches - in ONE launch: three memsets were three 5 us kernels, a tenth of the
// forward pass of a 16-heliostat field.
// It also looks at the stream's work counters (kCountersPerStream of them): a counter is zero whenever no launch is using it - the
// last fetch of a launch resets it - and the launches of a stream are ordered, so a non-zero counter here means that an earlier
// launch on this stream did not make all its fetches (a fault, an abort).  Every later launch would silently skip or repeat
// items; instead the counters are put back to zero and the status word says so (bit 2: ART_EQUEUE).
__global__ void trace_fwd_prep_kernel(unsigned* __restrict__ counts, int n, unsigned* __restrict__ work_counters, unsigned* __restrict__ status)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) counts[i] = 0u;
    if (blockIdx.x == 0 && threadIdx.x < kCountersPerStream && work_counters[threadIdx.x] != 0u) {
        work_counters[threadIdx.x] = 0u;
        if (status != nullptr) atomicOr(status, 4u);
    }
}

// counts (uint32, rows 0,1 of factors) -> fractions (heliostat_ray_tracer.py:498-506).
__global__ void finalize_factors_kernel(float* factors, int H, float rays_per_heliostat, int blocking, const int32_t* unblocked_if_empty,
                                        const int32_t* cand_count, int Cmax)
{
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= H) return;
    // More candidate rectangles than the tables hold (art_blocking_filter kept the count): the heliostat was traced with an
    // incomplete blocking mask.  Its factors are NaN (poison_overflow_kernel does the same to its bitmap), so the call that
    // overflowed cannot be mistaken for a result even by a caller who never asks art_async_status.
    if (cand_count != nullptr && cand_count[h] > Cmax) {
        const float nan = __builtin_nanf("");
        factors[h] = nan; factors[H + h] = nan; factors[2 * H + h] = nan;
        return;
    }
    const unsigned* c = reinterpret_cast<const unsigned*>(factors);
    const unsigned n_int = c[h], n_on = c[H + h], n_free = c[2 * H + h];
    factors[h] = (float)n_int / rays_per_heliostat;
    factors[H + h] = (float)n_on / rays_per_heliostat;
    // blocking off: blocked == 0 everywhere
    // (split launches: a heliostat with an empty candidate list went through the lean kernel, which does not count free rays)
    if (unblocked_if_empty != nullptr && unblocked_if_empty[h] == 0) blocking = 0;
    factors[2 * H + h] = (blocking ? (float)n_free : rays_per_heliostat) / rays_per_heliostat;
}

// The bitmap of a heliostat whose candidate list overflowed (see finalize_factors_kernel) becomes NaN: row h in mode 0, its
// target's bitmap in mode 1.  One workgroup per heliostat; all but the overflowed ones exit at once.  Blocking calls only.
__global__ __launch_bounds__(256) void poison_overflow_kernel(const int32_t* __restrict__ cand_count, int Cmax,
                                                              const int32_t* __restrict__ target_idx, int n_targets,
                                                              float* __restrict__ flux, int64_t npix, int mode,
                                                              double* __restrict__ moments)
{
    const int h = blockIdx.x;
    if (cand_count[h] <= Cmax) return;
    int64_t map = h;
    if (mode == 1) {
        const int t = target_idx[h];
        if ((unsigned)t >= (unsigned)n_targets) return;
        map = t;
    }
    const float nan = __builtin_nanf("");
    for (int64_t i = threadIdx.x; i < npix; i += blockDim.x) flux[map * npix + i] = nan;
    if (moments != nullptr && threadIdx.x < kLossParts * 3) moments[map * kLossParts * 3 + threadIdx.x] = (double)nan;
}

// --------------------------------------------------------------------------------------------
// Backward of the plain formulation (ARTIST_HIP_FWD=global): thread owns a point and ALL its samples, accumulates dL/dd and
// dL/do in registers, plain stores.
// --------------------------------------------------------------------------------------------
template <bool INTERLEAVED>
__global__ __launch_bounds__(kBlock) void trace_bwd_kernel(TraceArgs a, const float* __restrict__ grad_flux,
                                                           float4* __restrict__ grad_origins,
                                                           float4* __restrict__ grad_normals)
{
    const int bid = blockIdx.x;
    const int ptile = bid % a.n_ptiles;
    const int rchunk = (bid / a.n_ptiles) % a.n_rchunks;
    const int h = bid / (a.n_ptiles * a.n_rchunks);
    const int p = ptile * kBlock + threadIdx.x;
    if (p >= a.P) return;

    const int t = a.target_idx[h];
    if (!target_in_range(a, t) || t >= a.T) return;
    const Plane pl = load_plane(a.centers, a.pnormals, a.dims, t, a.W, a.Hh, a.mag, a.k_ext, a.k_refl);
    const float* __restrict__ G = grad_flux + (int64_t)(a.mode == 0 ? h : t) * a.Hh * a.W;

    const float4 o = a.origins[(int64_t)h * a.P + p];
    const float4 n = a.normals[(int64_t)h * a.P + p];
    const float4 inc = a.incident[h];
    float4 d; float s;
    reflect(inc, n, d, s);
    const float numer = plane_numer(pl, o);
    const float kI = (pl.mag * pl.k_ext) * pl.k_refl;
    const float sx = pl.wm1 / pl.w, sz = pl.hm1 / pl.h;

    float gdx = 0.f, gdy = 0.f, gdz = 0.f, gox = 0.f, goy = 0.f, goz = 0.f;
    const int r0 = rchunk * a.r_chunk;
    const int r1 = min(r0 + a.r_chunk, a.R);
    int64_t off = (int64_t)h * a.sh + (int64_t)r0 * a.sr + (int64_t)p * a.sp;
    for (int r = r0; r < r1; ++r, off += a.sr) {
        float u, e;
        load_dist<INTERLEAVED>(a, off, u, e);
        const Rot m = make_rot(e, u);
        float rx, ry, rz;
        scatter(m, d, rx, ry, rz);
        const Hit hit = intersect(pl, o, numer, rx, ry, rz);
        if (!hit.valid) continue;
        const Splat sp = splat_weights(hit.be, hit.bu, a.W, a.Hh);
        if (!sp.on) continue;
        const float I = (hit.I0 * pl.k_ext) * pl.k_refl;
        const float* g_hi = G + (int64_t)(a.Hh - 2 - sp.iu) * a.W + sp.ie;
        const float* g_lo = g_hi + a.W;
        const float g1 = g_hi[0], g2 = g_hi[1], g3 = g_lo[1], g4 = g_lo[0];
        const float gI = sp.cle * sp.chu * g1 + sp.che * sp.chu * g2 + sp.che * sp.clu * g3 + sp.cle * sp.clu * g4;
        const float g_be = ((sp.chu * g2 + sp.clu * g3) - (sp.chu * g1 + sp.clu * g4)) * I;
        const float g_bu = ((sp.cle * g1 + sp.che * g2) - (sp.che * g3 + sp.cle * g4)) * I;
        const float g_hx = -g_be * sx;          // be = wm1 - te / w * wm1
        const float g_hz = g_bu * sz;
        const float g_t = g_hx * rx + g_hz * rz;
        const float inv_a = 1.0f / hit.a;
        const float tt = numer * inv_a;          // t = numer / a (front facing)
        const float g_a = -kI * gI - g_t * tt * inv_a;
        const float g_numer = g_t * inv_a;
        const float grx = g_hx * tt + g_a * pl.mx;
        const float gry = g_a * pl.my;
        const float grz = g_hz * tt + g_a * pl.mz;
        gox += g_hx - g_numer * pl.mx;
        goy += -g_numer * pl.my;
        goz += g_hz - g_numer * pl.mz;
        // g_d = M^T g_r ; rows of M: [cu,-su,0], [m10,m11,-se], [m20,m21,ce]
        gdx += m.cu * grx + m.m10 * gry + m.m20 * grz;
        gdy += -m.su * grx + m.m11 * gry + m.m21 * grz;
        gdz += -m.se * gry + m.ce * grz;
    }
    // d = i - 2 (i.n) n  ->  dL/dn = -2 ((g_d . n) i + (i . n) g_d)
    const float gdn = gdx * n.x + gdy * n.y + gdz * n.z;
    float4 go = make_float4(gox, goy, goz, 0.0f);
    float4 gn = make_float4(-2.0f * (gdn * inc.x + s * gdx), -2.0f * (gdn * inc.y + s * gdy),
                            -2.0f * (gdn * inc.z + s * gdz), -2.0f * (gdn * inc.w));
    const int64_t idx = (int64_t)h * a.P + p;
    grad_origins[idx] = go;
    grad_normals[idx] = gn;
}

// --------------------------------------------------------------------------------------------
// Backward, LDS-staged gradient window (the production kernel).  Same decomposition as the forward:
// a workgroup owns `p_block` points x a chunk of samples; the part of dL/dflux its rays can touch is
// copied once into LDS (row-contiguous loads) and the four per-ray gathers become LDS reads; rays
// outside the window read global memory.  Gradients are accumulated per point in registers.
// --------------------------------------------------------------------------------------------
// Adjoint of the blocking mask for one ray of every lane: `near` = the rectangles whose mask the lane's ray entered
// (0 for lanes without a gradient), g_sigma = dL/dsigma.  Ray-side gradients come back per lane.
// The RECTANGLES' gradients (12 floats per candidate: corner 0, span u, span v, normal) are summed in a fixed order, so that
// they are bit-reproducible like everything else (round 2 added them to LDS doubles and then to the tables with float
// atomics): the 64 lanes' contributions to one rectangle are reduced on the DPP network (a fixed tree), and the wave's
// running sums live in REGISTERS - lanes 2k and 2k + 1 own rectangle k's components 0-5 and 6-11, six registers per lane
// for the 32 candidates.  The waves' sums are combined in wave order at the end of the item (trace_bwd_item), the items'
// in item order by reduce_prim_grads_kernel.  Few rays sit in the soft edge of a rectangle, so all of this is off the
// common path: with up to four such rays in the wave their values go to the owner lanes ray by ray (v_readlane, lane
// order), with more through twelve reductions on the DPP network - either way a fixed function of the data.
// The function is a real call (eight inlined copies would not fit the instruction cache), so its interface is kept in
// registers: the table comes as an LDS pointer (a generic pointer costs a null check per access), the wave-uniform mask
// goes back to an SGPR, gradients and sums travel by value (references would go through scratch memory).
struct RayGrad { float ox, oy, oz, rx, ry, rz; };
struct PrimSums { float v[6]; };                      // this lane's share of the wave's rectangle gradients
struct AdjointOut { RayGrad ray; PrimSums sums; };
typedef const __attribute__((address_space(3))) Prim* LdsPrims;
// POINT_SLOT (the lean backward item, where this body is INLINED - its ring has two steps): the first TWO rectangles of the wave's
// mask - usually the only ones whose edges a wave's rays meet - are not handed over ray by ray: a lane adds its ray's twelve values
// to `point` / `point2` (its own registers, summed over the samples of its point) and the item reduces them across the wave once
// per point.  The ray-by-ray hand-over cost 2.1 of exact mode's 18.2 ms (ablation build); same box: the call of round 3's first
// half 18.2 ms, this body inlined 18.1, one slot 17.0, two 16.8.
struct PointSums { float v[12]; };
template <bool POINT_SLOT>
__device__ __forceinline__ AdjointOut block_adjoint_body(LdsPrims prims, PrimSums sums, PointSums& point, PointSums& point2, unsigned wave_mask, unsigned near,
                                                         float ox, float oy, float oz, float rx, float ry, float rz, float g_sigma,
                                                         float num0 = 0.0f, float num1 = 0.0f)
{
    AdjointOut out = {{0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, sums};
    const int lane = threadIdx.x & 63;
    [[maybe_unused]] int rect_no = 0;
    for (unsigned m = __builtin_amdgcn_readfirstlane(wave_mask); m != 0u; m &= m - 1u) {
        const int k = __builtin_ctz(m);
        [[maybe_unused]] const int slot = rect_no++;
        const bool on = (near >> k) & 1u;
        if (!wave_any(on)) continue;
        typedef float v4f __attribute__((ext_vector_type(4)));
        static_assert(sizeof(Prim) == 6 * sizeof(v4f), "Prim is six 128-bit LDS reads");
        const __attribute__((address_space(3))) v4f* src = (const __attribute__((address_space(3))) v4f*)(prims + k);
        const v4f words[6] = {src[0], src[1], src[2], src[3], src[4], src[5]};
        Prim q;
        __builtin_memcpy(&q, words, sizeof(Prim));
        SoftHit sh;
        bool in_front;
        if constexpr (POINT_SLOT) in_front = soft_plane(q, ox, oy, oz, rx, ry, rz, sh, slot < 2, slot == 0 ? num0 : num1);   // (the caller's per-point numerators)
        else in_front = soft_plane(q, ox, oy, oz, rx, ry, rz, sh);
        soft_uv(q, ox, oy, oz, rx, ry, rz, in_front, sh);
        SoftSig sg;
        (void)soft_sigma(sh, sg);
        SoftGrad g;
        soft_sigma_bwd(q, ox, oy, oz, rx, ry, rz, sh, sg, on ? g_sigma : 0.0f, g);
        if (on) {     // other lanes may hold non-finite intermediates: branch, do not multiply
            out.ray.ox += g.ox; out.ray.oy += g.oy; out.ray.oz += g.oz;
            out.ray.rx += g.rx; out.ray.ry += g.ry; out.ray.rz += g.rz;
        }
        // (lanes without a gradient contribute exact zeros - selected, not multiplied)
        const float part[12] = {on ? g.c0[0] : 0.f, on ? g.c0[1] : 0.f, on ? g.c0[2] : 0.f, on ? g.su[0] : 0.f, on ? g.su[1] : 0.f,
                                on ? g.su[2] : 0.f, on ? g.sv[0] : 0.f, on ? g.sv[1] : 0.f, on ? g.sv[2] : 0.f, on ? g.n[0] : 0.f,
                                on ? g.n[1] : 0.f, on ? g.n[2] : 0.f};
        if constexpr (POINT_SLOT) {
            if (slot == 0) {         // (wave-uniform) this lane's own sums: reduced once per point by the caller
#pragma unroll
                for (int c = 0; c < 12; ++c) point.v[c] += part[c];
                continue;
            }
            if (slot == 1) {
#pragma unroll
                for (int c = 0; c < 12; ++c) point2.v[c] += part[c];
                continue;
            }
        }
        const bool lo = lane == 2 * k, hi = lane == 2 * k + 1;
        const unsigned long long m_on = __builtin_amdgcn_ballot_w64(on);
        if (__popcll(m_on) <= 4) {
            // the usual case - one or two rays of the wave sit in this rectangle's soft edge: their twelve values are handed to
            // the owner lanes one ray after the other, in lane order (v_readlane: no reduction network)
            for (unsigned long long mm = m_on; mm != 0ull; mm &= mm - 1ull) {
                const int src = __builtin_ctzll(mm);
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    const float a_lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, part[c]), src));
                    const float a_hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, part[6 + c]), src));
                    out.sums.v[c] += lo ? a_lo : (hi ? a_hi : 0.0f);
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const float a_lo = wave_reduce<kSum>(part[c]), a_hi = wave_reduce<kSum>(part[6 + c]);   // wave-uniform
                out.sums.v[c] += lo ? a_lo : (hi ? a_hi : 0.0f);
            }
        }
    }
    return out;
}
// (the generic item calls it: eight inlined copies would not fit the instruction cache)
__device__ __attribute__((noinline)) AdjointOut block_adjoint(LdsPrims prims, PrimSums sums, unsigned wave_mask, unsigned near,
                                                              float ox, float oy, float oz, float rx, float ry, float rz,
                                                              float g_sigma)
{
    PointSums unused = {}, unused2 = {};
    return block_adjoint_body<false>(prims, sums, unused, unused2, wave_mask, near, ox, oy, oz, rx, ry, rz, g_sigma);
}

// The same adjoint for the listed candidates of a wide heliostat (see wide_ray; all lanes of the wave are active here): `wm`
// and `nr` - the wave's and this lane's masks - lose their bit 31, the ray side is returned, the rectangle side - twelve wave
// sums per rectangle with a lane in its soft edge - goes to the heliostat's row of `wide_grad` by fp64 atomics (the items of a
// heliostat share the row; the sums are rounded to fp32 once, by reduce_prim_grads_kernel: their order can move the result by
// an fp64 rounding, not by an fp32 one except in a tie).
__device__ __attribute__((noinline)) RayGrad wide_adjoint_listed(const WideTabs w, unsigned long long bits, float ox, float oy, float oz,
                                                                 float rx, float ry, float rz, float g_sigma, bool adj)
{
    RayGrad out = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int lane = threadIdx.x & 63;
    wide_for_each(w, bits, [&](int c) {
        const Prim q = wide_prim(w, c);
        SoftHit sh;
        const bool in_front = soft_plane(q, ox, oy, oz, rx, ry, rz, sh);
        soft_uv(q, ox, oy, oz, rx, ry, rz, in_front, sh);
        if (!wave_any(adj && sh.near)) return;
        SoftSig sg;
        (void)soft_sigma(sh, sg);
        const bool on = adj && sh.near && sg.sigma_raw != 1.0f;
        if (!wave_any(on)) return;
        SoftGrad g;
        soft_sigma_bwd(q, ox, oy, oz, rx, ry, rz, sh, sg, on ? g_sigma : 0.0f, g);
        if (on) {     // other lanes may hold non-finite intermediates: branch, do not multiply
            out.ox += g.ox; out.oy += g.oy; out.oz += g.oz;
            out.rx += g.rx; out.ry += g.ry; out.rz += g.rz;
        }
        const float part[12] = {on ? g.c0[0] : 0.f, on ? g.c0[1] : 0.f, on ? g.c0[2] : 0.f, on ? g.su[0] : 0.f, on ? g.su[1] : 0.f,
                                on ? g.su[2] : 0.f, on ? g.sv[0] : 0.f, on ? g.sv[1] : 0.f, on ? g.sv[2] : 0.f, on ? g.n[0] : 0.f,
                                on ? g.n[1] : 0.f, on ? g.n[2] : 0.f};
        double* __restrict__ cell = w.grad_row + (int64_t)(c - kWideFirst) * 12;
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const float total = wave_reduce<kSum>(part[j]);          // wave-uniform
            if (lane == 0 && w.grad_row != nullptr) atomicAdd(cell + j, (double)total);
        }
    });
    return out;
}
template <bool GRAD>
__device__ __forceinline__ RayGrad wide_ray_adjoint(const TraceArgs& a, int h, PrimTable<true, GRAD>& tab, unsigned& wm, unsigned& nr,
                                                    float ox, float oy, float oz, float rx, float ry, float rz, float g_sigma)
{
    RayGrad out = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int n_list = __builtin_amdgcn_readfirstlane(tab.n_wide);
    if (n_list == 0) return out;
    const bool adj = (nr >> 31) != 0u;
    wm &= 0x7FFFFFFFu; nr &= 0x7FFFFFFFu;
    if (!wave_any(adj)) return out;
    unsigned long long bits = tab.xmask[threadIdx.x >> 6];
    bits = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(bits >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)bits);
    return wide_adjoint_listed(wide_tabs(a, h, n_list), bits, ox, oy, oz, rx, ry, rz, g_sigma, adj);
}
template <bool GRAD> __device__ __forceinline__ RayGrad wide_ray_adjoint(const TraceArgs&, int, PrimTable<false, GRAD>&, unsigned&, unsigned&, float, float, float, float, float, float, float) { RayGrad z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; return z; }

// A heliostat that is skipped because its target index is outside the tables (ART_ETARGET, target_in_range) gets ZERO
// gradients for the item's points - `grad_origins` / `grad_normals` (or the item's chunk slab) are the caller's uninitialised
// memory, and the status word may not have reached the host before the optimiser reads them.  (Both launches of a split call
// may do this for the same heliostat: the same zeros.)
__device__ __forceinline__ void zero_block_gradients(const TraceArgs& a, const WorkItem item, float4* __restrict__ grad_origins,
                                                     float4* __restrict__ grad_normals)
{
    int p0, p1;
    block_range(a, item.pblock, p0, p1, item.tail);
    const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (int p = p0 + (int)threadIdx.x; p < p1; p += (int)blockDim.x) {
        grad_origins[(int64_t)item.h * a.P + p] = z;
        grad_normals[(int64_t)item.h * a.P + p] = z;
    }
}

typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(1))) const float glb_f32;

// The backward items' gradient window: pth rows of tw cells of dL/dflux (LDS row k = bitmap row at g - k W: the bitmap is
// flipped), one contiguous run of pth x tw floats at `tile`.  LDS-direct loads (global_load_lds_dword: lane l's dword lands at
// LDS address M0 + 4 l, no destination registers): a wave takes every nwaves-th group of 64 cells and keeps ALL its loads in
// flight - one round trip per item where sixteen registers per lane bought four rows per round trip (6.2 -> ~2 us per item,
// tools/timeline.sh), and the staging batch no longer is the kernel's register peak (lean item: 155 -> 118).
// Inlined: as a real call the function would read the dynamic-LDS base through the table the compiler builds for callees.
__device__ __forceinline__ void stage_grad_window(const float* __restrict__ g, int W, int tw, int pth, float* tile, int wave,
                                                  int lane, int nwaves)
{
    const int n_cells = pth * tw;
    const int twd = max(tw, 1);
    const int step = nwaves * 64;
    const int sq = step / twd, sr = step - sq * twd;
    const int i0 = wave * 64 + lane;
    int row = i0 / twd, col = i0 - row * twd;
    for (int base = wave * 64; base < n_cells; base += step) {
        if (base + lane < n_cells)
            __builtin_amdgcn_global_load_lds((glb_f32*)(g + ((int64_t)col - (int64_t)row * W)), (lds_f32*)(tile + base), 4, 0, 0);
        col += sr; row += sq;
        if (col >= twd) { col -= twd; ++row; }
    }
    // this wave's cells are in LDS before it reaches the caller's barrier (vmcnt(0); the compiler puts the same wait there for
    // the barrier's fence - stated here so that the staging does not depend on that)
    __builtin_amdgcn_s_waitcnt(0x0F70);
}

// (The cylinder and blocking instantiations keep ~60 more live values per ray; they run 768-thread workgroups =
// 168 VGPRs, see kCylBwdThreads.)
template <bool INTERLEAVED, bool ATOMIC_OUT, bool CYL, bool BLOCKING>
__device__ __forceinline__ void trace_bwd_item(const TraceArgs& a, const float* __restrict__ grad_flux,
                                               float4* __restrict__ grad_origins, float4* __restrict__ grad_normals,
                                               float* __restrict__ prim_slabs, const WorkItem item,
                                               unsigned int* __restrict__ work_counter, int* s_next)
{
    extern __shared__ __attribute__((aligned(16))) float gtile[];
    __shared__ float s_red[13][16];
    __shared__ Window s_win;
    __shared__ PrimTable<BLOCKING> s_tab;

    const int pblock = item.pblock;
    const int h = item.h;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    // ATOMIC_OUT (a name from round 1): the samples of a point are split over several items.  Each sample chunk then
    // writes its partial gradients with plain stores to a slab of its own - `grad_origins` / `grad_normals` point at
    // [n_rchunks,H,P] scratch arrays - and reduce_chunks_kernel adds the slabs in chunk order: bit-reproducible, where
    // float atomics onto one row were not.
    if constexpr (ATOMIC_OUT) {
        grad_origins += (int64_t)item.rchunk * a.H * a.P;
        grad_normals += (int64_t)item.rchunk * a.H * a.P;
    }

    const int t = a.target_idx[h];
    const bool bad_target = !target_in_range(a, t);
    if (bad_target || (t >= a.T) != CYL || item.r1 <= item.r0 || other_launch_owns(a, h)) {   // workgroup-uniform: a bad index, or another launch owns this heliostat
        if (bad_target) {
            zero_block_gradients(a, item, grad_origins, grad_normals);
            if constexpr (BLOCKING) {                // ... and no rectangle gradients from this item either
                float* __restrict__ slab = prim_slabs + ((int64_t)(item.h * a.n_pblocks + item.pblock) * a.n_rchunks + item.rchunk) * slab_rows(a) * 12;
                for (int c = tid; c < slab_rows(a) * 12; c += blockDim.x) slab[c] = 0.0f;
            }
        }
        if (tid == 0) *s_next = (int)(gridDim.x + fetch_work_item(work_counter, a));
        return;
    }
    Plane pl; Cyl cy;
    if constexpr (CYL) cy = load_cyl(a.cyl_centers, a.cyl_normals, a.cyl_axes, a.cyl_radii, a.cyl_heights, a.cyl_opening,
                                     t - a.T, a.W, a.Hh, a.mag, a.k_ext, a.k_refl);
    else pl = load_plane(a.centers, a.pnormals, a.dims, t, a.W, a.Hh, a.mag, a.k_ext, a.k_refl);
    const float* __restrict__ G = grad_flux + (int64_t)(a.mode == 0 ? h : t) * a.Hh * a.W;
    const float4 inc = a.incident[h];
    int p0, p1;
    block_range(a, pblock, p0, p1);
    const int r0 = item.r0;
    const int r1 = item.r1;
    const float4* __restrict__ org = a.origins + (int64_t)h * a.P;
    const float4* __restrict__ nrm = a.normals + (int64_t)h * a.P;
    const int64_t dbase = (int64_t)h * a.sh + (int64_t)r0 * a.sr;

    const int n_prims = load_prims<BLOCKING>(a, h, s_tab);
    compute_window<INTERLEAVED, CYL>(a, pl, cy, inc, org, nrm, p0, p1, dbase, s_red, &s_win);
    const Window win = s_win;
    unsigned next_item = 0u;
    PrimSums prim_sums = {{0.f, 0.f, 0.f, 0.f, 0.f, 0.f}};               // this wave's rectangle gradients (see block_adjoint)
  for (int pass = 0; pass < win.npass; ++pass) {
    const int pu0 = win.u0 + pass * (win.ths - 1);                       // first flat row of this pass
    const int pth = min(win.ths, win.u0 + win.th - pu0);
    const bool first = pass == 0;
    // the next work item is requested at the start of the last pass and published at its end
    if (tid == 0 && pass == win.npass - 1) next_item = fetch_work_item(work_counter, a);
    // stage dL/dflux rows (flat row k = output row Hh-1-k) into LDS, un-flipped
    // Four floats per lane (one 16-byte load, 4-byte aligned: a wave covers a 256-pixel row segment in one instruction)
    // and four rows per trip, all four loads in flight before the first LDS store.  One float and one row at a time
    // this staging was 48 exposed round trips per wave - 16 us of a 195 us workgroup (tools/timeline.sh), 12 us of it
    // waiting for memory.
    {
        const int64_t gbase = (int64_t)(a.Hh - 1 - pu0) * a.W + win.e0;      // flat row k sits at gbase - k W
        stage_grad_window(G + gbase, a.W, win.tw, pth, gtile, wave, lane, nwaves);    // (round 3: LDS-direct loads, see there)
    }
    if (tid < 2) gtile[a.tile_cap + tid] = 0.0f;
    __syncthreads();

    const float kI = (a.mag * a.k_ext) * a.k_refl;
    float sx = 0.0f, sz = 0.0f;
    if constexpr (!CYL) { sx = pl.wm1 / pl.w; sz = pl.hm1 / pl.h; }
    const float Wf = (float)a.W, Hf = (float)a.Hh;
    const unsigned twm1 = (unsigned)max(win.tw - 1, 0), thm1 = (unsigned)max(pth - 1, 0), uthm1 = (unsigned)max(win.th - 1, 0);
    const int dummy = a.tile_cap;                    // two spare cells holding 0
    // (blocking: a WAVE walks a trip as long as its first lane has a point - a lane beyond the block's end repeats the last
    //  point with its gradients masked out - because the rectangle gradients are handed to fixed owner lanes of the whole
    //  wave, block_adjoint; without blocking the loop ends lane by lane as before)
    for (int pt = p0 + tid; (BLOCKING ? pt - lane : pt) < p1; pt += blockDim.x) {
        const bool lane_live = pt < p1;
        const int p = lane_live ? pt : p1 - 1;
        const float4 o = org[p];
        const float4 n = nrm[p];
        float4 d; float s;
        reflect(inc, n, d, s);
        float numer = 0.0f; CylPoint cp;
        if constexpr (CYL) cp = cyl_point(cy, o); else numer = plane_numer(pl, o);
        float gdx = 0.f, gdy = 0.f, gdz = 0.f, gox = 0.f, goy = 0.f, goz = 0.f;   // (cylinder: go in its local frame)
        float bgx = 0.f, bgy = 0.f, bgz = 0.f;                                     // dL/do through the blocking mask (world)
        unsigned pmask = 0u, wmask = 0u;       // rectangles this point's / this wave's rays can touch
        if constexpr (BLOCKING) {
            if (n_prims > 0) {
                const float il = rsqrtf(fmaxf(d.x * d.x + d.y * d.y + d.z * d.z, 1e-30f));
                pmask = cone_mask(s_tab.prim, s_tab.aux, n_prims, o.x, o.y, o.z, d.x * il, d.y * il, d.z * il, a.cone_cos, a.cone_sin,
                                  a.slab_cull != 0);
                wmask = wave_or_mask(pmask, n_prims);
                wmask |= wide_point(a, h, s_tab, o.x, o.y, o.z, d.x, d.y, d.z);     // (bit 31: see there)
            }
        }
        // One ray.  The forward re-computation is the reference's arithmetic (it decides which cells the ray
        // touched); masks are reduced to "inside this pass's window?", strays and other bands' rays are handled by
        // a wave-uniform cold branch, and masked rays get zero gradient weights instead of an early exit.
        auto trace_one = [&](auto small_angles, const float u, const float e) {
            const Rot m = make_rot_t<decltype(small_angles)::value>(e, u);
            float rx, ry, rz;
            scatter(m, d, rx, ry, rz);
            float keep = 1.0f, trans = 1.0f, g_keep = 0.0f;        // 1 - blocked, exp(-alpha sum), dL/d(1 - blocked)
            unsigned near = 0u;                                    // rectangles whose soft mask this ray entered
            if constexpr (BLOCKING) {
                if (wmask != 0u) {
                    unsigned wm = wmask;
                    WideSum ws = {0.0f, 0};                        // the sigmas of the candidates beyond the tables (a wide heliostat)
                    if (__builtin_expect((wmask >> 31) != 0u, 0)) ws = wide_ray(a, h, s_tab, wm, o.x, o.y, o.z, rx, ry, rz);
                    trans = soft_transmittance(s_tab.prim, wm, pmask, o.x, o.y, o.z, rx, ry, rz, near, 0.0f, 0.0f, ws.sum);
                    near |= (unsigned)ws.near << 31;               // (bit 31 of a wide heliostat's masks stands for all of them)
                    keep = 1.0f - (1.0f - trans);                  // the reference's rounding (blocked = 1 - trans)
                }
            }
            // the mask's adjoint: ray side into this thread's accumulators, rectangle side into the tables
            auto mask_adjoint = [&]() {
                if constexpr (BLOCKING) {
                    // only rays inside some rectangle's mask that still carry light have a gradient through it
                    const bool adj = lane_live && near != 0u && g_keep != 0.0f && trans > 1e-30f;
                    if (wave_any(adj)) {
                        unsigned wm = wmask, nr = adj ? near : 0u;
                        RayGrad x = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        if (__builtin_expect((wmask >> 31) != 0u, 0))
                            x = wide_ray_adjoint(a, h, s_tab, wm, nr, o.x, o.y, o.z, rx, ry, rz, -kBlockAlpha * trans * g_keep);
                        const AdjointOut ao = block_adjoint((LdsPrims)s_tab.prim, prim_sums, wm, nr, o.x, o.y,
                                                            o.z, rx, ry, rz, adj ? -kBlockAlpha * trans * g_keep : 0.0f);
                        prim_sums = ao.sums;
                        RayGrad b = ao.ray;
                        b.ox += x.ox; b.oy += x.oy; b.oz += x.oz; b.rx += x.rx; b.ry += x.ry; b.rz += x.rz;
                        bgx += b.ox; bgy += b.oy; bgz += b.oz;
                        gdx += m.cu * b.rx + m.m10 * b.ry + m.m20 * b.rz;
                        gdy += m.m11 * b.ry + m.m21 * b.rz - m.su * b.rx;
                        gdz += m.ce * b.rz - m.se * b.ry;
                    }
                }
            };
            float be, bu, ah = 0.0f, tt = 0.0f; bool valid; CylHit ch;
            if constexpr (CYL) {
                ch = cyl_hit(cy, cp, rx, ry, rz);
                valid = ch.ok; be = ch.be; bu = ch.bu;
            } else {
                ah = (rx * pl.mx + ry * pl.my) + rz * pl.mz;
                const bool front = ah < 0.0f;
                tt = div_noscale(numer, front ? ah : 1.0f);
                const float hx = o.x + rx * tt, hz = o.z + rz * tt;
                const float be0 = div_const((hx + pl.half_w) - pl.cx, pl.w, pl.inv_w) * pl.wm1;
                bu = div_const((hz + pl.half_h) - pl.cz, pl.h, pl.inv_h) * pl.hm1;
                valid = front && be0 == __builtin_amdgcn_fmed3f(be0, 0.0f, pl.wm1) &&
                        bu == __builtin_amdgcn_fmed3f(bu, 0.0f, pl.hm1);
                be = pl.wm1 - be0;
            }
            const float tbe = truncf(be), tbu = truncf(bu);
            const float cle = (tbe + 1.0f) - be, clu = (tbu + 1.0f) - bu, che = be - tbe, chu = bu - tbu;
            const int ie = (int)tbe, iu = (int)tbu;
            const int le = ie - win.e0, lu = iu - pu0;
            const bool inwin = valid && (unsigned)le < twm1 && (unsigned)lu < thm1;
            const int cell_lo = inwin ? (int)__umul24(lu, win.tw) + le : dummy;
            const int cell_hi = inwin ? cell_lo + win.tw : dummy;
            float g1 = gtile[cell_hi], g2 = gtile[cell_hi + 1], g3 = gtile[cell_lo + 1], g4 = gtile[cell_lo];
            bool use = inwin;
            if (__builtin_expect((__builtin_amdgcn_ballot_w64(valid) & ~__builtin_amdgcn_ballot_w64(inwin)) != 0ull, 0)) {
                const bool on = (tbe + 1.0f < Wf) && (tbu + 1.0f < Hf);
                const bool in_union = (unsigned)le < twm1 && (unsigned)(iu - win.u0) < uthm1;
                if (first && valid && on && !in_union) {              // stray: global gather, once
                    const float* g_hi = G + (int64_t)(a.Hh - 2 - iu) * a.W + ie;
                    const float* g_lo = g_hi + a.W;
                    g1 = g_hi[0]; g2 = g_hi[1]; g3 = g_lo[1]; g4 = g_lo[0];
                    // wait for the gather HERE (see trace_bwd_item_lean)
                    asm volatile("" : "+v"(g1), "+v"(g2), "+v"(g3), "+v"(g4));
                    use = true;
                }
            }
            if constexpr (CYL) {
                if (use) {     // divergent, but a masked ray's intermediates are not finite: no zero-weight trick here
#pragma clang fp contract(fast)
                    const float I = ((ch.I0 * keep) * a.k_ext) * a.k_refl;
                    const float g_abs = cle * (chu * g1 + clu * g4) + che * (chu * g2 + clu * g3);     // dL/dI
                    const float gI = g_abs * (a.k_ext * a.k_refl) * keep;                                // dL/dI0
                    g_keep = g_abs * (a.k_ext * a.k_refl) * ch.I0;
                    const float g_be = (chu * (g2 - g1) + clu * (g3 - g4)) * I;
                    const float g_bu = (cle * (g1 - g4) + che * (g2 - g3)) * I;
                    float lx, ly, lz, grx, gry, grz;
                    cyl_hit_bwd(cy, cp, ch, g_be, g_bu, gI, lx, ly, lz, grx, gry, grz);
                    gox += lx; goy += ly; goz += lz;
                    gdx += m.cu * grx + m.m10 * gry + m.m20 * grz;
                    gdy += m.m11 * gry + m.m21 * grz - m.su * grx;
                    gdz += m.ce * grz - m.se * gry;
                }
                mask_adjoint();
                return;
            }
            // The gradient arithmetic is not tied to the reference's operation order (it is compared with a tolerance,
            // like autograd's own reassociations): intensity in one multiply, and the origin gradient kept as three
            // running sums (hit-point x, hit-point z, plane offset) that are combined once per point.
            const float kIm = use ? kI * keep : 0.0f;
            const float I = -(kIm * ah);                                      // mag (-a) keep k_ext k_refl
            {
#pragma clang fp contract(fast)
                const float gI = cle * (chu * g1 + clu * g4) + che * (chu * g2 + clu * g3);
                if constexpr (BLOCKING) g_keep = use ? gI * (kI * (-ah)) : 0.0f;       // I = mag (-a) keep k_ext k_refl
                const float g_be = (chu * (g2 - g1) + clu * (g3 - g4)) * I;
                const float g_bu = (cle * (g1 - g4) + che * (g2 - g3)) * I;
                const float g_hx = -g_be * sx;
                const float g_hz = g_bu * sz;
                const float g_t = g_hx * rx + g_hz * rz;
                const float g_numer = g_t * __builtin_amdgcn_rcpf(ah);
                const float g_a = -kIm * gI - g_numer * tt;
                const float grx = g_hx * tt + g_a * pl.mx;
                const float gry = g_a * pl.my;
                const float grz = g_hz * tt + g_a * pl.mz;
                gox += g_hx;                     // (planar: gox / goz / goy hold sum g_hx / sum g_hz / sum g_numer)
                goz += g_hz;
                goy += g_numer;
                gdx += m.cu * grx + m.m10 * gry + m.m20 * grz;
                gdy += m.m11 * gry + m.m21 * grz - m.su * grx;
                gdz += m.ce * grz - m.se * gry;
            }
            mask_adjoint();
        };
        // distortion stream prefetched in groups of four samples (see the forward kernel)
        const int lane_off = p * (int)a.sp;
        const int nr = r1 - r0;
        const float* __restrict__ bu_ = a.dist_u + dbase;     // wave-uniform
        const float* __restrict__ be_ = a.dist_e + dbase;
        float cu0, ce0, cu1, ce1, cu2, ce2, cu3, ce3;
        load_dist_row<INTERLEAVED>(bu_, be_, lane_off, cu0, ce0);
        load_dist_row<INTERLEAVED>(bu_ + (int64_t)min(1, nr - 1) * a.sr, be_ + (int64_t)min(1, nr - 1) * a.sr, lane_off, cu1, ce1);
        load_dist_row<INTERLEAVED>(bu_ + (int64_t)min(2, nr - 1) * a.sr, be_ + (int64_t)min(2, nr - 1) * a.sr, lane_off, cu2, ce2);
        load_dist_row<INTERLEAVED>(bu_ + (int64_t)min(3, nr - 1) * a.sr, be_ + (int64_t)min(3, nr - 1) * a.sr, lane_off, cu3, ce3);
        for (int k = 0; k < nr; k += 4) {
            float nu0, ne0, nu1, ne1, nu2, ne2, nu3, ne3;
            const int64_t o4 = (int64_t)min(k + 4, nr - 1) * a.sr, o5 = (int64_t)min(k + 5, nr - 1) * a.sr;
            const int64_t o6 = (int64_t)min(k + 6, nr - 1) * a.sr, o7 = (int64_t)min(k + 7, nr - 1) * a.sr;
            load_dist_row<INTERLEAVED>(bu_ + o4, be_ + o4, lane_off, nu0, ne0);
            load_dist_row<INTERLEAVED>(bu_ + o5, be_ + o5, lane_off, nu1, ne1);
            load_dist_row<INTERLEAVED>(bu_ + o6, be_ + o6, lane_off, nu2, ne2);
            load_dist_row<INTERLEAVED>(bu_ + o7, be_ + o7, lane_off, nu3, ne3);
            const float amax = fmaxf(fmaxf(fmaxf(fabsf(cu0), fabsf(ce0)), fmaxf(fabsf(cu1), fabsf(ce1))),
                                     fmaxf(fmaxf(fabsf(cu2), fabsf(ce2)), fmaxf(fabsf(cu3), fabsf(ce3))));
            if (__builtin_expect(wave_any(!(amax <= kSmallAngle)), 0)) {       // see the forward kernel
                trace_one(std::false_type{}, cu0, ce0);
                if (k + 1 < nr) trace_one(std::false_type{}, cu1, ce1);
                if (k + 2 < nr) trace_one(std::false_type{}, cu2, ce2);
                if (k + 3 < nr) trace_one(std::false_type{}, cu3, ce3);
            } else {
                trace_one(std::true_type{}, cu0, ce0);
                if (k + 1 < nr) trace_one(std::true_type{}, cu1, ce1);
                if (k + 2 < nr) trace_one(std::true_type{}, cu2, ce2);
                if (k + 3 < nr) trace_one(std::true_type{}, cu3, ce3);
            }
            cu0 = nu0; ce0 = ne0; cu1 = nu1; ce1 = ne1; cu2 = nu2; ce2 = ne2; cu3 = nu3; ce3 = ne3;
        }
        const float gdn = gdx * n.x + gdy * n.y + gdz * n.z;
        if constexpr (CYL) {   // local origin = R (o - c)  ->  dL/do = R^T dL/dlocal
            const float wx = gox * cy.r00 + goy * cy.r10 + goz * cy.r20;
            const float wy = gox * cy.r01 + goy * cy.r11 + goz * cy.r21;
            const float wz = gox * cy.r02 + goy * cy.r12 + goz * cy.r22;
            gox = wx; goy = wy; goz = wz;
        } else {               // hit = o + t r, t = (c - o).m / (r.m): dL/do = (g_hx, 0, g_hz) - m * sum g_numer
            const float sn = goy;
            gox = gox - sn * pl.mx; goy = -(sn * pl.my); goz = goz - sn * pl.mz;
        }
        if constexpr (BLOCKING) { gox += bgx; goy += bgy; goz += bgz; }
        const float4 go = make_float4(gox, goy, goz, 0.0f);
        const float4 gn = make_float4(-2.0f * (gdn * inc.x + s * gdx), -2.0f * (gdn * inc.y + s * gdy),
                                      -2.0f * (gdn * inc.z + s * gdz), -2.0f * (gdn * inc.w));
        const int64_t idx = (int64_t)h * a.P + p;
        if (!lane_live) {
            // (a padding lane of the blocking instantiation: nothing to store)
        } else if (first) {
            grad_origins[idx] = go;
            grad_normals[idx] = gn;
        } else {                                      // this thread owns the point in every pass: plain add
            const float4 o0 = grad_origins[idx], n0 = grad_normals[idx];
            grad_origins[idx] = make_float4(o0.x + go.x, o0.y + go.y, o0.z + go.z, 0.0f);
            grad_normals[idx] = make_float4(n0.x + gn.x, n0.y + gn.y, n0.z + gn.z, n0.w + gn.w);
        }
    }
    if (tid == 0 && pass == win.npass - 1) *s_next = (int)(gridDim.x + next_item);
    __syncthreads();   // every wave is done with this band before it is overwritten
  }
    if (win.npass < 1 && tid == 0) *s_next = (int)(gridDim.x + fetch_work_item(work_counter, a));   // (never: npass >= 1)
    if constexpr (BLOCKING) {
        // The waves' rectangle gradients, added in WAVE ORDER (s_tab.grad[k * 12 + j]: lane 2k holds j = 0..5, lane 2k + 1
        // j = 6..11, i.e. entry lane * 6 + c), then this item's slab [Cmax,12] of the caller's scratch: reduce_prim_grads_kernel
        // adds the items' slabs in item order.  No atomics: the rectangle gradients are bit-reproducible.
        for (int w = 0; w < nwaves; ++w) {
            if (wave == w && lane < 2 * n_prims) {
#pragma unroll
                for (int c = 0; c < 6; ++c) s_tab.grad[lane * 6 + c] += (double)prim_sums.v[c];
            }
            __syncthreads();
        }
        float* __restrict__ slab = prim_slabs + ((int64_t)(item.h * a.n_pblocks + item.pblock) * a.n_rchunks + item.rchunk) * slab_rows(a) * 12;
        for (int c = tid; c < n_prims * 12; c += blockDim.x) slab[c] = (float)s_tab.grad[c];
    }
}


// --------------------------------------------------------------------------------------------
// The planar, non-blocking backward item with the lean ray body and the distortion ring of trace_fwd_item_lean:
// the forward is re-computed with the reference's arithmetic up to the pixel coordinates, the window test and the
// LDS address are formed in floating point and clamped (a ray outside the window reads some in-range cell and is
// given zero weight), the gradient arithmetic shares its sub-expressions:
//   A = chu g1 + clu g4,  B = chu g2 + clu g3   ->  dL/dI = cle A + che B,  dL/dbe = (B - A) I
//   dL/dbu = (cle (g1 - g4) + che (g2 - g3)) I
// --------------------------------------------------------------------------------------------

// BLOCKING: the same item with the soft blocking mask recomputed per ray, its factor `keep` in the intensity, and the mask's
// adjoint (block_adjoint: ray side into this thread's sums, rectangle side into the wave's owner-lane registers, see there) -
// the launch of a split call that owns the heliostats WITH candidate rectangles.  A wave then walks whole trips (a lane beyond
// the block's end repeats the last point, masked) and the item ends like trace_bwd_item: wave order, slab.
// CYL: the same item with the cylinder hit (cyl_hit) in place of the plane's and its hand-derived adjoint (cyl_hit_bwd) - no edge
// packing (the partition uses the planes' chief-ray hit) and no zero-weight trick (a masked ray's intermediates are not finite):
// a lane in use takes a branch.
template <bool INTERLEAVED, bool ATOMIC_OUT, bool BLOCKING = false, bool CYL = false>
__device__ __forceinline__ void trace_bwd_item_lean(const TraceArgs& a, const float* __restrict__ grad_flux,
                                                    float4* __restrict__ grad_origins, float4* __restrict__ grad_normals,
                                                    const WorkItem item, unsigned int* __restrict__ work_counter, int* s_next,
                                                    float* __restrict__ prim_slabs = nullptr, const int queue_item = -1)
{
    extern __shared__ __attribute__((aligned(16))) float gtile[];
    __shared__ float s_red[13][16];
    __shared__ Window s_win;
    __shared__ int s_edge[kPackTrips * 16 + 1];      // edge points per (trip, wave), then their exclusive scan; [last] = total
    __shared__ PrimTable<BLOCKING> s_tab;

    const int pblock = item.pblock;
    const int h = item.h;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    if constexpr (ATOMIC_OUT) {                      // chunk slabs: see trace_bwd_item
        grad_origins += (int64_t)item.rchunk * a.H * a.P;
        grad_normals += (int64_t)item.rchunk * a.H * a.P;
    }
    const int t = a.target_idx[h];
    const bool bad_target = !target_in_range(a, t);
    static_assert(!(BLOCKING && CYL), "the lean body serves blocking on planes or cylinders without blocking");
    if (bad_target || (t >= a.T) != CYL || item.r1 <= item.r0 || other_launch_owns(a, h)) {
        if (bad_target) {
            zero_block_gradients(a, item, grad_origins, grad_normals);
            if constexpr (BLOCKING) {                // ... and no rectangle gradients from this item either
                float* __restrict__ slab = prim_slabs + ((int64_t)(item.h * a.n_pblocks + item.pblock) * a.n_rchunks + item.rchunk) * slab_rows(a) * 12;
                for (int c = tid; c < slab_rows(a) * 12; c += blockDim.x) slab[c] = 0.0f;
            }
        }
        if (tid == 0) *s_next = (int)(gridDim.x + fetch_work_item(work_counter, a));
        return;
    }
    Plane pl = {}; Cyl cy = {};
    if constexpr (CYL) cy = load_cyl(a.cyl_centers, a.cyl_normals, a.cyl_axes, a.cyl_radii, a.cyl_heights, a.cyl_opening, t - a.T, a.W, a.Hh,
                                     a.mag, a.k_ext, a.k_refl);
    else pl = load_plane(a.centers, a.pnormals, a.dims, t, a.W, a.Hh, a.mag, a.k_ext, a.k_refl);
    const float* __restrict__ G = grad_flux + (int64_t)(a.mode == 0 ? h : t) * a.Hh * a.W;
    const float4 inc = a.incident[h];
    int p0, p1;
    block_range(a, pblock, p0, p1, item.tail);
    const int r0 = item.r0;
    const int r1 = item.r1;
    const float4* __restrict__ org = a.origins + (int64_t)h * a.P;
    const float4* __restrict__ nrm = a.normals + (int64_t)h * a.P;
    const int64_t dbase = (int64_t)h * a.sh + (int64_t)r0 * a.sr;

    const int n_prims = load_prims<BLOCKING>(a, h, s_tab);
    if (a.win_table != nullptr && queue_item >= 0) {       // the launch came with its items' windows (window_table_kernel)
        if (tid == 0) s_win = static_cast<const Window*>(a.win_table)[queue_item];
        __syncthreads();
    } else
    compute_window<INTERLEAVED, CYL>(a, pl, cy, inc, org, nrm, p0, p1, dbase, s_red, &s_win);
    const Window win = s_win;
    unsigned next_item = 0u;
    [[maybe_unused]] PrimSums prim_sums = {{0.f, 0.f, 0.f, 0.f, 0.f, 0.f}};   // (blocking) this wave's rectangle gradients
    const float kI = (a.mag * a.k_ext) * a.k_refl;
    const float sx = pl.wm1 / pl.w, sz = pl.hm1 / pl.h;
    const float Wf = (float)a.W, Hf = (float)a.Hh;
    // ---- edge points to the back of the block --------------------------------------------------------------------
    // A ray outside the window gathers its four dL/dflux values from global memory and its wave waits for them - and,
    // loads retiring in order, for its whole distortion ring.  The strays come from the points whose chief ray lands near
    // the window's border (or beyond it): with the points in mirror order every wave holds a few of them and stalls in
    // almost half of its ray steps.  A stable partition - interior points first, edge points last, both in mirror order -
    // gathers them in the block's last waves, so that the other waves run without a stall (and the interior waves' stream
    // stays coalesced: they skip a point here and there).  perm[j] = index within the block of the point in slot j.
    unsigned short* perm = reinterpret_cast<unsigned short*>(gtile + a.tile_cap + 2);
    const int n_pts = p1 - p0;
    const bool packed = !CYL && a.pack_edge != 0 && win.npass == 1 && n_pts <= kPackPoints && n_pts <= kPackTrips * (int)blockDim.x &&
                        win.tw >= 2 && win.th >= 2;
    if (packed) pack_edge_points(a, pl, inc, org, nrm, p0, n_pts, win, a.pack_edge, s_edge, perm);
         // (diagnostic build: end of the edge partition; slot 5 is read out of order by tools/timeline_report.py)
    [[maybe_unused]] const unsigned wm1_bits = f32_bits(pl.wm1), hm1_bits = f32_bits(pl.hm1);
    const float lds_base = (float)(unsigned)(size_t)(lds_f32*)gtile;
    const float e0f = (float)win.e0, tw4f = (float)(4 * win.tw), u0f = (float)win.u0;
    const unsigned twm2_bits = f32_bits((float)(win.tw - 2)), uthm2_bits = f32_bits((float)(win.th - 2));
  for (int pass = 0; pass < win.npass; ++pass) {
    const int pu0 = win.u0 + pass * (win.ths - 1);
    const int pth = min(win.ths, win.u0 + win.th - pu0);
    const bool first = pass == 0;
    if (tid == 0 && pass == win.npass - 1) next_item = fetch_work_item(work_counter, a);
    {   // stage dL/dflux rows (flat row k = output row Hh-1-k) into LDS, un-flipped: stage_grad_window
        // (until late in round 3 through registers, four rows per wave in flight - the #else branch: 6-8 us per item,
        //  tools/timeline.sh; eight rows measured 3.65 against 3.53 ms for the kernel, twelve 4.4 - the batch's registers are
        //  allocated on top of the ray loop's)
        const int64_t gbase = (int64_t)(a.Hh - 1 - pu0) * a.W + win.e0;
        stage_grad_window(G + gbase, a.W, win.tw, pth, gtile, wave, lane, nwaves);
    }
    // an empty window (no chief ray of the block reaches the target; or a degenerate one of a single row / column) holds no
    // ray: every valid ray is then a stray.  Its masked rays still READ two cells - clamped onto cells 0 and 1, see addr_hi_f -
    // and multiply them by zero weights, so those cells must hold finite values: nothing was staged there, and whatever
    // the CU's previous workgroup left in its LDS may look like a NaN (0 x NaN = NaN in every gradient of the block: the
    // rare, box-dependent failure of tests/test_gpu_parity.py::test_random_scenes_split_calls[8-12-77-17-*] in round 3).
    const unsigned long long win_ok = (win.tw >= 2 && pth >= 2) ? ~0ull : 0ull;
    if (win_ok == 0ull && tid < 2) gtile[tid] = 0.0f;
    __syncthreads();
    if (first) { }      // (diagnostic build: window + edge partition | staging | trace)

    const float pu0f = (float)pu0;
    const unsigned thm2_bits = f32_bits((float)(pth - 2));
    const float addr_hi_f = win_ok != 0ull ? lds_base + (float)(4 * (win.tw * (pth - 2) + win.tw - 2)) : lds_base;
    const unsigned tw4 = win_ok != 0ull ? 4u * (unsigned)win.tw : 0u;      // (degenerate window: both rows of a read are cells 0, 1)
    // (blocking: a WAVE walks a trip as long as its first lane has a point; lanes beyond the end repeat the last slot, masked)
    for (int jt = tid; (BLOCKING ? jt - lane : jt) < n_pts; jt += blockDim.x) {
        const bool lane_live = jt < n_pts;
        const int j = lane_live ? jt : n_pts - 1;
        const int p = p0 + (packed ? (int)perm[j] : j);
        const float4 o = org[p];
        const float4 n = nrm[p];
        float4 d; float s;
        reflect(inc, n, d, s);
        [[maybe_unused]] float numer = 0.0f;
        [[maybe_unused]] CylPoint cp = {};
        if constexpr (CYL) cp = cyl_point(cy, o); else numer = plane_numer(pl, o);
        float gdx = 0.f, gdy = 0.f, gdz = 0.f, gox = 0.f, goy = 0.f, goz = 0.f;   // gox / goz / goy: sums of g_hx / g_hz / g_numer (cylinders: of dL/d(local origin))
        [[maybe_unused]] float bgx = 0.f, bgy = 0.f, bgz = 0.f;                   // dL/do through the blocking mask (world)
        [[maybe_unused]] unsigned pmask = 0u, wmask = 0u;                         // rectangles this point's / this wave's rays can touch
        [[maybe_unused]] PointSums point_sums = {}, point_sums2 = {};             // this lane's gradient of the wave mask's first (second) rectangle (see block_adjoint_body)
        [[maybe_unused]] bool point_touched = false;                              // (wave-uniform)
        if constexpr (BLOCKING) {
            if (n_prims > 0) {
                const float il = rsqrtf(fmaxf(d.x * d.x + d.y * d.y + d.z * d.z, 1e-30f));
                pmask = cone_mask(s_tab.prim, s_tab.aux, n_prims, o.x, o.y, o.z, d.x * il, d.y * il, d.z * il, a.cone_cos, a.cone_sin,
                                  a.slab_cull != 0);
                wmask = wave_or_mask(pmask, n_prims);
            }
        }
        [[maybe_unused]] float bnum0 = 0.0f, bnum1 = 0.0f;
        if constexpr (BLOCKING) {
            if (wmask != 0u) {
                bnum0 = soft_plane_num(s_tab.prim[__builtin_ctz(wmask)], o.x, o.y, o.z);
                const unsigned rest = wmask & (wmask - 1u);
                if (rest != 0u) bnum1 = soft_plane_num(s_tab.prim[__builtin_ctz(rest)], o.x, o.y, o.z);
            }
        }
        if constexpr (BLOCKING) {          // (after the two numerators: they belong to tabled rectangles)
            if (n_prims > 0) wmask |= wide_point(a, h, s_tab, o.x, o.y, o.z, d.x, d.y, d.z);     // (bit 31: see there)
        }
        auto trace_one = [&](const float u, const float e, const unsigned long long live) {
            // sun-shape angles are milliradians: the Taylor kernels serve every lane almost always; only the rotation's
            // sines and cosines sit behind the (wave-uniform) branch - two copies of the whole ray body made the compiler
            // merge their tails and spill what crossed the join
            // (the Taylor values are computed first and REPLACED on the rare path: as two arms of a branch the common arm ended
            //  in register copies; |u| + |e| bounds the larger angle with one instruction instead of three)
            Rot m = make_rot_t<true>(e, u);
            if (__builtin_expect(wave_any(!(fabsf(u) + fabsf(e) <= kSmallAngle)), 0)) m = make_rot(e, u);
            float rx, ry, rz;
            scatter(m, d, rx, ry, rz);
            [[maybe_unused]] float ah = 0.0f, y = 0.0f, tt = 0.0f;          // (planes) r.m, ~ 1 / (r.m), the path length
            float be, bu;
            unsigned long long m_front, m_valid;
            [[maybe_unused]] CylHit ch = {};
            if constexpr (CYL) {
                ch = cyl_hit(cy, cp, rx, ry, rz);
                be = ch.be; bu = ch.bu;
                m_valid = ballot64(ch.ok) & live;
                m_front = m_valid;
            } else {
            ah = (rx * pl.mx + ry * pl.my) + rz * pl.mz;
            m_front = ballot64(ah < 0.0f) & live;
            // (the denominator is made safe, unlike in the forward: a masked ray's terms are multiplied by zero
            //  below and must therefore be finite)
            tt = div_noscale_rcp(numer, select_mask(m_front, ah, 1.0f), y);
            const float hx = o.x + rx * tt, hz = o.z + rz * tt;
            const float be0 = div_const((hx + pl.half_w) - pl.cx, pl.w, pl.inv_w) * pl.wm1;
            bu = div_const((hz + pl.half_h) - pl.cz, pl.h, pl.inv_h) * pl.hm1;
            be = pl.wm1 - be0;
            m_valid = m_front & ballot64(f32_bits(be0) <= wm1_bits) & ballot64(f32_bits(bu) <= hm1_bits);
            }
            const float tbe = truncf(be), tbu = truncf(bu);
            const float che = be - tbe, chu = bu - tbu;
            const float cle = 1.0f - che, clu = 1.0f - chu;
            const float lef = tbe - e0f, luf = tbu - pu0f;
            const unsigned long long m_in = m_front & win_ok & ballot64(f32_bits(lef) <= twm2_bits) & ballot64(f32_bits(luf) <= thm2_bits);
            const float af = __builtin_amdgcn_fmed3f(fmaf(luf, tw4f, fmaf(lef, 4.0f, lds_base)), lds_base, addr_hi_f);
            const unsigned addr_lo = (unsigned)af;
            const lds_f32* lo = (const lds_f32*)(size_t)addr_lo;
            const lds_f32* up = (const lds_f32*)(size_t)(addr_lo + tw4);
            float g1 = up[0], g2 = up[1], g3 = lo[1], g4 = lo[0];
            unsigned long long m_use = m_in;
            if (__builtin_expect((m_valid & ~m_in) != 0ull, 0)) {
                // valid, outside this pass's window: a stray of the union window gathers from global memory, once
                const bool valid = (m_valid >> lane) & 1ull, inwin = (m_in >> lane) & 1ull;
                const bool on = (tbe + 1.0f < Wf) && (tbu + 1.0f < Hf);
                const bool in_union = win_ok != 0ull && f32_bits(lef) <= twm2_bits && f32_bits(tbu - u0f) <= uthm2_bits;
                const bool stray = first && valid && !inwin && on && !in_union;
                if (stray) {
                    const float* g_hi = G + (int64_t)(a.Hh - 2 - (int)tbu) * a.W + (int)tbe;
                    const float* g_lo = g_hi + a.W;
                    g1 = g_hi[0]; g2 = g_hi[1]; g3 = g_lo[1]; g4 = g_lo[0];
                    // The wait for these four loads has to stand INSIDE the branch: the (empty) statement reads them, so the
                    // compiler puts its s_waitcnt vmcnt(0) here.  Without it the wait lands after the join, before the first
                    // use of g1 - on the path of EVERY ray, where vmcnt(0) also waits for the whole distortion ring (loads
                    // retire in order): the ring was drained at every step and the kernel ran 20 % slower.
                    asm volatile("" : "+v"(g1), "+v"(g2), "+v"(g3), "+v"(g4));
                }
                m_use |= ballot64(stray);
            }
            if constexpr (CYL) {
                if ((m_use >> lane) & 1ull) {     // divergent: a masked ray's intermediates are not finite, so no zero-weight trick here
#pragma clang fp contract(fast)
                    const float I = (ch.I0 * a.k_ext) * a.k_refl;
                    const float g_abs = cle * (chu * g1 + clu * g4) + che * (chu * g2 + clu * g3);     // dL/dI
                    const float gI0 = g_abs * (a.k_ext * a.k_refl);                                      // dL/dI0
                    const float g_be = (chu * (g2 - g1) + clu * (g3 - g4)) * I;
                    const float g_bu = (cle * (g1 - g4) + che * (g2 - g3)) * I;
                    float lx, ly, lz, grx, gry, grz;
                    cyl_hit_bwd(cy, cp, ch, g_be, g_bu, gI0, lx, ly, lz, grx, gry, grz);
                    gox += lx; goy += ly; goz += lz;
                    gdx += m.cu * grx + m.m10 * gry + m.m20 * grz;
                    gdy += m.m11 * gry + m.m21 * grz - m.su * grx;
                    gdz += m.ce * grz - m.se * gry;
                }
            } else {
#pragma clang fp contract(fast)
                float kIm = select_or_zero(m_use, kI);
                [[maybe_unused]] float trans = 1.0f, g_keep_scale = 0.0f;
                [[maybe_unused]] unsigned near = 0u;
                if constexpr (BLOCKING) {
                    if (wmask != 0u) {
                        unsigned wm = wmask;
                        WideSum ws = {0.0f, 0};                               // the sigmas of the candidates beyond the tables (a wide heliostat)
                        if (__builtin_expect((wmask >> 31) != 0u, 0)) ws = wide_ray(a, h, s_tab, wm, o.x, o.y, o.z, rx, ry, rz);
                        trans = soft_transmittance<true>(s_tab.prim, wm, pmask, o.x, o.y, o.z, rx, ry, rz, near, bnum0, bnum1, ws.sum);
                        near |= (unsigned)ws.near << 31;                      // (bit 31 of a wide heliostat's masks stands for all of them)
                        g_keep_scale = kIm * (-ah);                           // dI / d(keep) = mag (-a) k_ext k_refl for a ray in use
                        kIm *= 1.0f - (1.0f - trans);                         // keep, with the reference's rounding (blocked = 1 - trans)
                    }
                }
                const float I = -(kIm * ah);                                  // mag (-a) keep k_ext k_refl (one rounding: gradient side)
                const float A = chu * g1 + clu * g4, B = chu * g2 + clu * g3;
                const float gI = cle * A + che * B;
                const float g_be = (B - A) * I;
                const float g_bu = (cle * (g1 - g4) + che * (g2 - g3)) * I;
                const float g_hx = -g_be * sx;
                const float g_hz = g_bu * sz;
                const float g_numer = (g_hx * rx + g_hz * rz) * y;
                const float g_a = -kIm * gI - g_numer * tt;
                const float grx = g_hx * tt + g_a * pl.mx;
                const float gry = g_a * pl.my;
                const float grz = g_hz * tt + g_a * pl.mz;
                gox += g_hx;
                goz += g_hz;
                goy += g_numer;
                gdx += m.cu * grx + m.m10 * gry + m.m20 * grz;
                gdy += m.m11 * gry + m.m21 * grz - m.su * grx;
                gdz += m.ce * grz - m.se * gry;
                if constexpr (BLOCKING) {
                    // the mask's adjoint: only rays inside some rectangle's soft edge that still carry light have a gradient
                    // through it (dL/d(keep) = dL/dI x dI/d(keep); keep = trans up to its rounding)
                    const float g_keep = gI * g_keep_scale;
                    const bool adj = lane_live && near != 0u && g_keep != 0.0f && trans > 1e-30f;
                    if (wave_any(adj)) {
                        unsigned wm = wmask, nr = adj ? near : 0u;
                        RayGrad x = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        if (__builtin_expect((wmask >> 31) != 0u, 0))
                            x = wide_ray_adjoint(a, h, s_tab, wm, nr, o.x, o.y, o.z, rx, ry, rz, -kBlockAlpha * trans * g_keep);
                        AdjointOut ao = block_adjoint_body<true>((LdsPrims)s_tab.prim, prim_sums, point_sums, point_sums2, wm, nr, o.x, o.y,
                                                                 o.z, rx, ry, rz, adj ? -kBlockAlpha * trans * g_keep : 0.0f, bnum0, bnum1);
                        point_touched = true;
                        prim_sums = ao.sums;
                        ao.ray.ox += x.ox; ao.ray.oy += x.oy; ao.ray.oz += x.oz; ao.ray.rx += x.rx; ao.ray.ry += x.ry; ao.ray.rz += x.rz;
                        bgx += ao.ray.ox; bgy += ao.ray.oy; bgz += ao.ray.oz;
                        gdx += m.cu * ao.ray.rx + m.m10 * ao.ray.ry + m.m20 * ao.ray.rz;
                        gdy += m.m11 * ao.ray.ry + m.m21 * ao.ray.rz - m.su * ao.ray.rx;
                        gdz += m.ce * ao.ray.rz - m.se * ao.ray.ry;
                    }
                }
            }
            // The sums are pinned here: volatile statements keep their order, so this ray's gradient arithmetic cannot
            // sink below the next ring step's head (it did, and everything that was alive across it went to scratch).
            asm volatile("" : "+v"(gdx), "+v"(gdy), "+v"(gdz), "+v"(gox), "+v"(goy), "+v"(goz));
        };
        const int lane_off = p * (int)a.sp;
        const int nr = r1 - r0;
        const float* __restrict__ bu_ = a.dist_u + dbase;     // walks from sample to sample (see trace_fwd_item_lean)
        const float* __restrict__ be_ = a.dist_e + dbase;
        int next_r = 0;
        auto request = [&](int, float& u, float& e) {
            load_dist_stream<INTERLEAVED>(bu_, be_, lane_off, u, e);
            const int64_t step = next_r + 1 < nr ? a.sr : 0;
            bu_ += step; be_ += step; ++next_r;
        };
        if (nr >= 8) {                                // the distortion ring of trace_fwd_item_lean
            [[maybe_unused]] float su0, se0, su1, se1, su2, se2, su3, se3, su4, se4, su5, se5, su6, se6, su7, se7;
            request(0, su0, se0); request(1, su1, se1);
#define ART_RING_STEP(j)                                                            \
            {                                                                       \
                float u, e;                                                         \
                asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=v"(u), "=v"(e) : "v"(su##j), "v"(se##j) : "memory"); \
                request(k + j + 2, su##j, se##j);                      \
                trace_one(u, e, k + j < nr ? ~0ull : 0ull);                          \
                /* a ray's arithmetic stays inside its step: hoisting the next step's head above it made the */ \
                /* compiler spill the ray's live values around the hoisted code */   \
                __builtin_amdgcn_sched_barrier(0);                                  \
            }
            for (int k = 0; k < nr; k += 2) {
                ART_RING_STEP(0) ART_RING_STEP(1)
            }
#undef ART_RING_STEP
        } else {
            for (int r = 0; r < nr; ++r) {
                float u, e;
                request(r, u, e);
                trace_one(u, e, ~0ull);
            }
        }
        if constexpr (BLOCKING) {
            // the lanes' sums for the wave mask's first rectangle -> its two owner lanes: twelve reductions on the DPP network,
            // once per point (a wave walks whole trips: every lane is here)
            if (point_touched) {
                const int k0 = __builtin_ctz(wmask);
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    const float a_lo = wave_reduce<kSum>(point_sums.v[c]), a_hi = wave_reduce<kSum>(point_sums.v[6 + c]);
                    prim_sums.v[c] += lane == 2 * k0 ? a_lo : (lane == 2 * k0 + 1 ? a_hi : 0.0f);
                }
                const unsigned rest = wmask & (wmask - 1u);
                if (rest != 0u) {
                    const int k1 = __builtin_ctz(rest);
#pragma unroll
                    for (int c = 0; c < 6; ++c) {
                        const float a_lo = wave_reduce<kSum>(point_sums2.v[c]), a_hi = wave_reduce<kSum>(point_sums2.v[6 + c]);
                        prim_sums.v[c] += lane == 2 * k1 ? a_lo : (lane == 2 * k1 + 1 ? a_hi : 0.0f);
                    }
                }
            }
        }
        const float gdn = gdx * n.x + gdy * n.y + gdz * n.z;
        float4 go;
        if constexpr (CYL) {   // local origin = R (o - c)  ->  dL/do = R^T dL/dlocal
            go = make_float4(gox * cy.r00 + goy * cy.r10 + goz * cy.r20, gox * cy.r01 + goy * cy.r11 + goz * cy.r21,
                             gox * cy.r02 + goy * cy.r12 + goz * cy.r22, 0.0f);
        } else {
            // hit = o + t r, t = (c - o).m / (r.m): dL/do = (g_hx, 0, g_hz) - m * sum g_numer
            const float sn = goy;
            go = make_float4(gox - sn * pl.mx, -(sn * pl.my), goz - sn * pl.mz, 0.0f);
        }
        if constexpr (BLOCKING) { go.x += bgx; go.y += bgy; go.z += bgz; }
        const float4 gn = make_float4(-2.0f * (gdn * inc.x + s * gdx), -2.0f * (gdn * inc.y + s * gdy),
                                      -2.0f * (gdn * inc.z + s * gdz), -2.0f * (gdn * inc.w));
        const int64_t idx = (int64_t)h * a.P + p;
        if (!lane_live) {
            // (a padding lane of the blocking instantiation: nothing to store)
        } else if (first) {
            grad_origins[idx] = go;
            grad_normals[idx] = gn;
        } else {
            const float4 o0 = grad_origins[idx], n0 = grad_normals[idx];
            grad_origins[idx] = make_float4(o0.x + go.x, o0.y + go.y, o0.z + go.z, 0.0f);
            grad_normals[idx] = make_float4(n0.x + gn.x, n0.y + gn.y, n0.z + gn.z, n0.w + gn.w);
        }
    }
    if (tid == 0 && pass == win.npass - 1) *s_next = (int)(gridDim.x + next_item);
    __syncthreads();
  }
    if constexpr (BLOCKING) {          // the waves' rectangle gradients in wave order, then this item's slab (see trace_bwd_item)
        for (int w = 0; w < nwaves; ++w) {
            if (wave == w && lane < 2 * n_prims) {
#pragma unroll
                for (int c = 0; c < 6; ++c) s_tab.grad[lane * 6 + c] += (double)prim_sums.v[c];
            }
            __syncthreads();
        }
        float* __restrict__ slab = prim_slabs + ((int64_t)(item.h * a.n_pblocks + item.pblock) * a.n_rchunks + item.rchunk) * slab_rows(a) * 12;
        for (int c = tid; c < n_prims * 12; c += blockDim.x) slab[c] = (float)s_tab.grad[c];
    }
}

// Persistent workgroups over the work-item queue, like the forward kernel.
// The lean instantiation is compiled for 1024-thread workgroups (118 registers since the gradient window is staged with LDS-direct
// loads - the staging batch's sixteen registers per lane were the kernel's peak, 155) and launched with 768 or 1024 threads:
// items of 2048 points and more (a facet of the metric field: 2500 = 1024 + 1024 + 452 instead of 768 + 768 + 768 + 196)
// take 1024 - 3.61 -> 3.38 ms on the metric field, same box, interleaved -, smaller items
// (one of eight ranks' share: 1250 points = 768 + 482) stay at 768 (0.512 against 0.529 ms).  The block GEOMETRY is always
// computed for 768 threads, so the items - and with them every bit of the results - do not depend on the choice.
constexpr int kLeanBwdThreads = 1024;       // launch bound
constexpr int kLeanBwdGeometryThreads = 768;
constexpr int kLeanBwdWidePoints = 2048;
static int lean_bwd_threads(int p_block)
{
    const int forced = debug_env_int("ARTIST_HIP_BWD_THREADS", 0);       // (tests, A/B: 768 or 1024)
    if (forced == 768 || forced == 1024) return std::min(forced, kLeanBwdThreads);
    return std::min(p_block >= kLeanBwdWidePoints ? 1024 : 768, kLeanBwdThreads);
}
// Points per item: every item stages a 158 KB gradient window and runs a window phase (~16 us), so fewer, larger blocks pay
// as long as the chip still gets its rounds (window_geometry adds blocks when it does not).  Same-box sweep on the metric
// field (10 000 points per heliostat): 7 blocks of 1429 points 4.16 ms, 5 x 2000 3.92, 4 x 2500 3.85-3.94, 3 x 3334 4.09-4.35,
// 2 x 5000 4.08.  The trips of a block need not be full: 2500 points = 768 + 768 + 768 + 196 threads beat four equal trips of
// 640 threads (4.28 ms) - a trip costs what its active waves issue, not a fixed time.
constexpr int kLeanBwdPoints = 2560;
// (768-thread workgroups for the cylinder adjoint and the planar blocking instantiation: they keep ~60 more values alive per
//  ray than the plain body, and three waves per SIMD with a few spills beat two without - same-box, tools/cylinder_bench.py:
//  512 -> 768 threads 18.6 -> 17.0 ms forward + backward, 1024 threads 17.9; tools/blocking_bench.py exact mode 30.5 -> 29.0)
constexpr int kCylBwdThreads = 768, kBlockingBwdThreads = 768;
constexpr int kLeanBlockBwdThreads = 768;     // the lean backward item with the mask and its inlined adjoint
// static LDS the rectangle tables add to the lean backward kernel (PrimTable<true>: rectangles, cull data, fp64 gradient sums)
constexpr int kLeanBlockBwdStatic = (int)sizeof(PrimTable<true>);
template <bool INTERLEAVED, bool ATOMIC_OUT, bool CYL, bool BLOCKING, bool LEAN = false>
__global__ __launch_bounds__(CYL ? kCylBwdThreads : (BLOCKING ? (LEAN ? kLeanBlockBwdThreads : kBlockingBwdThreads) : (LEAN ? kLeanBwdThreads : 1024))) void trace_bwd_lds_kernel(TraceArgs a, const float* __restrict__ grad_flux,
                                                             float4* __restrict__ grad_origins,
                                                             float4* __restrict__ grad_normals,
                                                             float* __restrict__ prim_slabs,
                                                             unsigned int* __restrict__ work_counter)
{
    __shared__ int s_next;
    const int n_items = work_item_count(a);
    int item = blockIdx.x;
    if constexpr ((CYL && !LEAN && !kCylPersistentBwd) || (!CYL && BLOCKING && !LEAN && !kBlockingPersistentBwd)) {     // one item per workgroup, as in the forward kernel
        if (item >= n_items) return;
        if constexpr (CYL && !BLOCKING) {
            if (a.split == 3) {                      // the heliostats that aim at a cylinder first (see art_trace_fwd)
                __shared__ int s_scan[18];
                const int per = a.n_pblocks * a.n_rchunks;
                const int hk = kth_blocked_heliostat<true>(a, item / per, s_scan);
                if (hk < 0) return;
                item = hk * per + item % per;
            }
        }
        trace_bwd_item<INTERLEAVED, ATOMIC_OUT, CYL, BLOCKING>(a, grad_flux, grad_origins, grad_normals, prim_slabs,
                                                               decode_work_item(a, item), nullptr, &s_next);
        return;
    }
    const bool reverse = a.reverse_bwd != 0;
    while (item < n_items) {
        if constexpr (LEAN)
            trace_bwd_item_lean<INTERLEAVED, ATOMIC_OUT, BLOCKING, CYL>(a, grad_flux, grad_origins, grad_normals,
                                                                   decode_work_item(a, item, reverse), work_counter, &s_next, prim_slabs, item);
        else
            trace_bwd_item<INTERLEAVED, ATOMIC_OUT, CYL, BLOCKING>(a, grad_flux, grad_origins, grad_normals, prim_slabs,
                                                                   decode_work_item(a, item, reverse), work_counter, &s_next);
        __syncthreads();
        item = __builtin_amdgcn_readfirstlane(s_next);       // wave-uniform by construction: say so (the item's fields then live in SGPRs)
        __syncthreads();
    }
}

// grads[i] = sum_c slabs[c][i] in chunk order (two [n_chunks,n] float4 slab sets: origins, normals).
__global__ __launch_bounds__(256) void reduce_chunks_kernel(const float4* __restrict__ slabs_o, const float4* __restrict__ slabs_n,
                                                            int n_chunks, int64_t n, float4* __restrict__ out_o,
                                                            float4* __restrict__ out_n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 so = slabs_o[i], sn = slabs_n[i];
    for (int c = 1; c < n_chunks; ++c) {
        const float4 a = slabs_o[(int64_t)c * n + i], b = slabs_n[(int64_t)c * n + i];
        so.x += a.x; so.y += a.y; so.z += a.z; so.w += a.w;
        sn.x += b.x; sn.y += b.y; sn.z += b.z; sn.w += b.w;
    }
    out_o[i] = so; out_n[i] = sn;
}

// Rectangle gradients: the items' slabs -> the three gradient tables, every element written (zeros where nothing flows: a
// rectangle nobody's rays met, corners 1-3, the w components).  One workgroup per rectangle k; thread t looks at the
// heliostats h = t, t + 256, ... (their candidate lists, <= Cmax entries each), adds the slabs of h's items in item order,
// and the 256 partial sums are added in thread order by a fixed tree: bit-reproducible.  A heliostat whose list is empty
// (the lean launch of a split call traced it) or overflowed beyond Cmax (truncated; reported) contributes its first
// min(count, Cmax) candidates like the kernels that filled the slabs.
// (a wide heliostat - more than kMaxCand candidates -: the sums of its candidates c >= kWideFirst are its fp64 row of `wide`)
__global__ __launch_bounds__(256) void wide_grad_zero_kernel(const int32_t* __restrict__ cand_count, int Cmax, double* __restrict__ wide)
{
    const int h = blockIdx.x;
    const int listed = min(cand_count[h], Cmax);
    if (listed <= kMaxCand) return;
    double* __restrict__ row = wide + (int64_t)h * (Cmax - kWideFirst) * 12;
    for (int i = threadIdx.x; i < (listed - kWideFirst) * 12; i += blockDim.x) row[i] = 0.0;
}
__global__ __launch_bounds__(256) void reduce_prim_grads_kernel(const float* __restrict__ slabs, const double* __restrict__ wide,
                                                                const int32_t* __restrict__ cand,
                                                                const int32_t* __restrict__ cand_count, int H, int Cmax,
                                                                int items_per_heliostat, float* __restrict__ g_corners,
                                                                float* __restrict__ g_spans, float* __restrict__ g_pnormals)
{
    __shared__ float s_part[256][13];              // (+1: no bank conflicts in the tree)
    const int k = blockIdx.x, tid = threadIdx.x;
    const int rows = min(Cmax, kMaxCand);          // of an item's slab
    float acc[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) acc[j] = 0.0f;
    for (int h = tid; h < H; h += 256) {
        const int nc = min(cand_count[h], Cmax);
        for (int c = 0; c < nc; ++c) {
            if (cand[(int64_t)h * Cmax + c] != k) continue;
            if (nc > kMaxCand && c >= kWideFirst) {
                const double* __restrict__ v = wide + ((int64_t)h * (Cmax - kWideFirst) + (c - kWideFirst)) * 12;
#pragma unroll
                for (int j = 0; j < 12; ++j) acc[j] += (float)v[j];
                continue;
            }
            for (int it = 0; it < items_per_heliostat; ++it) {
                const float* __restrict__ v = slabs + (((int64_t)h * items_per_heliostat + it) * rows + c) * 12;
#pragma unroll
                for (int j = 0; j < 12; ++j) acc[j] += v[j];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 12; ++j) s_part[tid][j] = acc[j];
    __syncthreads();
    for (int half = 128; half > 0; half >>= 1) {
        if (tid < half)
#pragma unroll
            for (int j = 0; j < 12; ++j) s_part[tid][j] += s_part[tid + half][j];
        __syncthreads();
    }
    if (tid < 16) g_corners[16 * (int64_t)k + tid] = tid < 3 ? s_part[0][tid] : 0.0f;                       // corner 0
    if (tid < 8) g_spans[8 * (int64_t)k + tid] = (tid & 3) < 3 ? s_part[0][3 + 3 * (tid >> 2) + (tid & 3)] : 0.0f;   // span u, span v
    if (tid < 4) g_pnormals[4 * (int64_t)k + tid] = tid < 3 ? s_part[0][9 + tid] : 0.0f;
}

// pixel accumulators -> fp32 bitmap (one rounding per pixel), and the accumulators are left zero for the next call.
// sign_unit = sign(mag k_ext k_refl) 2^(ex_g - 28).  Two pixels per thread: 16-byte loads, 8-byte stores.
__global__ __launch_bounds__(256) void accum_to_flux_kernel(unsigned long long* __restrict__ accum, float* __restrict__ flux,
                                                            int64_t npix, float sign_unit)
{
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i + 1 < npix) {
        const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(accum + i);
        *reinterpret_cast<float2*>(flux + i) = make_float2((float)v.x * sign_unit, (float)v.y * sign_unit);
        if ((v.x | v.y) != 0ull) *reinterpret_cast<ulonglong2*>(accum + i) = make_ulonglong2(0ull, 0ull);
    } else if (i < npix) {
        const unsigned long long v = accum[i];
        flux[i] = (float)v * sign_unit;
        if (v != 0ull) accum[i] = 0ull;
    }
}

// The same conversion, one workgroup per (quarter of a bitmap's rows, bitmap), that also leaves the bitmap's centre-of-mass sums
// behind - moments[map][part][sum f, sum x f, sum y f] in fp64, x / y in normalised coordinates - formed exactly as
// flux_com_parts_kernel forms them from the finished bitmap (flux_moments.hpp: same thread <-> pixel mapping, same order, same
// tree): the crop + loss pass that follows in a reconstruction epoch then starts from them instead of reading every bitmap
// once more (0.08 ms per 1000 bitmaps, 12 us at 125).
__global__ __launch_bounds__(kMomentsBlock) void accum_to_flux_moments_kernel(unsigned long long* __restrict__ accum, float* __restrict__ flux,
                                                                             int Hh, int W, float sign_unit, double* __restrict__ moments)
{
    __shared__ double s_red[16 * 3];
    const int v = blockIdx.x;
    const int64_t map = blockIdx.y;
    unsigned long long* __restrict__ acc = accum + map * Hh * W;
    float* __restrict__ out = flux + map * Hh * W;
    auto load4 = [&](int64_t k) {
        const ulonglong2 lo = *reinterpret_cast<const ulonglong2*>(acc + 4 * k), hi = *reinterpret_cast<const ulonglong2*>(acc + 4 * k + 2);
        const float4 q = make_float4((float)lo.x * sign_unit, (float)lo.y * sign_unit, (float)hi.x * sign_unit, (float)hi.y * sign_unit);
        reinterpret_cast<float4*>(out)[k] = q;
        if ((lo.x | lo.y) != 0ull) *reinterpret_cast<ulonglong2*>(acc + 4 * k) = make_ulonglong2(0ull, 0ull);
        if ((hi.x | hi.y) != 0ull) *reinterpret_cast<ulonglong2*>(acc + 4 * k + 2) = make_ulonglong2(0ull, 0ull);
        return q;
    };
    auto load1 = [&](int64_t k) {
        const unsigned long long a1 = acc[k];
        const float q = (float)a1 * sign_unit;
        out[k] = q;
        if (a1 != 0ull) acc[k] = 0ull;
        return q;
    };
    double s, xs, ys;
    com_part_sums_from(load4, load1, Hh, W, v, s_red, s, xs, ys);
    if (threadIdx.x == 0) {
        double* m = moments + (map * kLossParts + v) * 3;
        m[0] = s; m[1] = xs; m[2] = ys;
    }
}

// out[t] = sum_h [target_idx[h] == t] bitmaps[h]   (heliostat_ray_tracer.py:593-608)
// One thread per (t, VEC pixels); heliostats are added in index order (deterministic) with 8-64 loads in flight.
// A 256 x 256 bitmap has too few pixels to fill the chip with 4-pixel threads: VEC = 4 only for large bitmaps.
template <int VEC>
__global__ __launch_bounds__(256) void per_target_sum_kernel(const float* __restrict__ bitmaps,
                                                             const int32_t* __restrict__ target_idx, int H, int T,
                                                             int64_t npix, float* __restrict__ out)
{
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
    if (i >= npix) return;
    const int t = blockIdx.y;
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.0f;
    constexpr int kInFlight = VEC == 1 ? 64 : 8;      // (a thread of a 256 x 256 bitmap has 64 loads in flight: 80 -> 69 us per 1000 bitmaps against 16)
    for (int h0 = 0; h0 < H; h0 += kInFlight) {
        float val[kInFlight][VEC];
#pragma unroll
        for (int q = 0; q < kInFlight; ++q) {
            const int h = h0 + q;
            const bool mine = h < H && target_idx[h] == t;          // wave-uniform
#pragma unroll
            for (int v = 0; v < VEC; ++v) val[q][v] = 0.0f;
            if (mine) {
                if constexpr (VEC == 4) {
                    const float4 x = *reinterpret_cast<const float4*>(bitmaps + (int64_t)h * npix + i);
                    val[q][0] = x.x; val[q][1] = x.y; val[q][2] = x.z; val[q][3] = x.w;
                } else {
                    val[q][0] = bitmaps[(int64_t)h * npix + i];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < kInFlight; ++q)
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] += val[q][v];       // adding 0 for foreign heliostats changes nothing
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) out[(int64_t)t * npix + i + v] = acc[v];
}

// Pick the sample-chunk so that the grid has a few thousand blocks (>> 256 CUs) without
// shrinking chunks below what amortises the per-point prologue.
static void choose_chunks(TraceArgs& a, int target_blocks, int min_chunk)
{
    const int64_t base = (int64_t)a.H * a.n_ptiles;
    int64_t want = (target_blocks + base - 1) / (base > 0 ? base : 1);
    if (want < 1) want = 1;
    int chunk = (int)((a.R + want - 1) / want);
    if (chunk < min_chunk) chunk = min_chunk;
    if (chunk > a.R) chunk = a.R;
    a.r_chunk = chunk;
    a.n_rchunks = (a.R + chunk - 1) / chunk;
}

// Launch geometry of the forward kernel.  Defaults are the tuned MI355X values; the environment
// overrides exist for A/B measurements and tests only and are read ONLY when ARTIST_HIP_DEBUG=1 (debug_env_int: a caller's
// environment does not change what the product launches; ARTIST_HIP_FWD=global selects the plain global-atomic
// kernel, ARTIST_HIP_FWD_BLOCK / _TILE_KB / _BLOCKS / _MINCHUNK the geometry).
struct FwdConfig {
    int variant;        // 0 = LDS window, 1 = global atomics
    int block;          // threads per workgroup
    int tile_cap;       // window capacity in pixels
    int target_blocks;  // grid size to aim for when chunking samples
    int min_chunk;      // fewest samples per workgroup worth a window build + flush
    int p_block;        // target points per workgroup, forward
    int p_block_bwd;    // ... backward (gathers are cheaper than atomics when a ray strays: larger blocks pay)
    bool p_block_fixed, p_block_bwd_fixed;   // set by the environment: no adaptation to the grid size
    int multipass_ratio;
    int min_rays;       // rays per workgroup worth a window build + flush
    bool exact_pblock;  // balanced point blocks that need not fill every lane (workgroup sizes that do not divide P)
    int facet_points;   // the caller's hint: consecutive points that form one facet (0: unknown)
};

static FwdConfig fwd_config()
{
    FwdConfig c;
    const char* v = debug_env_str("ARTIST_HIP_FWD");
    c.variant = (v && v[0] == 'g') ? 1 : 0;
    c.block = debug_env_int("ARTIST_HIP_FWD_BLOCK", 1024);
    if (c.block < 64 || c.block > 1024 || (c.block % 64) != 0) c.block = 1024;
    int kb = debug_env_int("ARTIST_HIP_FWD_TILE_KB", 158);
    if (kb < 4) kb = 4;
    if (kb > 158) kb = 158;   // 160 KB per CU minus < 1 KB of static LDS (the blocking instantiations cap it further)
    c.tile_cap = kb * 256;   // 4-byte fixed-point cells
    c.target_blocks = debug_env_int("ARTIST_HIP_FWD_BLOCKS", 512);
    c.min_chunk = debug_env_int("ARTIST_HIP_FWD_MINCHUNK", 4);
    c.min_rays = debug_env_int("ARTIST_HIP_FWD_MINRAYS", 100000);
    // A footprint larger than the window: up to this multiple of the capacity the window keeps the densest part and the
    // tails stray; beyond it the footprint is swept in several passes.  Round 1 set 2 (a stray then cost ~30 window rays);
    // with parked (forward) and packed (backward) strays trimming wins much further out - metric field with shrunken
    // windows, forward / backward ms: 80 KB (footprint ~4x) 34.8 / 11.9 swept vs 12.5 / 4.4 trimmed; 40 KB (~8x) 71 / 21 vs
    // 39.5 / 5.5.
    c.multipass_ratio = debug_env_int("ARTIST_HIP_FWD_MULTIPASS", 8);
    if (c.multipass_ratio < 1) c.multipass_ratio = 1;
    c.p_block_fixed = debug_env_str("ARTIST_HIP_FWD_PBLOCK") != nullptr;
    c.p_block_bwd_fixed = debug_env_str("ARTIST_HIP_BWD_PBLOCK") != nullptr;
    c.p_block = debug_env_int("ARTIST_HIP_FWD_PBLOCK", 1024);
    if (c.p_block < 64) c.p_block = 64;
    c.exact_pblock = debug_env_int("ARTIST_HIP_PBLOCK_EXACT", 1) != 0;   // balanced blocks; a partial trip costs what its active waves issue
    c.facet_points = 0;
    c.p_block_bwd = debug_env_int("ARTIST_HIP_BWD_PBLOCK", 2048);
    if (c.p_block_bwd < 64) c.p_block_bwd = 64;
    return c;
}

static void window_geometry_for(TraceArgs& a, const FwdConfig& cfg, int p_block_target);

// Work counters of the persistent kernels: eight 4-byte counters per (device, stream) - forward: planar, cylinder and lean
// launch of a split call; backward: main, lean, (spare) - carved from 4 KB pages of device memory that the library allocates
// on first use and keeps.  A counter is zero whenever no launch is using it: the launch's last fetch resets it
// (fetch_work_item), and launches that share a counter are ordered by their stream.  So there is no per-launch memset, and no
// bound on how many launches may be queued (round 2 took the next of 1024 slots per launch, unguarded).
static unsigned* stream_work_counters(hipStream_t stream)
{
    struct Entry { int dev; hipStream_t stream; unsigned* base; };
    static std::vector<Entry> table;
    static unsigned* page = nullptr;
    static int page_dev = -1, page_used = 0;
    static std::mutex lock;
    constexpr int kPageEntries = 4096 / (int)(sizeof(unsigned) * kCountersPerStream);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> guard(lock);
    for (const Entry& e : table)
        if (e.dev == dev && e.stream == stream) return e.base;
    if (page == nullptr || page_dev != dev || page_used == kPageEntries) {
        unsigned* fresh = nullptr;
        if (hipMalloc(reinterpret_cast<void**>(&fresh), 4096) != hipSuccess) return nullptr;
        if (hipMemset(fresh, 0, 4096) != hipSuccess) { (void)hipFree(fresh); return nullptr; }
        page = fresh; page_dev = dev; page_used = 0;
    }
    unsigned* base = page + (page_used++) * kCountersPerStream;
    table.push_back({dev, stream, base});
    return base;
}


// A split call (blocking on: lean launch for the heliostats with no candidate rectangle, blocking instantiation for the
// others) runs its two launches side by side: with the reference's tree the blocking launch has a few dozen items for 256
// CUs and would otherwise hold the stream for the length of one item (0.5 ms forward, 1 ms backward at the metric size);
// in exact mode the second launch fills the first one's tail.  One side stream + two events per host thread and device,
// created on first use and kept; fork/join by events, so it also works under stream capture.  (Two host threads that drive
// the SAME caller stream would share that stream's work counters: one stream, one thread at a time - as for any HIP stream.)
struct SideStream {
    hipStream_t side = nullptr; hipEvent_t fork = nullptr, join = nullptr; int dev = -1; bool pending = false;
    hipStream_t main = nullptr;
    // the side launch depends on everything enqueued on `stream` SO FAR ...
    bool begin(hipStream_t stream)
    {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess) return false;
        if (dev != d) {
            if (dev >= 0) return false;                 // (never: side_stream() keeps one set per device)
            if (hipStreamCreateWithFlags(&side, hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&fork, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&join, hipEventDisableTiming) != hipSuccess) { dev = -2; return false; }
            dev = d;
        }
        if (hipEventRecord(fork, stream) != hipSuccess) return false;
        main = stream;
        return true;
    }
    // ... and is submitted later (after the main stream's launch): the side stream's work starts here
    bool start()
    {
        if (hipStreamWaitEvent(side, fork, 0) != hipSuccess) return false;
        pending = true;
        return true;
    }
    // `stream` continues after the side launch
    void end()
    {
        if (!pending) return;
        pending = false;
        if (hipEventRecord(join, side) == hipSuccess) (void)hipStreamWaitEvent(main, join, 0);
        else (void)hipStreamSynchronize(side);
    }
};
struct SideJoin { SideStream* s; ~SideJoin() { if (s) s->end(); } };      // error returns join too
static SideStream* side_stream()
{
    constexpr int kMaxDevices = 64;
    thread_local SideStream s[kMaxDevices];          // one side stream + two events per host thread AND device
    int d = 0;
    if (debug_env_int("ARTIST_HIP_BLOCKING_CONCURRENT", 1) == 0 || hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDevices) return nullptr;
    return &s[d];
}


// Workgroups that run at once: one per CU (a window fills the CU's LDS).
static int resident_workgroups()
{
    static int n = 0;
    if (n == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            cus <= 0)
            cus = 256;
        n = cus;
        const int cap = debug_env_int("ARTIST_HIP_WORKGROUPS", 0);      // (diagnostic: fewer persistent workgroups than CUs, tools/timeline.sh)
        if (cap > 0 && cap < n) n = cap;
    }
    return n;
}

// Launch geometry of the windowed kernels.  The configured point-block size is the best one for a field that fills the
// chip many times over; for a small share of a field (one of eight ranks: 125 heliostats) the number of ROUNDS of
// resident workgroups decides: 625 workgroups of 2048 points are 2.44 rounds on 256 CUs and cost 3, 1250 of 1024
// points are 4.88 rounds and cost 5 of half the length.  Model: rounds x (rays per workgroup + window build and flush,
// worth ~1.5e4 rays); candidates = the configured size and half of it.
static void window_geometry(TraceArgs& a, const FwdConfig& cfg, int p_block_target, bool fixed)
{
    window_geometry_for(a, cfg, p_block_target);
    if (fixed || (!cfg.exact_pblock && p_block_target / 2 < cfg.block)) return;
    auto cost = [&](const TraceArgs& g) {
        const int64_t wgs = (int64_t)g.H * g.n_pblocks * g.n_rchunks;
        const int64_t rounds = (wgs + resident_workgroups() - 1) / resident_workgroups();
        const int points = (g.facet_points + g.blocks_per_facet - 1) / g.blocks_per_facet;
        // + 0.5: the queue ends when the slowest CU has finished its last item, about half an item after the average one
        // (measured at 125 heliostats, lean backward: 8 blocks per heliostat 0.635 ms < 6 blocks 0.669 < 4 blocks 0.671 <
        //  7 blocks 0.699, the order this expression gives)
        return ((double)rounds + 0.5) * ((double)points * g.r_chunk + 1.5e4);
    };
    if (cfg.exact_pblock) {
        // balanced blocks of any size: also try one to three blocks more per heliostat (measured, 125 heliostats, lean
        // backward: 7 blocks = 875 items = 3.4 rounds 0.699 ms, 8 blocks = 1000 items = 3.9 rounds 0.635 ms)
        const TraceArgs base = a;
        double best = 0.97 * cost(base);
        for (int extra = 1; extra <= 3; ++extra) {
            const int nblk = base.blocks_per_facet + extra;
            TraceArgs g = base;
            window_geometry_for(g, cfg, (base.facet_points + nblk - 1) / nblk);
            if (g.blocks_per_facet == nblk && cost(g) < best) { best = cost(g); a = g; }
        }
        return;
    }
    TraceArgs half = a;
    window_geometry_for(half, cfg, p_block_target / 2);
    if (cost(half) < 0.97 * cost(a)) a = half;
}

// work_item_count() on the host
static int64_t host_item_count(const TraceArgs& a)
{
    if (a.tail_h > 0) return (int64_t)(a.H - a.tail_h) * a.n_pblocks + (int64_t)a.tail_h * a.tail_npb;
    return (int64_t)(a.h_group > 1 ? a.n_groups : a.H) * a.n_pblocks * a.n_rchunks;
}

// The window tables of a stream's launches (window_table_kernel): kWindowTableItems entries per role - forward / backward, main /
// lean launch of a split call -, allocated once per (device, stream) like the counters below.  A launch with more items than
// that works its windows out in its items, as every launch did before round 4.
constexpr int kWindowTableItems = 4096;      // (the table serves small fields only: 640 KB per stream that ever launched one)
static Window* stream_window_tables(hipStream_t stream)
{
    struct Entry { int dev; hipStream_t stream; Window* base; };
    static std::vector<Entry> table;
    static std::mutex lock;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> guard(lock);
    for (const Entry& e : table)
        if (e.dev == dev && e.stream == stream) return e.base;
    Window* base = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&base), sizeof(Window) * 4 * kWindowTableItems) != hipSuccess) return nullptr;
    table.push_back({dev, stream, base});
    return base;
}
// Fills the table of launch `a` (role 0 ... 3) and points the launch at it; leaves a.win_table NULL when there is no table to be had.
static void launch_window_table(TraceArgs& a, int role, int order, hipStream_t stream)
{
    a.win_table = nullptr;
    const int64_t items = host_item_count(a);
    // Small fields only.  Measured same-box, step of the bench with / without the table: 125 heliostats (1.25e6 points) 1.198 /
    // 1.224 ms, 250: 1.945 / 1.925, 500: 3.83 / 3.82, 1000: 7.89 / 7.87 - the trace kernels get 64 + 72 us shorter at the metric
    // size, and the two table kernels take 77 us each: their sampled points are 400 MB of cache lines that the items' own window
    // phase reads too, but as a prefetch of what the item traces next.  A field whose points fit the last-level cache (and whose
    // CUs see four items each, so that every prologue shows) is where the table pays.  (ARTIST_HIP_WINDOW_TABLE = 2 forces it: tests)
    const int mode = debug_env_int("ARTIST_HIP_WINDOW_TABLE", 1);
    if (mode == 0 || items < 1 || items > kWindowTableItems || a.h_group > 1) return;
    if (mode != 2 && (int64_t)a.H * a.P > 1500000) return;
    Window* base = stream_window_tables(stream);
    if (base == nullptr) return;
    Window* tab = base + (int64_t)role * kWindowTableItems;
    if (interleaved_layout(a)) hipLaunchKernelGGL(window_table_kernel<true>, dim3((unsigned)items), dim3(256), 0, stream, a, tab, order);
    else hipLaunchKernelGGL(window_table_kernel<false>, dim3((unsigned)items), dim3(256), 0, stream, a, tab, order);
    a.win_table = tab;
}

// The queue's finer-grained end (lean kernels, whole samples per item): the heliostats of the LAST round of resident
// workgroups are cut into twice as many point blocks.  With ~2 items per CU (one of eight ranks' share of the metric field:
// 500 items of ~200 us on 256 CUs) the queue otherwise ends with half an item of idle time per CU on average; splitting by
// POINTS keeps every point's samples in one item, so bitmaps and gradients are the same bits with or without it
// (ARTIST_HIP_TAIL=0 switches it off; ARTIST_HIP_TAIL=2 forces it for the backward kernel as well: tests).  Costs one more
// window build / clear / flush per extra item.
static void set_queue_tail(TraceArgs& a, int min_points)
{
    a.tail_h = 0; a.tail_bpf = a.blocks_per_facet; a.tail_pblock = a.p_block; a.tail_npb = a.n_pblocks;
    const int mode = debug_env_int("ARTIST_HIP_TAIL", 1);
    if (mode == 0 || a.n_rchunks != 1 || a.h_group > 1 || a.n_pblocks < 1) return;
    const bool forward = min_points >= 512;
    if (mode == 2) min_points = 64;                                   // (tests: also in the backward kernel, any field size)
    const int64_t items = (int64_t)a.H * a.n_pblocks;
    // Measured (same box, forward / backward ms, tail off -> on): 1000 heliostats 3.23 -> 3.17 / 3.43 -> 3.43, 250 heliostats
    // 0.848 -> 0.828 / 0.873 -> 0.885, 125 heliostats 0.506 -> 0.516 / 0.516 -> 0.585: every extra item pays a window build,
    // a clear / staging pass and a flush (more in the backward kernel: 155 KB of gradient window + the edge-point
    // partition), which eats what the shorter tail gives unless the queue has at least three rounds - and in the backward
    // kernel even then.  So: forward only (min_points == 512 is the forward's call), three rounds or more.
    if (mode != 2 && (items < 3 * (int64_t)resident_workgroups() || !forward)) return;
    if (items < 2) return;
    const int bpf = 2 * a.blocks_per_facet;
    const int pb = (a.facet_points + bpf - 1) / bpf;
    if (pb < min_points || (a.facet_points + pb - 1) / pb != bpf) return;
    a.tail_h = (int)std::min<int64_t>(a.H, (resident_workgroups() + a.n_pblocks - 1) / a.n_pblocks);
    a.tail_bpf = bpf; a.tail_pblock = pb;
    a.tail_npb = (a.P + a.facet_points - 1) / a.facet_points * bpf;
}

// Geometry for one point-block size.  p_block: a multiple of the block size close to
// P / ceil(P / 2048) so that point blocks are balanced; samples are chunked only as far as needed to
// fill the chip (each chunk pays a window build + flush).
static void window_geometry_for(TraceArgs& a, const FwdConfig& cfg, int p_block_target)
{
    const int bs = cfg.block;
    // a window build + flush costs about as much as 1e4 rays: keep >= ~1e5 rays per workgroup when the sun has
    // few samples per point (R = 1 field-scale prediction: one workgroup per heliostat)
    if ((int64_t)p_block_target * a.R < cfg.min_rays) p_block_target = (int)((cfg.min_rays + a.R - 1) / a.R);
    if (p_block_target > a.P) p_block_target = a.P;
    // the unit that blocks subdivide: a facet (the caller's hint), several whole facets when a block is to hold more points
    // than a facet has, or the whole heliostat
    int unit = a.P;
    if (cfg.facet_points > 0 && cfg.facet_points < a.P && a.P % cfg.facet_points == 0)
        unit = (int)std::min<int64_t>(a.P, (int64_t)cfg.facet_points * std::max(1, p_block_target / cfg.facet_points));
    if (p_block_target > unit) p_block_target = unit;
    int nblk = (unit + p_block_target - 1) / p_block_target;
    // a field too small to fill the chip is cut into more POINT blocks first (down to half a trip of the workgroup) and
    // into sample chunks only then: a chunk pays a window build and a flush like a block, and in the backward pass a
    // slab of partial gradients on top
    const int64_t units = ((int64_t)a.P + unit - 1) / unit * a.H;
    while (units * nblk < cfg.target_blocks && (unit + nblk) / (nblk + 1) >= bs / 2) ++nblk;
    // (exact: balanced blocks that need not fill every lane - for workgroup sizes that do not divide the unit)
    const int pb = cfg.exact_pblock ? (unit + nblk - 1) / nblk : ((unit + nblk - 1) / nblk + bs - 1) / bs * bs;
    a.p_block = pb;
    a.facet_points = unit;
    a.blocks_per_facet = (unit + pb - 1) / pb;
    a.n_pblocks = (a.P + unit - 1) / unit * a.blocks_per_facet;
    a.tile_cap = cfg.tile_cap;
    a.multipass_ratio = cfg.multipass_ratio;
    a.win_sample = debug_env_int("ARTIST_HIP_WINDOW_SAMPLE", 1) != 0;
    const int64_t base = (int64_t)a.H * a.n_pblocks;
    int64_t want = (cfg.target_blocks + base - 1) / base;
    if (want < 1) want = 1;
    int chunk = (int)((a.R + want - 1) / want);
    if (chunk < cfg.min_chunk) chunk = cfg.min_chunk;
    if (chunk > a.R) chunk = a.R;
    a.r_chunk = chunk;
    a.n_rchunks = (a.R + chunk - 1) / chunk;
    a.reverse_bwd = debug_env_int("ARTIST_HIP_BWD_REVERSE", 0);
    a.reverse_items = debug_env_int("ARTIST_HIP_REVERSE", -1);       // -1: decided on the device (farther_end_is_last)
}

// The geometry of art_trace_bwd's MAIN launch - decided in ONE place for art_trace_bwd and for the two entry points that
// size its scratch buffer (round 3 sized the rectangle-gradient slabs from the generic geometry while the call took the lean
// one, whose facet-sized items are more numerous: a buffer of exactly the reported size was rejected).
//   lean        planar receivers, no blocking: trace_bwd_item_lean, one block per facet
//   lean_block  planar receivers, blocking: the same body with the soft mask and its adjoint
//   otherwise   the generic item (cylinders; blocking with ARTIST_HIP_BLOCK_LEAN=0)
//   whole_samples: never cut a point's samples into chunks (no room for the [n_rchunks,H,P] slabs)
static size_t bwd_main_geometry(TraceArgs& a, FwdConfig& cfg, bool lean, bool lean_block, int64_t facet_points, bool whole_samples)
{
    size_t perm_bytes = 0;
    if (lean || lean_block) {
        cfg.block = lean_block ? kLeanBlockBwdThreads : kLeanBwdGeometryThreads;
        cfg.exact_pblock = true;
        if (lean || debug_env_int("ARTIST_HIP_BLOCK_FACETS", 1) != 0) {
            cfg.facet_points = (int)facet_points;      // (lean kernels only: see art_trace_fwd)
            if (!cfg.p_block_bwd_fixed) cfg.p_block_bwd = kLeanBwdPoints;
        }
        // edge points packed into the block's last waves (trace_bwd_item_lean): the permutation lives behind the window
        a.pack_edge = std::min(std::max(debug_env_int("ARTIST_HIP_BWD_PACK", 32), 0), 256);
        if (a.pack_edge != 0) {
            perm_bytes = 2 * kPackPoints;
            cfg.tile_cap = std::min<int>(cfg.tile_cap, (int)((160 * 1024 - 1408 - (lean_block ? kLeanBlockBwdStatic : 0) - perm_bytes - 8) / 4) / 64 * 64);
        }
    }
    if (whole_samples) cfg.target_blocks = 1;
    // (cylinders, the generic item: facet-sized balanced blocks measured slower - 12.7 vs 12.45 ms forward + backward,
    //  tools/cylinder_bench.py - so they keep round 2's 2048-point blocks)
    window_geometry(a, cfg, cfg.p_block_bwd, cfg.p_block_bwd_fixed);
    return perm_bytes;
}

// Scratch floats a backward launch of geometry `a` needs: [n_rchunks,H,P,8] slabs of partial point gradients when the samples
// are cut into chunks, then one [Cmax,12] slab of rectangle gradients per item when blocking is on.
// (a candidate list longer than the tables - Cmax > kMaxCand -: the slabs hold the tables' rows, and the candidates beyond them
//  have one fp64 row [Cmax - kWideFirst, 12] per heliostat behind the slabs, TraceArgs::wide_grad)
struct BwdScratch { int64_t chunk_floats, slab_floats, wide_floats; };
static BwdScratch bwd_scratch_need(const TraceArgs& a, bool blocking, int64_t Cmax)
{
    BwdScratch n;
    n.chunk_floats = a.n_rchunks > 1 ? (int64_t)a.n_rchunks * a.H * a.P * 8 : 0;
    n.slab_floats = blocking ? (int64_t)a.H * a.n_pblocks * a.n_rchunks * std::min<int64_t>(Cmax, kMaxCand) * 12 : 0;
    n.wide_floats = blocking && Cmax > kMaxCand ? (int64_t)a.H * (Cmax - kWideFirst) * 12 * 2 : 0;
    return n;
}

static bool bwd_uses_lean(bool blocking, int64_t T, int64_t Tc) { return !blocking && debug_env_int("ARTIST_HIP_LEAN", 1) != 0 && T > 0 && Tc == 0; }
static bool bwd_uses_lean_block(bool blocking, int64_t T, int64_t Tc)
{
    return blocking && debug_env_int("ARTIST_HIP_LEAN", 1) != 0 && T > 0 && Tc == 0 && debug_env_int("ARTIST_HIP_BLOCK_LEAN", 1) != 0;
}

}  // namespace art

using namespace art;

extern "C" int art_trace_fwd(const float* origins, const float* normals, const float* incident,
                             const float* dist_u, const float* dist_e, int64_t dist_sh, int64_t dist_sr,
                             int64_t dist_sp, const int32_t* target_idx, const float* plane_centers,
                             const float* plane_normals, const float* plane_dims, const float* cyl_centers,
                             const float* cyl_normals, const float* cyl_axes, const float* cyl_radii,
                             const float* cyl_heights, const float* cyl_opening, const float* prim_corners,
                             const float* prim_spans, const float* prim_normals, const int32_t* cand,
                             const int32_t* cand_count, int64_t Cmax, double max_scatter_angle,
                             double ray_magnitude, double extinction, double reflectivity, int64_t H, int64_t R,
                             int64_t P, int64_t facet_points, int64_t T, int64_t Tc, int64_t W, int64_t Hh, int mode, float* flux,
                             float* factors, uint64_t* accum, double* moments, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    TraceArgs a;
    if (H == 0) {   // empty field: nothing to trace; a per-target bitmap is still all zeros
        if (mode == 1 && flux && T >= 0 && Tc >= 0 && T + Tc > 0 && W > 0 && Hh > 0) {
            ART_HIP(hipMemsetAsync(flux, 0, sizeof(float) * (T + Tc) * Hh * W, stream));
            if (moments) ART_HIP(hipMemsetAsync(moments, 0, sizeof(double) * (T + Tc) * kLossParts * 3, stream));
        }
        return ART_OK;
    }
    if (!flux || !factors ||
        !fill_args(a, origins, normals, incident, dist_u, dist_e, dist_sh, dist_sr, dist_sp, target_idx,
                   plane_centers, plane_normals, plane_dims, cyl_centers, cyl_normals, cyl_axes, cyl_radii, cyl_heights,
                   cyl_opening, ray_magnitude, extinction, reflectivity, H, R, P, T, Tc, W, Hh, mode))
        return ART_EINVAL;
    const StatusWord status = status_word();
    if (status.host != nullptr && (*status.host & 1u)) return ART_ETARGET;     // an earlier launch met a bad target index
    if (status.host != nullptr && (*status.host & 2u)) return ART_ECANDIDATES; // ... or more candidate rectangles than the tables hold
    if (status.host != nullptr && (*status.host & 4u)) return ART_EQUEUE;      // ... or a work counter that an aborted launch left behind
    a.status = status.dev;
    const bool blocking = prim_corners != nullptr;
    if (blocking) {
        if (!prim_spans || !prim_normals || !cand || !cand_count || Cmax < 1 || Cmax > (1 << 22)) return ART_EINVAL;
        a.prim_corners = prim_corners; a.prim_spans = prim_spans; a.prim_normals = prim_normals;
        a.cand = cand; a.cand_count = cand_count; a.Cmax = (int)Cmax;
        set_cone(a, max_scatter_angle);
    }
    const int64_t n_maps = mode == 0 ? H : T + Tc;
    unsigned* counts = reinterpret_cast<unsigned*>(factors);
    // planar launch, cylinder launch, lean launch of a split call
    unsigned* const wc_base = stream_work_counters(stream);
    if (wc_base == nullptr) {
        g_last_hip_error = (int)hipErrorOutOfMemory;
        return ART_ELAUNCH;
    }
    unsigned* work_counters[3] = {wc_base, wc_base + 1, wc_base + 2};
    if (debug_env_int("ARTIST_HIP_CORRUPT_COUNTER", 0) != 0)          // (tests: what an aborted launch leaves behind)
        ART_HIP(hipMemsetAsync(wc_base, 0x07, sizeof(unsigned), stream));
    hipLaunchKernelGGL(trace_fwd_prep_kernel, dim3((unsigned)((3 * H + 255) / 256)), dim3(256), 0, stream, counts, (int)(3 * H), wc_base,
                       status.dev);
    FwdConfig cfg = fwd_config();
    if (facet_points < 0 || (facet_points > 0 && P % facet_points != 0)) return ART_EINVAL;
    if (blocking && cfg.tile_cap > 154 * 256) cfg.tile_cap = 154 * 256;   // room for the rectangle tables in LDS (no gradient sums here: 4.2 KB + 1 KB of static LDS)
    if (accum == nullptr || (reinterpret_cast<uintptr_t>(accum) % 16) != 0) return ART_EINVAL;
    // unit of the pixel accumulators: 2^(ex_g - 28) with 2^ex_g > |mag k_ext k_refl|
    {
        const float kI = (a.mag * a.k_ext) * a.k_refl;
        int ex = 0;
        if (kI != 0.0f && fabsf(kI) < 3.0e38f) (void)frexpf(fabsf(kI) * 1.001f, &ex);
        a.ex_g = std::min(std::max(ex, -80), 80);
        a.scale_g = ldexpf(1.0f, 28 - a.ex_g);
        a.accum = reinterpret_cast<unsigned long long*>(accum);
    }
    SideJoin side = {nullptr};
    std::function<int()> launch_lean;
    if (cfg.variant == 0) {
        // the lean ray body (trace_fwd_item_lean): planar receivers, no blocking, positive and sanely scaled intensity
        // factors - then a valid ray is known to carry intensity and one counter serves both factors
        const bool lean_ok = debug_env_int("ARTIST_HIP_LEAN", 1) != 0 && a.mag >= 1e-6f && a.k_ext >= 1e-6f && a.k_refl >= 1e-6f &&
                             a.mag <= 1e6f && a.k_ext <= 1e6f && a.k_refl <= 1e6f;
        const bool lean = !blocking && lean_ok;
        // Blocking on: the heliostats with an empty candidate list (with the reference's tree, almost all of them) go through
        // the lean kernel in a launch of their own; the blocking instantiation below skips them.
        // A tower with planar AND cylindrical receivers (no blocking) is a split call of the same kind: the lean launch, with
        // its own geometry, for the heliostats that aim at a plane (the others skip themselves), the cylinder launch below.
        const bool mixed_split = !blocking && lean_ok && T > 0 && Tc > 0;
        bool planar_done = false;
        if ((mixed_split || (blocking && lean_ok && T > 0 && Tc == 0)) && debug_env_int("ARTIST_HIP_BLOCKING_SPLIT", 1) != 0) {
            TraceArgs al = a;
            al.split = mixed_split ? 0 : 1;
            FwdConfig cl = fwd_config();
            cl.facet_points = (int)facet_points;
            cl.block = kLeanFwdThreads;
            cl.exact_pblock = true;
            if (!cl.p_block_fixed) cl.p_block = cl.facet_points > 0 ? kLeanFwdPoints : kLeanFwdThreads;
            window_geometry(al, cl, cl.p_block, cl.p_block_fixed);
            set_queue_tail(al, 512);
            const int64_t items_l = host_item_count(al);
            if (items_l > 2147483647LL - 65536) return ART_EINVAL;
            const int64_t blocks_l = (debug_env_int("ARTIST_HIP_PERSISTENT", 3) & 1) ? std::min<int64_t>(items_l, resident_workgroups()) : items_l;
            const size_t lds_l = ((size_t)al.tile_cap + 2) * sizeof(unsigned);
            if (T > 0) launch_window_table(al, 1, -1, stream);          // (on `stream`, ahead of the side stream's start)
            const FwdLaunch launch = {al, flux, counts, work_counters[2]};
            const bool il_l = interleaved_layout(al);
            const int threads_l = cl.block;
            // submitted AFTER the blocking launch (whose few long workgroups then start at once, on CUs of their own)
            SideStream* ss = side_stream();
            if (ss != nullptr && !ss->begin(stream)) ss = nullptr;
            launch_lean = [=, &side]() -> int {
                hipStream_t lean_stream = stream;
                if (ss != nullptr && ss->start()) { side.s = ss; lean_stream = ss->side; }
                if (il_l) {
                    ART_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&trace_fwd_lds_kernel<true, false, false, 1>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_l));
                    hipLaunchKernelGGL((trace_fwd_lds_kernel<true, false, false, 1>), dim3((unsigned)blocks_l), dim3(threads_l),
                                       lds_l, lean_stream, launch);
                } else {
                    ART_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&trace_fwd_lds_kernel<false, false, false, 1>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_l));
                    hipLaunchKernelGGL((trace_fwd_lds_kernel<false, false, false, 1>), dim3((unsigned)blocks_l), dim3(threads_l),
                                       lds_l, lean_stream, launch);
                }
                return ART_OK;
            };
            if (mixed_split) planar_done = true; else a.split = 2;
        }
        if (lean && T > 0 && Tc == 0) {
            // (the facet hint serves the lean kernels only: with blocking on, facet-sized items measured SLOWER - 37.8 vs
            //  30.5 ms forward + backward on the blocking bench's field - and the cylinder kernels were not measured)
            cfg.facet_points = (int)facet_points;
            // Balanced blocks whose trips need not be full (a trip costs what its active waves issue), and - when the
            // caller says where the facets are - one block per facet or per equal part of a facet: every item pays a window
            // phase, a clear and a flush (~17 us), so fewer and larger blocks win as long as no block straddles two facets.
            // Same-box sweep on the metric field (4 facets of 2500 points; blocks per heliostat -> forward ms):
            // 4 -> 3.32, 6 -> 3.43, 8 -> 3.52, 10 -> 3.57 (the round-1 geometry, 1024-point blocks: 3.75), and with most
            // blocks across a facet boundary 5 -> 4.04, 7 -> 3.85, 9 -> 3.96.
            cfg.block = kLeanFwdThreads;
            cfg.exact_pblock = true;
            if (!cfg.p_block_fixed) cfg.p_block = cfg.facet_points > 0 ? kLeanFwdPoints : kLeanFwdThreads;
        }
        // Blocking on, planar receivers: the heliostats WITH candidate rectangles take the lean ray body with the soft mask
        // (trace_fwd_item_lean<.., BLOCKING>; 768-thread persistent workgroups) instead of the generic item.
        const bool lean_block = blocking && lean_ok && T > 0 && Tc == 0 && debug_env_int("ARTIST_HIP_BLOCK_LEAN", 1) != 0;
        // Cylindrical receivers (no blocking): the lean item with the cylinder hit in place of the plane's (trace_fwd_item_lean<..,
        // false, true>; 768-thread persistent workgroups, facet-sized items) instead of the generic one-item workgroups.  On a mixed
        // tower this is the cylinder launch beside the planes' lean launch (a.split == 3: the receiver type decides who skips).
        const bool lean_cyl = !blocking && lean_ok && Tc > 0 && debug_env_int("ARTIST_HIP_CYL_LEAN", 1) != 0;
        if (lean_cyl) {
            cfg.block = kLeanCylFwdThreads;
            cfg.exact_pblock = true;
            cfg.facet_points = (int)facet_points;
            if (!cfg.p_block_fixed) cfg.p_block = cfg.facet_points > 0 ? kLeanFwdPoints : kLeanCylFwdThreads;
        }
        if (lean_block) {
            cfg.block = kLeanBlockFwdThreads;
            cfg.exact_pblock = true;
            if (debug_env_int("ARTIST_HIP_BLOCK_FACETS", 1) != 0) {       // (facet-sized items: slower with the generic body, faster here)
                cfg.facet_points = (int)facet_points;
                if (!cfg.p_block_fixed) cfg.p_block = cfg.facet_points > 0 ? kLeanFwdPoints : kLeanBlockFwdThreads;
            }
        }
        window_geometry(a, cfg, cfg.p_block, cfg.p_block_fixed);
        // Field-scale prediction (per-target bitmaps, a handful of samples per point, whole heliostats per item): items are
        // GROUPS of consecutive heliostats that share a window and its flush (trace_fwd_item_field).  Group size by the round
        // model of window_geometry: rounds x (rays per item + ~1.5e4 ray-equivalents per window build / clear / flush +
        // ~1.5e3 per heliostat for its share of the window phase and its counters).
        bool field_groups = false;
        if (lean && T > 0 && Tc == 0 && mode == 1 && a.R < 8 && a.n_pblocks == 1 && a.n_rchunks == 1 && a.H > 1) {
            int K = debug_env_int("ARTIST_HIP_FIELD_GROUP", -1);                   // -1: chosen here; 0 / 1: off
            if (K < 0) {
                double best = 0.0;
                K = 1;
                for (int k = 1; k <= 32 && k <= a.H; ++k) {
                    const int64_t groups = (a.H + k - 1) / k;
                    const int64_t rounds = (groups + resident_workgroups() - 1) / resident_workgroups();
                    const double cost = ((double)rounds + 0.5) * ((double)k * ((double)a.P * a.R + 1.5e3) + 1.5e4);
                    if (k == 1 || cost < 0.97 * best) { best = cost; K = k; }
                }
            }
            if (K > 1) { a.h_group = std::min(K, a.H); a.n_groups = (a.H + a.h_group - 1) / a.h_group; field_groups = true; }
        }
        if (lean && T > 0 && Tc == 0) set_queue_tail(a, 512);
        if (lean && T > 0 && Tc == 0 && !field_groups) launch_window_table(a, 0, -1, stream);
        if (debug_env_int("ARTIST_HIP_PRINT_GEOMETRY", 0))
            fprintf(stderr, "art_trace_fwd: H %d P %d unit %d blocks/unit %d p_block %d n_pblocks %d r_chunk %d n_rchunks %d threads %d lean %d group %d tail %d x %d\n",
                    a.H, a.P, a.facet_points, a.blocks_per_facet, a.p_block, a.n_pblocks, a.r_chunk, a.n_rchunks, cfg.block, (int)lean,
                    a.h_group, a.tail_h, a.tail_npb);
        const int64_t items = host_item_count(a);
        if (items > 2147483647LL - 65536) return ART_EINVAL;
        // persistent: one workgroup per CU (ARTIST_HIP_PERSISTENT bit 0 cleared: one workgroup per item, for A/B runs)
        const int64_t persistent_blocks = (debug_env_int("ARTIST_HIP_PERSISTENT", 3) & 1) ? std::min<int64_t>(items, resident_workgroups()) : items;
        const size_t lds = ((size_t)a.tile_cap + 2) * sizeof(unsigned);
        // one launch per receiver type present in the tables; a workgroup whose heliostat aims at the other type
        // exits at once (the type is only known on the device)
#define ART_LAUNCH_FWD(IL, CY, BL, LN)                                                                           \
        do {                                                                                                     \
            const int64_t blocks = ((CY && LN == 0) || (BL && LN == 0 && !kBlockingPersistentFwd)) ? items : persistent_blocks;     \
            ART_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&trace_fwd_lds_kernel<IL, CY, BL, LN>),    \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                  \
            const FwdLaunch launch = {a, flux, counts, work_counters[CY ? 1 : 0]};                               \
            hipLaunchKernelGGL((trace_fwd_lds_kernel<IL, CY, BL, LN>), dim3((unsigned)blocks), dim3(CY ? std::min(cfg.block, kCylFwdThreads) : cfg.block),  \
                               lds, stream, launch);                                                             \
        } while (0)
#define ART_LAUNCH_FWD_TYPE(CY)                                                                                  \
        do {                                                                                                     \
            if (il && blocking) ART_LAUNCH_FWD(true, CY, true, 0);                                               \
            else if (il) ART_LAUNCH_FWD(true, CY, false, 0);                                                     \
            else if (blocking) ART_LAUNCH_FWD(false, CY, true, 0);                                               \
            else ART_LAUNCH_FWD(false, CY, false, 0);                                                            \
        } while (0)
        const bool il = interleaved_layout(a);
        if (T > 0 && !planar_done) {
            if (lean && field_groups && il) ART_LAUNCH_FWD(true, false, false, 2);
            else if (lean && field_groups) ART_LAUNCH_FWD(false, false, false, 2);
            else if (lean && il) ART_LAUNCH_FWD(true, false, false, 1);
            else if (lean) ART_LAUNCH_FWD(false, false, false, 1);
            else if (lean_block && il) ART_LAUNCH_FWD(true, false, true, 1);
            else if (lean_block) ART_LAUNCH_FWD(false, false, true, 1);
            else ART_LAUNCH_FWD_TYPE(false);
        }
        if (planar_done) a.split = 3;                // the cylinder launch of a mixed tower: cylinder heliostats first
        if (Tc > 0 && lean_cyl && il) ART_LAUNCH_FWD(true, true, false, 1);
        else if (Tc > 0 && lean_cyl) ART_LAUNCH_FWD(false, true, false, 1);
        else if (Tc > 0) ART_LAUNCH_FWD_TYPE(true);
#undef ART_LAUNCH_FWD_TYPE
#undef ART_LAUNCH_FWD
        ART_HIP(hipGetLastError());
        if (launch_lean) { const int rc = launch_lean(); if (rc != ART_OK) return rc; }
    } else {
        if (Tc > 0 || blocking) return ART_EUNSUPPORTED;   // the global-atomic A/B variant: planar, no blocking
        choose_chunks(a, 4096, 8);
        const int64_t blocks = (int64_t)a.H * a.n_rchunks * a.n_ptiles;
        if (blocks > 2147483647LL) return ART_EINVAL;
        if (interleaved_layout(a))
            hipLaunchKernelGGL(trace_fwd_kernel<true>, dim3((unsigned)blocks), dim3(kBlock), 0, stream, a, flux, counts);
        else
            hipLaunchKernelGGL(trace_fwd_kernel<false>, dim3((unsigned)blocks), dim3(kBlock), 0, stream, a, flux, counts);
    }
    ART_HIP(hipGetLastError());
    if (side.s) { side.s->end(); side.s = nullptr; }
    bool moments_valid = moments != nullptr;
    {   // accumulators -> fp32 bitmaps (every pixel is written: no memset of `flux`); accumulators back to zero
        const int64_t npix = n_maps * Hh * W;
        const float kI = (a.mag * a.k_ext) * a.k_refl;
        const float sign_unit = (kI < 0.0f ? -1.0f : 1.0f) * ldexpf(1.0f, a.ex_g - 28);
        // (the accumulators of a map are 16-byte aligned for the float4 path when Hh * W is even; odd bitmaps take the plain pass)
        if (moments != nullptr && n_maps <= 65535 && Hh >= kLossParts && (((int64_t)Hh * W) & 1) == 0)
            hipLaunchKernelGGL(accum_to_flux_moments_kernel, dim3((unsigned)kLossParts, (unsigned)n_maps), dim3(kMomentsBlock), 0, stream,
                               a.accum, flux, (int)Hh, (int)W, sign_unit, moments);
        else {
            if (moments != nullptr) moments_valid = false;
            hipLaunchKernelGGL(accum_to_flux_kernel, dim3((unsigned)((npix / 2 + 1 + 255) / 256)), dim3(256), 0, stream, a.accum, flux, npix,
                               sign_unit);
        }
    }
    if (blocking)
        hipLaunchKernelGGL(poison_overflow_kernel, dim3((unsigned)H), dim3(256), 0, stream, a.cand_count, a.Cmax, a.target_idx,
                           a.T + a.Tc, flux, (int64_t)Hh * W, mode, moments_valid ? moments : nullptr);
    hipLaunchKernelGGL(finalize_factors_kernel, dim3((unsigned)((H + 255) / 256)), dim3(256), 0, stream, factors,
                       (int)H, (float)(R * P), blocking ? 1 : 0, a.split == 2 ? a.cand_count : nullptr,
                       blocking ? a.cand_count : nullptr, a.Cmax);
    ART_HIP(hipGetLastError());
    if (moments != nullptr && !moments_valid)       // (shapes the sums are not formed for: say so in the buffer itself)
        ART_HIP(hipMemsetAsync(moments, 0xFF, sizeof(double) * n_maps * kLossParts * 3, stream));     // all-ones bits = NaN
    return ART_OK;
}

extern "C" int art_trace_bwd(const float* origins, const float* normals, const float* incident,
                             const float* dist_u, const float* dist_e, int64_t dist_sh, int64_t dist_sr,
                             int64_t dist_sp, const int32_t* target_idx, const float* plane_centers,
                             const float* plane_normals, const float* plane_dims, const float* cyl_centers,
                             const float* cyl_normals, const float* cyl_axes, const float* cyl_radii,
                             const float* cyl_heights, const float* cyl_opening, const float* prim_corners,
                             const float* prim_spans, const float* prim_normals, const int32_t* cand,
                             const int32_t* cand_count, int64_t Cmax, int64_t N, double max_scatter_angle,
                             double ray_magnitude,
                             double extinction, double reflectivity, int64_t H, int64_t R, int64_t P, int64_t facet_points, int64_t T,
                             int64_t Tc, int64_t W, int64_t Hh, int mode, const float* grad_flux, float* grad_origins,
                             float* grad_normals, float* grad_prim_corners, float* grad_prim_spans,
                             float* grad_prim_normals, float* grad_scratch, int64_t grad_scratch_floats, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    TraceArgs a;
    if (H == 0) return ART_OK;
    if (!grad_flux || !grad_origins || !grad_normals ||
        !fill_args(a, origins, normals, incident, dist_u, dist_e, dist_sh, dist_sr, dist_sp, target_idx,
                   plane_centers, plane_normals, plane_dims, cyl_centers, cyl_normals, cyl_axes, cyl_radii, cyl_heights,
                   cyl_opening, ray_magnitude, extinction, reflectivity, H, R, P, T, Tc, W, Hh, mode))
        return ART_EINVAL;
    {
        const StatusWord status = status_word();
        if (status.host != nullptr && (*status.host & 1u)) return ART_ETARGET;
        if (status.host != nullptr && (*status.host & 2u)) return ART_ECANDIDATES;
        if (status.host != nullptr && (*status.host & 4u)) return ART_EQUEUE;
        a.status = status.dev;
    }
    const bool blocking = prim_corners != nullptr;
    if (blocking) {
        if (!prim_spans || !prim_normals || !cand || !cand_count || Cmax < 1 || Cmax > (1 << 22) || N <= 0 ||
            !grad_prim_corners || !grad_prim_spans || !grad_prim_normals)
            return ART_EINVAL;
        a.prim_corners = prim_corners; a.prim_spans = prim_spans; a.prim_normals = prim_normals;
        a.cand = cand; a.cand_count = cand_count; a.Cmax = (int)Cmax;
        set_cone(a, max_scatter_angle);
    }
    float4* go = reinterpret_cast<float4*>(grad_origins);
    float4* gn = reinterpret_cast<float4*>(grad_normals);
    const bool il = interleaved_layout(a);
    FwdConfig cfg = fwd_config();
    if (facet_points < 0 || (facet_points > 0 && P % facet_points != 0)) return ART_EINVAL;
    if (blocking && cfg.tile_cap > 148 * 256) cfg.tile_cap = 148 * 256;   // room for the rectangle tables in LDS
    if (cfg.variant == 0) {
        // the lean ray body (trace_bwd_item_lean): planar receivers, no blocking; 768-thread workgroups, one block per facet
        const bool lean = bwd_uses_lean(blocking, T, Tc);
        // ... and with blocking on, the heliostats WITH candidate rectangles take the same body with the soft mask and its
        // adjoint (trace_bwd_item_lean<.., BLOCKING>) instead of the generic item
        const bool lean_block = bwd_uses_lean_block(blocking, T, Tc);
        const FwdConfig cfg0 = cfg;
        size_t perm_bytes = bwd_main_geometry(a, cfg, lean, lean_block, facet_points, false);
        // A small field is cut into sample chunks to fill the chip; the chunks of a point then write partial gradients
        // to [n_rchunks,H,P] slabs in the caller's scratch buffer and reduce_chunks_kernel adds them in chunk order
        // (no float atomics: bit-reproducible gradients).  Without (enough) scratch the samples stay in one item.
        const bool scratch_usable = grad_scratch != nullptr && (reinterpret_cast<uintptr_t>(grad_scratch) % 16) == 0;
        BwdScratch need = bwd_scratch_need(a, blocking, Cmax);
        if (a.n_rchunks > 1 && (!scratch_usable || grad_scratch_floats < need.chunk_floats + need.slab_floats + need.wide_floats)) {
            cfg = cfg0;
            perm_bytes = bwd_main_geometry(a, cfg, lean, lean_block, facet_points, true);
            need = bwd_scratch_need(a, blocking, Cmax);
        }
        if (debug_env_int("ARTIST_HIP_PRINT_GEOMETRY", 0))
            fprintf(stderr, "art_trace_bwd: H %d P %d unit %d blocks/unit %d p_block %d n_pblocks %d r_chunk %d n_rchunks %d threads %d lean %d\n",
                    a.H, a.P, a.facet_points, a.blocks_per_facet, a.p_block, a.n_pblocks, a.r_chunk, a.n_rchunks, lean ? lean_bwd_threads(a.p_block) : cfg.block, (int)lean);
        const bool atomic_out = a.n_rchunks > 1;      // (round 1's name: today the chunks write slabs, nothing is atomic)
        // Blocking: every item of the blocking launch leaves its rectangle gradients in a slab [Cmax,12] of the scratch buffer
        // (behind the chunk slabs); reduce_prim_grads_kernel adds them in item order.
        float* prim_slabs = nullptr;
        if (blocking) {
            if (!scratch_usable || grad_scratch_floats < need.chunk_floats + need.slab_floats + need.wide_floats)
                return ART_EINVAL;                    // (art_trace_bwd_scratch_floats says how much)
            prim_slabs = grad_scratch + need.chunk_floats;
            if (need.wide_floats > 0) {               // (both float counts before it are multiples of four: 8-byte aligned)
                a.wide_grad = reinterpret_cast<double*>(prim_slabs + need.slab_floats);
                hipLaunchKernelGGL(wide_grad_zero_kernel, dim3((unsigned)H), dim3(256), 0, stream, a.cand_count, a.Cmax, a.wide_grad);
            }
        }
        SideJoin side = {nullptr};                    // (joins `stream` when this scope is left, errors included)
        std::function<int()> launch_lean;
        // Blocking on: the heliostats with an empty candidate list take the lean kernel in a launch of their own (see
        // art_trace_fwd) - for fields large enough that neither launch cuts a point's samples into chunks (the two would
        // need slabs of their own).
        // (a tower with planar and cylindrical receivers, no blocking, is split the same way: see art_trace_fwd)
        const bool mixed_split = !blocking && T > 0 && Tc > 0;
        bool planar_done = false;
        if ((mixed_split || (blocking && Tc == 0)) && !atomic_out && T > 0 && debug_env_int("ARTIST_HIP_LEAN", 1) != 0 &&
            debug_env_int("ARTIST_HIP_BLOCKING_SPLIT", 1) != 0) {
            TraceArgs al = a;
            al.split = mixed_split ? 0 : 1;
            FwdConfig cl = fwd_config();
            cl.block = kLeanBwdGeometryThreads;
            cl.exact_pblock = true;
            cl.facet_points = (int)facet_points;
            if (!cl.p_block_bwd_fixed) cl.p_block_bwd = kLeanBwdPoints;
            size_t perm_l = 0;
            al.pack_edge = std::min(std::max(debug_env_int("ARTIST_HIP_BWD_PACK", 32), 0), 256);
            if (al.pack_edge != 0) {
                perm_l = 2 * kPackPoints;
                cl.tile_cap = std::min<int>(cl.tile_cap, (int)((160 * 1024 - 1408 - perm_l - 8) / 4) / 64 * 64);
            }
            window_geometry(al, cl, cl.p_block_bwd, cl.p_block_bwd_fixed);
            if (al.n_rchunks == 1) {
                set_queue_tail(al, 384);
                const int64_t items_l = host_item_count(al);
                if (items_l > 2147483647LL - 65536) return ART_EINVAL;
                const int64_t blocks_l = (debug_env_int("ARTIST_HIP_PERSISTENT", 3) & 2) ? std::min<int64_t>(items_l, resident_workgroups()) : items_l;
                const size_t lds_l = ((size_t)al.tile_cap + 2) * sizeof(float) + perm_l;
                unsigned* wc = stream_work_counters(stream);
                if (wc == nullptr) { g_last_hip_error = (int)hipErrorOutOfMemory; return ART_ELAUNCH; }
                wc += 4;                                 // backward, lean launch of a split call
                const int threads_l = lean_bwd_threads(al.p_block);
                launch_window_table(al, 3, al.reverse_bwd != 0 ? 1 : 0, stream);
                // submitted AFTER the blocking launch (see art_trace_fwd)
                SideStream* ss = side_stream();
                if (ss != nullptr && !ss->begin(stream)) ss = nullptr;
                launch_lean = [=, &side]() -> int {
                    hipStream_t lean_stream = stream;
                    if (ss != nullptr && ss->start()) { side.s = ss; lean_stream = ss->side; }
                    if (il) {
                        ART_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&trace_bwd_lds_kernel<true, false, false, false, true>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_l));
                        hipLaunchKernelGGL((trace_bwd_lds_kernel<true, false, false, false, true>), dim3((unsigned)blocks_l),
                                           dim3(threads_l), lds_l, lean_stream, al, grad_flux, go, gn, prim_slabs, wc);
                    } else {
                        ART_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&trace_bwd_lds_kernel<false, false, false, false, true>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_l));
                        hipLaunchKernelGGL((trace_bwd_lds_kernel<false, false, false, false, true>), dim3((unsigned)blocks_l),
                                           dim3(threads_l), lds_l, lean_stream, al, grad_flux, go, gn, prim_slabs, wc);
                    }
                    return ART_OK;
                };
                if (mixed_split) planar_done = true; else a.split = 2;
            }
        }
        if (lean && a.n_rchunks == 1) set_queue_tail(a, 384);
        if (lean && T > 0 && Tc == 0) launch_window_table(a, 2, a.reverse_bwd != 0 ? 1 : 0, stream);
        if (lean) cfg.block = lean_bwd_threads(a.p_block);       // (the geometry is fixed by now)
        const int64_t items = host_item_count(a);
        if (items > 2147483647LL - 65536) return ART_EINVAL;
        const int64_t persistent_blocks = (debug_env_int("ARTIST_HIP_PERSISTENT", 3) & 2) ? std::min<int64_t>(items, resident_workgroups()) : items;
        const size_t lds = ((size_t)a.tile_cap + 2) * sizeof(float) + perm_bytes;
        if (atomic_out) {
            go = reinterpret_cast<float4*>(grad_scratch);
            gn = go + (int64_t)a.n_rchunks * H * P;
        }
#define ART_LAUNCH_BWD(IL, AT, CY, BL, LN)                                                                       \
        do {                                                                                                     \
            const int64_t blocks = ((CY && !LN && !kCylPersistentBwd) || (!CY && BL && !LN && !kBlockingPersistentBwd)) ? items : persistent_blocks; \
            ART_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&trace_bwd_lds_kernel<IL, AT, CY, BL, LN>),\
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                  \
            unsigned* work_counter = stream_work_counters(stream);                                               \
            if (work_counter == nullptr) { g_last_hip_error = (int)hipErrorOutOfMemory; return ART_ELAUNCH; }    \
            work_counter += CY ? 6 : 5;                       /* backward: planar / cylinder launch */          \
            hipLaunchKernelGGL((trace_bwd_lds_kernel<IL, AT, CY, BL, LN>), dim3((unsigned)blocks),               \
                               dim3(std::min(cfg.block, CY ? kCylBwdThreads : (BL ? (LN ? kLeanBlockBwdThreads : kBlockingBwdThreads) : 1024))), lds, stream, a, grad_flux, \
                               go, gn, prim_slabs, work_counter);                                                \
        } while (0)
#define ART_LAUNCH_BWD_BL(CY, BL, LN)                                                                            \
        do {                                                                                                     \
            if (il && atomic_out) ART_LAUNCH_BWD(true, true, CY, BL, LN);                                        \
            else if (il) ART_LAUNCH_BWD(true, false, CY, BL, LN);                                                \
            else if (atomic_out) ART_LAUNCH_BWD(false, true, CY, BL, LN);                                        \
            else ART_LAUNCH_BWD(false, false, CY, BL, LN);                                                       \
        } while (0)
#define ART_LAUNCH_BWD_TYPE(CY)                                                                                  \
        do {                                                                                                     \
            if (blocking) ART_LAUNCH_BWD_BL(CY, true, false); else ART_LAUNCH_BWD_BL(CY, false, false);          \
        } while (0)
        if (T > 0 && Tc == 0 && lean) ART_LAUNCH_BWD_BL(false, false, true);
        else if (lean_block) ART_LAUNCH_BWD_BL(false, true, true);
        else
        if (T > 0 && !planar_done) ART_LAUNCH_BWD_TYPE(false);
        if (planar_done) a.split = 3;
        // cylinders without blocking: the lean item with the cylinder hit (persistent 768-thread workgroups on the work queue)
        if (Tc > 0 && !blocking && debug_env_int("ARTIST_HIP_LEAN", 1) != 0 && debug_env_int("ARTIST_HIP_CYL_LEAN", 1) != 0) ART_LAUNCH_BWD_BL(true, false, true);
        else
        if (Tc > 0) ART_LAUNCH_BWD_TYPE(true);
#undef ART_LAUNCH_BWD_TYPE
#undef ART_LAUNCH_BWD_BL
#undef ART_LAUNCH_BWD
        ART_HIP(hipGetLastError());
        if (launch_lean) { const int rc = launch_lean(); if (rc != ART_OK) return rc; }
        if (blocking) {
            if (side.s) { side.s->end(); side.s = nullptr; }        // (the lean launch wrote no slabs, but `stream` must own what follows)
            hipLaunchKernelGGL(reduce_prim_grads_kernel, dim3((unsigned)N), dim3(256), 0, stream, prim_slabs, a.wide_grad, a.cand, a.cand_count, a.H,
                               a.Cmax, a.n_pblocks * a.n_rchunks, grad_prim_corners, grad_prim_spans, grad_prim_normals);
            ART_HIP(hipGetLastError());
        }
        if (atomic_out) {
            const int64_t n = H * P;
            hipLaunchKernelGGL(reduce_chunks_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, go, gn, a.n_rchunks, n,
                               reinterpret_cast<float4*>(grad_origins), reinterpret_cast<float4*>(grad_normals));
            ART_HIP(hipGetLastError());
        }
        return ART_OK;
    }
    if (Tc > 0 || blocking) return ART_EUNSUPPORTED;
    a.r_chunk = a.R; a.n_rchunks = 1;                       // a thread owns all samples of its point: no atomics
    const int64_t blocks = (int64_t)a.H * a.n_ptiles;
    if (blocks > 2147483647LL) return ART_EINVAL;
    if (il) hipLaunchKernelGGL(trace_bwd_kernel<true>, dim3((unsigned)blocks), dim3(kBlock), 0, stream, a, grad_flux, go, gn);
    else hipLaunchKernelGGL(trace_bwd_kernel<false>, dim3((unsigned)blocks), dim3(kBlock), 0, stream, a, grad_flux, go, gn);
    ART_HIP(hipGetLastError());
    return ART_OK;
}

// What art_trace_bwd asks of its scratch buffer for ONE receiver configuration (T planar, Tc cylindrical areas; Cmax > 0:
// blocking on), when it is given at least that much: the floats of the geometry it then takes.  Host arithmetic only.
extern "C" int64_t art_trace_bwd_scratch_need(int64_t H, int64_t R, int64_t P, int64_t facet_points, int64_t T, int64_t Tc, int64_t Cmax)
{
    if (H <= 0 || R <= 0 || P <= 0 || H > (1 << 24) || R > (1 << 24) || P > (1 << 26) || Cmax < 0 || Cmax > (1 << 22) || T < 0 || Tc < 0) return 0;
    FwdConfig cfg = fwd_config();
    if (cfg.variant != 0) return 0;
    if (facet_points < 0 || (facet_points > 0 && P % facet_points != 0)) return 0;
    const bool blocking = Cmax > 0;
    TraceArgs a = {};
    a.H = (int)H; a.R = (int)R; a.P = (int)P; a.Cmax = (int)Cmax;
    if (blocking && cfg.tile_cap > 148 * 256) cfg.tile_cap = 148 * 256;
    (void)bwd_main_geometry(a, cfg, bwd_uses_lean(blocking, T, Tc), bwd_uses_lean_block(blocking, T, Tc), facet_points, false);
    const BwdScratch n = bwd_scratch_need(a, blocking, Cmax);
    return n.chunk_floats + n.slab_floats + n.wide_floats;
}

// The caller's side of it: enough for every receiver configuration (the tables' types are not known when the buffer is
// allocated) = the largest need over the geometries art_trace_bwd can take for these sizes.
extern "C" int64_t art_trace_bwd_scratch_floats(int64_t H, int64_t R, int64_t P, int64_t facet_points, int64_t Cmax)
{
    int64_t most = 0;
    for (int cfgno = 0; cfgno < 3; ++cfgno)       // planar only / cylinders only / both
        most = std::max(most, art_trace_bwd_scratch_need(H, R, P, facet_points, cfgno == 1 ? 0 : 1, cfgno == 0 ? 0 : 1, Cmax));
    return most;
}

extern "C" int art_per_target_sum(const float* bitmaps, const int32_t* target_idx, int64_t H, int64_t T,
                                  int64_t npix, float* out, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!out || T <= 0 || npix <= 0 || H < 0 || T > 65535 || (H > 0 && (!bitmaps || !target_idx))) return ART_EINVAL;
    const bool vec4 = npix >= (int64_t)1 << 20 && (npix % 4) == 0 && (reinterpret_cast<uintptr_t>(bitmaps) % 16) == 0 &&
                      (reinterpret_cast<uintptr_t>(out) % 16) == 0;
    if (vec4)
        hipLaunchKernelGGL(per_target_sum_kernel<4>, dim3((unsigned)((npix / 4 + 255) / 256), (unsigned)T), dim3(256), 0,
                           stream, bitmaps, target_idx, (int)H, (int)T, npix, out);
    else
        hipLaunchKernelGGL(per_target_sum_kernel<1>, dim3((unsigned)((npix + 255) / 256), (unsigned)T), dim3(256), 0, stream,
                           bitmaps, target_idx, (int)H, (int)T, npix, out);
    ART_HIP(hipGetLastError());
    return ART_OK;
}

extern "C" int art_async_status(void* stream_, int clear)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const StatusWord status = status_word();
    if (status.host == nullptr) { g_last_hip_error = (int)hipErrorOutOfMemory; return ART_ELAUNCH; }
    ART_HIP(hipStreamSynchronize(stream));
    const unsigned word = *status.host;
    if (clear) *status.host = 0u;
    return (word & 1u) ? ART_ETARGET : ((word & 2u) ? ART_ECANDIDATES : ((word & 4u) ? ART_EQUEUE : ART_OK));
}
