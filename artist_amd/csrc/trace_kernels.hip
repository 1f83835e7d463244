// trace_kernels.hip - fused heliostat ray trace (forward + backward) for gfx950 / MI355X.
//
// One thread owns one surface point (= ray origin) of one heliostat and walks a chunk of the R
// distortion samples of that point in registers: reflect once per point, then per ray
// scatter -> plane hit -> bilinear splat.  No per-ray intermediate ever reaches HBM: the only
// per-ray traffic is the 8 B (u,e) distortion pair, read coalesced (consecutive lanes =
// consecutive points of the same sample r).
//
// Replaces (ARTIST v2.0.0): heliostat_ray_tracer.py:285-290 (reflect), :328-335 + :510-561
// (scatter_rays), :390-409 (line_plane_intersections), :482-487 (intensities), :489-494 +
// :610-778 (bilinear_splatting), :498-506 (factors), :563-608 (per-target sums, mode 1).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ray_math.hpp"
#include "launch_common.hpp"

namespace art {

constexpr int kBlock = 256;   // 4 waves; one wave per SIMD, several blocks per CU

struct TraceArgs {
    const float4* origins;    // [H,P]
    const float4* normals;    // [H,P]
    const float4* incident;   // [H]
    const float* dist_u;
    const float* dist_e;
    int64_t sh, sr, sp;       // element strides of the distortion views
    const int32_t* target_idx;
    const float* centers;
    const float* pnormals;
    const float* dims;
    float mag, k_ext, k_refl;
    int H, R, P, T, W, Hh;
    int mode;                 // 0: bitmap per heliostat, 1: bitmap per target
    int r_chunk;              // samples per block
    int n_rchunks;            // ceil(R / r_chunk)
    int n_ptiles;             // ceil(P / kBlock)
};

// Distortion fetch.  INTERLEAVED: (u,e) adjacent floats of one [H,R,P,2] buffer -> one 8-byte load.
template <bool INTERLEAVED>
__device__ __forceinline__ void load_dist(const TraceArgs& a, int64_t off, float& u, float& e)
{
    if constexpr (INTERLEAVED) {
        const float2 v = *reinterpret_cast<const float2*>(a.dist_u + off);
        u = v.x; e = v.y;
    } else {
        u = a.dist_u[off]; e = a.dist_e[off];
    }
}

// --------------------------------------------------------------------------------------------
// Forward, global-atomic splat.
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
    const Plane pl = load_plane(a.centers, a.pnormals, a.dims, t, a.W, a.Hh, a.mag, a.k_ext, a.k_refl);
    float* __restrict__ bitmap = flux + (int64_t)(a.mode == 0 ? h : t) * a.Hh * a.W;

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
                float* row_hi = bitmap + (int64_t)(a.Hh - 2 - sp.iu) * a.W + sp.ie;   // flat row iu+1
                float* row_lo = row_hi + a.W;                                          // flat row iu
                atomicAdd(row_hi, sp.cle * sp.chu * I);        // pixel 1
                atomicAdd(row_hi + 1, sp.che * sp.chu * I);    // pixel 2
                atomicAdd(row_lo + 1, sp.che * sp.clu * I);    // pixel 3
                atomicAdd(row_lo, sp.cle * sp.clu * I);        // pixel 4
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

// counts (uint32, rows 0,1 of factors) -> fractions (heliostat_ray_tracer.py:498-506).
__global__ void finalize_factors_kernel(float* factors, int H, float rays_per_heliostat)
{
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= H) return;
    const unsigned* c = reinterpret_cast<const unsigned*>(factors);
    const unsigned n_int = c[h], n_on = c[H + h];
    factors[h] = (float)n_int / rays_per_heliostat;
    factors[H + h] = (float)n_on / rays_per_heliostat;
    factors[2 * H + h] = rays_per_heliostat / rays_per_heliostat;   // blocked == 0 everywhere (blocking off)
}

// --------------------------------------------------------------------------------------------
// Backward: thread owns a point, accumulates dL/dd and dL/do over its samples in registers.
// ATOMIC_OUT: several sample-chunks per point -> atomicAdd into (pre-zeroed) outputs.
// --------------------------------------------------------------------------------------------
template <bool INTERLEAVED, bool ATOMIC_OUT>
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
    if constexpr (ATOMIC_OUT) {
        float* po = reinterpret_cast<float*>(grad_origins + idx);
        float* pn = reinterpret_cast<float*>(grad_normals + idx);
        atomicAdd(po + 0, go.x); atomicAdd(po + 1, go.y); atomicAdd(po + 2, go.z);
        atomicAdd(pn + 0, gn.x); atomicAdd(pn + 1, gn.y); atomicAdd(pn + 2, gn.z); atomicAdd(pn + 3, gn.w);
    } else {
        grad_origins[idx] = go;
        grad_normals[idx] = gn;
    }
}

// out[t] = sum_h [target_idx[h] == t] bitmaps[h]   (heliostat_ray_tracer.py:593-608)
// One thread per (t, pixel); heliostats summed in index order (deterministic).
__global__ void per_target_sum_kernel(const float* __restrict__ bitmaps, const int32_t* __restrict__ target_idx,
                                      int H, int T, int64_t npix, float* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const int t = blockIdx.y;
    float acc = 0.0f;
    for (int h = 0; h < H; ++h)
        if (target_idx[h] == t) acc += bitmaps[(int64_t)h * npix + i];
    out[(int64_t)t * npix + i] = acc;
}

static bool fill_args(TraceArgs& a, const float* origins, const float* normals, const float* incident,
                      const float* dist_u, const float* dist_e, int64_t sh, int64_t sr, int64_t sp,
                      const int32_t* target_idx, const float* centers, const float* pnormals, const float* dims,
                      double mag, double ext, double refl, int64_t H, int64_t R, int64_t P, int64_t T, int64_t W,
                      int64_t Hh, int mode)
{
    if (!origins || !normals || !incident || !dist_u || !dist_e || !target_idx || !centers || !pnormals || !dims)
        return false;
    if (H < 0 || R <= 0 || P <= 0 || T <= 0 || W < 2 || Hh < 2 || (mode != 0 && mode != 1)) return false;
    if (H > (1 << 24) || R > (1 << 24) || P > (1 << 26) || W > 32768 || Hh > 32768) return false;
    if ((double)R * (double)P >= 4294967296.0) return false;   // uint32 ray counters
    a.origins = reinterpret_cast<const float4*>(origins);
    a.normals = reinterpret_cast<const float4*>(normals);
    a.incident = reinterpret_cast<const float4*>(incident);
    a.dist_u = dist_u; a.dist_e = dist_e; a.sh = sh; a.sr = sr; a.sp = sp;
    a.target_idx = target_idx; a.centers = centers; a.pnormals = pnormals; a.dims = dims;
    a.mag = (float)mag; a.k_ext = (float)(1.0 - ext); a.k_refl = (float)refl;
    a.H = (int)H; a.R = (int)R; a.P = (int)P; a.T = (int)T; a.W = (int)W; a.Hh = (int)Hh; a.mode = mode;
    a.n_ptiles = (int)((P + kBlock - 1) / kBlock);
    return true;
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

static bool interleaved_layout(const TraceArgs& a)
{
    return a.dist_e == a.dist_u + 1 && a.sp == 2 && (a.sr % 2) == 0 && (a.sh % 2) == 0 &&
           (reinterpret_cast<uintptr_t>(a.dist_u) % 8) == 0;
}

}  // namespace art

using namespace art;

extern "C" int art_trace_fwd(const float* origins, const float* normals, const float* incident,
                             const float* dist_u, const float* dist_e, int64_t dist_sh, int64_t dist_sr,
                             int64_t dist_sp, const int32_t* target_idx, const float* plane_centers,
                             const float* plane_normals, const float* plane_dims, double ray_magnitude,
                             double extinction, double reflectivity, int64_t H, int64_t R, int64_t P, int64_t T,
                             int64_t W, int64_t Hh, int mode, float* flux, float* factors, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    TraceArgs a;
    if (!flux || !factors ||
        !fill_args(a, origins, normals, incident, dist_u, dist_e, dist_sh, dist_sr, dist_sp, target_idx,
                   plane_centers, plane_normals, plane_dims, ray_magnitude, extinction, reflectivity, H, R, P, T, W,
                   Hh, mode))
        return ART_EINVAL;
    const int64_t n_maps = mode == 0 ? H : T;
    ART_HIP(hipMemsetAsync(flux, 0, sizeof(float) * n_maps * Hh * W, stream));
    if (H == 0) return ART_OK;
    ART_HIP(hipMemsetAsync(factors, 0, sizeof(float) * 3 * H, stream));
    choose_chunks(a, 4096, 8);
    const int64_t blocks = (int64_t)a.H * a.n_rchunks * a.n_ptiles;
    if (blocks > 2147483647LL) return ART_EINVAL;
    unsigned* counts = reinterpret_cast<unsigned*>(factors);
    if (interleaved_layout(a))
        hipLaunchKernelGGL(trace_fwd_kernel<true>, dim3((unsigned)blocks), dim3(kBlock), 0, stream, a, flux, counts);
    else
        hipLaunchKernelGGL(trace_fwd_kernel<false>, dim3((unsigned)blocks), dim3(kBlock), 0, stream, a, flux, counts);
    ART_HIP(hipGetLastError());
    hipLaunchKernelGGL(finalize_factors_kernel, dim3((unsigned)((H + 255) / 256)), dim3(256), 0, stream, factors,
                       (int)H, (float)(R * P));
    ART_HIP(hipGetLastError());
    return ART_OK;
}

extern "C" int art_trace_bwd(const float* origins, const float* normals, const float* incident,
                             const float* dist_u, const float* dist_e, int64_t dist_sh, int64_t dist_sr,
                             int64_t dist_sp, const int32_t* target_idx, const float* plane_centers,
                             const float* plane_normals, const float* plane_dims, double ray_magnitude,
                             double extinction, double reflectivity, int64_t H, int64_t R, int64_t P, int64_t T,
                             int64_t W, int64_t Hh, int mode, const float* grad_flux, float* grad_origins,
                             float* grad_normals, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    TraceArgs a;
    if (!grad_flux || !grad_origins || !grad_normals ||
        !fill_args(a, origins, normals, incident, dist_u, dist_e, dist_sh, dist_sr, dist_sp, target_idx,
                   plane_centers, plane_normals, plane_dims, ray_magnitude, extinction, reflectivity, H, R, P, T, W,
                   Hh, mode))
        return ART_EINVAL;
    if (H == 0) return ART_OK;
    choose_chunks(a, 2048, 16);
    const int64_t blocks = (int64_t)a.H * a.n_rchunks * a.n_ptiles;
    if (blocks > 2147483647LL) return ART_EINVAL;
    float4* go = reinterpret_cast<float4*>(grad_origins);
    float4* gn = reinterpret_cast<float4*>(grad_normals);
    const bool il = interleaved_layout(a);
    if (a.n_rchunks > 1) {
        ART_HIP(hipMemsetAsync(grad_origins, 0, sizeof(float) * 4 * H * P, stream));
        ART_HIP(hipMemsetAsync(grad_normals, 0, sizeof(float) * 4 * H * P, stream));
        if (il) hipLaunchKernelGGL((trace_bwd_kernel<true, true>), dim3((unsigned)blocks), dim3(kBlock), 0, stream, a, grad_flux, go, gn);
        else hipLaunchKernelGGL((trace_bwd_kernel<false, true>), dim3((unsigned)blocks), dim3(kBlock), 0, stream, a, grad_flux, go, gn);
    } else {
        if (il) hipLaunchKernelGGL((trace_bwd_kernel<true, false>), dim3((unsigned)blocks), dim3(kBlock), 0, stream, a, grad_flux, go, gn);
        else hipLaunchKernelGGL((trace_bwd_kernel<false, false>), dim3((unsigned)blocks), dim3(kBlock), 0, stream, a, grad_flux, go, gn);
    }
    ART_HIP(hipGetLastError());
    return ART_OK;
}

extern "C" int art_per_target_sum(const float* bitmaps, const int32_t* target_idx, int64_t H, int64_t T,
                                  int64_t npix, float* out, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!out || T <= 0 || npix <= 0 || H < 0 || T > 65535 || (H > 0 && (!bitmaps || !target_idx))) return ART_EINVAL;
    hipLaunchKernelGGL(per_target_sum_kernel, dim3((unsigned)((npix + 255) / 256), (unsigned)T), dim3(256), 0, stream,
                       bitmaps, target_idx, (int)H, (int)T, npix, out);
    ART_HIP(hipGetLastError());
    return ART_OK;
}
