// nurbs_kernels.hip - NURBS surface points + normals (forward) and control-point gradients
// (backward) for gfx950 / MI355X.
//
// One workgroup per (heliostat, facet) x tile of evaluation points.  The facet's control-point
// net (nu*nv*3 floats, 1.2 KB at 10x10) and its two knot vectors are staged once in LDS and
// then gathered from there (16 control points per evaluation at degree 3); one thread owns one
// evaluation point.  Backward privatises the facet's gradient net in LDS and writes it out once
// with plain stores - no global atomics.  The LDS accumulator is DOUBLE: on gfx950 ds_add_f64 retires
// a wave instruction in ~25 cycles whereas ds_add_f32 needs ~193 (tools/lds_atomic_bench.hip), and the
// wider accumulator makes the sum insensitive to the order of the adds at fp32 output precision.
//
// Replaces (ARTIST v2.0.0): artist/nurbs/surfaces.py:157-245 (find_spans), :247-417
// (basis_functions_and_derivatives, NURBS Book A2.3), :419-473 + :578-613 (gather + A3.6),
// :615-672 (cross product, homogeneous divide, normalise), :674-687 with
// artist/geometry/transforms.py:276-347 (canting rotation + facet translation).
// Operation order follows the reference so that points come out bit-identical to the
// PyTorch-CPU path (the file is compiled with -ffp-contract=off).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "launch_common.hpp"
#include "ray_math.hpp"      // div_noscale: n / a bit for bit, without the range scaling of the IEEE sequence

namespace art {

constexpr int kMaxDeg = 7;
constexpr int kNurbsBlock = 256;

struct NurbsArgs {
    const float* cp;        // [H,F,nu,nv,3]
    const float* uv;        // element (h,f,m,c) at uv[h*uv_sh + f*uv_sf + 2m + c]
    int64_t uv_sh, uv_sf;
    const float* knots_u;   // [H,F,nu+p+1]
    const float* knots_v;   // [H,F,nv+q+1]
    const float* canting;   // [H,F,2,4] or null
    const float* transl;    // [H,F,4] or null
    const float* orientation;   // [H,4,4] or null: the alignment (points @ M^T, normals @ M^T) applied in the epilogue /
                                // undone in the backward's prologue (heliostat_group_rigid_body.py:217-222, 265-270)
    int p, q, uniform;
    int n_unique_u, n_unique_v;
    int H, F, M, nu, nv;
    int n_mtiles;
    int tiles_per_wg;       // forward: tiles of 256 points one workgroup evaluates
};

// surfaces.py:198-207 (uniform) / :209-243 (search).
__device__ __forceinline__ int find_span(float x, const float* knots, int n, int deg, int uniform, int n_unique)
{
    int span;
    if (uniform) {
        span = (int)floorf(x * (float)(n_unique - 1)) + deg;
    } else {
        span = deg;
        for (int k = deg; k < n; ++k)
            if (x >= knots[k] && x < knots[k + 1]) { span = k; break; }
        const float last = knots[n];
        if (fabsf(x - last) <= 1e-5f + 1e-5f * fabsf(last)) span = n - 1;
    }
    // The reference would raise an IndexError outside [deg, n-1]; keep the LDS gathers in range.
    return min(max(span, deg), n - 1);
}

// surfaces.py:294-417 for nth_derivative = 1.  DEG > 0: compile-time degree (registers);
// DEG == 0: runtime degree `deg` (arrays may live in scratch - rare shapes only).
template <int DEG>
__device__ __forceinline__ void basis(float x, const float* knots, int span, int deg, float* N, float* D)
{
    constexpr int S = (DEG > 0 ? DEG : kMaxDeg) + 1;
    const int pdeg = DEG > 0 ? DEG : deg;
    float ndu[S][S], left[S], right[S];
    ndu[0][0] = 1.0f;
#pragma unroll
    for (int j = 1; j < S; ++j) {
        if (j > pdeg) break;
        left[j] = x - knots[span + 1 - j];
        right[j] = knots[span + j] - x;
        float saved = 0.0f;
#pragma unroll
        for (int r = 0; r < S - 1; ++r) {
            if (r >= j) break;
            ndu[j][r] = right[r + 1] + left[j - r];
            // (knot differences and basis values are far inside the normal range: div_noscale == '/')
            const float tmp = div_noscale(ndu[r][j - 1], ndu[j][r]);
            ndu[r][j] = saved + right[r + 1] * tmp;
            saved = left[j - r] * tmp;
        }
        ndu[j][j] = saved;
    }
#pragma unroll
    for (int j = 0; j < S; ++j) {
        if (j > pdeg) break;
        N[j] = ndu[j][pdeg];
    }
    const int pk = pdeg - 1;
#pragma unroll
    for (int r = 0; r < S; ++r) {
        if (r > pdeg) break;
        float d = 0.0f;
        if (r >= 1) {
            const float a0 = div_noscale(1.0f, ndu[pk + 1][r - 1]);
            d = a0 * ndu[r - 1][pk];
        }
        if (r <= pk) {
            const float a1 = div_noscale(-1.0f, ndu[pk + 1][r]);
            d += a1 * ndu[r][pk];
        }
        D[r] = d * (float)pdeg;
    }
}

__device__ __forceinline__ float norm3(float x, float y, float z) { return sqrtf((x * x + y * y) + z * z); }

// transforms.py:320-340.  B[0..2] = e, B[3..5] = n_ortho, B[6..8] = u.
__device__ __forceinline__ void canting_basis(const float* cant, float* B)
{
    float ex = cant[0], ey = cant[1], ez = cant[2];
    const float nx = cant[4], ny = cant[5], nz = cant[6];
    const float ne = fmaxf(norm3(ex, ey, ez), 1e-12f);
    ex = ex / ne; ey = ey / ne; ez = ez / ne;
    float ux = ey * nz - ez * ny, uy = ez * nx - ex * nz, uz = ex * ny - ey * nx;
    const float nu_ = fmaxf(norm3(ux, uy, uz), 1e-8f);
    ux = ux / nu_; uy = uy / nu_; uz = uz / nu_;
    float ox = uy * ez - uz * ey, oy = uz * ex - ux * ez, oz = ux * ey - uy * ex;
    const float no = fmaxf(norm3(ox, oy, oz), 1e-8f);
    ox = ox / no; oy = oy / no; oz = oz / no;
    B[0] = ex; B[1] = ey; B[2] = ez; B[3] = ox; B[4] = oy; B[5] = oz; B[6] = ux; B[7] = uy; B[8] = uz;
}

// float index (even) at which the backward's double accumulator starts inside the dynamic LDS block
__host__ __device__ inline int nurbs_f64_offset(const NurbsArgs& a)
{
    const int n = a.nu * a.nv * 3 + (a.nu + a.p + 1) + (a.nv + a.q + 1) + 12;
    return (n + 1) & ~1;
}

template <int DEG>
struct Eval {
    static constexpr int S = (DEG > 0 ? DEG : kMaxDeg) + 1;
    int su, sv;
    float Nu[S], Du[S], Nv[S], Dv[S];
    float S0[4], Su[3], Sv[3];   // S0 = homogeneous point (w in [3])
};

// surfaces.py:592-613 with the reference's loop order (k, s, r) and zero-initialised accumulators.
template <int DEG>
__device__ __forceinline__ void evaluate(const NurbsArgs& a, const float* s_cp, const float* s_ku, const float* s_kv,
                                         float x, float y, Eval<DEG>& E)
{
    constexpr int S = Eval<DEG>::S;
    const int p = DEG > 0 ? DEG : a.p, q = DEG > 0 ? DEG : a.q;
    E.su = find_span(x, s_ku, a.nu, p, a.uniform, a.n_unique_u);
    E.sv = find_span(y, s_kv, a.nv, q, a.uniform, a.n_unique_v);
    basis<DEG>(x, s_ku, E.su, p, E.Nu, E.Du);
    basis<DEG>(y, s_kv, E.sv, q, E.Nv, E.Dv);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        float temp[S][4];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (s > q) break;
            float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
#pragma unroll
            for (int r = 0; r < S; ++r) {
                if (r > p) break;
                const float b = k ? E.Du[r] : E.Nu[r];
                const float* c3 = s_cp + ((E.su - p + r) * a.nv + (E.sv - q + s)) * 3;
                t0 += b * c3[0]; t1 += b * c3[1]; t2 += b * c3[2];
                t3 += b * 1.0f;          // control-point weights are all ones (surfaces.py:524-537)
            }
            temp[s][0] = t0; temp[s][1] = t1; temp[s][2] = t2; temp[s][3] = t3;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (t > 1 - k) break;
            float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                if (s > q) break;
                const float b = t ? E.Dv[s] : E.Nv[s];
                d0 += b * temp[s][0]; d1 += b * temp[s][1]; d2 += b * temp[s][2]; d3 += b * temp[s][3];
            }
            if (k == 0 && t == 0) { E.S0[0] = d0; E.S0[1] = d1; E.S0[2] = d2; E.S0[3] = d3; }
            else if (k == 1) { E.Su[0] = d0; E.Su[1] = d1; E.Su[2] = d2; }
            else { E.Sv[0] = d0; E.Sv[1] = d1; E.Sv[2] = d2; }
        }
    }
}

// Stage one facet's control net + knots (+ canting basis) in LDS.
// LDS layout: [cp nu*nv*3][knots_u nu+p+1][knots_v nv+q+1][B 9]
__device__ __forceinline__ void stage_facet(const NurbsArgs& a, int hf, float* lds, float*& s_cp, float*& s_ku,
                                            float*& s_kv, float*& s_B)
{
    const int ncp = a.nu * a.nv * 3, nku = a.nu + a.p + 1, nkv = a.nv + a.q + 1;
    s_cp = lds; s_ku = s_cp + ncp; s_kv = s_ku + nku; s_B = s_kv + nkv;
    const float* g_cp = a.cp + (int64_t)hf * ncp;
    for (int i = threadIdx.x; i < ncp; i += blockDim.x) s_cp[i] = g_cp[i];
    for (int i = threadIdx.x; i < nku; i += blockDim.x) s_ku[i] = a.knots_u[(int64_t)hf * nku + i];
    for (int i = threadIdx.x; i < nkv; i += blockDim.x) s_kv[i] = a.knots_v[(int64_t)hf * nkv + i];
    if (a.canting && threadIdx.x == 0) canting_basis(a.canting + (int64_t)hf * 8, s_B);
    __syncthreads();
}

template <int DEG>
__global__ __launch_bounds__(kNurbsBlock) void nurbs_fwd_kernel(NurbsArgs a, float4* __restrict__ points,
                                                                float4* __restrict__ normals)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // a workgroup evaluates tiles_per_wg consecutive tiles of 256 points of one facet: staging the control net and building
    // the canting basis (one thread, ~100 dependent instructions) cost more than evaluating one tile
    const int groups = (a.n_mtiles + a.tiles_per_wg - 1) / a.tiles_per_wg;
    const int hf = blockIdx.x / groups;
    const int mt0 = (blockIdx.x % groups) * a.tiles_per_wg;
    float *s_cp, *s_ku, *s_kv, *s_B;
    stage_facet(a, hf, lds, s_cp, s_ku, s_kv, s_B);
    const int h = hf / a.F, f = hf % a.F;
  for (int mt = mt0; mt < min(mt0 + a.tiles_per_wg, a.n_mtiles); ++mt) {
    const int m = mt * kNurbsBlock + threadIdx.x;
    if (m >= a.M) return;
    const float2 xy = *reinterpret_cast<const float2*>(a.uv + (int64_t)h * a.uv_sh + (int64_t)f * a.uv_sf + 2 * m);
    Eval<DEG> E;
    evaluate<DEG>(a, s_cp, s_ku, s_kv, xy.x, xy.y, E);
    // surfaces.py:615-632
    const float cx = E.Su[1] * E.Sv[2] - E.Su[2] * E.Sv[1];
    const float cy = E.Su[2] * E.Sv[0] - E.Su[0] * E.Sv[2];
    const float cz = E.Su[0] * E.Sv[1] - E.Su[1] * E.Sv[0];
    // :642-657
    const float px = E.S0[0] / E.S0[3], py = E.S0[1] / E.S0[3], pz = E.S0[2] / E.S0[3];
    // :659-661 (F.normalize, eps = 1e-12)
    const float nc = fmaxf(norm3(cx, cy, cz), 1e-12f);
    const float nx = cx / nc, ny = cy / nc, nz = cz / nc;
    float4 po, no;
    if (a.canting) {
        // data @ R^T, R columns = e, n_ortho, u (transforms.py:334-347), then + translation (:678-683)
        const float* tr = a.transl + (int64_t)hf * 4;
        po.x = (((px * s_B[0] + py * s_B[3]) + pz * s_B[6]) + 1.0f * 0.0f) + tr[0];
        po.y = (((px * s_B[1] + py * s_B[4]) + pz * s_B[7]) + 1.0f * 0.0f) + tr[1];
        po.z = (((px * s_B[2] + py * s_B[5]) + pz * s_B[8]) + 1.0f * 0.0f) + tr[2];
        po.w = 1.0f + tr[3];
        no.x = (nx * s_B[0] + ny * s_B[3]) + nz * s_B[6];
        no.y = (nx * s_B[1] + ny * s_B[4]) + nz * s_B[7];
        no.z = (nx * s_B[2] + ny * s_B[5]) + nz * s_B[8];
        no.w = 0.0f;
    } else {
        po = make_float4(px, py, pz, 1.0f);
        no = make_float4(nx, ny, nz, 0.0f);
    }
    if (a.orientation) {          // the arithmetic of align_fwd_kernel: the fused result equals the two-kernel one bit for bit
        const float* Mo = a.orientation + (int64_t)h * 16;
        po = apply_mt(po, Mo);
        no = apply_mt(no, Mo);
    }
    points[(int64_t)hf * a.M + m] = po;
    normals[(int64_t)hf * a.M + m] = no;
  }
}

// One workgroup per (h,f); threads stride over the M evaluation points; gradient net in LDS.
template <int DEG>
__global__ __launch_bounds__(kNurbsBlock) void nurbs_bwd_kernel(NurbsArgs a, const float4* __restrict__ g_points,
                                                                const float4* __restrict__ g_normals,
                                                                float* __restrict__ g_cp)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int S = Eval<DEG>::S;
    const int hf = blockIdx.x;
    float *s_cp, *s_ku, *s_kv, *s_B;
    stage_facet(a, hf, lds, s_cp, s_ku, s_kv, s_B);
    const int ncp = a.nu * a.nv * 3;
    double* s_g = reinterpret_cast<double*>(lds + nurbs_f64_offset(a));   // 8-byte aligned tail of the LDS block
    for (int i = threadIdx.x; i < ncp; i += blockDim.x) s_g[i] = 0.0;
    __syncthreads();
    const int p = DEG > 0 ? DEG : a.p, q = DEG > 0 ? DEG : a.q;
    const int h = hf / a.F, f = hf % a.F;
    // A thread owns a RUN of consecutive evaluation points and keeps the (p+1)(q+1) x 3 sums of the control points of the
    // current knot-span cell in registers; they go to the LDS accumulator - one ds_add_f64 each, the pipe that bounds this
    // kernel - only when the run leaves the cell and at its end.  On a row-major evaluation grid (50 x 50 points over 7 x 7
    // cells) a run of ten points meets ~2.4 cells: 4 x fewer LDS atomics than one set per point (0.47 -> see DESIGN.md 4.3);
    // scattered evaluation points flush after every point, as before.  Neighbouring runs share their control points, so lanes
    // that took neighbouring runs would all add to the same LDS cells (up to 64-way serialisation): lane t takes run
    // (t K) mod n_runs, K prime and coprime to n_runs (a bijection), which puts the lanes of a wave in different cells; the
    // gradient loads become gathers of 32 B per point and are L2-resident.
    const int run_len = (a.M + (int)blockDim.x - 1) / (int)blockDim.x;
    const int n_runs = (a.M + run_len - 1) / run_len;
    int K = 1;
    {
        const int primes[8] = {61, 59, 53, 47, 43, 41, 37, 31};
#pragma unroll
        for (int i = 7; i >= 0; --i)
            if (n_runs % primes[i] != 0) K = primes[i];
    }
    float acc[S][S][3];
    int cur_u = -1, cur_v = -1;
    auto flush = [&]() {
        if (cur_u < 0) return;
#pragma unroll
        for (int r = 0; r < S; ++r) {
            if (r > p) break;
#pragma unroll
            for (int s_ = 0; s_ < S; ++s_) {
                if (s_ > q) break;
                double* g3 = s_g + ((cur_u - p + r) * a.nv + (cur_v - q + s_)) * 3;
#pragma unroll
                for (int k = 0; k < 3; ++k) atomicAdd(g3 + k, (double)acc[r][s_][k]);
            }
        }
    };
    if ((int)threadIdx.x < n_runs) {
      const int run = (int)(((int64_t)threadIdx.x * K) % n_runs);
      for (int m = run * run_len; m < min(a.M, (run + 1) * run_len); ++m) {
        const float2 xy = *reinterpret_cast<const float2*>(a.uv + (int64_t)h * a.uv_sh + (int64_t)f * a.uv_sf + 2 * m);
        Eval<DEG> E;
        evaluate<DEG>(a, s_cp, s_ku, s_kv, xy.x, xy.y, E);
        float4 gp = g_points[(int64_t)hf * a.M + m];
        float4 gn = g_normals[(int64_t)hf * a.M + m];
        if (a.orientation) {      // align_bwd_kernel's arithmetic
            const float* Mo = a.orientation + (int64_t)h * 16;
            gp = apply_m(gp, Mo);
            gn = apply_m(gn, Mo);
        }
        float gpt[3], gnr[3];
        if (a.canting) {   // out_j = sum_k data_k B[k][j]  ->  g_data_k = sum_j g_out_j B[k][j]
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                gpt[k] = gp.x * s_B[3 * k] + gp.y * s_B[3 * k + 1] + gp.z * s_B[3 * k + 2];
                gnr[k] = gn.x * s_B[3 * k] + gn.y * s_B[3 * k + 1] + gn.z * s_B[3 * k + 2];
            }
        } else {
            gpt[0] = gp.x; gpt[1] = gp.y; gpt[2] = gp.z; gnr[0] = gn.x; gnr[1] = gn.y; gnr[2] = gn.z;
        }
        const float cx = E.Su[1] * E.Sv[2] - E.Su[2] * E.Sv[1];
        const float cy = E.Su[2] * E.Sv[0] - E.Su[0] * E.Sv[2];
        const float cz = E.Su[0] * E.Sv[1] - E.Su[1] * E.Sv[0];
        const float nc = norm3(cx, cy, cz);
        float gc[3];
        if (nc < 1e-12f) {
            gc[0] = gnr[0] / 1e-12f; gc[1] = gnr[1] / 1e-12f; gc[2] = gnr[2] / 1e-12f;
        } else {
            const float inv = 1.0f / nc;
            const float nx = cx * inv, ny = cy * inv, nz = cz * inv;
            const float dot = nx * gnr[0] + ny * gnr[1] + nz * gnr[2];
            gc[0] = (gnr[0] - nx * dot) * inv; gc[1] = (gnr[1] - ny * dot) * inv; gc[2] = (gnr[2] - nz * dot) * inv;
        }
        // c = Su x Sv : gSu = Sv x gc ; gSv = gc x Su
        const float gSu[3] = {E.Sv[1] * gc[2] - E.Sv[2] * gc[1], E.Sv[2] * gc[0] - E.Sv[0] * gc[2],
                              E.Sv[0] * gc[1] - E.Sv[1] * gc[0]};
        const float gSv[3] = {gc[1] * E.Su[2] - gc[2] * E.Su[1], gc[2] * E.Su[0] - gc[0] * E.Su[2],
                              gc[0] * E.Su[1] - gc[1] * E.Su[0]};
        const float iw = 1.0f / E.S0[3];
        const float gS[3] = {gpt[0] * iw, gpt[1] * iw, gpt[2] * iw};
        const bool same = E.su == cur_u && E.sv == cur_v;
        if (!same) { flush(); cur_u = E.su; cur_v = E.sv; }
#pragma unroll
        for (int r = 0; r < S; ++r) {
            if (r > p) break;
#pragma unroll
            for (int s_ = 0; s_ < S; ++s_) {
                if (s_ > q) break;
                const float w00 = E.Nu[r] * E.Nv[s_], w10 = E.Du[r] * E.Nv[s_], w01 = E.Nu[r] * E.Dv[s_];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float c = w00 * gS[k] + w10 * gSu[k] + w01 * gSv[k];
                    acc[r][s_][k] = same ? acc[r][s_][k] + c : c;
                }
            }
        }
      }
      flush();
    }
    __syncthreads();
    float* out = g_cp + (int64_t)hf * ncp;
    for (int i = threadIdx.x; i < ncp; i += blockDim.x) out[i] = (float)s_g[i];
}

static bool fill_nurbs(NurbsArgs& a, const float* cp, const float* uv, int64_t uv_sh, int64_t uv_sf,
                       const float* ku, const float* kv, const float* canting, const float* transl, int p, int q,
                       int uniform, int64_t nuq_u, int64_t nuq_v, int64_t H, int64_t F, int64_t M, int64_t nu,
                       int64_t nv)
{
    if (!cp || !uv || !ku || !kv) return false;
    if (canting && !transl) return false;
    if (p < 1 || q < 1 || p > kMaxDeg || q > kMaxDeg || nu <= p || nv <= q) return false;
    if (H < 0 || F <= 0 || M <= 0 || nu > 4096 || nv > 4096 || H * F > 2147483647LL || M > 2147483647LL) return false;
    if (uniform && (nuq_u < 2 || nuq_v < 2)) return false;
    a.cp = cp; a.uv = uv; a.uv_sh = uv_sh; a.uv_sf = uv_sf; a.knots_u = ku; a.knots_v = kv;
    a.canting = canting; a.transl = transl; a.orientation = nullptr; a.p = p; a.q = q; a.uniform = uniform;
    a.n_unique_u = (int)nuq_u; a.n_unique_v = (int)nuq_v;
    a.H = (int)H; a.F = (int)F; a.M = (int)M; a.nu = (int)nu; a.nv = (int)nv;
    a.n_mtiles = (int)((M + kNurbsBlock - 1) / kNurbsBlock);
    a.tiles_per_wg = 1;
    return true;
}

static size_t nurbs_lds_bytes(const NurbsArgs& a, bool bwd)
{
    const size_t ncp = (size_t)a.nu * a.nv * 3;
    size_t n = ncp + (a.nu + a.p + 1) + (a.nv + a.q + 1) + 12;
    if (bwd) n = (size_t)nurbs_f64_offset(a) + 2 * ncp;
    return n * sizeof(float);
}

}  // namespace art

using namespace art;

#define ART_DISPATCH_DEG(KERNEL, grid, lds, stream, ...)                                                        \
    do {                                                                                                        \
        const int deg__ = (a.p == a.q && a.p <= 4) ? a.p : 0;                                                   \
        switch (deg__) {                                                                                        \
            case 1: hipLaunchKernelGGL(KERNEL<1>, grid, dim3(kNurbsBlock), lds, stream, __VA_ARGS__); break;     \
            case 2: hipLaunchKernelGGL(KERNEL<2>, grid, dim3(kNurbsBlock), lds, stream, __VA_ARGS__); break;     \
            case 3: hipLaunchKernelGGL(KERNEL<3>, grid, dim3(kNurbsBlock), lds, stream, __VA_ARGS__); break;     \
            case 4: hipLaunchKernelGGL(KERNEL<4>, grid, dim3(kNurbsBlock), lds, stream, __VA_ARGS__); break;     \
            default: hipLaunchKernelGGL(KERNEL<0>, grid, dim3(kNurbsBlock), lds, stream, __VA_ARGS__); break;    \
        }                                                                                                       \
    } while (0)

extern "C" int art_nurbs_fwd(const float* control_points, const float* eval_points, int64_t uv_sh, int64_t uv_sf,
                             const float* knots_u, const float* knots_v, const float* canting,
                             const float* translations, int p, int q, int uniform, int64_t n_unique_u,
                             int64_t n_unique_v, int64_t H, int64_t F, int64_t M, int64_t nu, int64_t nv,
                             const float* orientation, float* points, float* normals, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    NurbsArgs a;
    if (H == 0) return ART_OK;
    if (!points || !normals ||
        !fill_nurbs(a, control_points, eval_points, uv_sh, uv_sf, knots_u, knots_v, canting, translations, p, q,
                    uniform, n_unique_u, n_unique_v, H, F, M, nu, nv))
        return ART_EINVAL;
    a.orientation = orientation;
    if (H == 0) return ART_OK;
    const size_t lds = nurbs_lds_bytes(a, false);
    if (lds > 64 * 1024) return ART_EUNSUPPORTED;
    // enough workgroups to fill the chip a few times over, as few stagings as that allows
    {
        static const int env_tiles = getenv("ARTIST_HIP_NURBS_TILES") ? atoi(getenv("ARTIST_HIP_NURBS_TILES")) : 0;
        int t = env_tiles;
        if (t <= 0) {      // as few workgroups per facet as still give ~1000 workgroups, tiles dealt evenly
            int groups = 1;
            while (groups < a.n_mtiles && (int64_t)H * F * groups < 1024) ++groups;
            t = (a.n_mtiles + groups - 1) / groups;
        }
        a.tiles_per_wg = t;      // (1000 heliostats x 4 facets x 2500 points: 1 / 2 / 5 / 10 tiles = 0.257 / 0.235 / 0.225 / 0.220 ms)
    }
    const int64_t blocks = (int64_t)H * F * ((a.n_mtiles + a.tiles_per_wg - 1) / a.tiles_per_wg);
    if (blocks > 2147483647LL) return ART_EINVAL;
    ART_DISPATCH_DEG(nurbs_fwd_kernel, dim3((unsigned)blocks), lds, stream, a, reinterpret_cast<float4*>(points),
                     reinterpret_cast<float4*>(normals));
    ART_HIP(hipGetLastError());
    return ART_OK;
}

extern "C" int art_nurbs_bwd(const float* control_points, const float* eval_points, int64_t uv_sh, int64_t uv_sf,
                             const float* knots_u, const float* knots_v, const float* canting, int p, int q,
                             int uniform, int64_t n_unique_u, int64_t n_unique_v, int64_t H, int64_t F, int64_t M,
                             int64_t nu, int64_t nv, const float* orientation, const float* grad_points,
                             const float* grad_normals, float* grad_control_points, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    NurbsArgs a;
    static const float dummy_transl[4] = {0, 0, 0, 0};
    if (H == 0) return ART_OK;
    if (!grad_points || !grad_normals || !grad_control_points ||
        !fill_nurbs(a, control_points, eval_points, uv_sh, uv_sf, knots_u, knots_v, canting,
                    canting ? dummy_transl : nullptr, p, q, uniform, n_unique_u, n_unique_v, H, F, M, nu, nv))
        return ART_EINVAL;
    a.transl = nullptr;   // unused by the backward
    a.orientation = orientation;
    if (H == 0) return ART_OK;
    const size_t lds = nurbs_lds_bytes(a, true);
    if (lds > 64 * 1024) return ART_EUNSUPPORTED;
    ART_DISPATCH_DEG(nurbs_bwd_kernel, dim3((unsigned)(H * F)), lds, stream, a,
                     reinterpret_cast<const float4*>(grad_points), reinterpret_cast<const float4*>(grad_normals),
                     grad_control_points);
    ART_HIP(hipGetLastError());
    return ART_OK;
}
