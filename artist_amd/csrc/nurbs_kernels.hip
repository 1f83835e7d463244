// nurbs_kernels.hip - NURBS surface points + normals (forward) and control-point gradients
// (backward) for gfx950 / MI355X.
//
// Two evaluation schemes behind one entry point, chosen per workgroup ON THE DEVICE:
//
//  * TENSOR-PRODUCT (round 4).  ARTIST's evaluation points are a cartesian grid (artist/nurbs/utils.py:37-49:
//    cartesian_prod(u_lin, v_lin), point m = i Mv + j at (u_i, v_j)), so S = B_u CP B_v^T factorises: the basis functions
//    are computed once per grid ROW and COLUMN (Mu + Mv evaluations of A2.3 with its divisions instead of 2 Mu Mv), stage 1
//    contracts the control net with the row bases (temp[i][c] = sum_r Nu_i[r] CP[su_i-p+r][c], and the same with Du), stage 2
//    contracts with the column bases per point.  The sums run in the reference's order (r, then s, from a zero
//    accumulator, no FMA contraction), so points are bit-identical to the scattered scheme and to the reference.
//    The backward is the adjoint of the two stages with every output element owned by ONE thread that adds its terms in
//    index order - no atomics of any kind, control-point gradients are bit-reproducible.  The workgroup discovers the grid
//    itself (row length = index of the first point whose u differs from point 0's; then every point is compared with
//    (u of its row's first point, v of its column's first point)): no flag in the API, no host synchronisation, and points
//    that are not a grid simply take the other scheme.
//  * SCATTERED.  Arbitrary evaluation points (SurfaceGenerator.fit_nurbs evaluates at deflectometry points,
//    artist/field/surface_generator.py:133-202): one thread owns one evaluation point, gathers its (p+1)(q+1) control points
//    from LDS; backward privatises the facet's gradient net in LDS in DOUBLE (ds_add_f64 retires a wave instruction in ~25
//    cycles, ds_add_f32 needs ~193: tools/lds_atomic_bench.hip) - the order of those adds is not fixed, the result is the same
//    to fp32 output precision but not bit-reproducible.
//
// MFMA: not used.  gfx950's f32-input MFMA (v_mfma_f32_16x16x4_f32) issues at 64 FLOP/clk/SIMD - exactly the v_fma_f32 rate
// (MI355X_MICROARCH.md: 157.3 TFLOP/s both) - the basis matrices are banded (p+1 = 4 of nv = 10 columns non-zero), so a dense
// contraction does 2.5x the multiply-adds, and its fused accumulation breaks the bit-parity of the points.  Measured:
// tools/nurbs_mfma_bench.hip, profiles/r04_nurbs_bench.json, DESIGN.md 4.3.
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

#include <algorithm>

#include "launch_common.hpp"
#include "ray_math.hpp"      // div_noscale: n / a bit for bit, without the range scaling of the IEEE sequence

namespace art {

constexpr int kMaxDeg = 7;
constexpr int kNurbsFwdBlock = 64;      // forward: one wave per (facet, group of grid rows) - no workgroup barriers
constexpr int kNurbsBwdBlock = 512;     // backward: launch bound; 4 or 8 waves per facet (art_nurbs_bwd)

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
    int groups;             // forward: workgroups per facet (each takes a run of grid rows / of points)
    int grid_mode;          // 1: look for a cartesian grid (tensor-product scheme), 0: scattered scheme only
    int lds_floats;         // dynamic LDS of the launch, in floats
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

// ---- shared per-point pieces ---------------------------------------------------------------------------------------

// surfaces.py:615-689 for one evaluation point: cross product, homogeneous divide, normalisation, canting rotation + facet
// translation (transforms.py:334-347), and - when fused - the alignment (heliostat_group_rigid_body.py:217-222).
__device__ __forceinline__ void finish_point(const NurbsArgs& a, const float* S0, const float* Su, const float* Sv,
                                             const float* s_B, int hf, int h, int m, float4* __restrict__ points,
                                             float4* __restrict__ normals)
{
    // surfaces.py:615-632
    const float cx = Su[1] * Sv[2] - Su[2] * Sv[1];
    const float cy = Su[2] * Sv[0] - Su[0] * Sv[2];
    const float cz = Su[0] * Sv[1] - Su[1] * Sv[0];
    // :642-657
    const float px = S0[0] / S0[3], py = S0[1] / S0[3], pz = S0[2] / S0[3];
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

// The adjoint of finish_point for one evaluation point: dL/dS (position), dL/dSu, dL/dSv from the two incoming gradients.
__device__ __forceinline__ void point_adjoint(const NurbsArgs& a, float w, const float* Su, const float* Sv, const float* s_B,
                                              int h, float4 gp, float4 gn, float* gS, float* gSu, float* gSv)
{
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
    const float cx = Su[1] * Sv[2] - Su[2] * Sv[1];
    const float cy = Su[2] * Sv[0] - Su[0] * Sv[2];
    const float cz = Su[0] * Sv[1] - Su[1] * Sv[0];
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
    gSu[0] = Sv[1] * gc[2] - Sv[2] * gc[1]; gSu[1] = Sv[2] * gc[0] - Sv[0] * gc[2]; gSu[2] = Sv[0] * gc[1] - Sv[1] * gc[0];
    gSv[0] = gc[1] * Su[2] - gc[2] * Su[1]; gSv[1] = gc[2] * Su[0] - gc[0] * Su[2]; gSv[2] = gc[0] * Su[1] - gc[1] * Su[0];
    const float iw = 1.0f / w;
    gS[0] = gpt[0] * iw; gS[1] = gpt[1] * iw; gS[2] = gpt[2] * iw;
}

// ---- tensor-product scheme: grid discovery, row / column bases, stage 1 ----------------------------------------------

// Writes of one wave to LDS become visible to its other lanes: the LDS unit executes a wave's instructions in order, so all it
// takes is to wait for the wave's own LDS operations and to keep the compiler from moving memory accesses across this point.
// No hardware barrier - the waves of a workgroup work independently - and no wait for outstanding GLOBAL loads (a
// wavefront-scope fence made the compiler wait for those as well, which serialised the prefetch of the next strip).
__device__ __forceinline__ void wave_sync()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// Row length of a cartesian point list m = i Mv + j at (u_i, v_j): the index of the first point whose u differs from
// point 0's (M when there is none: one row).  Found by every wave for itself (64 points per step: one step for ARTIST's 50).
// Returns 0 when point 0's u is a NaN (it differs from itself).
__device__ __forceinline__ int find_row_length(const float* __restrict__ uvp, int M)
{
    const int lane = threadIdx.x & 63;
    const float u0 = uvp[0];
    for (int base = 0; base < M; base += 64) {
        const int m = base + lane;
        const bool differs = m < M && uvp[2 * (int64_t)m] != u0;
        const unsigned long long mask = __ballot(differs);
        if (mask != 0ull) return base + (int)__builtin_ctzll(mask);
    }
    return M;
}

// Row / column basis records in LDS, BasisRec<DEG>::words words per grid line:
//   [span (int bits), N[0..S-1], D[0..S-1], sum_r N[r] * 1, the line's coordinate]
template <int DEG> struct BasisRec { static constexpr int S = (DEG > 0 ? DEG : kMaxDeg) + 1; static constexpr int words = 2 * S + 3; };

template <int DEG>
__device__ __forceinline__ void line_basis(const NurbsArgs& a, float x, const float* knots, int n, int deg, int n_unique, float* rec)
{
    constexpr int S = BasisRec<DEG>::S;
    float N[S], D[S];
    const int span = find_span(x, knots, n, deg, a.uniform, n_unique);
    basis<DEG>(x, knots, span, deg, N, D);
    rec[0] = __int_as_float(span);
    float w = 0.f;
#pragma unroll
    for (int r = 0; r < S; ++r) {
        if (r > deg) break;
        rec[1 + r] = N[r]; rec[1 + S + r] = D[r];
        w += N[r] * 1.0f;          // the homogeneous coordinate's inner sum (weights are all ones, surfaces.py:524-537)
    }
    rec[1 + 2 * S] = w;
    rec[2 + 2 * S] = x;
}

// stage 1 for grid rows [r0, r0 + rs): temp[il][c][0..2] = sum_r Nu[r] CP[su-p+r][c], temp[il][c][3..5] the same with Du
// (surfaces.py:592-603: loop order r from a zero accumulator).  Threads t0, t0 + stride, ... of the caller's team.
template <int DEG>
__device__ __forceinline__ void stage1_rows(const NurbsArgs& a, const float* s_cp, const float* s_bu, int bu_row0, int r0, int rs,
                                            float* s_temp, int t0, int stride)
{
    constexpr int S = BasisRec<DEG>::S, W = BasisRec<DEG>::words;
    const int p = DEG > 0 ? DEG : a.p;
    const float inv_nv = 1.0f / (float)a.nv;
    for (int idx = t0; idx < rs * a.nv; idx += stride) {
        const int il = (int)(((float)idx + 0.5f) * inv_nv), c = idx - il * a.nv;      // (exact: idx < 2^22)
        const float* rec = s_bu + (r0 + il - bu_row0) * W;
        const int su = __float_as_int(rec[0]);
        float t[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < S; ++r) {
            if (r > p) break;
            const float* c3 = s_cp + ((su - p + r) * a.nv + c) * 3;
            const float bn = rec[1 + r], bd = rec[1 + S + r];
            t[0] += bn * c3[0]; t[1] += bn * c3[1]; t[2] += bn * c3[2];
            t[3] += bd * c3[0]; t[4] += bd * c3[1]; t[5] += bd * c3[2];
        }
        float* o = s_temp + (il * a.nv + c) * 6;
#pragma unroll
        for (int k = 0; k < 6; ++k) o[k] = t[k];
    }
}

// stage 2 for one point: S0 (with w), Su, Sv from its row's temps and its column's basis record (surfaces.py:604-613)
template <int DEG>
__device__ __forceinline__ void stage2_point(const NurbsArgs& a, const float* trow, const float* recv, float wrow, float* S0,
                                             float* Su, float* Sv)
{
    constexpr int S = BasisRec<DEG>::S;
    const int q = DEG > 0 ? DEG : a.q;
    const int sv = __float_as_int(recv[0]);
    float d0[4] = {0.f, 0.f, 0.f, 0.f}, du[3] = {0.f, 0.f, 0.f}, dv[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < S; ++s) {
        if (s > q) break;
        const float* t = trow + (sv - q + s) * 6;
        const float bn = recv[1 + s], bd = recv[1 + S + s];
        d0[0] += bn * t[0]; d0[1] += bn * t[1]; d0[2] += bn * t[2]; d0[3] += bn * wrow;
        du[0] += bn * t[3]; du[1] += bn * t[4]; du[2] += bn * t[5];
        dv[0] += bd * t[0]; dv[1] += bd * t[1]; dv[2] += bd * t[2];
    }
    S0[0] = d0[0]; S0[1] = d0[1]; S0[2] = d0[2]; S0[3] = d0[3];
    Su[0] = du[0]; Su[1] = du[1]; Su[2] = du[2];
    Sv[0] = dv[0]; Sv[1] = dv[1]; Sv[2] = dv[2];
}

// The scattered scheme's point: its own u- and v-records (line_basis, in the thread's LDS slots) contracted with the control
// net - the arithmetic of stage 1 + stage 2 for a single point, i.e. surfaces.py:592-613 in the reference's order.
template <int DEG>
__device__ __forceinline__ void eval_from_records(const NurbsArgs& a, const float* s_cp, const float* recu, const float* recv,
                                                  float* S0, float* Su, float* Sv)
{
    constexpr int S = BasisRec<DEG>::S;
    const int p = DEG > 0 ? DEG : a.p, q = DEG > 0 ? DEG : a.q;
    const int su = __float_as_int(recu[0]), sv = __float_as_int(recv[0]);
    float d0[4] = {0.f, 0.f, 0.f, 0.f}, du[3] = {0.f, 0.f, 0.f}, dv[3] = {0.f, 0.f, 0.f};
#pragma unroll 1      // (the rare scheme: keep its register footprint below the tensor-product scheme's)
    for (int s = 0; s <= q; ++s) {
        float t[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < S; ++r) {
            if (r > p) break;
            const float* c3 = s_cp + ((su - p + r) * a.nv + (sv - q + s)) * 3;
            const float bn = recu[1 + r], bd = recu[1 + S + r];
            t[0] += bn * c3[0]; t[1] += bn * c3[1]; t[2] += bn * c3[2];
            t[3] += bd * c3[0]; t[4] += bd * c3[1]; t[5] += bd * c3[2];
        }
        const float bn = recv[1 + s], bd = recv[1 + S + s];
        d0[0] += bn * t[0]; d0[1] += bn * t[1]; d0[2] += bn * t[2]; d0[3] += bn * recu[1 + 2 * S];
        du[0] += bn * t[3]; du[1] += bn * t[4]; du[2] += bn * t[5];
        dv[0] += bd * t[0]; dv[1] += bd * t[1]; dv[2] += bd * t[2];
    }
    S0[0] = d0[0]; S0[1] = d0[1]; S0[2] = d0[2]; S0[3] = d0[3];
    Su[0] = du[0]; Su[1] = du[1]; Su[2] = du[2];
    Sv[0] = dv[0]; Sv[1] = dv[1]; Sv[2] = dv[2];
}

constexpr int kScatterThreads = 256;     // threads of a workgroup that take points in the scattered scheme (LDS records each)

// ---- forward -------------------------------------------------------------------------------------------------------------

// One WAVE (a 64-thread workgroup: its barriers are free) per (facet, group of grid rows); grid = H * F * groups.  Group g
// takes grid rows [g ceil(Mu / groups), ...), i.e. points [row0 Mv, row1 Mv).  The wave finds the row length, builds the
// column bases and its rows' bases (one grid line per lane), then alternates stage 1 (a pass of rows) and stage 2 (their
// points).  Every point's coordinates are compared with its row's u and its column's v as it is evaluated; if any point of
// the wave's rows is off the grid, the wave evaluates its points again one by one (scattered scheme) - so a point list that
// is not a grid costs one wasted pass, never a wrong result, and nobody has to say what kind of list it is.
template <int DEG>
__global__ __launch_bounds__(kNurbsFwdBlock) void nurbs_fwd_kernel(NurbsArgs a, float4* __restrict__ points,
                                                                   float4* __restrict__ normals)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int W = BasisRec<DEG>::words;
    const int hf = blockIdx.x / a.groups, group = blockIdx.x % a.groups;
    float *s_cp, *s_ku, *s_kv, *s_B;
    stage_facet(a, hf, lds, s_cp, s_ku, s_kv, s_B);
    const int h = hf / a.F, f = hf % a.F;
    const float* uvp = a.uv + (int64_t)h * a.uv_sh + (int64_t)f * a.uv_sf;
    const int p = DEG > 0 ? DEG : a.p, q = DEG > 0 ? DEG : a.q;
    const int n0 = nurbs_f64_offset(a);
    // the points this wave evaluates when the list is not a grid
    const int per = (a.M + a.groups - 1) / a.groups;
    int m0 = min(a.M, group * per), m1 = min(a.M, m0 + per);
    const int Mv = a.grid_mode ? find_row_length(uvp, a.M) : 0;
    if (Mv > 0 && a.M % Mv == 0) {
        const int Mu = a.M / Mv;
        const int rpg = (Mu + a.groups - 1) / a.groups;
        const int i0 = min(Mu, group * rpg), i1 = min(Mu, i0 + rpg);
        // LDS behind the facet tables: [column bases Mv][row bases rpg][temps rows_pass x nv x 6]
        float* s_bv = lds + n0;
        float* s_bu = s_bv + Mv * W;
        float* s_temp = s_bu + rpg * W;
        const int room = a.lds_floats - (int)(s_temp - lds);
        int rows_pass = room > 0 ? min(rpg, room / (a.nv * 6)) : 0;
        rows_pass = min(rows_pass, max(1, 128 / a.nv));          // ~two lane-steps of (row, column) pairs per pass
        if (rows_pass >= 1) {                                     // (all groups of a facet decide alike: sized by rpg)
            m0 = i0 * Mv; m1 = i1 * Mv;
            if (i0 >= i1) return;
            for (int t = threadIdx.x; t < (i1 - i0) + Mv; t += blockDim.x) {
                if (t < Mv) line_basis<DEG>(a, uvp[2 * (int64_t)t + 1], s_kv, a.nv, q, a.n_unique_v, s_bv + t * W);
                else line_basis<DEG>(a, uvp[2 * (int64_t)(i0 + t - Mv) * Mv], s_ku, a.nu, p, a.n_unique_u, s_bu + (t - Mv) * W);
            }
            __syncthreads();
            const float inv_mv = 1.0f / (float)Mv;
            bool off_grid = false;
            for (int r0 = i0; r0 < i1; r0 += rows_pass) {
                const int rs = min(rows_pass, i1 - r0);
                stage1_rows<DEG>(a, s_cp, s_bu, i0, r0, rs, s_temp, threadIdx.x, blockDim.x);
                __syncthreads();
                for (int idx = threadIdx.x; idx < rs * Mv; idx += blockDim.x) {
                    const int il = (int)(((float)idx + 0.5f) * inv_mv), j = idx - il * Mv;
                    const float* recu = s_bu + (r0 + il - i0) * W;
                    const float* recv = s_bv + j * W;
                    const int m = (r0 + il) * Mv + j;
                    const float2 x = *reinterpret_cast<const float2*>(uvp + 2 * (int64_t)m);
                    off_grid |= !(x.x == recu[W - 1]) || !(x.y == recv[W - 1]);
                    float S0[4], Su[3], Sv[3];
                    stage2_point<DEG>(a, s_temp + il * a.nv * 6, recv, recu[W - 2], S0, Su, Sv);
                    finish_point(a, S0, Su, Sv, s_B, hf, h, m, points, normals);
                }
                __syncthreads();
            }
            if (__syncthreads_or(off_grid ? 1 : 0) == 0) return;
        }
    }
    // scattered scheme (the thread's two basis records live in LDS, behind the facet tables)
    float* recu = lds + n0 + threadIdx.x * 2 * W;
    float* recv = recu + W;
    for (int m = m0 + threadIdx.x; m < m1; m += blockDim.x) {
        const float2 xy = *reinterpret_cast<const float2*>(uvp + 2 * (int64_t)m);
        line_basis<DEG>(a, xy.x, s_ku, a.nu, p, a.n_unique_u, recu);
        line_basis<DEG>(a, xy.y, s_kv, a.nv, q, a.n_unique_v, recv);
        float S0[4], Su[3], Sv[3];
        eval_from_records<DEG>(a, s_cp, recu, recv, S0, Su, Sv);
        finish_point(a, S0, Su, Sv, s_B, hf, h, m, points, normals);
    }
}

// ---- backward ------------------------------------------------------------------------------------------------------------

// Scattered scheme: one point per thread and step, 3 (p+1)(q+1) double LDS atomics per point into the facet's gradient net
// (ds_add_f64; the order of the adds is not fixed - the sums are the same to fp32 output precision, not bit-reproducible).
// (Round 3 merged the sums of consecutive points of a knot-span cell in registers first - 48 accumulators that set the
// kernel's register count; the scheme now serves SurfaceGenerator.fit_nurbs only, not the epoch.)
// LDS: [facet tables][gradient net: ncp doubles][two basis records per working thread]
template <int DEG>
__device__ __forceinline__ void nurbs_bwd_scattered(const NurbsArgs& a, int hf, float* lds, const float* s_cp, const float* s_ku,
                                                    const float* s_kv, const float* s_B, const float4* __restrict__ g_points,
                                                    const float4* __restrict__ g_normals, float* __restrict__ g_cp)
{
    constexpr int S = BasisRec<DEG>::S, W = BasisRec<DEG>::words;
    const int ncp = a.nu * a.nv * 3;
    const int n0 = nurbs_f64_offset(a);
    double* s_g = reinterpret_cast<double*>(lds + n0);                     // 8-byte aligned
    __syncthreads();
    for (int i = threadIdx.x; i < ncp; i += blockDim.x) s_g[i] = 0.0;
    __syncthreads();
    const int p = DEG > 0 ? DEG : a.p, q = DEG > 0 ? DEG : a.q;
    const int h = hf / a.F, f = hf % a.F;
    const float* uvp = a.uv + (int64_t)h * a.uv_sh + (int64_t)f * a.uv_sf;
    // neighbouring points share their control points: thread t takes points (t K) mod M, ((t + T) K) mod M, ... with K
    // coprime to M (a bijection), so that the lanes of a wave add to different cells
    int K = 1;
    {
        const int primes[8] = {61, 59, 53, 47, 43, 41, 37, 31};
#pragma unroll
        for (int i = 7; i >= 0; --i)
            if (a.M % primes[i] != 0) K = primes[i];
    }
    const int team = min((int)blockDim.x, kScatterThreads);
    if ((int)threadIdx.x < team) {
        float* recu = lds + n0 + 2 * ncp + threadIdx.x * 2 * W;
        float* recv = recu + W;
        for (int t = threadIdx.x; t < a.M; t += team) {
            const int m = (int)(((int64_t)t * K) % a.M);
            const float2 xy = *reinterpret_cast<const float2*>(uvp + 2 * (int64_t)m);
            line_basis<DEG>(a, xy.x, s_ku, a.nu, p, a.n_unique_u, recu);
            line_basis<DEG>(a, xy.y, s_kv, a.nv, q, a.n_unique_v, recv);
            float S0[4], Su[3], Sv[3], gS[3], gSu[3], gSv[3];
            eval_from_records<DEG>(a, s_cp, recu, recv, S0, Su, Sv);
            point_adjoint(a, S0[3], Su, Sv, s_B, h, g_points[(int64_t)hf * a.M + m], g_normals[(int64_t)hf * a.M + m], gS, gSu, gSv);
            const int su = __float_as_int(recu[0]), sv = __float_as_int(recv[0]);
#pragma unroll 1
            for (int r = 0; r <= p; ++r) {
#pragma unroll 1
                for (int s_ = 0; s_ <= q; ++s_) {
                    const float nu_r = recu[1 + r], du_r = recu[1 + S + r], nv_s = recv[1 + s_], dv_s = recv[1 + S + s_];
                    const float w00 = nu_r * nv_s, w10 = du_r * nv_s, w01 = nu_r * dv_s;
                    double* g3 = s_g + ((su - p + r) * a.nv + (sv - q + s_)) * 3;
#pragma unroll
                    for (int k = 0; k < 3; ++k) atomicAdd(g3 + k, (double)(w00 * gS[k] + w10 * gSu[k] + w01 * gSv[k]));
                }
            }
        }
    }
    __syncthreads();
    float* out = g_cp + (int64_t)hf * ncp;
    for (int i = threadIdx.x; i < ncp; i += blockDim.x) out[i] = (float)s_g[i];
}

// One workgroup per (h,f).  Tensor-product scheme = the adjoint of the forward's two stages.  The workgroup's WAVES take
// strips of grid rows on their own - no workgroup barrier while the points are processed:
//   temps    stage 1 of the strip's rows (recomputation)
//   points   (lane <-> point)  Su, Sv, w from the temps; (gS, gSu, gSv) -> the wave's LDS strip [point][9]
//   stage A  (lane <-> (row, control column c, component k) x half of the column range)
//            gT[i][c][k] = sum_j Nv_j[c - sv_j + q] gS[i][j][k] + Dv_j[..] gSv[i][j][k],  gT[i][c][3 + k] = sum_j Nv_j[..] gSu[i][j][k]
//            j in index order over the grid columns whose span covers c (two halves, lower + upper)
// then ONE barrier and
//   stage B  (thread <-> control-point component)
//            gCP[a][c][k] = sum_i Nu_i[a - su_i + p] gT[i][c][k] + Du_i[..] gT[i][c][3 + k]      i in index order
// Every output element has ONE owner that adds in a fixed order: no atomics, bit-reproducible gradients.
template <int DEG>
__global__ __launch_bounds__(kNurbsBwdBlock) void nurbs_bwd_kernel(NurbsArgs a, const float4* __restrict__ g_points,
                                                                   const float4* __restrict__ g_normals,
                                                                   float* __restrict__ g_cp)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int S = BasisRec<DEG>::S, W = BasisRec<DEG>::words;
    const int hf = blockIdx.x;
    float *s_cp, *s_ku, *s_kv, *s_B;
    stage_facet(a, hf, lds, s_cp, s_ku, s_kv, s_B);
    const int h = hf / a.F, f = hf % a.F;
    const float* uvp = a.uv + (int64_t)h * a.uv_sh + (int64_t)f * a.uv_sf;
    const int p = DEG > 0 ? DEG : a.p, q = DEG > 0 ? DEG : a.q;
    const int n0 = nurbs_f64_offset(a);
    const int ncp = a.nu * a.nv * 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int Mv = a.grid_mode ? find_row_length(uvp, a.M) : 0;          // (every wave for itself: the same number)
    if (Mv > 0 && a.M % Mv == 0) {
        const int Mu = a.M / Mv;
        // LDS behind the facet tables: [flag][jlo, jhi per control column: 2 nv ints][ilo, ihi per control row: 2 nu ints]
        //   [row bases Mu][column bases Mv][gT Mu x nv x 6][per wave: temps rs x nv x 6, point gradients rs x Mv x 9]
        int* s_flag = reinterpret_cast<int*>(lds + n0);
        int* s_jr = s_flag + 2;
        int* s_ir = s_jr + 2 * a.nv;
        float* s_bu = reinterpret_cast<float*>(s_ir + 2 * a.nu);
        float* s_bv = s_bu + Mu * W;
        float* s_gt = s_bv + Mv * W;
        float* s_wave = s_gt + Mu * a.nv * 6;
        const int rs = max(1, 64 / Mv);                                   // rows per strip: one lane-step of points
        const int per_wave = rs * (a.nv * 6 + Mv * 9);
        if ((int)(s_wave - lds) + nwaves * per_wave <= a.lds_floats) {
            if (threadIdx.x == 0) s_flag[0] = 0;
            for (int t = threadIdx.x; t < 2 * a.nv + 2 * a.nu; t += blockDim.x) s_jr[t] = (t & 1) ? -1 : 0x7fffffff;
            for (int t = threadIdx.x; t < Mu + Mv; t += blockDim.x) {
                if (t < Mu) line_basis<DEG>(a, uvp[2 * (int64_t)t * Mv], s_ku, a.nu, p, a.n_unique_u, s_bu + t * W);
                else line_basis<DEG>(a, uvp[2 * (int64_t)(t - Mu) + 1], s_kv, a.nv, q, a.n_unique_v, s_bv + (t - Mu) * W);
            }
            __syncthreads();
            // which grid columns / rows touch control column c / control row a (any order of the grid lines is fine: the range
            // only bounds the loops, the band test inside them decides)
            for (int t = threadIdx.x; t < Mu + Mv; t += blockDim.x) {
                const bool is_u = t < Mu;
                const int line = is_u ? t : t - Mu;
                const int span = __float_as_int((is_u ? s_bu : s_bv)[line * W]);
                const int deg = is_u ? p : q;
                int* rng = is_u ? s_ir : s_jr;
                for (int r = 0; r <= deg; ++r) {
                    atomicMin(&rng[2 * (span - deg + r)], line);
                    atomicMax(&rng[2 * (span - deg + r) + 1], line);
                }
            }
            __syncthreads();
            float* s_temp = s_wave + wave * per_wave;
            float* s_pg = s_temp + rs * a.nv * 6;
            const float inv_mv = 1.0f / (float)Mv;
            const int n_strips = (Mu + rs - 1) / rs;
            bool off_grid = false;
            // a strip is at most one lane-step of points when a row has at most 64 of them: then the NEXT strip's gradients and
            // coordinates are requested before this strip is worked on (their latency would otherwise be exposed once per strip)
            const bool one_step = rs * Mv <= 64;
            const float4* __restrict__ gpf = g_points + (int64_t)hf * a.M;
            const float4* __restrict__ gnf = g_normals + (int64_t)hf * a.M;
            float4 gp_next = make_float4(0.f, 0.f, 0.f, 0.f), gn_next = gp_next;
            float2 x_next = make_float2(0.f, 0.f);
            if (one_step && wave < n_strips) {
                const int m = wave * rs * Mv + lane;
                if (lane < min(rs, Mu - wave * rs) * Mv) { gp_next = gpf[m]; gn_next = gnf[m]; x_next = *reinterpret_cast<const float2*>(uvp + 2 * (int64_t)m); }
            }
            for (int strip = wave; strip < n_strips; strip += nwaves) {
                const int r0 = strip * rs, rows = min(rs, Mu - r0);
                const float4 gp_now = gp_next, gn_now = gn_next;
                const float2 x_now = x_next;
                if (one_step && strip + nwaves < n_strips) {
                    const int r0n = (strip + nwaves) * rs;
                    const int m = r0n * Mv + lane;
                    if (lane < min(rs, Mu - r0n) * Mv) { gp_next = gpf[m]; gn_next = gnf[m]; x_next = *reinterpret_cast<const float2*>(uvp + 2 * (int64_t)m); }
                }
                stage1_rows<DEG>(a, s_cp, s_bu, 0, r0, rows, s_temp, lane, 64);
                wave_sync();
                for (int idx = lane; idx < rows * Mv; idx += 64) {
                    const int il = (int)(((float)idx + 0.5f) * inv_mv), j = idx - il * Mv;
                    const float* recu = s_bu + (r0 + il) * W;
                    const float* recv = s_bv + j * W;
                    const int m = r0 * Mv + idx;
                    const float2 x = one_step ? x_now : *reinterpret_cast<const float2*>(uvp + 2 * (int64_t)m);
                    off_grid |= !(x.x == recu[W - 1]) || !(x.y == recv[W - 1]);
                    float S0[4], Su[3], Sv[3], gS[3], gSu[3], gSv[3];
                    stage2_point<DEG>(a, s_temp + il * a.nv * 6, recv, recu[W - 2], S0, Su, Sv);
                    point_adjoint(a, S0[3], Su, Sv, s_B, h, one_step ? gp_now : gpf[m], one_step ? gn_now : gnf[m], gS, gSu, gSv);
                    float* o = s_pg + idx * 9;
                    o[0] = gS[0]; o[1] = gS[1]; o[2] = gS[2]; o[3] = gSu[0]; o[4] = gSu[1]; o[5] = gSu[2];
                    o[6] = gSv[0]; o[7] = gSv[1]; o[8] = gSv[2];
                }
                wave_sync();
                const int n_out = rows * a.nv * 3;
                if (n_out <= 32) {
                    // two lanes per output: lane o adds the lower half of the column range, lane o + 32 the upper half
                    const int o = lane & 31, half = lane >> 5;
                    float acc_n = 0.f, acc_d = 0.f;
                    int il = 0, c = 0, k = 0;
                    if (o < n_out) {
                        il = o / (a.nv * 3);
                        const int rem = o - il * (a.nv * 3);
                        c = rem / 3; k = rem - c * 3;
                        const int jlo = s_jr[2 * c], jhi = s_jr[2 * c + 1];
                        const int jmid = (jlo + jhi + 1) >> 1;
                        const int ja = half ? jmid : jlo, jb = half ? jhi : jmid - 1;
                        for (int j = ja; j <= jb; ++j) {
                            const float* rec = s_bv + j * W;
                            const int s = c - (__float_as_int(rec[0]) - q);
                            if ((unsigned)s > (unsigned)q) continue;
                            const float* pg = s_pg + (il * Mv + j) * 9;
                            acc_n = fmaf(rec[1 + S + s], pg[6 + k], fmaf(rec[1 + s], pg[k], acc_n));
                            acc_d = fmaf(rec[1 + s], pg[3 + k], acc_d);
                        }
                    }
                    const float up_n = __shfl_xor(acc_n, 32, 64), up_d = __shfl_xor(acc_d, 32, 64);
                    if (half == 0 && o < n_out) {
                        float* gt = s_gt + ((r0 + il) * a.nv + c) * 6;
                        gt[k] = acc_n + up_n; gt[3 + k] = acc_d + up_d;
                    }
                } else {
                    for (int o = lane; o < n_out; o += 64) {
                        const int il = o / (a.nv * 3), rem = o - il * (a.nv * 3);
                        const int c = rem / 3, k = rem - c * 3;
                        const int jlo = s_jr[2 * c], jhi = s_jr[2 * c + 1];
                        float acc_n = 0.f, acc_d = 0.f;
                        for (int j = jlo; j <= jhi; ++j) {
                            const float* rec = s_bv + j * W;
                            const int s = c - (__float_as_int(rec[0]) - q);
                            if ((unsigned)s > (unsigned)q) continue;
                            const float* pg = s_pg + (il * Mv + j) * 9;
                            acc_n = fmaf(rec[1 + S + s], pg[6 + k], fmaf(rec[1 + s], pg[k], acc_n));
                            acc_d = fmaf(rec[1 + s], pg[3 + k], acc_d);
                        }
                        float* gt = s_gt + ((r0 + il) * a.nv + c) * 6;
                        gt[k] = acc_n; gt[3 + k] = acc_d;
                    }
                }
                wave_sync();       // (the next strip overwrites the temps and the point gradients)
            }
            if (__syncthreads_or(off_grid ? 1 : 0) == 0) {
                float* out = g_cp + (int64_t)hf * ncp;
                for (int o = threadIdx.x; o < ncp; o += blockDim.x) {
                    const int ar = o / (a.nv * 3), rem = o - ar * (a.nv * 3);
                    const int c = rem / 3, k = rem - c * 3;
                    const int ilo = s_ir[2 * ar], ihi = s_ir[2 * ar + 1];
                    float acc = 0.f;
                    for (int i = ilo; i <= ihi; ++i) {
                        const float* rec = s_bu + i * W;
                        const int r = ar - (__float_as_int(rec[0]) - p);
                        if ((unsigned)r > (unsigned)p) continue;
                        const float* gt = s_gt + (i * a.nv + c) * 6;
                        acc = fmaf(rec[1 + S + r], gt[3 + k], fmaf(rec[1 + r], gt[k], acc));
                    }
                    out[o] = acc;
                }
                return;
            }
        }
    }
    nurbs_bwd_scattered<DEG>(a, hf, lds, s_cp, s_ku, s_kv, s_B, g_points, g_normals, g_cp);
}

static bool fill_nurbs(NurbsArgs& a, const float* cp, const float* uv, int64_t uv_sh, int64_t uv_sf,
                       const float* ku, const float* kv, const float* canting, const float* transl, int p, int q,
                       int uniform, int64_t nuq_u, int64_t nuq_v, int64_t H, int64_t F, int64_t M, int64_t nu,
                       int64_t nv)
{
    if (!cp || !uv || !ku || !kv) return false;
    if (canting && !transl) return false;
    if (p < 1 || q < 1 || p > kMaxDeg || q > kMaxDeg || nu <= p || nv <= q) return false;
    if (H < 0 || F <= 0 || M <= 0 || nu > 4096 || nv > 4096 || H * F > 2147483647LL || M > 1073741823LL) return false;
    if (uniform && (nuq_u < 2 || nuq_v < 2)) return false;
    a.cp = cp; a.uv = uv; a.uv_sh = uv_sh; a.uv_sf = uv_sf; a.knots_u = ku; a.knots_v = kv;
    a.canting = canting; a.transl = transl; a.orientation = nullptr; a.p = p; a.q = q; a.uniform = uniform;
    a.n_unique_u = (int)nuq_u; a.n_unique_v = (int)nuq_v;
    a.H = (int)H; a.F = (int)F; a.M = (int)M; a.nu = (int)nu; a.nv = (int)nv;
    a.groups = 1;
    a.grid_mode = debug_env_int("ARTIST_HIP_NURBS_GRID", 1) != 0 ? 1 : 0;
    a.lds_floats = 0;
    return true;
}

// Dynamic LDS of a launch in bytes: what the scattered scheme needs, or - when the grid scheme is on - room for a roughly
// square grid of M points (column bases, the bases of a group's rows and a pass of stage-1 temps; in the backward all row
// bases, the gradient temps and a strip per wave); a grid that does not fit (long and thin) takes the scattered scheme.
static size_t nurbs_lds_bytes(const NurbsArgs& a, bool bwd, int groups, int bwd_waves)
{
    const size_t ncp = (size_t)a.nu * a.nv * 3;
    const size_t n0 = (size_t)nurbs_f64_offset(a);
    const int deg = (a.p == a.q && a.p <= 4) ? a.p : 0;
    const size_t W = 2 * (size_t)((deg > 0 ? deg : kMaxDeg) + 1) + 3;
    size_t scattered = bwd ? n0 + 2 * ncp + (size_t)kScatterThreads * 2 * W : n0 + (size_t)kNurbsFwdBlock * 2 * W;
    size_t n = scattered;
    if (a.grid_mode) {
        size_t side = 1;
        while (side * side < (size_t)a.M) ++side;
        size_t grid;
        if (!bwd) {
            const size_t rows = (side + groups - 1) / groups;
            grid = n0 + (rows + side) * W + std::min<size_t>(rows, std::max<size_t>(1, 128 / a.nv)) * a.nv * 6;
        } else {
            const size_t rs = std::max<size_t>(1, 64 / side);
            grid = n0 + 2 + 2 * (size_t)(a.nu + a.nv) + 2 * side * W + side * a.nv * 6 + bwd_waves * rs * (a.nv * 6 + side * 9);
        }
        grid += 64;
        if (grid * sizeof(float) <= 64 * 1024) n = std::max(n, grid);
    }
    return n * sizeof(float);
}

}  // namespace art

using namespace art;

#define ART_DISPATCH_DEG(KERNEL, grid, block, lds, stream, ...)                                                 \
    do {                                                                                                        \
        const int deg__ = (a.p == a.q && a.p <= 4) ? a.p : 0;                                                   \
        switch (deg__) {                                                                                        \
            case 1: hipLaunchKernelGGL(KERNEL<1>, grid, dim3(block), lds, stream, __VA_ARGS__); break;          \
            case 2: hipLaunchKernelGGL(KERNEL<2>, grid, dim3(block), lds, stream, __VA_ARGS__); break;          \
            case 3: hipLaunchKernelGGL(KERNEL<3>, grid, dim3(block), lds, stream, __VA_ARGS__); break;          \
            case 4: hipLaunchKernelGGL(KERNEL<4>, grid, dim3(block), lds, stream, __VA_ARGS__); break;          \
            default: hipLaunchKernelGGL(KERNEL<0>, grid, dim3(block), lds, stream, __VA_ARGS__); break;         \
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
    // waves per facet: as few as still give every SIMD ~4 waves (each wave stages the control net, builds the canting basis
    // and the column bases for itself), at least ~128 points each
    {
        int groups = debug_env_int("ARTIST_HIP_NURBS_GROUPS", 0);
        if (groups <= 0) {
            groups = 1;
            while ((int64_t)H * F * groups < 4096 && M / (groups + 1) >= 128) ++groups;
        }
        a.groups = (int)std::min<int64_t>(groups, M);
    }
    const size_t lds = nurbs_lds_bytes(a, false, a.groups, 0);
    if (lds > 64 * 1024) return ART_EUNSUPPORTED;
    a.lds_floats = (int)(lds / sizeof(float));
    const int64_t blocks = (int64_t)H * F * a.groups;
    if (blocks > 2147483647LL) return ART_EINVAL;
    ART_DISPATCH_DEG(nurbs_fwd_kernel, dim3((unsigned)blocks), kNurbsFwdBlock, lds, stream, a, reinterpret_cast<float4*>(points),
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
    // waves per workgroup (= per facet): 4, or 8 for small fields (fewer facets than the chip holds workgroups)
    int block = debug_env_int("ARTIST_HIP_NURBS_BWD_BLOCK", 0);
    if (block != 256 && block != 512 && block != 128 && block != 64) block = H * F < 1024 ? 512 : 256;
    const size_t lds = nurbs_lds_bytes(a, true, 1, block / 64);
    if (lds > 64 * 1024) return ART_EUNSUPPORTED;
    a.lds_floats = (int)(lds / sizeof(float));
    ART_DISPATCH_DEG(nurbs_bwd_kernel, dim3((unsigned)(H * F)), block, lds, stream, a,
                     reinterpret_cast<const float4*>(grad_points), reinterpret_cast<const float4*>(grad_normals),
                     grad_control_points);
    ART_HIP(hipGetLastError());
    return ART_OK;
}
