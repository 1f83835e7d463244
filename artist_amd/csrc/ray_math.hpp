// ray_math.hpp - per-ray arithmetic of the heliostat trace, shared by the forward and backward
// kernels.  gfx950 device code only.
//
// The op ORDER and the separate roundings follow the reference's eager-PyTorch chain (each
// ATen op rounds to fp32), so results agree with the reference CPU path to the last bits that
// sinf/cosf allow.  The translation unit is compiled with -ffp-contract=off; where a fused
// multiply-add is wanted (backward accumulations) it is written explicitly with fmaf().
//
// Reference (ARTIST v2.0.0):
//   reflect                artist/raytracing/geometry.py:32-41
//   scatter                artist/geometry/transforms.py:52-83 + heliostat_ray_tracer.py:547-552
//   plane intersection     artist/raytracing/geometry.py:116-197
//   bilinear weights       artist/raytracing/heliostat_ray_tracer.py:674-728
#pragma once
#include <hip/hip_runtime.h>

namespace art {

// Per-heliostat constants (wave-uniform -> SGPRs).
struct Plane {
    float cx, cy, cz;       // plane centre
    float mx, my, mz;       // plane normal
    float w, h;             // plane dimensions (east, up)
    float half_w, half_h;   // dims / 2                       (geometry.py:150,156)
    float inv_w, inv_h;     // RN(1/w), RN(1/h) for div_const()
    float wm1, hm1;         // float(resolution - 1)          (geometry.py:168,173)
    float mag;              // ray magnitude                  (heliostat_ray_tracer.py:556-560)
    float k_ext, k_refl;    // float(1 - extinction), float(reflectivity)  (:485-486)
};

__device__ __forceinline__ Plane load_plane(const float* __restrict__ centers, const float* __restrict__ pnormals,
                                            const float* __restrict__ dims, int t, int W, int Hh, float mag,
                                            float k_ext, float k_refl)
{
    Plane pl;
    pl.cx = centers[4 * t + 0]; pl.cy = centers[4 * t + 1]; pl.cz = centers[4 * t + 2];
    pl.mx = pnormals[4 * t + 0]; pl.my = pnormals[4 * t + 1]; pl.mz = pnormals[4 * t + 2];
    pl.w = dims[2 * t + 0]; pl.h = dims[2 * t + 1];
    pl.half_w = pl.w / 2.0f; pl.half_h = pl.h / 2.0f;
    pl.inv_w = 1.0f / pl.w; pl.inv_h = 1.0f / pl.h;
    pl.wm1 = (float)(W - 1); pl.hm1 = (float)(Hh - 1);
    pl.mag = mag; pl.k_ext = k_ext; pl.k_refl = k_refl;
    return pl;
}

// a / b for a divisor b whose correctly rounded reciprocal y = RN(1/b) is known (Markstein): q0 = RN(a y),
// r = a - b q0 (exact in an FMA), q = RN(q0 + r y).  The result is the IEEE-correctly-rounded quotient -
// bit-identical to `a / b` - for normal operands (checked on 2e9 random pairs incl. all-ones significands,
// tools/div_const_check.c), at 3 instructions instead of the ~10 of the generic division expansion.
__device__ __forceinline__ float div_const(float a, float b, float y)
{
    const float q0 = a * y;
    const float r = fmaf(-b, q0, a);
    return fmaf(r, y, q0);
}

// geometry.py:32-41: d = i - 2 (i.n) n over all four components (torch.sum over dim -1).
__device__ __forceinline__ void reflect(const float4 i, const float4 n, float4& d, float& s)
{
    s = ((i.x * n.x + i.y * n.y) + i.z * n.z) + i.w * n.w;
    const float s2 = 2.0f * s;
    d.x = i.x - s2 * n.x; d.y = i.y - s2 * n.y; d.z = i.z - s2 * n.z; d.w = i.w - s2 * n.w;
}

// geometry.py:126-128: (c - o).m
__device__ __forceinline__ float plane_numer(const Plane& pl, const float4 o)
{
    return ((pl.cx - o.x) * pl.mx + (pl.cy - o.y) * pl.my) + (pl.cz - o.z) * pl.mz;
}

struct Rot {   // rows of rotate_distortions(e,u), transforms.py:67-74
    float cu, su, ce, se, m10, m11, m20, m21;
};

// sin/cos of a scatter angle.  Sun-shape angles are milliradians, so the common case is a short
// Taylor kernel evaluated with fused multiply-adds: for |x| <= 2^-3 the truncation error is below
// 2e-12 and the result is the correctly rounded value except in ~1 % of the cases (then 1 ULP) -
// the same quality as the libm the reference calls.  Larger angles take the full-range OCML path.
constexpr float kSmallAngle = 0.125f;

__device__ __forceinline__ void sincos_full(float x, float& s, float& c)
{
    s = sinf(x);
    c = cosf(x);
}

__device__ __forceinline__ void sincos_angle(float x, float& s, float& c)
{
    if (__builtin_expect(fabsf(x) <= kSmallAngle, 1)) {
        const float z = x * x;
        const float ps = fmaf(z, fmaf(z, -1.98412698e-4f, 8.33333333e-3f), -1.66666667e-1f);
        const float pc = fmaf(z, fmaf(z, -1.38888889e-3f, 4.16666667e-2f), -0.5f);
        s = fmaf(x * z, ps, x);
        c = fmaf(z, pc, 1.0f);
    } else {
        sincos_full(x, s, c);
    }
}

__device__ __forceinline__ void sincos_small(float x, float& s, float& c)
{
    const float z = x * x;
    const float ps = fmaf(z, fmaf(z, -1.98412698e-4f, 8.33333333e-3f), -1.66666667e-1f);
    const float pc = fmaf(z, fmaf(z, -1.38888889e-3f, 4.16666667e-2f), -0.5f);
    s = fmaf(x * z, ps, x);
    c = fmaf(z, pc, 1.0f);
}

// true iff `cond` holds for at least one active lane; the result is wave-uniform, so `if (wave_any(c))`
// compiles to a scalar branch (s_cbranch_scc*) that the hot path falls through without touching EXEC.
__device__ __forceinline__ bool wave_any(bool cond) { return __builtin_amdgcn_ballot_w64(cond) != 0ull; }

__device__ __forceinline__ Rot make_rot(float e, float u)
{
    Rot m;
    // One wave-uniform test for both angles keeps the full-range code (and its EXEC juggling) off the
    // hot path; NaN angles compare false and take the full-range branch.
    const bool small = fmaxf(fabsf(e), fabsf(u)) <= kSmallAngle;
    if (__builtin_expect(wave_any(!small), 0)) {
        sincos_angle(e, m.se, m.ce);
        sincos_angle(u, m.su, m.cu);
    } else {
        sincos_small(e, m.se, m.ce);
        sincos_small(u, m.su, m.cu);
    }
    m.m10 = m.ce * m.su; m.m11 = m.ce * m.cu; m.m20 = m.se * m.su; m.m21 = m.se * m.cu;
    return m;
}

// Same with the range test hoisted by the caller (one test per group of four rays): SMALL = every angle of the
// group is within the Taylor kernel's range on every lane.
template <bool SMALL>
__device__ __forceinline__ Rot make_rot_t(float e, float u)
{
    if constexpr (!SMALL) return make_rot(e, u);
    Rot m;
    sincos_small(e, m.se, m.ce);
    sincos_small(u, m.su, m.cu);
    m.m10 = m.ce * m.su; m.m11 = m.ce * m.cu; m.m20 = m.se * m.su; m.m21 = m.se * m.cu;
    return m;
}

// n / a, correctly rounded, without the range scaling of the generic IEEE sequence (v_div_scale / v_div_fixup): the
// same Newton + two residual corrections, valid while neither operand nor the quotient leaves the normal range -
// path lengths and direction cosines here are ~1e-3 .. 1e3.  Bit-identical to n / a on 6e8 random operand pairs
// with every reciprocal seed within 1 ULP (tools/fastdiv_check.c); a denormal cosine yields NaN instead of a
// huge quotient, and both fail the validity test the same way.
__device__ __forceinline__ float div_noscale(float n, float a)
{
    const float y0 = __builtin_amdgcn_rcpf(a);
    const float e = fmaf(-a, y0, 1.0f);
    const float y = fmaf(e, y0, y0);
    float q = n * y;
    float r = fmaf(-a, q, n);
    q = fmaf(r, y, q);
    r = fmaf(-a, q, n);
    return fmaf(r, y, q);
}

// The same quotient, bit for bit, with the refined reciprocal y ~ 1 / a handed back (the backward pass multiplies by it
// where autograd divides: gradients are compared with a tolerance, the quotient is not).
__device__ __forceinline__ float div_noscale_rcp(float n, float a, float& y)
{
    const float y0 = __builtin_amdgcn_rcpf(a);
    const float e = fmaf(-a, y0, 1.0f);
    y = fmaf(e, y0, y0);
    float q = n * y;
    float r = fmaf(-a, q, n);
    q = fmaf(r, y, q);
    r = fmaf(-a, q, n);
    return fmaf(r, y, q);
}

// heliostat_ray_tracer.py:547-552: 4-term dot products, k = 0..3 sequential; the w column of the
// matrix is zero for rows 0..2 and d.w is finite, so "+ 0*d.w" is an exact no-op apart from the
// sign of a zero result, which nothing downstream observes.
__device__ __forceinline__ void scatter(const Rot& m, const float4 d, float& rx, float& ry, float& rz)
{
    rx = m.cu * d.x + (-m.su) * d.y;
    ry = (m.m10 * d.x + m.m11 * d.y) + (-m.se) * d.z;
    rz = (m.m20 * d.x + m.m21 * d.y) + m.ce * d.z;
}

struct Hit {
    float a;        // r . m  (geometry.py:116-118)
    float t;        // intersection distance before the valid mask (geometry.py:131)
    float be, bu;   // bitmap coordinates after mask + e-flip (geometry.py:186-197)
    float I0;       // mag * (-a) * valid (geometry.py:139,189)
    bool valid;     // geometry.py:178-184
};

__device__ __forceinline__ Hit intersect(const Plane& pl, const float4 o, float numer, float rx, float ry, float rz)
{
    Hit h;
    h.a = (rx * pl.mx + ry * pl.my) + rz * pl.mz;
    const bool front = h.a < 0.0f;
    const float den = front ? h.a : 1.0f;
    h.t = (numer / den) * (front ? 1.0f : 0.0f);
    const float hx = o.x + rx * h.t;
    const float hz = o.z + rz * h.t;
    const float te = (hx + pl.half_w) - pl.cx;
    const float tu = (hz + pl.half_h) - pl.cz;
    const float be0 = div_const(te, pl.w, pl.inv_w) * pl.wm1;
    const float bu0 = div_const(tu, pl.h, pl.inv_h) * pl.hm1;
    h.valid = (0.0f <= be0) && (be0 <= pl.wm1) && (0.0f <= bu0) && (bu0 <= pl.hm1) && front;
    const float v = h.valid ? 1.0f : 0.0f;
    h.be = pl.wm1 - be0 * v;
    h.bu = bu0 * v;
    h.I0 = (pl.mag * (-h.a)) * v;
    return h;
}

struct Splat {
    int ie, iu;                  // .long() truncation (heliostat_ray_tracer.py:674-675)
    float cle, clu, che, chu;    // :694-700
    bool on;                     // :723-728
};

__device__ __forceinline__ Splat splat_weights(float be, float bu, int W, int Hh)
{
    Splat s;
    // be, bu are finite and within [0, res-1] for valid rays; invalid rays carry (res-1, 0).
    s.ie = (int)be; s.iu = (int)bu;
    s.cle = (float)(s.ie + 1) - be;
    s.clu = (float)(s.iu + 1) - bu;
    s.che = be - (float)s.ie;
    s.chu = bu - (float)s.iu;
    s.on = (0 <= s.ie) && (s.ie + 1 < W) && (0 <= s.iu) && (s.iu + 1 < Hh);
    return s;
}

// ---------------------------------------------------------------------------------------------------
// Fused hit + bilinear weights for the production kernels.  Arithmetic of a ray that reaches the
// bitmap is exactly that of intersect() + splat_weights() above (same operations, same order, same
// roundings); what is dropped are the multiplications by the 0/1 masks, which are exact no-ops for
// such a ray, and all work for rays the reference zeroes out (not front-facing, off the plane, or on
// the last pixel row/column): those end with `on == false` and contribute nothing, as in the reference
// (geometry.py:186-197 moves them to pixel (W-1, 0), which fails heliostat_ray_tracer.py:723-728).
// ---------------------------------------------------------------------------------------------------
struct RaySplat {
    float a;                     // r . m
    float t;                     // numer / a for front-facing rays
    float I0;                    // mag * (-a)                                   (valid rays)
    float cle, clu, che, chu;    // bilinear weights                             (valid rays)
    int ie, iu;                  // low pixel indices (un-flipped flat rows)     (valid rays)
    bool valid;                  // geometry.py:178-184
    bool on;                     // valid && heliostat_ray_tracer.py:723-728
};

__device__ __forceinline__ RaySplat hit_and_weights(const Plane& pl, const float4 o, float numer, float rx, float ry,
                                                    float rz, float Wf, float Hf)
{
    RaySplat h;
    h.a = (rx * pl.mx + ry * pl.my) + rz * pl.mz;
    const bool front = h.a < 0.0f;
    const float den = front ? h.a : 1.0f;
    h.t = numer / den;
    const float hx = o.x + rx * h.t;
    const float hz = o.z + rz * h.t;
    const float te = (hx + pl.half_w) - pl.cx;
    const float tu = (hz + pl.half_h) - pl.cz;
    const float be0 = div_const(te, pl.w, pl.inv_w) * pl.wm1;
    const float bu = div_const(tu, pl.h, pl.inv_h) * pl.hm1;
    h.valid = (0.0f <= be0) && (be0 <= pl.wm1) && (0.0f <= bu) && (bu <= pl.hm1) && front;
    const float be = pl.wm1 - be0;               // e-flip (geometry.py:195-197); be, bu in [0, res-1]
    h.I0 = pl.mag * (-h.a);
    const float tbe = truncf(be), tbu = truncf(bu);          // == float(int(be)) for 0 <= be < 2^24
    const float tbe1 = tbe + 1.0f, tbu1 = tbu + 1.0f;        // == float(ie + 1)
    h.cle = tbe1 - be; h.clu = tbu1 - bu;
    h.che = be - tbe; h.chu = bu - tbu;
    h.ie = (int)tbe; h.iu = (int)tbu;
    h.on = h.valid && (tbe1 < Wf) && (tbu1 < Hf);            // ie + 1 < W, iu + 1 < Hh; ie, iu >= 0 when valid
    return h;
}

// ---------------------------------------------------------------------------------------------------
// Cylindrical receivers: artist/raytracing/geometry.py:207-445, in the reference's operation order.
// Note that in fp32 this intersection is ill-conditioned (b^2 - 4ac cancels ~(distance/radius)^2
// fold); the reference has the same property and parity is asserted against that yardstick.
// ---------------------------------------------------------------------------------------------------
struct Cyl {
    float cx, cy, cz;                 // centre
    float r00, r01, r02;              // rows of `rotations`: u = n x axis   (:299-300)
    float r10, r11, r12;              //                      n
    float r20, r21, r22;              //                      axis
    float r2;                         // radius ** 2
    float height, half_height, inv_height;
    float opening, inv_opening;
    float ang0;                       // atan2(n_y, n_x) - opening / 2 (world components, :399-405)
    float wm1, hm1;
    float mag, k_ext, k_refl;
};

__device__ __forceinline__ Cyl load_cyl(const float* __restrict__ centers, const float* __restrict__ normals,
                                        const float* __restrict__ axes, const float* __restrict__ radii,
                                        const float* __restrict__ heights, const float* __restrict__ opening, int t,
                                        int W, int Hh, float mag, float k_ext, float k_refl)
{
    Cyl c;
    c.cx = centers[4 * t]; c.cy = centers[4 * t + 1]; c.cz = centers[4 * t + 2];
    const float nx = normals[4 * t], ny = normals[4 * t + 1], nz = normals[4 * t + 2];
    const float ax = axes[4 * t], ay = axes[4 * t + 1], az = axes[4 * t + 2];
    c.r00 = ny * az - nz * ay; c.r01 = nz * ax - nx * az; c.r02 = nx * ay - ny * ax;
    c.r10 = nx; c.r11 = ny; c.r12 = nz;
    c.r20 = ax; c.r21 = ay; c.r22 = az;
    c.r2 = radii[t] * radii[t];
    c.height = heights[t]; c.half_height = heights[t] / 2.0f; c.inv_height = 1.0f / heights[t];
    c.opening = opening[t]; c.inv_opening = 1.0f / opening[t];
    c.ang0 = atan2f(ny, nx) - opening[t] / 2.0f;
    c.wm1 = (float)(W - 1); c.hm1 = (float)(Hh - 1);
    c.mag = mag; c.k_ext = k_ext; c.k_refl = k_refl;
    return c;
}

// v @ rotations^T, k sequential
__device__ __forceinline__ void cyl_rot(const Cyl& c, float vx, float vy, float vz, float& ox, float& oy, float& oz)
{
    ox = (vx * c.r00 + vy * c.r01) + vz * c.r02;
    oy = (vx * c.r10 + vy * c.r11) + vz * c.r12;
    oz = (vx * c.r20 + vy * c.r21) + vz * c.r22;
}

struct CylPoint {        // per surface point (:303-305, :320)
    float ox, oy, oz;    // origin in the cylinder frame
    float c;             // ox^2 + oy^2 - r^2
};

__device__ __forceinline__ CylPoint cyl_point(const Cyl& cy, const float4 o)
{
    CylPoint p;
    cyl_rot(cy, o.x - cy.cx, o.y - cy.cy, o.z - cy.cz, p.ox, p.oy, p.oz);
    p.c = (p.ox * p.ox + p.oy * p.oy) - cy.r2;
    return p;
}

struct CylHit {
    float dx, dy, dz;        // ray direction in the cylinder frame
    float a, b, sq, t;       // quadratic, sqrt(disc + 1e-12), selected root
    float x, y, rho, nx, ny; // hit point (local xy), its radius, outward normal
    float abi;               // clamp(-d.n, 0)
    float be, bu, I0;        // outputs (masked like the reference: 0 when the ray misses)
    bool near, ok;           // root choice; hit && inside the sector
};

// sqrtf(x), correctly rounded, for 2^-96 <= x < inf (and x = 0): the compiler's own sequence - v_sqrt_f32, then the neighbours
// below / above tested with one FMA each - without its scaling of tiny arguments and its fix-up of 0 / inf / NaN (15 -> 9
// instructions).  The cylinder's discriminant carries + 1e-12 and the hit's distance from the axis is the radius.
__device__ __forceinline__ float sqrt_noscale(float x)
{
    float s = __builtin_amdgcn_sqrtf(x);
    const float sd = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, s) - 1u);
    const float su = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, s) + 1u);
    const float rd = fmaf(-sd, s, x), ru = fmaf(-su, s, x);
    s = rd <= 0.0f ? sd : s;
    s = ru > 0.0f ? su : s;
    return s;
}

// atan2(y, x) for the receiver's azimuth (geometry.py:399): |smaller| / |larger| by div_noscale, atan(a) = a P(a^2) on [0, 1]
// (degree 8 in a^2, fitted error 6e-9), then the octant.  Within 3.3 ULP of the exact value on 2e6 random pairs - what glibc's
// float atan2 shows on the same pairs by the same measure (3.2) - in ~26 instructions instead of the library routine's ~60.
// (0, 0) gives 0; signed zeros and the receiver's seam behave as atan2's for the purposes of geometry.py:407-412.
__device__ __forceinline__ float atan2_poly(float y, float x)
{
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float a = mx > 0.0f ? div_noscale(mn, mx) : 0.0f;
    const float z = a * a;
    float pz = 0.0024567015934735537f;
    pz = fmaf(pz, z, -0.01440125796943903f);
    pz = fmaf(pz, z, 0.03978104144334793f);
    pz = fmaf(pz, z, -0.07234840095043182f);
    pz = fmaf(pz, z, 0.10498936474323273f);
    pz = fmaf(pz, z, -0.14161226153373718f);
    pz = fmaf(pz, z, 0.19985906779766083f);
    pz = fmaf(pz, z, -0.33332598209381104f);
    pz = fmaf(pz, z, 0.9999998807907104f);
    float r = a * pz;
    r = ay > ax ? 1.57079632679489662f - r : r;
    r = x < 0.0f ? 3.14159265358979324f - r : r;
    return y < 0.0f ? -r : r;
}

__device__ __forceinline__ CylHit cyl_hit(const Cyl& cy, const CylPoint& p, float rx, float ry, float rz)
{
    CylHit h;
    cyl_rot(cy, rx, ry, rz, h.dx, h.dy, h.dz);                                   // :306
    h.a = h.dx * h.dx + h.dy * h.dy;                                             // :318
    h.b = 2.0f * (p.ox * h.dx + p.oy * h.dy);                                    // :319
    const float disc = h.b * h.b - (4.0f * h.a) * p.c;                           // :322
    const bool mask = (disc >= 0.0f) && (fabsf(h.a) > 1e-8f);                    // :326
    h.sq = sqrt_noscale(disc * (mask ? 1.0f : 0.0f) + 1e-12f);                   // :336
    const float two_a = 2.0f * h.a;
    // (the quotients of this function are div_noscale: the IEEE quotient bit for bit while operands and quotient stay in the
    //  normal range - metres over metres here; a ray (almost) parallel to the axis, whose quotient may leave it, is masked by
    //  |a| > 1e-8 and its roots are replaced below)
    float tn = div_noscale(-h.b - h.sq, two_a);                                  // :343-348
    // The far root is only looked at where the near one is not positive (a = dx^2 + dy^2 >= 0 and sq >= 0, so it is never the
    // smaller one): a wave whose near roots are all positive - every ray that starts outside the cylinder and runs towards
    // it - skips the second IEEE division.  Where it is computed it is the reference's value; elsewhere "+inf" selects the near
    // root exactly as the real value would.
    float tf = __builtin_inff();
    if (wave_any(!(tn > 0.0f)))
        tf = div_noscale(-h.b + h.sq, two_a);
    tn = tn > 0.0f ? tn : __builtin_inff();                                      // :351-355
    tf = tf > 0.0f ? tf : __builtin_inff();
    h.near = tn <= tf;
    float t = h.near ? tn : tf;                                                  // :356
    const bool hit = (fabsf(t) < __builtin_inff()) && mask;                      // :357-359 (isfinite)
    t = hit ? t : 0.0f;                                                          // :360-364
    h.t = t;
    h.x = p.ox + t * h.dx; h.y = p.oy + t * h.dy;                                // :374-381
    float z = p.oz + t * h.dz;
    h.rho = sqrt_noscale(h.x * h.x + h.y * h.y);                                 // :384-385
    h.nx = div_noscale(h.x, h.rho); h.ny = div_noscale(h.y, h.rho);
    const float dot = (-h.dx) * h.nx + (-h.dy) * h.ny;                           // :388-390 (z term is +-0)
    h.abi = dot < 0.0f ? 0.0f : dot;                                             // clamp(min=0); NaN passes through
    z = z + cy.half_height;                                                      // :397
    const float ang = atan2_poly(h.y, h.x) - cy.ang0;                            // :399-405
    const bool on = (z >= 0.0f) && (z <= cy.height) && (ang >= 0.0f) && (ang <= cy.opening);   // :407-412
    h.ok = on && hit;
    const float m = h.ok ? 1.0f : 0.0f;
    h.bu = (div_const(z, cy.height, cy.inv_height) * cy.hm1) * m;                // :414-429
    h.be = (div_const(ang, cy.opening, cy.inv_opening) * cy.wm1) * m;
    h.I0 = (cy.mag * h.abi) * m;                                                 // :433-438
    return h;
}

// Backward of cyl_hit: (g_be, g_bu, g_I0) -> gradient w.r.t. the local origin (gox, goy, goz) and the
// WORLD ray direction (grx, gry, grz).  Only called for rays with h.ok.
__device__ __forceinline__ void cyl_hit_bwd(const Cyl& cy, const CylPoint& p, const CylHit& h, float g_be, float g_bu,
                                            float g_I0, float& gox, float& goy, float& goz, float& grx, float& gry,
                                            float& grz)
{
#pragma clang fp contract(fast)
    // (reciprocals by v_rcp + one Newton step, ~1 ULP: the adjoint is compared with a tolerance, and five generic divisions were
    //  a tenth of this function's instructions)
    auto rcp_nr = [](float x) { const float y0 = __builtin_amdgcn_rcpf(x); return fmaf(fmaf(-x, y0, 1.0f), y0, y0); };
    const float g_ang = g_be * cy.wm1 * cy.inv_opening;
    const float g_z = g_bu * cy.hm1 * cy.inv_height;
    const float irho = rcp_nr(h.rho);
    const float irho2 = irho * irho;
    float g_x = -g_ang * h.y * irho2, g_y = g_ang * h.x * irho2;
    float gdx = 0.0f, gdy = 0.0f;
    if (h.abi > 0.0f) {
        const float g_abi = g_I0 * cy.mag;
        gdx = -g_abi * h.nx; gdy = -g_abi * h.ny;
        const float gnx = -g_abi * h.dx, gny = -g_abi * h.dy;
        const float dotn = h.nx * gnx + h.ny * gny;
        g_x += (gnx - h.nx * dotn) * irho; g_y += (gny - h.ny * dotn) * irho;
    }
    gox = g_x; goy = g_y; goz = g_z;
    gdx += g_x * h.t; gdy += g_y * h.t;
    const float gdz = g_z * h.t;
    const float g_t = g_x * h.dx + g_y * h.dy + g_z * h.dz;
    const float inv2a = rcp_nr(2.0f * h.a);
    float g_b = -g_t * inv2a;
    const float g_sq = (h.near ? -g_t : g_t) * inv2a;
    float g_a = -g_t * h.t * (2.0f * inv2a);
    const float g_disc = g_sq * rcp_nr(2.0f * h.sq);
    g_b += 2.0f * h.b * g_disc;
    g_a -= 4.0f * p.c * g_disc;
    const float g_c = -4.0f * h.a * g_disc;
    gdx += 2.0f * h.dx * g_a + 2.0f * p.ox * g_b;
    gdy += 2.0f * h.dy * g_a + 2.0f * p.oy * g_b;
    gox += 2.0f * h.dx * g_b + 2.0f * p.ox * g_c;
    goy += 2.0f * h.dy * g_b + 2.0f * p.oy * g_c;
    grx = gdx * cy.r00 + gdy * cy.r10 + gdz * cy.r20;
    gry = gdx * cy.r01 + gdy * cy.r11 + gdz * cy.r21;
    grz = gdx * cy.r02 + gdy * cy.r12 + gdz * cy.r22;
}

// ---------------------------------------------------------------------------------------------------
// Blocking: soft_ray_blocking_mask (artist/raytracing/blocking.py:212-354) for one ray and one rectangle.
// The reference evaluates every ray against every filtered primitive; here a ray first passes two
// exact-arithmetic rejection tests whose thresholds put the skipped contribution below 1e-11 (the mask is
// 1 - exp(-100 sum sigma), so a skipped term cannot change an fp32 result), and only rays near or inside a
// rectangle pay for the five sigmoids.
// ---------------------------------------------------------------------------------------------------
constexpr float kBlockSoftness = 1000.0f;     // blocking.py:218
constexpr float kBlockAlpha = 100.0f;         // :219
constexpr float kBlockOffset = 0.05f;         // :220
constexpr float kBlockEps = 1e-12f;           // :217
constexpr float kBlockMargin = 0.026f;        // sigmoid(-1000 * 0.026) = 5e-12

struct alignas(16) Prim {   // one blocking rectangle, 24 floats in LDS (six 128-bit reads)
    float c0x, c0y, c0z;     // corner 0
    float sux, suy, suz;     // span u = corner 1 - corner 0
    float svx, svy, svz;     // span v = corner 3 - corner 0
    float nx, ny, nz;        // plane normal
    float suu, svv, suv;     // :333-335
    float det_safe;          // :338-339
    float cx, cy, cz, rho;   // bounding sphere of the rectangle + its soft edge (per-point culling)
    float inv_det;           // RN(1 / det_safe) for div_const(): the two divisions of :340-345 in three instructions each
    float pad0, pad1, pad2;
};

__device__ __forceinline__ Prim make_prim(const float* __restrict__ corners, const float* __restrict__ spans,
                                          const float* __restrict__ normals, int k)
{
    Prim q;
    q.c0x = corners[16 * k]; q.c0y = corners[16 * k + 1]; q.c0z = corners[16 * k + 2];
    q.sux = spans[8 * k]; q.suy = spans[8 * k + 1]; q.suz = spans[8 * k + 2];
    q.svx = spans[8 * k + 4]; q.svy = spans[8 * k + 5]; q.svz = spans[8 * k + 6];
    q.nx = normals[4 * k]; q.ny = normals[4 * k + 1]; q.nz = normals[4 * k + 2];
    q.suu = (q.sux * q.sux + q.suy * q.suy) + q.suz * q.suz;
    q.svv = (q.svx * q.svx + q.svy * q.svy) + q.svz * q.svz;
    q.suv = (q.sux * q.svx + q.suy * q.svy) + q.suz * q.svz;
    const float det = q.suu * q.svv - q.suv * q.suv;
    const float sgn = det > 0.0f ? 1.0f : (det < 0.0f ? -1.0f : 0.0f);
    q.det_safe = fabsf(det) < kBlockEps ? sgn * kBlockEps : det;
    q.cx = q.c0x + 0.5f * (q.sux + q.svx); q.cy = q.c0y + 0.5f * (q.suy + q.svy); q.cz = q.c0z + 0.5f * (q.suz + q.svz);
    // half diagonals |su + sv| / 2 and |su - sv| / 2; the mask reaches 2.6 % of a span beyond the edges
    const float d1 = q.suu + q.svv + 2.0f * q.suv, d2 = q.suu + q.svv - 2.0f * q.suv;
    q.rho = 0.5f * sqrtf(fmaxf(fmaxf(d1, d2), 0.0f)) * 1.06f + 2e-3f;
    q.inv_det = 1.0f / q.det_safe;
    q.pad0 = q.pad1 = q.pad2 = 0.0f;
    return q;
}

__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

struct SoftHit {
    float den, den_safe, d;
    float offx, offy, offz, pu, pv, u, v;
    bool near;               // passes both rejection tests
};

// Plane hit (:315-324) and local coordinates (:328-345), reference operation order, in two steps so that a wave
// whose rays all pass behind / too close to the rectangle's plane skips the second one.
// (c0 - o).n of :315-324: a function of the POINT and the rectangle, not of the ray - the lean forward item computes it once per
// point for the first two rectangles of the wave's mask and hands it to soft_transmittance (same expression, same bits)
__device__ __forceinline__ float soft_plane_num(const Prim& q, float ox, float oy, float oz)
{
    return ((q.c0x - ox) * q.nx + (q.c0y - oy) * q.ny) + (q.c0z - oz) * q.nz;
}

__device__ __forceinline__ bool soft_plane(const Prim& q, float ox, float oy, float oz, float rx, float ry, float rz,
                                           SoftHit& s, bool have_num = false, float num_in = 0.0f)
{
    s.den = (rx * q.nx + ry * q.ny) + rz * q.nz;
    s.den_safe = fabsf(s.den) < kBlockEps ? (s.den >= 0.0f ? kBlockEps : -kBlockEps) : s.den;
    const float num = have_num ? num_in : soft_plane_num(q, ox, oy, oz);
    // (the IEEE quotient without the generic sequence's range scaling: |den_safe| >= 1e-12 and |num| is a distance in
    //  metres, so neither operand nor quotient leaves the normal range - bit-identical, see div_noscale)
    s.d = div_noscale(num, s.den_safe);
    return s.d > kBlockOffset - kBlockMargin;
}

__device__ __forceinline__ void soft_uv(const Prim& q, float ox, float oy, float oz, float rx, float ry, float rz,
                                        bool in_front, SoftHit& s)
{
    s.offx = (ox + s.d * rx) - q.c0x; s.offy = (oy + s.d * ry) - q.c0y; s.offz = (oz + s.d * rz) - q.c0z;
    s.pu = (s.offx * q.sux + s.offy * q.suy) + s.offz * q.suz;
    s.pv = (s.offx * q.svx + s.offy * q.svy) + s.offz * q.svz;
    // (division by a per-rectangle constant whose correctly rounded reciprocal is in the table: the IEEE quotient, bit for
    //  bit, in three instructions - div_const)
    s.u = div_const(s.pu * q.svv - s.pv * q.suv, q.det_safe, q.inv_det);
    s.v = div_const(s.pv * q.suu - s.pu * q.suv, q.det_safe, q.inv_det);
    // NaN coordinates (degenerate rectangle) fail the comparisons and are skipped; the reference would carry NaN
    s.near = in_front && s.u > -kBlockMargin && s.u < 1.0f + kBlockMargin && s.v > -kBlockMargin &&
             s.v < 1.0f + kBlockMargin;
}

struct SoftSig { float f, Au, Bu, Av, Bv, sigma_raw; };

__device__ __forceinline__ float soft_sigma(const SoftHit& s, SoftSig& g)     // :325-327, :353-361
{
    g.f = sigmoid_fast(kBlockSoftness * (s.d - kBlockOffset));
    g.Au = sigmoid_fast(kBlockSoftness * s.u); g.Bu = sigmoid_fast(kBlockSoftness * (1.0f - s.u));
    g.Av = sigmoid_fast(kBlockSoftness * s.v); g.Bv = sigmoid_fast(kBlockSoftness * (1.0f - s.v));
    g.sigma_raw = ((g.Au * g.Bu) * (g.Av * g.Bv)) * g.f;
    return __builtin_amdgcn_fmed3f(g.sigma_raw, 0.0f, 1.0f);
}

// The local coordinates of soft_uv are two linear functionals of the hit point: u(X) = (X - c0).a, v(X) = (X - c0).b.
struct alignas(16) PrimAux { float ax, ay, az, an, bx, by, bz, bn; };     // a, |a|, b, |b|

__device__ __forceinline__ PrimAux make_prim_aux(const Prim& q)
{
    PrimAux x;
    const float idet = 1.0f / q.det_safe;
    x.ax = (q.sux * q.svv - q.svx * q.suv) * idet; x.ay = (q.suy * q.svv - q.svy * q.suv) * idet;
    x.az = (q.suz * q.svv - q.svz * q.suv) * idet;
    x.bx = (q.svx * q.suu - q.sux * q.suv) * idet; x.by = (q.svy * q.suu - q.suy * q.suv) * idet;
    x.bz = (q.svz * q.suu - q.suz * q.suv) * idet;
    x.an = sqrtf(x.ax * x.ax + x.ay * x.ay + x.az * x.az);
    x.bn = sqrtf(x.bx * x.bx + x.by * x.by + x.bz * x.bz);
    return x;
}

// Which of the n rectangles can a ray leaving o within `theta` of the unit direction (dx,dy,dz) touch at all?
//  1. Sphere against cone: the centre's distance to the cone surface is perp cos(theta) - t sin(theta).  (cos_t, sin_t) =
//     (0, 0) accepts everything (no bound on the scatter angle known).
//  2. The sphere is a loose hull of a rectangle, and the 64 points of a wave spread over a facet's width, so a wave would
//     evaluate many rectangles its rays pass beside.  A ray of the cone that enters the mask does so at a point X of the
//     rectangle's plane with |X - o| <= |w| + rho, u(X), v(X) in (-margin, 1 + margin); X is within
//     delta = (|w| + rho) sin(theta) of some point Y = o + tau d (tau >= 0) of the chief ray, hence
//     |(Y - c0).n| <= |n| delta, u(Y) within |a| delta of that interval, v(Y) within |b| delta: three slabs in tau.
//     Conservative (delta carries 1 % + 1 mm of slack for the fp32 evaluation); a lane without the bit contributes
//     nothing for that rectangle, exactly like a ray that fails soft_uv's own test.
__device__ __forceinline__ unsigned cone_mask(const Prim* __restrict__ prims, const PrimAux* __restrict__ aux, int n, float ox,
                                              float oy, float oz, float dx, float dy, float dz, float cos_t, float sin_t,
                                              bool slabs = true)
{
#pragma clang fp contract(fast)
    unsigned mask = 0u;
    for (int k = 0; k < n; ++k) {
        const float wx = prims[k].cx - ox, wy = prims[k].cy - oy, wz = prims[k].cz - oz;
        const float l2 = wx * wx + wy * wy + wz * wz;
        const float t = wx * dx + wy * dy + wz * dz;
        const float perp = sqrtf(fmaxf(l2 - t * t, 0.0f));
        const float rho = prims[k].rho;
        if (!(perp * cos_t - t * sin_t <= rho || l2 <= rho * rho)) continue;
        if (slabs && cos_t > 0.0f) {
            const Prim q = prims[k];
            const PrimAux x = aux[k];
            const float delta = (sqrtf(l2) + rho) * sin_t * 1.01f + 1e-3f;
            const float rx = ox - q.c0x, ry = oy - q.c0y, rz = oz - q.c0z;
            const float f[3] = {rx * q.nx + ry * q.ny + rz * q.nz, rx * x.ax + ry * x.ay + rz * x.az,
                                rx * x.bx + ry * x.by + rz * x.bz};
            const float g[3] = {dx * q.nx + dy * q.ny + dz * q.nz, dx * x.ax + dy * x.ay + dz * x.az,
                                dx * x.bx + dy * x.by + dz * x.bz};
            const float nn = sqrtf(q.nx * q.nx + q.ny * q.ny + q.nz * q.nz);
            const float lo[3] = {-nn * delta, -kBlockMargin - x.an * delta, -kBlockMargin - x.bn * delta};
            const float hi[3] = {nn * delta, 1.0f + kBlockMargin + x.an * delta, 1.0f + kBlockMargin + x.bn * delta};
            float entry = 0.0f, exit_ = 3.0e38f;         // tau >= 0
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float inv = 1.0f / (fabsf(g[c]) > 1e-12f ? g[c] : 1e-12f);
                const float t0 = (lo[c] - f[c]) * inv, t1 = (hi[c] - f[c]) * inv;
                entry = fmaxf(entry, fminf(t0, t1));
                exit_ = fminf(exit_, fmaxf(t0, t1));
            }
            if (!(exit_ >= entry)) continue;
        }
        mask |= 1u << k;
    }
    return mask;
}

// exp(-alpha sum sigma) over the rectangles in `wave_mask` (wave-uniform union of the lanes' `mask`); `near` gets
// the rectangles whose mask this ray actually entered with a gradient (not the rays deep inside one: sigma == 1).
// (PRE: num0 / num1 = soft_plane_num of the first / second rectangle of wave_mask for this lane's point)
template <bool PRE = false>
__device__ __forceinline__ float soft_transmittance(const Prim* __restrict__ prims, unsigned wave_mask, unsigned mask,
                                                    float ox, float oy, float oz, float rx, float ry, float rz,
                                                    unsigned& near, float num0 = 0.0f, float num1 = 0.0f, float tail = 0.0f)
{
    float sum = 0.0f;
    near = 0u;
    [[maybe_unused]] int it = 0;
    for (unsigned m = wave_mask; m != 0u; m &= m - 1u) {
        const int k = __builtin_ctz(m);
        const Prim q = prims[k];                       // wave-uniform LDS address: broadcast reads
        SoftHit s;
        bool in_front;
        if constexpr (PRE) {
            in_front = soft_plane(q, ox, oy, oz, rx, ry, rz, s, it < 2, it == 0 ? num0 : num1) && ((mask >> k) & 1u);
            ++it;
        } else
        in_front = soft_plane(q, ox, oy, oz, rx, ry, rz, s) && ((mask >> k) & 1u);
        if (!wave_any(in_front)) continue;
        soft_uv(q, ox, oy, oz, rx, ry, rz, in_front, s);
        if (!wave_any(s.near)) continue;
        SoftSig g;
        const float sg = soft_sigma(s, g);
        sum += s.near ? sg : 0.0f;
        // all five sigmoids saturated at exactly 1 (x >= 17): sigma = 1 with a gradient of exactly zero - not "near"
        const bool flat = g.sigma_raw == 1.0f;
        near |= s.near && !flat ? 1u << k : 0u;
    }
    // (tail: the sigmas of the candidates beyond the tables - trace_kernels.hip, "wide" heliostats; + 0.0f otherwise: the same bits)
    return __expf(-(kBlockAlpha * (sum + tail)));
}

// wave-uniform OR of a per-lane mask over the ACTIVE lanes (n <= 32 ballots; once per point, not per ray)
__device__ __forceinline__ unsigned wave_or_mask(unsigned mask, int n)
{
    unsigned out = 0u;
    for (int k = 0; k < n; ++k)
        if (__builtin_amdgcn_ballot_w64((mask >> k) & 1u) != 0ull) out |= 1u << k;
    return out;
}

// Adjoint of sigma w.r.t. the ray (origin, direction) and the rectangle (corner 0, spans, normal).
struct SoftGrad { float ox, oy, oz, rx, ry, rz, c0[3], su[3], sv[3], n[3]; };

__device__ __forceinline__ void soft_sigma_bwd(const Prim& q, float ox, float oy, float oz, float rx, float ry, float rz,
                                               const SoftHit& s, const SoftSig& g, float g_sigma, SoftGrad& out)
{
#pragma clang fp contract(fast)
    const float k = kBlockSoftness;
    const float iu = g.Au * g.Bu, iv = g.Av * g.Bv;
    const float g_u = g_sigma * iv * g.f * (k * iu * (g.Bu - g.Au));
    const float g_v = g_sigma * iu * g.f * (k * iv * (g.Bv - g.Av));
    float g_d = g_sigma * iu * iv * (g.f * (1.0f - g.f) * k);
    const float idet = __builtin_amdgcn_rcpf(q.det_safe);       // (adjoint only: 1 ulp reciprocals, no division sequences)
    const float g_pu = (g_u * q.svv - g_v * q.suv) * idet;
    const float g_pv = (g_v * q.suu - g_u * q.suv) * idet;
    float g_svv = g_u * s.pu * idet, g_suu = g_v * s.pv * idet, g_suv = -(g_u * s.pv + g_v * s.pu) * idet;
    if (fabsf(q.det_safe) > kBlockEps) {           // det itself was used (:339)
        const float g_det = -(g_u * s.u + g_v * s.v) * idet;
        g_suu += g_det * q.svv; g_svv += g_det * q.suu; g_suv -= 2.0f * q.suv * g_det;
    }
    const float su[3] = {q.sux, q.suy, q.suz}, sv[3] = {q.svx, q.svy, q.svz}, nn[3] = {q.nx, q.ny, q.nz};
    const float off[3] = {s.offx, s.offy, s.offz}, dir[3] = {rx, ry, rz};
    const float rel[3] = {q.c0x - ox, q.c0y - oy, q.c0z - oz};
    float g_off[3], go[3], gr[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        g_off[c] = g_pu * su[c] + g_pv * sv[c];
        out.su[c] = g_pu * off[c] + 2.0f * g_suu * su[c] + g_suv * sv[c];
        out.sv[c] = g_pv * off[c] + 2.0f * g_svv * sv[c] + g_suv * su[c];
        g_d += g_off[c] * dir[c];
    }
    const float g_num = g_d * __builtin_amdgcn_rcpf(s.den_safe);
    const float g_den = fabsf(s.den) < kBlockEps ? 0.0f : -g_num * s.d;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        go[c] = g_off[c] - g_num * nn[c];
        gr[c] = s.d * g_off[c] + g_den * nn[c];
        out.c0[c] = g_num * nn[c] - g_off[c];
        out.n[c] = g_num * rel[c] + g_den * dir[c];
    }
    out.ox = go[0]; out.oy = go[1]; out.oz = go[2];
    out.rx = gr[0]; out.ry = gr[1]; out.rz = gr[2];
}

}  // namespace art
