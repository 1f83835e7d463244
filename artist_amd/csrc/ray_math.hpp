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
    pl.wm1 = (float)(W - 1); pl.hm1 = (float)(Hh - 1);
    pl.mag = mag; pl.k_ext = k_ext; pl.k_refl = k_refl;
    return pl;
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

__device__ __forceinline__ Rot make_rot(float e, float u)
{
    Rot m;
    sincos_angle(e, m.se, m.ce);
    sincos_angle(u, m.su, m.cu);
    m.m10 = m.ce * m.su; m.m11 = m.ce * m.cu; m.m20 = m.se * m.su; m.m21 = m.se * m.cu;
    return m;
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
    const float be0 = (te / pl.w) * pl.wm1;
    const float bu0 = (tu / pl.h) * pl.hm1;
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

}  // namespace art
