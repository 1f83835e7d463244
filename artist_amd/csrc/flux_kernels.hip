// flux_kernels.hip - the flux epilogue that runs on the ray tracer's bitmaps every optimisation epoch (gfx950).
//
//   crop   crop_flux_distributions_around_center (artist/flux/bitmap.py:121-246): centre of mass -> affine grid
//          -> bilinear grid_sample (align_corners=True, zeros padding), forward and backward.  The sampling grid is
//          an axis-aligned affine map, so the backward w.r.t. the bitmap is written as a GATHER (every input pixel
//          sums the few output pixels that sampled it): deterministic, no atomics - torch's grid_sample backward
//          scatters with atomics.
//   loss   PixelLoss / KLDivergenceLoss (artist/optim/loss.py:251-410) with the reduction over the two bitmap
//          dimensions, forward and backward, one workgroup per sample.
//
// All of it is HBM-bound elementwise / reduction work on [B,Hh,W] fp32: one read and one write of the bitmaps.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <mutex>
#include <vector>

#include "launch_common.hpp"
#include "flux_moments.hpp"

namespace art {

constexpr int kFluxBlock = 256;
constexpr int kReduceBlock = kMomentsBlock;     // per-bitmap reductions: one workgroup per bitmap, 16 waves to hide latency

__device__ __forceinline__ double block_sum(double v, double* s_red)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    double r = 0.0;
    for (int w = 0; w < nw; ++w) r += s_red[w];
    return r;
}

struct CropMap {       // pixel j of the output samples input coordinate ix(j); same arithmetic in every kernel
    float sx, sy, xc, yc;
    int W, Hh;
    __device__ __forceinline__ float ix(int j) const { return (((sx * lin11(j, W) + xc) + 1.0f) / 2.0f) * (float)(W - 1); }
    __device__ __forceinline__ float iy(int i) const { return (((sy * lin11(i, Hh) + yc) + 1.0f) / 2.0f) * (float)(Hh - 1); }
};

__device__ __forceinline__ CropMap make_map(const float* __restrict__ dims, const float* __restrict__ com, int b, int W,
                                            int Hh, float crop_w, float crop_h)
{
    CropMap m;
    m.sx = crop_w / fmaxf(dims[2 * b], 1e-8f);           // bitmap.py:218-225
    m.sy = crop_h / fmaxf(dims[2 * b + 1], 1e-8f);
    m.xc = com[3 * b]; m.yc = com[3 * b + 1];
    m.W = W; m.Hh = Hh;
    return m;
}

// com[b] = (x centre, y centre, sum + 1e-8) of bitmap b in normalised coordinates (:165-182).  One pass, sums in
// fp64: sum x (f / S) and (sum x f) / S differ by far less than the fp32 rounding of the reference's own sums.
__global__ __launch_bounds__(kReduceBlock) void flux_com_kernel(const float* __restrict__ flux, int Hh, int W,
                                                              float* __restrict__ com)
{
    __shared__ double s_red[16];
    const int b = blockIdx.x;
    const float* __restrict__ f = flux + (int64_t)b * Hh * W;
    double s = 0.0, xs = 0.0, ys = 0.0;
    if ((W & 3) == 0) {
        // four pixels of one row per load; (y, x) advanced without a division
        const int W4 = W >> 2;
        int x4 = threadIdx.x % W4, y = threadIdx.x / W4;
        const int dx = blockDim.x % W4, dy = blockDim.x / W4;
        const float4* __restrict__ f4 = reinterpret_cast<const float4*>(f);
        for (int k = threadIdx.x; k < Hh * W4; k += blockDim.x) {
            const float4 v = f4[k];
            const int x = 4 * x4;
            s += (double)((v.x + v.y) + (v.z + v.w));
            xs += (double)((lin11(x, W) * v.x + lin11(x + 1, W) * v.y) + (lin11(x + 2, W) * v.z + lin11(x + 3, W) * v.w));
            ys += (double)(lin11(y, Hh) * ((v.x + v.y) + (v.z + v.w)));
            x4 += dx; y += dy;
            if (x4 >= W4) { x4 -= W4; ++y; }
        }
    } else {
        int x = threadIdx.x % W, y = threadIdx.x / W;
        const int dx = blockDim.x % W, dy = blockDim.x / W;
        for (int k = threadIdx.x; k < Hh * W; k += blockDim.x) {
            const float v = f[k];
            s += (double)v;
            xs += (double)(lin11(x, W) * v);
            ys += (double)(lin11(y, Hh) * v);
            x += dx; y += dy;
            if (x >= W) { x -= W; ++y; }
        }
    }
    s = block_sum(s, s_red);
    xs = block_sum(xs, s_red);
    ys = block_sum(ys, s_red);
    if (threadIdx.x == 0) {
        const float S = (float)s + 1e-8f;
        com[3 * b] = (float)(xs / (double)S); com[3 * b + 1] = (float)(ys / (double)S); com[3 * b + 2] = S;
    }
}

// grid (ceil(W / 256), ceil(Hh / kCropRows), B): a thread walks kCropRows output pixels of one column (the column's
// sampling coordinate and weights are computed once; a row's are wave-uniform).  One row per workgroup was 256 000
// workgroups of one pixel per thread for the metric field - dispatch and map set-up dominated (0.30 ms for 0.13 ms of
// HBM traffic).
constexpr int kCropRows = 8;
__global__ __launch_bounds__(kFluxBlock) void flux_crop_fwd_kernel(const float* __restrict__ flux,
                                                                   const float* __restrict__ dims,
                                                                   const float* __restrict__ com, int Hh, int W,
                                                                   float crop_w, float crop_h, float* __restrict__ out)
{
    const int b = blockIdx.z;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= W) return;
    const CropMap m = make_map(dims, com, b, W, Hh, crop_w, crop_h);
    const float* __restrict__ f = flux + (int64_t)b * Hh * W;
    const float ix = m.ix(j);
    const float x0f = floorf(ix);
    const float tx = ix - x0f;
    const int x0 = (int)x0f;
    const bool xa = x0 >= 0 && x0 < W, xb = x0 + 1 >= 0 && x0 + 1 < W;
    const int i_end = min((int)(blockIdx.y + 1) * kCropRows, Hh);
    for (int i = blockIdx.y * kCropRows; i < i_end; ++i) {
        const float iy = m.iy(i);
        const float y0f = floorf(iy);
        const float ty = iy - y0f;
        const int y0 = (int)y0f;
        const bool ya = y0 >= 0 && y0 < Hh, yb = y0 + 1 >= 0 && y0 + 1 < Hh;
        float acc = 0.0f;
        if (ya && xa) acc += f[y0 * W + x0] * ((1.0f - tx) * (1.0f - ty));
        if (ya && xb) acc += f[y0 * W + x0 + 1] * (tx * (1.0f - ty));
        if (yb && xa) acc += f[(y0 + 1) * W + x0] * ((1.0f - tx) * ty);
        if (yb && xb) acc += f[(y0 + 1) * W + x0 + 1] * (tx * ty);
        out[(int64_t)b * Hh * W + i * W + j] = acc;
    }
}

// gcom[b] = (dL/dxc, dL/dyc): grid_sample's gradient w.r.t. its grid, summed through affine_grid's translation
__global__ __launch_bounds__(kReduceBlock) void flux_crop_bwd_com_kernel(const float* __restrict__ flux,
                                                                       const float* __restrict__ dims,
                                                                       const float* __restrict__ com,
                                                                       const float* __restrict__ grad_out, int Hh, int W,
                                                                       float crop_w, float crop_h,
                                                                       float* __restrict__ gcom)
{
    __shared__ double s_red[16];
    const int b = blockIdx.x;
    const CropMap m = make_map(dims, com, b, W, Hh, crop_w, crop_h);
    const float* __restrict__ f = flux + (int64_t)b * Hh * W;
    const float* __restrict__ g = grad_out + (int64_t)b * Hh * W;
    double gx = 0.0, gy = 0.0;
    // Four pixels per trip, their twenty loads issued before the first is used: one workgroup per bitmap leaves the
    // memory system nearly idle on a small field, so the loop lives on loads in flight (47 -> 20 us for 125 bitmaps).
    // The per-thread summation order is that of the one-pixel loop: results do not change.
    constexpr int U = 4;
    const int npx = Hh * W;
    for (int k0 = threadIdx.x; k0 < npx; k0 += U * blockDim.x) {
        float v00[U], v01[U], v10[U], v11[U], go[U], tx[U], ty[U];
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int k = k0 + q * blockDim.x;
            const bool live = k < npx;
            const int i = live ? k / W : 0, j = live ? k - i * W : 0;
            const float ix = m.ix(j), iy = m.iy(i);
            const float x0f = floorf(ix), y0f = floorf(iy);
            tx[q] = ix - x0f; ty[q] = iy - y0f;
            const int x0 = (int)x0f, y0 = (int)y0f;
            const bool xa = live && x0 >= 0 && x0 < W, xb = live && x0 + 1 >= 0 && x0 + 1 < W;
            const bool ya = y0 >= 0 && y0 < Hh, yb = y0 + 1 >= 0 && y0 + 1 < Hh;
            v00[q] = ya && xa ? f[y0 * W + x0] : 0.0f; v01[q] = ya && xb ? f[y0 * W + x0 + 1] : 0.0f;
            v10[q] = yb && xa ? f[(y0 + 1) * W + x0] : 0.0f; v11[q] = yb && xb ? f[(y0 + 1) * W + x0 + 1] : 0.0f;
            go[q] = live ? g[k] : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < U; ++q) {
            if (k0 + q * (int)blockDim.x >= npx) break;
            gx += (double)(go[q] * ((v01[q] - v00[q]) * (1.0f - ty[q]) + (v11[q] - v10[q]) * ty[q]));
            gy += (double)(go[q] * ((v10[q] - v00[q]) * (1.0f - tx[q]) + (v11[q] - v01[q]) * tx[q]));
        }
    }
    gx = block_sum(gx, s_red);
    gy = block_sum(gy, s_red);
    if (threadIdx.x == 0) {
        gcom[2 * b] = (float)(gx * (double)((float)(W - 1) / 2.0f));
        gcom[2 * b + 1] = (float)(gy * (double)((float)(Hh - 1) / 2.0f));
    }
}

// Weight with which output pixel `o` (sampling coordinate c = map(o)) read input pixel `p`.
__device__ __forceinline__ float tap(float c, int p)
{
    const float c0 = floorf(c);
    const int p0 = (int)c0;
    const float t = c - c0;
    return p == p0 ? 1.0f - t : (p == p0 + 1 ? t : 0.0f);
}

// sum_i wy(i) sum_q wx[q] g[i, j0 + q] with the K column weights held in registers
template <int K>
__device__ __forceinline__ float gather_rows(const CropMap& m, const float* __restrict__ g, int W, int x, int y, int j0,
                                             int j1, int i0, int i1)
{
    float wx[K];
#pragma unroll
    for (int q = 0; q < K; ++q) wx[q] = j0 + q <= j1 ? tap(m.ix(j0 + q), x) : 0.0f;
    float acc = 0.0f;
    for (int i = i0; i <= i1; ++i) {
        const float wy = tap(m.iy(i), y);
        if (wy == 0.0f) continue;
        const float* __restrict__ grow = g + i * W + j0;
        float row = 0.0f;
#pragma unroll
        for (int q = 0; q < K; ++q)
            if (wx[q] != 0.0f) row += grow[q] * wx[q];
        acc += row * wy;
    }
    return acc;
}

// Gradient of one INPUT pixel (x, y): direct part = sum over the output pixels that sampled it (the map is separable
// and monotone, so they form a small index rectangle), plus the part through the centre of mass.  General form, any
// crop scale; the tiled kernel below uses it for bitmaps whose scale needs more than four taps per axis.
__device__ __forceinline__ float crop_bwd_pixel(const CropMap& m, const float* __restrict__ g, float S,
                                                const float* __restrict__ gcom, int b, int Hh, int W, int x, int y)
{
    // ix(j) = sx (j - (W-1)/2) + (xc + 1)(W-1)/2 up to rounding (<< 1e-3 pixel for bitmaps up to 32768 wide):
    // candidates j with |ix(j) - x| < 1 + 4e-3; tap() decides exactly
    int j0 = 0, j1 = W - 1, i0 = 0, i1 = Hh - 1;
    if (m.sx > 1e-6f && m.sx < 1e6f) {
        const float bx = (m.xc + 1.0f) * 0.5f * (float)(W - 1) - m.sx * 0.5f * (float)(W - 1);
        const float lo = ((float)x - 1.004f - bx) / m.sx, hi = ((float)x + 1.004f - bx) / m.sx;
        if (lo > -2.0e9f && lo < 2.0e9f && hi > -2.0e9f && hi < 2.0e9f) { j0 = max(0, (int)ceilf(lo)); j1 = min(W - 1, (int)floorf(hi)); }
    }
    if (m.sy > 1e-6f && m.sy < 1e6f) {
        const float by = (m.yc + 1.0f) * 0.5f * (float)(Hh - 1) - m.sy * 0.5f * (float)(Hh - 1);
        const float lo = ((float)y - 1.004f - by) / m.sy, hi = ((float)y + 1.004f - by) / m.sy;
        if (lo > -2.0e9f && lo < 2.0e9f && hi > -2.0e9f && hi < 2.0e9f) { i0 = max(0, (int)ceilf(lo)); i1 = min(Hh - 1, (int)floorf(hi)); }
    }
    float acc = 0.0f;
    if (j1 - j0 < 4) acc = gather_rows<4>(m, g, W, x, y, j0, j1, i0, i1);
    else if (j1 - j0 < 8) acc = gather_rows<8>(m, g, W, x, y, j0, j1, i0, i1);
    else {
        for (int i = i0; i <= i1; ++i) {
            const float wy = tap(m.iy(i), y);
            if (wy == 0.0f) continue;
            float row = 0.0f;
            for (int j = j0; j <= j1; ++j) {
                const float wx = tap(m.ix(j), x);
                if (wx != 0.0f) row += g[i * W + j] * wx;
            }
            acc += row * wy;
        }
    }
    return acc + gcom[2 * b] * (lin11(x, W) - m.xc) / S + gcom[2 * b + 1] * (lin11(y, Hh) - m.yc) / S;
}

// Candidate output indices whose sampling coordinate can lie within one pixel of input index `p` (see above).
__device__ __forceinline__ void tap_range(float scale, float centre, int n, int p, int& lo_i, int& hi_i)
{
    lo_i = 0; hi_i = n - 1;
    if (scale > 1e-6f && scale < 1e6f) {
        const float b = (centre + 1.0f) * 0.5f * (float)(n - 1) - scale * 0.5f * (float)(n - 1);
        const float lo = ((float)p - 1.004f - b) / scale, hi = ((float)p + 1.004f - b) / scale;
        if (lo > -2.0e9f && lo < 2.0e9f && hi > -2.0e9f && hi < 2.0e9f) { lo_i = max(0, (int)ceilf(lo)); hi_i = min(n - 1, (int)floorf(hi)); }
    }
}

// The same gradient organised by tiles of 64 x 32 input pixels: the map is separable, so the
// first output index and the (at most four) weights of a column are shared by the 16 pixels of that column and those
// of a row by its 64 pixels - they are computed once per tile into LDS - and the sum factorises into a horizontal
// pass (4 global loads per output row and column, kept in LDS) and a vertical pass (4 LDS reads per pixel).  A bitmap
// whose crop scale needs more than four taps per axis (scale < ~0.6) takes the per-pixel form.
constexpr int kTileX = 64, kTileY = 32, kTaps = 4, kTileRows = 64, kRowUnroll = 4;
// LOSS = true: the fused crop + PixelLoss adjoint.  grad_out is then the RESIDUAL crop - truth that the forward pass kept, gcom
// the gradient of the two centre coordinates per unit of 2 gl / sum(truth), and the whole pixel is scaled by that factor at
// the end (everything here is linear in it): dL/dflux in ONE pass over the residual - no second sampling of the crop, no
// dL/dcrop written and read back (round 3: two kernels, 188 + 297 us at 1000 bitmaps).  com is then the forward's record
// [B,4] = (x centre, y centre, sum + 1e-8, sum of the measured flux).
template <bool LOSS>
__global__ __launch_bounds__(256) void flux_crop_bwd_tiled_kernel(const float* __restrict__ dims, const float* __restrict__ com,
                                                                  const float* __restrict__ gcom,
                                                                  const float* __restrict__ grad_out, int Hh, int W,
                                                                  float crop_w, float crop_h, float* __restrict__ grad_flux,
                                                                  const float* __restrict__ grad_loss, int grad_loss_stride = 1)
{
    constexpr int CS = LOSS ? 4 : 3;                 // stride of the centre record
    __shared__ int s_j0[kTileX], s_i0[kTileY];
    __shared__ float s_wx[kTileX][kTaps], s_wy[kTileY][kTaps];
    __shared__ float s_t[kTileRows][kTileX];        // horizontal pass: T[i][x] = sum_q wx[x][q] g[i][j0(x) + q]
    __shared__ int s_wide, s_ilo, s_ihi;
    __shared__ float s_gy[kTileY];                   // the row's share of the centre-of-mass term (one division per row, not per pixel)
    const int b = blockIdx.z;
    const int x0 = blockIdx.x * kTileX, y0 = blockIdx.y * kTileY;
    CropMap m;
    m.sx = crop_w / fmaxf(dims[2 * b], 1e-8f); m.sy = crop_h / fmaxf(dims[2 * b + 1], 1e-8f);
    m.xc = com[CS * b]; m.yc = com[CS * b + 1]; m.W = W; m.Hh = Hh;
    const float S = com[CS * b + 2];
    const float scale = LOSS ? (grad_loss[(int64_t)b * grad_loss_stride] * 2.0f) / com[CS * b + 3] : 1.0f;     // (stride 0: the gradient of a summed loss)
    if (threadIdx.x == 0) { s_wide = 0; s_ilo = 0x7fffffff; s_ihi = -1; }
    __syncthreads();
    if (threadIdx.x < kTileX + kTileY) {
        const bool col = threadIdx.x < kTileX;
        const int t = col ? threadIdx.x : threadIdx.x - kTileX;
        const int p = col ? x0 + t : y0 + t;
        const int n = col ? W : Hh;
        int lo_i = 0, hi_i = -1;
        if (p < n) tap_range(col ? m.sx : m.sy, col ? m.xc : m.yc, n, p, lo_i, hi_i);
        if (hi_i - lo_i >= kTaps) s_wide = 1;
        (col ? s_j0 : s_i0)[t] = lo_i;
        for (int q = 0; q < kTaps; ++q) {
            const int o = lo_i + q;
            (col ? s_wx[t] : s_wy[t])[q] = o <= hi_i ? tap(col ? m.ix(o) : m.iy(o), p) : 0.0f;
        }
        if (!col && hi_i >= lo_i) { atomicMin(&s_ilo, lo_i); atomicMax(&s_ihi, hi_i); }
        if (!col) s_gy[t] = gcom[2 * b + 1] * (lin11(p, Hh) - m.yc) / S;
    }
    __syncthreads();
    const int ilo = s_ilo, ihi = s_ihi;
    const float* __restrict__ g = grad_out + (int64_t)b * Hh * W;
    if (s_wide || ihi - ilo >= kTileRows) {    // strong magnification: per-pixel form for this tile
        for (int k = threadIdx.x; k < kTileX * kTileY; k += 256) {
            const int x = x0 + (k & (kTileX - 1)), y = y0 + k / kTileX;
            if (x < W && y < Hh) grad_flux[((int64_t)b * Hh + y) * W + x] = scale * crop_bwd_pixel(m, g, S, gcom, b, Hh, W, x, y);
        }
        return;
    }
    const int tx = threadIdx.x & (kTileX - 1);
    const int x = x0 + tx;
    const bool in_x = x < W;
    const int j0 = s_j0[tx];
    const float wx0 = s_wx[tx][0], wx1 = s_wx[tx][1], wx2 = s_wx[tx][2], wx3 = s_wx[tx][3];
    const int c0 = min(j0, W - 1), c1 = min(j0 + 1, W - 1), c2 = min(j0 + 2, W - 1), c3 = min(j0 + 3, W - 1);   // weight 0 there
    // horizontal pass over the output rows this tile's input rows sampled from
    // (all four taps are loaded and weighted unconditionally - absent taps have weight 0 and a clamped, valid index - so
    //  that the loads of several rows are in flight together: with a branch per tap every load was waited for in turn
    //  and a workgroup took 26 us for 6 000 loads)
    const int xr = in_x ? 1 : 0;
    if (W >= 4) {
        // the four taps of a row are four consecutive floats: ONE 16-byte load at 4-byte alignment (from column min(j0, W - 4);
        // the weights move with it - absent taps carry weight 0 - so the non-zero products are added in the same order: same bits)
        struct __attribute__((packed, aligned(4))) Taps { float t0, t1, t2, t3; };
        const int cb = in_x ? min(j0, W - 4) : 0, sh = in_x ? j0 - cb : 0;
        const float w0 = sh == 0 ? wx0 : 0.0f, w1 = sh == 0 ? wx1 : (sh == 1 ? wx0 : 0.0f);
        const float w2 = sh == 0 ? wx2 : (sh == 1 ? wx1 : (sh == 2 ? wx0 : 0.0f));
        const float w3 = sh == 0 ? wx3 : (sh == 1 ? wx2 : (sh == 2 ? wx1 : wx0));
        for (int r = threadIdx.x / kTileX; r <= ihi - ilo; r += kRowUnroll * (256 / kTileX)) {
            float v[kRowUnroll];
#pragma unroll
            for (int q = 0; q < kRowUnroll; ++q) {
                const int rr = min(r + q * (256 / kTileX), ihi - ilo);
                const Taps t = *reinterpret_cast<const Taps*>(g + (int64_t)(ilo + rr) * W + cb);
                v[q] = ((t.t0 * w0 + t.t1 * w1) + t.t2 * w2) + t.t3 * w3;
            }
#pragma unroll
            for (int q = 0; q < kRowUnroll; ++q) {
                const int rr = r + q * (256 / kTileX);
                if (rr <= ihi - ilo) s_t[rr][tx] = in_x ? v[q] : 0.0f;
            }
        }
    } else
    for (int r = threadIdx.x / kTileX; r <= ihi - ilo; r += kRowUnroll * (256 / kTileX)) {
        float v[kRowUnroll];
#pragma unroll
        for (int q = 0; q < kRowUnroll; ++q) {
            const int rr = min(r + q * (256 / kTileX), ihi - ilo);
            const float* __restrict__ row = g + (int64_t)(ilo + rr) * W;
            v[q] = ((row[c0 * xr] * wx0 + row[c1 * xr] * wx1) + row[c2 * xr] * wx2) + row[c3 * xr] * wx3;
        }
#pragma unroll
        for (int q = 0; q < kRowUnroll; ++q) {
            const int rr = r + q * (256 / kTileX);
            if (rr <= ihi - ilo) s_t[rr][tx] = in_x ? v[q] : 0.0f;
        }
    }
    __syncthreads();
    if (!in_x) return;
    const float gx = gcom[2 * b] * (lin11(x, W) - m.xc) / S;
    for (int ty = threadIdx.x / kTileX; ty < kTileY; ty += 256 / kTileX) {
        const int y = y0 + ty;
        if (y >= Hh) break;
        const int r0 = s_i0[ty] - ilo;
        float acc = 0.0f;
#pragma unroll
        for (int a = 0; a < kTaps; ++a) {
            const float wy = s_wy[ty][a];
            if (wy != 0.0f) acc += s_t[r0 + a][tx] * wy;      // uniform across the 64 lanes of a row
        }
        acc += gx + s_gy[ty];
        grad_flux[((int64_t)b * Hh + y) * W + x] = LOSS ? scale * acc : acc;
    }
}

// ---------------------------------------------------------------------------------------------------
// Losses: one workgroup per sample.  mode 0 = PixelLoss, 1 = KLDivergenceLoss.
// ---------------------------------------------------------------------------------------------------
// Element loop over one sample with four elements per load when the sample is 16-byte aligned: fewer, wider loads
// in flight per thread (a sample is streamed by ONE workgroup, so the loop is latency-bound when B is small).
template <typename F>
__device__ __forceinline__ void for_each_pair(const float* __restrict__ p, const float* __restrict__ g, int64_t npix, F&& f)
{
    const bool vec = (npix & 3) == 0 && ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g)) & 15) == 0;
    if (vec) {
        const float4* __restrict__ p4 = reinterpret_cast<const float4*>(p);
        const float4* __restrict__ g4 = reinterpret_cast<const float4*>(g);
        for (int64_t k = threadIdx.x; k < (npix >> 2); k += blockDim.x) {
            const float4 a = p4[k], c = g4[k];
            f(4 * k, a.x, c.x); f(4 * k + 1, a.y, c.y); f(4 * k + 2, a.z, c.z); f(4 * k + 3, a.w, c.w);
        }
    } else {
        for (int64_t k = threadIdx.x; k < npix; k += blockDim.x) f(k, p[k], g[k]);
    }
}

__global__ __launch_bounds__(kReduceBlock) void flux_loss_kernel(const float* __restrict__ pred, const float* __restrict__ truth,
                                                               int64_t npix, int mode, float* __restrict__ loss,
                                                               const float* __restrict__ grad_loss,
                                                               float* __restrict__ grad_pred)
{
    __shared__ double s_red[16];
    const int b = blockIdx.x;
    const float* __restrict__ p = pred + (int64_t)b * npix;
    const float* __restrict__ g = truth + (int64_t)b * npix;
    float* __restrict__ gp = grad_pred ? grad_pred + (int64_t)b * npix : nullptr;
    if (mode == 0) {                                           // loss.py:312-318
        double se = 0.0, sg = 0.0;
        for_each_pair(p, g, npix, [&](int64_t, float pk, float gk) { const float d = pk - gk; se += (double)(d * d); sg += (double)gk; });
        se = block_sum(se, s_red);
        const float sgf = (float)block_sum(sg, s_red);
        if (threadIdx.x == 0 && loss) loss[b] = (float)se / sgf;
        if (gp) {
            const float gl = grad_loss[b];
            for_each_pair(p, g, npix, [&](int64_t k, float pk, float gk) { gp[k] = gl * (2.0f * (pk - gk)) / sgf; });
        }
        return;
    }
    const float eps = 1e-12f;                                  // loss.py:385-410
    double np_ = 0.0, ng = 0.0;
    for_each_pair(p, g, npix, [&](int64_t, float pk, float gk) { np_ += (double)fabsf(pk); ng += (double)fabsf(gk); });
    const float npf = (float)block_sum(np_, s_red), ngf = (float)block_sum(ng, s_red);
    const float dp = fmaxf(npf, eps), dg = fmaxf(ngf, eps);
    double acc = 0.0, dot = 0.0;
    for_each_pair(p, g, npix, [&](int64_t, float pk, float gk) {
        const float t = logf(gk / dg + eps), q = logf(pk / dp + eps);
        const float et = expf(t);
        acc += (double)(et * (t - q));
        dot += (double)((-et / (pk / dp + eps)) * pk);
    });
    acc = block_sum(acc, s_red);
    const float dotf = (float)block_sum(dot, s_red);
    if (threadIdx.x == 0 && loss) loss[b] = (float)acc;
    if (gp) {
        const float gl = grad_loss[b];
        for_each_pair(p, g, npix, [&](int64_t k, float pk, float gk) {
            const float t = logf(gk / dg + eps);
            const float a = -expf(t) / (pk / dp + eps);
            const float sgn = pk > 0.0f ? 1.0f : (pk < 0.0f ? -1.0f : 0.0f);
            const float through_norm = npf > eps ? sgn * dotf / (dp * dp) : 0.0f;
            gp[k] = gl * (a / dp - through_norm);
        });
    }
}

// ---------------------------------------------------------------------------------------------------
// Sampling helpers of the fused crop + loss kernels (the arithmetic of flux_crop_fwd_kernel, tap for tap).
// ---------------------------------------------------------------------------------------------------
struct CropTap { int x0, y0; float tx, ty; bool xa, xb, ya, yb; };
__device__ __forceinline__ float crop_sample(const float* __restrict__ f, const CropMap& m, int i, int j, float& v00, float& v01,
                                             float& v10, float& v11, float& tx, float& ty)
{
    const float ix = m.ix(j), iy = m.iy(i);
    const float x0f = floorf(ix), y0f = floorf(iy);
    tx = ix - x0f; ty = iy - y0f;
    const int x0 = (int)x0f, y0 = (int)y0f;
    const bool xa = x0 >= 0 && x0 < m.W, xb = x0 + 1 >= 0 && x0 + 1 < m.W;
    const bool ya = y0 >= 0 && y0 < m.Hh, yb = y0 + 1 >= 0 && y0 + 1 < m.Hh;
    v00 = ya && xa ? f[y0 * m.W + x0] : 0.0f; v01 = ya && xb ? f[y0 * m.W + x0 + 1] : 0.0f;
    v10 = yb && xa ? f[(y0 + 1) * m.W + x0] : 0.0f; v11 = yb && xb ? f[(y0 + 1) * m.W + x0 + 1] : 0.0f;
    // the accumulation order of flux_crop_fwd_kernel (absent taps add nothing there either)
    float acc = 0.0f;
    if (ya && xa) acc += v00 * ((1.0f - tx) * (1.0f - ty));
    if (ya && xb) acc += v01 * (tx * (1.0f - ty));
    if (yb && xa) acc += v10 * ((1.0f - tx) * ty);
    if (yb && xb) acc += v11 * (tx * ty);
    return acc;
}


// The same sample with the column's part of the work done once per thread: when the workgroup size is a multiple of
// the bitmap width a thread stays in ONE column (pixel k = tid + n * blockDim: j = tid % W, i = tid / W + n * blockDim / W),
// so ix(j), its floor, weights and bounds leave the loop, and the row's iy(i) is wave-uniform.
struct CropColumn { int x0; float tx; bool xa, xb; };
__device__ __forceinline__ CropColumn crop_column(const CropMap& m, int j)
{
    CropColumn c;
    const float ix = m.ix(j);
    const float x0f = floorf(ix);
    c.tx = ix - x0f; c.x0 = (int)x0f;
    c.xa = c.x0 >= 0 && c.x0 < m.W; c.xb = c.x0 + 1 >= 0 && c.x0 + 1 < m.W;
    return c;
}
// (`pixel(k)`: element k of the bitmap - from global memory, or from the rows a workgroup has staged in LDS)
template <typename Pixel>
__device__ __forceinline__ float crop_sample_col_from(Pixel&& pixel, const CropMap& m, const CropColumn& c, int i, float& v00,
                                                      float& v01, float& v10, float& v11, float& ty)
{
    const float iy = m.iy(i);
    const float y0f = floorf(iy);
    ty = iy - y0f;
    const int y0 = (int)y0f;
    const bool ya = y0 >= 0 && y0 < m.Hh, yb = y0 + 1 >= 0 && y0 + 1 < m.Hh;
    v00 = ya && c.xa ? pixel(y0 * m.W + c.x0) : 0.0f; v01 = ya && c.xb ? pixel(y0 * m.W + c.x0 + 1) : 0.0f;
    v10 = yb && c.xa ? pixel((y0 + 1) * m.W + c.x0) : 0.0f; v11 = yb && c.xb ? pixel((y0 + 1) * m.W + c.x0 + 1) : 0.0f;
    float acc = 0.0f;
    if (ya && c.xa) acc += v00 * ((1.0f - c.tx) * (1.0f - ty));
    if (ya && c.xb) acc += v01 * (c.tx * (1.0f - ty));
    if (yb && c.xa) acc += v10 * ((1.0f - c.tx) * ty);
    if (yb && c.xb) acc += v11 * (c.tx * ty);
    return acc;
}
__device__ __forceinline__ float crop_sample_col(const float* __restrict__ f, const CropMap& m, const CropColumn& c, int i, float& v00,
                                                 float& v01, float& v10, float& v11, float& ty)
{
    return crop_sample_col_from([&](int k) { return f[k]; }, m, c, i, v00, v01, v10, v11, ty);
}

// ---------------------------------------------------------------------------------------------------
// The fused crop + PixelLoss pair.  A bitmap's rows are ALWAYS summed in kLossParts = 4 parts, whatever the batch size: the
// parts' fp64 sums (per-thread order, block tree) are added in part order, so the loss, the centre and the gradient are the
// same bits whether a bitmap has one workgroup (large batches: all four parts in one workgroup, one launch) or two / four
// (small batches - one of eight ranks' 125 bitmaps on 256 CUs; as many workgroups as still have a CU each, the parts
// carried from kernel to kernel through a small scratch).  Round 3's one-workgroup kernels summed all rows in one go and
// differed from the part kernels in the last bits: the same data gave another loss at another batch size (advisor).
//   flux_com_parts_kernel           grid (P, B): the centre-of-mass sums of each part -> parts[b][v][0..2]       (P > 1 only)
//   flux_crop_pixel_loss_fwd_kernel grid (P, B): [P == 1: the same sums first, in the workgroup] the four parts in part order
//                                   -> the centre; crops and compares its parts' rows; per part (sum d^2, sum truth) and -
//                                   when a backward pass will follow - the RESIDUAL d = crop - truth [B,Hh,W] and the two
//                                   sums that make the gradient of the centre; [P == 1: loss and records written here]
//   flux_crop_pixel_loss_final_kernel  one thread per bitmap adds the parts                                      (P > 1 only)
//   backward                        flux_crop_bwd_tiled_kernel<true>: ONE pass over the residual (see there)
// (A "last workgroup adds the parts" ticket was built first: its device-scope fences write back and invalidate the L2 of
//  every XCD - 0.24 instead of 0.04 ms at 125 bitmaps.  Kernel boundaries carry the parts instead.)
// ---------------------------------------------------------------------------------------------------
constexpr int kMaxPartBitmaps = 512;
struct PartScratch { double com[kMaxPartBitmaps][kLossParts][3]; double acc[kMaxPartBitmaps][kLossParts][4]; };

__global__ __launch_bounds__(kReduceBlock) void flux_com_parts_kernel(const float* __restrict__ flux, int Hh, int W, int parts_per_wg,
                                                                    PartScratch* __restrict__ ws)
{
    __shared__ double s_red[16 * 4];
    const int b = blockIdx.y;
    const float* __restrict__ f = flux + (int64_t)b * Hh * W;
    for (int v = blockIdx.x * parts_per_wg; v < (int)(blockIdx.x + 1) * parts_per_wg; ++v) {
        double s, xs, ys;
        com_part_sums(f, Hh, W, v, s_red, s, xs, ys);
        if (threadIdx.x == 0) { ws->com[b][v][0] = s; ws->com[b][v][1] = xs; ws->com[b][v][2] = ys; }
    }
}

// the centre of a bitmap from its four parts (part order); every workgroup of the bitmap computes the same three numbers
__device__ __forceinline__ void com_from_parts(const double (*parts)[3], float& xc, float& yc, float& S)
{
    double s = 0.0, xs = 0.0, ys = 0.0;
    for (int v = 0; v < kLossParts; ++v) { s += parts[v][0]; xs += parts[v][1]; ys += parts[v][2]; }
    S = (float)s + 1e-8f;
    xc = (float)(xs / (double)S); yc = (float)(ys / (double)S);
}

// loss, the record the backward pass reads and - with a residual - the centre's gradient per unit of 2 gl / sum(truth)
__device__ __forceinline__ void loss_from_parts(const double (*acc)[4], float xc, float yc, float S, int b, int Hh, int W,
                                                float* __restrict__ loss, float* __restrict__ com4, float* __restrict__ gunit)
{
    double se = 0.0, sg = 0.0, gx = 0.0, gy = 0.0;
    for (int v = 0; v < kLossParts; ++v) { se += acc[v][0]; sg += acc[v][1]; gx += acc[v][2]; gy += acc[v][3]; }
    const float sgf = (float)sg;
    loss[b] = (float)se / sgf;                              // loss.py:312-318
    com4[4 * b] = xc; com4[4 * b + 1] = yc; com4[4 * b + 2] = S; com4[4 * b + 3] = sgf;
    if (gunit) {
        gunit[2 * b] = (float)(gx * (double)((float)(W - 1) / 2.0f));
        gunit[2 * b + 1] = (float)(gy * (double)((float)(Hh - 1) / 2.0f));
    }
}

template <bool WHOLE>       // WHOLE: one workgroup per bitmap does all four parts, the centre and the final sums itself
__global__ __launch_bounds__(kReduceBlock) void flux_crop_pixel_loss_fwd_kernel(const float* __restrict__ flux,
                                                                               const float* __restrict__ dims,
                                                                               const float* __restrict__ truth, int Hh, int W,
                                                                               float crop_w, float crop_h, int parts_per_wg,
                                                                               PartScratch* ws, float* __restrict__ residual,
                                                                               float* __restrict__ loss, float* __restrict__ com4,
                                                                               float* __restrict__ gunit, const double* __restrict__ moments,
                                                                               int stage_rows)
{
    extern __shared__ __attribute__((aligned(16))) float s_stage[];      // stage_rows x W floats (see the column loop)
    __shared__ double s_red[16 * 4];
    __shared__ double s_com[kLossParts][3], s_acc[kLossParts][4];
    const int b = blockIdx.y;
    const float* __restrict__ f = flux + (int64_t)b * Hh * W;
    const float* __restrict__ g = truth + (int64_t)b * Hh * W;
    float* __restrict__ res = residual ? residual + (int64_t)b * Hh * W : nullptr;
    CropMap m;
    float S;
    if (moments != nullptr) {          // the sums came with the bitmaps (the trace's conversion pass formed them: flux_moments.hpp)
        com_from_parts(reinterpret_cast<const double(*)[3]>(moments + (int64_t)b * kLossParts * 3), m.xc, m.yc, S);
    } else if constexpr (WHOLE) {
        for (int v = 0; v < kLossParts; ++v) {
            double s, xs, ys;
            com_part_sums(f, Hh, W, v, s_red, s, xs, ys);
            if (threadIdx.x == 0) { s_com[v][0] = s; s_com[v][1] = xs; s_com[v][2] = ys; }
        }
        __syncthreads();
        com_from_parts(s_com, m.xc, m.yc, S);
    } else {
        com_from_parts(ws->com[b], m.xc, m.yc, S);
    }
    m.sx = crop_w / fmaxf(dims[2 * b], 1e-8f); m.sy = crop_h / fmaxf(dims[2 * b + 1], 1e-8f);
    m.W = W; m.Hh = Hh;
    for (int v = blockIdx.x * parts_per_wg; v < (int)(blockIdx.x + 1) * parts_per_wg; ++v) {
        int r0, r1;
        part_rows(Hh, v, r0, r1);
        double a[4] = {0.0, 0.0, 0.0, 0.0};             // sum d^2, sum truth, and the two sums of the centre's gradient
        if ((int)blockDim.x % W == 0) {                 // a thread owns one column
            const int j = threadIdx.x % W, di = blockDim.x / W;
            const CropColumn col = crop_column(m, j);
            // The bitmap rows this part's output rows sample, staged in LDS with 16-byte loads when they fit (stage_rows of them:
            // the launch's dynamic LDS): the four taps of a pixel are then LDS reads - a fifth of the vector-memory instructions
            // (four predicated 4-byte taps per pixel were four tag look-ups of the same two cache lines per wave): 0.298 -> 0.257 ms
            // per 1000 bitmaps.  Same values, same arithmetic: same bits.
            int ylo = 0, nrows = 0;
            if (stage_rows > 0 && (W & 3) == 0) {
                const float ya_ = m.iy(r0), yb_ = m.iy(r1 - 1);
                if (ya_ > -1.0e9f && ya_ < 1.0e9f && yb_ > -1.0e9f && yb_ < 1.0e9f) {
                    ylo = max(0, (int)floorf(fminf(ya_, yb_)));
                    const int yhi = min(Hh - 1, (int)floorf(fmaxf(ya_, yb_)) + 1);
                    nrows = yhi - ylo + 1;
                    if (nrows > stage_rows) nrows = 0;
                }
            }
            // four rows per step, every tap and the measured flux loaded BEFORE the first residual is stored: the loop is a chain
            // of L2 round trips otherwise (a store between two rows' loads keeps the compiler from having them in flight together)
            constexpr int kRows = 4;
            auto rows = [&](auto&& pixel) {
            for (int i0 = r0 + threadIdx.x / W; i0 < r1; i0 += kRows * di) {
                float c[kRows], t[kRows], gxs[kRows], gys[kRows];
#pragma unroll
                for (int u = 0; u < kRows; ++u) {
                    const int i = min(i0 + u * di, r1 - 1);                 // (a clamped row is loaded and not used)
                    float v00, v01, v10, v11, ty;
                    c[u] = crop_sample_col_from(pixel, m, col, i, v00, v01, v10, v11, ty);
                    t[u] = g[i * W + j];
                    gxs[u] = (v01 - v00) * (1.0f - ty) + (v11 - v10) * ty;
                    gys[u] = (v10 - v00) * (1.0f - col.tx) + (v11 - v01) * col.tx;
                }
#pragma unroll
                for (int u = 0; u < kRows; ++u) {
                    const int i = i0 + u * di;
                    if (i >= r1) break;
                    const float d = c[u] - t[u];
                    a[0] += (double)(d * d); a[1] += (double)t[u];
                    if (res) {
                        res[i * W + j] = d;
                        a[2] += (double)(d * gxs[u]);
                        a[3] += (double)(d * gys[u]);
                    }
                }
            }
            };
            if (nrows > 0) {                            // (workgroup-uniform)
                __syncthreads();                        // the previous part's readers are done with the staged rows
                const float4* __restrict__ src = reinterpret_cast<const float4*>(f + (int64_t)ylo * W);
                float4* dst = reinterpret_cast<float4*>(s_stage);
                for (int k = threadIdx.x; k < nrows * (W >> 2); k += blockDim.x) dst[k] = src[k];
                __syncthreads();
                const float* staged = s_stage - ylo * W;      // (row y of the bitmap at staged + y W)
                rows([&](int k) { return staged[k]; });
            } else {
                rows([&](int k) { return f[k]; });
            }
        } else {
            for (int k = r0 * W + threadIdx.x; k < r1 * W; k += blockDim.x) {
                const int i = k / W, j = k - i * W;
                float v00, v01, v10, v11, tx, ty;
                const float c = crop_sample(f, m, i, j, v00, v01, v10, v11, tx, ty);
                const float t = g[k];
                const float d = c - t;
                a[0] += (double)(d * d); a[1] += (double)t;
                if (res) {
                    res[k] = d;
                    a[2] += (double)(d * ((v01 - v00) * (1.0f - ty) + (v11 - v10) * ty));
                    a[3] += (double)(d * ((v10 - v00) * (1.0f - tx) + (v11 - v01) * tx));
                }
            }
        }
        block_sum_n<4>(a, s_red);
        if (threadIdx.x == 0) {
            double* out = WHOLE ? s_acc[v] : ws->acc[b][v];
            out[0] = a[0]; out[1] = a[1]; out[2] = a[2]; out[3] = a[3];
        }
    }
    if constexpr (WHOLE) {
        if (threadIdx.x == 0) loss_from_parts(s_acc, m.xc, m.yc, S, b, Hh, W, loss, com4, gunit);     // (its own writes)
    }
}

// the parts of the bitmaps that were shared among workgroups -> loss and records: one thread per bitmap
__global__ __launch_bounds__(256) void flux_crop_pixel_loss_final_kernel(const PartScratch* __restrict__ ws, int B, int Hh, int W,
                                                                         float* __restrict__ loss, float* __restrict__ com4,
                                                                         float* __restrict__ gunit, const double* __restrict__ moments)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float xc, yc, S;
    com_from_parts(moments ? reinterpret_cast<const double(*)[3]>(moments + (int64_t)b * kLossParts * 3) : ws->com[b], xc, yc, S);
    loss_from_parts(ws->acc[b], xc, yc, S, b, Hh, W, loss, com4, gunit);
}

// ---------------------------------------------------------------------------------------------------
// get_center_of_mass (artist/flux/bitmap.py:12-71): PIXEL coordinates (e, u) of each bitmap's centre of mass,
// sum_j j f / (sum f + 1e-8) - what FocalSpotLoss (artist/optim/loss.py:124-250) and the kinematics reconstructor's
// validation (kinematics_reconstructor.py:120) ask of the tracer's bitmaps.  com[b] = (e px, u px, sum + 1e-8).
// One streaming pass, fp64 sums; backward = one elementwise pass: d e_com / d f_ij = (j - e_com) / (S + 1e-8).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kReduceBlock) void flux_com_px_kernel(const float* __restrict__ flux, int Hh, int W,
                                                                 float* __restrict__ com)
{
    __shared__ double s_red[16];
    const int b = blockIdx.x;
    const float* __restrict__ f = flux + (int64_t)b * Hh * W;
    double s = 0.0, xs = 0.0, ys = 0.0;
    int x = threadIdx.x % W, y = threadIdx.x / W;
    const int dx = blockDim.x % W, dy = blockDim.x / W;
    for (int k = threadIdx.x; k < Hh * W; k += blockDim.x) {
        const float v = f[k];
        s += (double)v;
        xs += (double)((float)x * v);
        ys += (double)((float)y * v);
        x += dx; y += dy;
        if (x >= W) { x -= W; ++y; }
    }
    s = block_sum(s, s_red);
    xs = block_sum(xs, s_red);
    ys = block_sum(ys, s_red);
    if (threadIdx.x == 0) {
        const float S = (float)s + 1e-8f;
        com[3 * b] = (float)(xs / (double)S); com[3 * b + 1] = (float)(ys / (double)S); com[3 * b + 2] = S;
    }
}

__global__ __launch_bounds__(kFluxBlock) void flux_com_px_bwd_kernel(const float* __restrict__ com, const float* __restrict__ grad_com,
                                                                     int Hh, int W, float* __restrict__ grad_flux)
{
    const int b = blockIdx.y;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= Hh * W) return;
    const int i = k / W, j = k - i * W;
    const float S = com[3 * b + 2];
    grad_flux[(int64_t)b * Hh * W + k] = (grad_com[2 * b] * ((float)j - com[3 * b]) + grad_com[2 * b + 1] * ((float)i - com[3 * b + 1])) / S;
}

// ---------------------------------------------------------------------------------------------------
// Fused crop around the centre of mass + KLDivergenceLoss (artist/flux/bitmap.py:121-246 followed by
// artist/optim/loss.py:321-410 with the reduction over the two bitmap dimensions), one workgroup per bitmap, the same
// arithmetic as flux_com_kernel / flux_crop_fwd_kernel / flux_loss_kernel(mode 1): the cropped bitmap never reaches
// HBM.  The KL terms need the crop's L1 norm first, so the crop is sampled twice (the taps hit L2: the workgroup has
// just streamed the bitmap).  rec8[b] = (x centre, y centre, sum + 1e-8, |crop|_1, |truth|_1, dot, 0, 0) for the
// backward pass, where dot = sum_k a_k crop_k with a_k = -exp(t_k) / (crop_k / dp + eps).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void block_com(const float* __restrict__ f, int Hh, int W, double* s_red, float* s_com)
{
    double s = 0.0, xs = 0.0, ys = 0.0;
    if ((W & 3) == 0) {                                 // flux_com_kernel's loop
        const int W4 = W >> 2;
        int x4 = threadIdx.x % W4, y = threadIdx.x / W4;
        const int dx = blockDim.x % W4, dy = blockDim.x / W4;
        const float4* __restrict__ f4 = reinterpret_cast<const float4*>(f);
#pragma unroll 4
        for (int k = threadIdx.x; k < Hh * W4; k += blockDim.x) {
            const float4 v = f4[k];
            const int x = 4 * x4;
            s += (double)((v.x + v.y) + (v.z + v.w));
            xs += (double)((lin11(x, W) * v.x + lin11(x + 1, W) * v.y) + (lin11(x + 2, W) * v.z + lin11(x + 3, W) * v.w));
            ys += (double)(lin11(y, Hh) * ((v.x + v.y) + (v.z + v.w)));
            x4 += dx; y += dy;
            if (x4 >= W4) { x4 -= W4; ++y; }
        }
    } else {
        int x = threadIdx.x % W, y = threadIdx.x / W;
        const int dx = blockDim.x % W, dy = blockDim.x / W;
        for (int k = threadIdx.x; k < Hh * W; k += blockDim.x) {
            const float v = f[k];
            s += (double)v; xs += (double)(lin11(x, W) * v); ys += (double)(lin11(y, Hh) * v);
            x += dx; y += dy;
            if (x >= W) { x -= W; ++y; }
        }
    }
    s = block_sum(s, s_red); xs = block_sum(xs, s_red); ys = block_sum(ys, s_red);
    if (threadIdx.x == 0) {
        const float S = (float)s + 1e-8f;
        s_com[0] = (float)(xs / (double)S); s_com[1] = (float)(ys / (double)S); s_com[2] = S;
    }
    __syncthreads();
}

// f(k, crop_k, v00, v01, v10, v11, tx, ty) for every output pixel k of the crop of bitmap `f`, each thread in its own
// fixed order (a thread owns one column when the workgroup size is a multiple of the bitmap width)
template <typename F>
__device__ __forceinline__ void for_each_crop_pixel(const float* __restrict__ f, const CropMap& m, F&& fn)
{
    const int W = m.W, Hh = m.Hh;
    if ((int)blockDim.x % W == 0) {
        const int j = threadIdx.x % W, di = blockDim.x / W;
        const CropColumn col = crop_column(m, j);
#pragma unroll 4
        for (int i = threadIdx.x / W; i < Hh; i += di) {
            float v00, v01, v10, v11, ty;
            const float c = crop_sample_col(f, m, col, i, v00, v01, v10, v11, ty);
            fn(i * W + j, c, v00, v01, v10, v11, col.tx, ty);
        }
    } else {
        for (int k = threadIdx.x; k < Hh * W; k += blockDim.x) {
            const int i = k / W, j = k - i * W;
            float v00, v01, v10, v11, tx, ty;
            const float c = crop_sample(f, m, i, j, v00, v01, v10, v11, tx, ty);
            fn(k, c, v00, v01, v10, v11, tx, ty);
        }
    }
}

__global__ __launch_bounds__(kReduceBlock) void flux_crop_kl_loss_fwd_kernel(const float* __restrict__ flux,
                                                                            const float* __restrict__ dims,
                                                                            const float* __restrict__ truth, int Hh, int W,
                                                                            float crop_w, float crop_h, float* __restrict__ loss,
                                                                            float* __restrict__ rec8)
{
    __shared__ double s_red[16];
    __shared__ float s_com[3];
    const int b = blockIdx.x;
    const float* __restrict__ f = flux + (int64_t)b * Hh * W;
    const float* __restrict__ g = truth + (int64_t)b * Hh * W;
    block_com(f, Hh, W, s_red, s_com);
    CropMap m;
    m.sx = crop_w / fmaxf(dims[2 * b], 1e-8f); m.sy = crop_h / fmaxf(dims[2 * b + 1], 1e-8f);
    m.xc = s_com[0]; m.yc = s_com[1]; m.W = W; m.Hh = Hh;
    const float eps = 1e-12f;                                  // loss.py:385-410, as flux_loss_kernel mode 1
    double np_ = 0.0, ng = 0.0;
    for_each_crop_pixel(f, m, [&](int k, float c, float, float, float, float, float, float) {
        np_ += (double)fabsf(c); ng += (double)fabsf(g[k]);
    });
    const float npf = (float)block_sum(np_, s_red), ngf = (float)block_sum(ng, s_red);
    const float dp = fmaxf(npf, eps), dg = fmaxf(ngf, eps);
    double acc = 0.0, dot = 0.0;
    for_each_crop_pixel(f, m, [&](int k, float c, float, float, float, float, float, float) {
        const float t = logf(g[k] / dg + eps), q = logf(c / dp + eps);
        const float et = expf(t);
        acc += (double)(et * (t - q));
        dot += (double)((-et / (c / dp + eps)) * c);
    });
    acc = block_sum(acc, s_red);
    const float dotf = (float)block_sum(dot, s_red);
    if (threadIdx.x == 0) {
        loss[b] = (float)acc;
        float* r = rec8 + 8 * b;
        r[0] = s_com[0]; r[1] = s_com[1]; r[2] = s_com[2]; r[3] = npf; r[4] = ngf; r[5] = dotf; r[6] = 0.0f; r[7] = 0.0f;
    }
}

// grad_crop[b] = gl[b] dKL/dcrop (written), gcom[b] = its gradient w.r.t. the two centre coordinates, com3 for the tiled gather
__global__ __launch_bounds__(kReduceBlock) void flux_crop_kl_loss_bwd_kernel(const float* __restrict__ flux,
                                                                            const float* __restrict__ dims,
                                                                            const float* __restrict__ truth,
                                                                            const float* __restrict__ rec8,
                                                                            const float* __restrict__ grad_loss, int Hh, int W,
                                                                            float crop_w, float crop_h,
                                                                            float* __restrict__ grad_crop, float* __restrict__ com3,
                                                                            float* __restrict__ gcom)
{
    __shared__ double s_red[16];
    const int b = blockIdx.x;
    const float* __restrict__ f = flux + (int64_t)b * Hh * W;
    const float* __restrict__ g = truth + (int64_t)b * Hh * W;
    float* __restrict__ gc = grad_crop + (int64_t)b * Hh * W;
    const float* r = rec8 + 8 * b;
    CropMap m;
    m.sx = crop_w / fmaxf(dims[2 * b], 1e-8f); m.sy = crop_h / fmaxf(dims[2 * b + 1], 1e-8f);
    m.xc = r[0]; m.yc = r[1]; m.W = W; m.Hh = Hh;
    const float eps = 1e-12f;
    const float npf = r[3], ngf = r[4], dotf = r[5], gl = grad_loss[b];
    const float dp = fmaxf(npf, eps), dg = fmaxf(ngf, eps);
    double gx = 0.0, gy = 0.0;
    for_each_crop_pixel(f, m, [&](int k, float c, float v00, float v01, float v10, float v11, float tx, float ty) {
        const float t = logf(g[k] / dg + eps);                    // flux_loss_kernel's gradient (mode 1)
        const float a = -expf(t) / (c / dp + eps);
        const float sgn = c > 0.0f ? 1.0f : (c < 0.0f ? -1.0f : 0.0f);
        const float through_norm = npf > eps ? sgn * dotf / (dp * dp) : 0.0f;
        const float go = gl * (a / dp - through_norm);
        gc[k] = go;
        gx += (double)(go * ((v01 - v00) * (1.0f - ty) + (v11 - v10) * ty));     // flux_crop_bwd_com_kernel
        gy += (double)(go * ((v10 - v00) * (1.0f - tx) + (v11 - v01) * tx));
    });
    gx = block_sum(gx, s_red);
    gy = block_sum(gy, s_red);
    if (threadIdx.x == 0) {
        gcom[2 * b] = (float)(gx * (double)((float)(W - 1) / 2.0f));
        gcom[2 * b + 1] = (float)(gy * (double)((float)(Hh - 1) / 2.0f));
        com3[3 * b] = m.xc; com3[3 * b + 1] = m.yc; com3[3 * b + 2] = r[2];
    }
}

}  // namespace art

using namespace art;

// Scratch of the part kernels: one PartScratch per (device, stream), allocated on first use and kept (every entry is written
// before it is read: no initialisation).  nullptr when the allocation fails: the callers give a bitmap one workgroup.
static PartScratch* flux_parts_scratch(hipStream_t stream)
{
    struct Entry { int dev; hipStream_t stream; PartScratch* ws; };
    static std::vector<Entry> table;
    static std::mutex lock;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> guard(lock);
    for (const Entry& e : table)
        if (e.dev == dev && e.stream == stream) return e.ws;
    PartScratch* ws = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&ws), sizeof(PartScratch)) != hipSuccess) return nullptr;
    table.push_back({dev, stream, ws});
    return ws;
}

// Workgroups per bitmap of the fused crop + loss pair: as many of 4 / 2 / 1 as still leave every workgroup a CU of its own
// (the passes are bound by what ONE CU's texture path delivers: two workgroups on a CU gain nothing and the extra launches
// cost ~4 us each).  Measured, forward / backward in us (tools/flux_bench.py; 1, 2, 4 workgroups per bitmap): 64 bitmaps
// 40 / 71, 35 / 57, 22 / 45; 125 bitmaps 42 / 84, 37 / 73, 37 / 74; 180 bitmaps 43 / 98, 63 / 112, 52 / 100; 250 bitmaps
// 45 / 114, 64 / 128, 66 / 130.  (ARTIST_HIP_DEBUG=1 ARTIST_HIP_LOSS_PARTS = 1 / 2 / 4 forces a value: tests - the results are the same bits.)
static int loss_workgroups_per_bitmap(int64_t B, int64_t Hh)
{
    const int forced = debug_env_int("ARTIST_HIP_LOSS_PARTS", 0);
    if (B > kMaxPartBitmaps || Hh < kLossParts) return 1;
    if (forced == 1 || forced == 2 || forced == 4) return forced;
    int cus = 256;
    {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
            cus = n;
    }
    return 4 * B <= cus ? 4 : (2 * B <= cus ? 2 : 1);
}

static bool crop_args_ok(const void* a, const void* b, const void* c, const void* d, int64_t B, int64_t Hh, int64_t W)
{
    return a && b && c && d && B >= 0 && Hh >= 1 && W >= 1 && Hh <= 65535 && Hh * W <= (int64_t)1 << 30 && B <= 65535;
}

extern "C" int art_flux_crop_fwd(const float* flux, const float* target_dims, int64_t B, int64_t Hh, int64_t W,
                                 double crop_width, double crop_height, float* out, float* centers, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!crop_args_ok(flux, target_dims, out, centers, B, Hh, W)) return ART_EINVAL;
    if (B == 0) return ART_OK;
    hipLaunchKernelGGL(flux_com_kernel, dim3((unsigned)B), dim3(kReduceBlock), 0, stream, flux, (int)Hh, (int)W, centers);
    hipLaunchKernelGGL(flux_crop_fwd_kernel,
                       dim3((unsigned)((W + kFluxBlock - 1) / kFluxBlock), (unsigned)((Hh + kCropRows - 1) / kCropRows), (unsigned)B),
                       dim3(kFluxBlock), 0, stream, flux, target_dims, centers, (int)Hh, (int)W, (float)crop_width,
                       (float)crop_height, out);
    ART_HIP(hipGetLastError());
    return ART_OK;
}

extern "C" int art_flux_crop_bwd(const float* flux, const float* target_dims, const float* centers, int64_t B,
                                 int64_t Hh, int64_t W, double crop_width, double crop_height, const float* grad_out,
                                 float* grad_flux, float* workspace, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!crop_args_ok(flux, target_dims, centers, grad_out, B, Hh, W) || !grad_flux || !workspace) return ART_EINVAL;
    if (B == 0) return ART_OK;
    hipLaunchKernelGGL(flux_crop_bwd_com_kernel, dim3((unsigned)B), dim3(kReduceBlock), 0, stream, flux, target_dims, centers,
                       grad_out, (int)Hh, (int)W, (float)crop_width, (float)crop_height, workspace);
    hipLaunchKernelGGL(flux_crop_bwd_tiled_kernel<false>,
                       dim3((unsigned)((W + kTileX - 1) / kTileX), (unsigned)((Hh + kTileY - 1) / kTileY), (unsigned)B),
                       dim3(256), 0, stream, target_dims, centers, workspace, grad_out, (int)Hh, (int)W,
                       (float)crop_width, (float)crop_height, grad_flux, (const float*)nullptr);
    ART_HIP(hipGetLastError());
    return ART_OK;
}

extern "C" int art_flux_loss(const float* prediction, const float* ground_truth, int64_t B, int64_t npix, int kind,
                             float* loss, const float* grad_loss, float* grad_prediction, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!prediction || !ground_truth || B < 0 || npix < 1 || (kind != 0 && kind != 1) || (!loss && !grad_prediction) ||
        (grad_prediction && !grad_loss))
        return ART_EINVAL;
    if (B == 0) return ART_OK;
    hipLaunchKernelGGL(flux_loss_kernel, dim3((unsigned)B), dim3(kReduceBlock), 0, stream, prediction, ground_truth, npix,
                       kind, loss, grad_loss, grad_prediction);
    ART_HIP(hipGetLastError());
    return ART_OK;
}

extern "C" int art_flux_crop_pixel_loss_fwd(const float* flux, const float* target_dims, const float* ground_truth, int64_t B,
                                            int64_t Hh, int64_t W, double crop_width, double crop_height, float* loss,
                                            float* centers4, float* residual, float* center_grad_unit, const double* moments,
                                            void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!crop_args_ok(flux, target_dims, ground_truth, loss, B, Hh, W) || !centers4 || (residual != nullptr) != (center_grad_unit != nullptr))
        return ART_EINVAL;
    if (B == 0) return ART_OK;
    const int P = loss_workgroups_per_bitmap(B, Hh);
    PartScratch* ws = P > 1 ? flux_parts_scratch(stream) : nullptr;
    // LDS for the rows a part samples (a crop never magnifies by much: the rows of a part + 3 cover every scale up to 1), two
    // workgroups per CU: at most 76 KB each
    int stage_rows = (int)std::min<int64_t>(Hh, (Hh + kLossParts - 1) / kLossParts + 4);
    if ((W & 3) != 0 || kReduceBlock % W != 0 || (int64_t)stage_rows * W * 4 > 76 * 1024) stage_rows = 0;
    const size_t stage_bytes = (size_t)stage_rows * W * sizeof(float);
    if (ws != nullptr) {
        if (moments == nullptr)
            hipLaunchKernelGGL(flux_com_parts_kernel, dim3((unsigned)P, (unsigned)B), dim3(kReduceBlock), 0, stream, flux, (int)Hh, (int)W,
                               kLossParts / P, ws);
        // (small batches - their bitmaps sit in the last-level cache - gain nothing from staged rows: 32 -> 35 us at 125 bitmaps)
        hipLaunchKernelGGL(flux_crop_pixel_loss_fwd_kernel<false>, dim3((unsigned)P, (unsigned)B), dim3(kReduceBlock), 0, stream, flux,
                           target_dims, ground_truth, (int)Hh, (int)W, (float)crop_width, (float)crop_height, kLossParts / P, ws, residual,
                           loss, centers4, center_grad_unit, moments, 0);
        hipLaunchKernelGGL(flux_crop_pixel_loss_final_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, stream, ws, (int)B, (int)Hh,
                           (int)W, loss, centers4, center_grad_unit, moments);
    } else
    {
        if (stage_bytes > 48 * 1024)
            ART_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&flux_crop_pixel_loss_fwd_kernel<true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)stage_bytes));
        hipLaunchKernelGGL(flux_crop_pixel_loss_fwd_kernel<true>, dim3(1u, (unsigned)B), dim3(kReduceBlock), stage_bytes, stream, flux, target_dims,
                           ground_truth, (int)Hh, (int)W, (float)crop_width, (float)crop_height, kLossParts, (PartScratch*)nullptr, residual,
                           loss, centers4, center_grad_unit, moments, stage_rows);
    }
    ART_HIP(hipGetLastError());
    return ART_OK;
}

extern "C" int art_flux_crop_pixel_loss_bwd(const float* target_dims, const float* centers4, const float* grad_loss,
                                            int64_t grad_loss_stride, const float* residual, const float* center_grad_unit, int64_t B,
                                            int64_t Hh, int64_t W, double crop_width, double crop_height, float* grad_flux, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!crop_args_ok(residual, target_dims, center_grad_unit, centers4, B, Hh, W) || !grad_loss || !grad_flux ||
        (grad_loss_stride != 0 && grad_loss_stride != 1))
        return ART_EINVAL;
    if (B == 0) return ART_OK;
    hipLaunchKernelGGL(flux_crop_bwd_tiled_kernel<true>,
                       dim3((unsigned)((W + kTileX - 1) / kTileX), (unsigned)((Hh + kTileY - 1) / kTileY), (unsigned)B),
                       dim3(256), 0, stream, target_dims, centers4, center_grad_unit, residual, (int)Hh, (int)W, (float)crop_width,
                       (float)crop_height, grad_flux, grad_loss, (int)grad_loss_stride);
    ART_HIP(hipGetLastError());
    return ART_OK;
}

extern "C" int art_flux_center_of_mass(const float* flux, int64_t B, int64_t Hh, int64_t W, float* com, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!crop_args_ok(flux, com, flux, com, B, Hh, W)) return ART_EINVAL;
    if (B == 0) return ART_OK;
    hipLaunchKernelGGL(flux_com_px_kernel, dim3((unsigned)B), dim3(kReduceBlock), 0, stream, flux, (int)Hh, (int)W, com);
    ART_HIP(hipGetLastError());
    return ART_OK;
}

extern "C" int art_flux_center_of_mass_bwd(const float* com, const float* grad_com, int64_t B, int64_t Hh, int64_t W,
                                           float* grad_flux, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!crop_args_ok(com, grad_com, grad_flux, com, B, Hh, W)) return ART_EINVAL;
    if (B == 0) return ART_OK;
    hipLaunchKernelGGL(flux_com_px_bwd_kernel, dim3((unsigned)((Hh * W + kFluxBlock - 1) / kFluxBlock), (unsigned)B), dim3(kFluxBlock),
                       0, stream, com, grad_com, (int)Hh, (int)W, grad_flux);
    ART_HIP(hipGetLastError());
    return ART_OK;
}

extern "C" int art_flux_crop_kl_loss_fwd(const float* flux, const float* target_dims, const float* ground_truth, int64_t B,
                                         int64_t Hh, int64_t W, double crop_width, double crop_height, float* loss,
                                         float* record8, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!crop_args_ok(flux, target_dims, ground_truth, loss, B, Hh, W) || !record8) return ART_EINVAL;
    if (B == 0) return ART_OK;
    hipLaunchKernelGGL(flux_crop_kl_loss_fwd_kernel, dim3((unsigned)B), dim3(kReduceBlock), 0, stream, flux, target_dims,
                       ground_truth, (int)Hh, (int)W, (float)crop_width, (float)crop_height, loss, record8);
    ART_HIP(hipGetLastError());
    return ART_OK;
}

extern "C" int art_flux_crop_kl_loss_bwd(const float* flux, const float* target_dims, const float* ground_truth,
                                         const float* record8, const float* grad_loss, int64_t B, int64_t Hh, int64_t W,
                                         double crop_width, double crop_height, float* grad_flux, float* workspace,
                                         void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!crop_args_ok(flux, target_dims, ground_truth, record8, B, Hh, W) || !grad_loss || !grad_flux || !workspace) return ART_EINVAL;
    if (B == 0) return ART_OK;
    float* grad_crop = workspace;                      // [B,Hh,W]
    float* com3 = workspace + B * Hh * W;              // [B,3]
    float* gcom = com3 + 3 * B;                        // [B,2]
    hipLaunchKernelGGL(flux_crop_kl_loss_bwd_kernel, dim3((unsigned)B), dim3(kReduceBlock), 0, stream, flux, target_dims,
                       ground_truth, record8, grad_loss, (int)Hh, (int)W, (float)crop_width, (float)crop_height, grad_crop, com3,
                       gcom);
    hipLaunchKernelGGL(flux_crop_bwd_tiled_kernel<false>,
                       dim3((unsigned)((W + kTileX - 1) / kTileX), (unsigned)((Hh + kTileY - 1) / kTileY), (unsigned)B),
                       dim3(256), 0, stream, target_dims, com3, gcom, grad_crop, (int)Hh, (int)W, (float)crop_width,
                       (float)crop_height, grad_flux, (const float*)nullptr);
    ART_HIP(hipGetLastError());
    return ART_OK;
}
