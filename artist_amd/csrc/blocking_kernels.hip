// blocking_kernels.hip - which blocking rectangles matter (gfx950 / MI355X).
//
// Replaces lbvh_filter_blocking_planes (artist/raytracing/blocking.py:832-995) and the tree it walks
// (build_linear_bounding_volume_hierarchies, :514-749).  The reference needs the tree because its soft mask is a
// dense [rays x primitives] tensor; here the mask is evaluated inside the trace kernels against a short
// per-heliostat candidate list, and this file produces those lists:
//
//   beam      one workgroup per heliostat: bounding sphere of its ray origins, mean reflected direction, largest
//             angle between any of its rays and that direction (mirror shape + largest scatter angle)
//   cull      heliostat x primitive: bounding sphere of the rectangle against the heliostat's cone -> <= Cmax
//             candidates per heliostat (conservative: a primitive outside the cone cannot be touched by any ray)
//   live      (reference compatibility) the reference's tree leaves most leaves unreachable from its root (its
//             split search halves the step by floor division, :640-650); only reachable primitives can ever be
//             returned, so the set is rebuilt here node by node: Morton codes, sort, ranges, splits, the
//             bottom-up box pass, reachability from node 0
//   filter    every ray against the boxes of its heliostat's candidates (:922-950): flags[k] = 1 iff some ray
//             of a heliostat other than k's own hits the box of k no later than its target
//   compact   candidates of a heliostat that are flagged -> the list the trace kernels loop over
//
// A leaf of the reference tree is reported exactly when the ray passes the leaf's own box test (every box on
// the path contains it and entry/exit distances are monotone in the bounds), so `filter` restricted to `live`
// primitives returns the reference's set without walking a tree.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "trace_common.hpp"

namespace art {

constexpr int kFilterBlock = 256;

struct Beam {                         // 8 floats per heliostat
    float cx, cy, cz, r;              // bounding sphere of the ray origins
    float dx, dy, dz, theta;          // unit mean reflected direction, cone half angle
};

__device__ __forceinline__ float block_reduce(float v, float* s_red, bool is_max)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_xor(v, off, 64);
        v = is_max ? fmaxf(v, o) : v + o;
    }
    __syncthreads();
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    float r = s_red[0];
    for (int w = 1; w < nw; ++w) r = is_max ? fmaxf(r, s_red[w]) : r + s_red[w];
    return r;
}

// max |angle| over both distortion views (only when the caller does not know the bound)
__global__ __launch_bounds__(256) void max_abs_kernel(const float* __restrict__ du, const float* __restrict__ de,
                                                      int64_t sh, int64_t sr, int64_t sp, int H, int R, int P,
                                                      unsigned* __restrict__ out_bits)
{
    __shared__ float s_red[16];
    const int64_t total = (int64_t)H * R * P;
    float m = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int p = (int)(i % P);
        const int r = (int)((i / P) % R);
        const int h = (int)(i / ((int64_t)P * R));
        const int64_t off = (int64_t)h * sh + (int64_t)r * sr + (int64_t)p * sp;
        m = fmaxf(m, fmaxf(fabsf(du[off]), fabsf(de[off])));
    }
    m = block_reduce(m, s_red, true);
    if (threadIdx.x == 0) atomicMax(out_bits, __float_as_uint(m));     // non-negative floats order like their bits
}

__global__ __launch_bounds__(256) void beam_kernel(TraceArgs a, float max_scatter, const unsigned* __restrict__ scatter_bits,
                                                   Beam* __restrict__ beams)
{
    __shared__ float s_red[16];
    const int h = blockIdx.x;
    const float4* __restrict__ org = a.origins + (int64_t)h * a.P;
    const float4* __restrict__ nrm = a.normals + (int64_t)h * a.P;
    const float4 inc = a.incident[h];
    float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f}, sd[3] = {0.f, 0.f, 0.f};
    for (int p = threadIdx.x; p < a.P; p += blockDim.x) {
        const float4 o = org[p];
        float4 d; float s;
        reflect(inc, nrm[p], d, s);
        const float il = rsqrtf(fmaxf(d.x * d.x + d.y * d.y + d.z * d.z, 1e-30f));
        lo[0] = fminf(lo[0], o.x); lo[1] = fminf(lo[1], o.y); lo[2] = fminf(lo[2], o.z);
        hi[0] = fmaxf(hi[0], o.x); hi[1] = fmaxf(hi[1], o.y); hi[2] = fmaxf(hi[2], o.z);
        sd[0] += d.x * il; sd[1] += d.y * il; sd[2] += d.z * il;
    }
    for (int c = 0; c < 3; ++c) {
        lo[c] = -block_reduce(-lo[c], s_red, true);
        hi[c] = block_reduce(hi[c], s_red, true);
        sd[c] = block_reduce(sd[c], s_red, false);
    }
    const float nl = rsqrtf(fmaxf(sd[0] * sd[0] + sd[1] * sd[1] + sd[2] * sd[2], 1e-30f));
    const float mx = sd[0] * nl, my = sd[1] * nl, mz = sd[2] * nl;
    float cmin = 1.0f;
    for (int p = threadIdx.x; p < a.P; p += blockDim.x) {
        float4 d; float s;
        reflect(inc, nrm[p], d, s);
        const float il = rsqrtf(fmaxf(d.x * d.x + d.y * d.y + d.z * d.z, 1e-30f));
        cmin = fminf(cmin, (d.x * mx + d.y * my + d.z * mz) * il);
    }
    cmin = -block_reduce(-cmin, s_red, true);
    if (threadIdx.x == 0) {
        const float scatter = max_scatter >= 0.0f ? max_scatter : __uint_as_float(*scatter_bits);
        Beam b;
        b.cx = 0.5f * (lo[0] + hi[0]); b.cy = 0.5f * (lo[1] + hi[1]); b.cz = 0.5f * (lo[2] + hi[2]);
        const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
        b.r = 0.5f * sqrtf(ex * ex + ey * ey + ez * ez) * 1.001f + 1e-4f;
        b.dx = mx; b.dy = my; b.dz = mz;
        // the two rotations of the scatter compose to at most sqrt(2) x the larger angle (+ slack for rounding)
        b.theta = acosf(fminf(fmaxf(cmin, -1.0f), 1.0f)) + 1.4143f * scatter * 1.001f + 1e-4f;
        beams[h] = b;
    }
}

// cand_count[h] = number of primitives inside heliostat h's cone, ids in cand[h] in ASCENDING order (the order in which the
// kernels add the rectangles' sigmas: the same in every run), as many as the row holds (Cmax; a longer list is reported).
__global__ __launch_bounds__(256) void cull_kernel(const Beam* __restrict__ beams, const float* __restrict__ corners, int N,
                                                   int Cmax, int* __restrict__ cand, int* __restrict__ cand_count)
{
    __shared__ int s_wave[4];
    const int h = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const Beam b = beams[h];
    const float tan_t = b.theta < 1.5f ? tanf(b.theta) : 1e30f;
    int base = 0;                                   // (workgroup-uniform) candidates among the rectangles before this round's
    for (int k0 = 0; k0 < N; k0 += 256) {
        const int k = k0 + (int)threadIdx.x;
        bool in_cone = false;
        if (k < N) {
            const float* c = corners + 16 * (int64_t)k;
            float lo[3], hi[3], cen[3];
            for (int ax = 0; ax < 3; ++ax) {
                const float v0 = c[ax], v1 = c[4 + ax], v2 = c[8 + ax], v3 = c[12 + ax];
                lo[ax] = fminf(fminf(v0, v1), fminf(v2, v3));
                hi[ax] = fmaxf(fmaxf(v0, v1), fmaxf(v2, v3));
                cen[ax] = 0.5f * (lo[ax] + hi[ax]);
            }
            // radius: the whole box (the filter tests the box) widened by the soft edge of the mask (3 % of a span)
            const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
            const float rho = 0.5f * sqrtf(ex * ex + ey * ey + ez * ez) * 1.04f + 2e-3f;
            const float wx = cen[0] - b.cx, wy = cen[1] - b.cy, wz = cen[2] - b.cz;
            const float ts = wx * b.dx + wy * b.dy + wz * b.dz;
            const float perp = sqrtf(fmaxf(wx * wx + wy * wy + wz * wz - ts * ts, 0.0f));
            const float reach = b.r + rho;
            in_cone = ts >= -reach && perp <= reach + (ts + reach) * tan_t;
        }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(in_cone);
        if (lane == 0) s_wave[wave] = __popcll(m);
        __syncthreads();
        int before = base;
        for (int w = 0; w < wave; ++w) before += s_wave[w];
        if (in_cone) {
            const int slot = before + __popcll(m & ((1ull << lane) - 1ull));
            if (slot < Cmax) cand[(int64_t)h * Cmax + slot] = k;
        }
        base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) cand_count[h] = base;
}

// ---------------------------------------------------------------------------------------------------
// live[k]: can the reference's tree reach primitive k from its root?  One workgroup, phases separated by barriers.
// ws: keys[M] (uint64, M = next power of two >= N), codes[N], order[N], left[N], right[N], complete[N], reach[2N].
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lcp30(const int* __restrict__ codes, int N, int i, int j)     // blocking.py:446-510
{
    if (j < 0 || j >= N) return -1;
    const unsigned x = (unsigned)(codes[i] ^ codes[j]);
    return x == 0u ? 30 : 29 - (31 - __clz((int)x));
}

__device__ __forceinline__ unsigned expand_bits10(unsigned v)                               // :357-389
{
    unsigned x = v & 0x3FFu;
    x = (x | (x << 16)) & 0x030000FFu;
    x = (x | (x << 8)) & 0x0300F00Fu;
    x = (x | (x << 4)) & 0x030C30C3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}

__global__ __launch_bounds__(1024) void lbvh_live_kernel(const float* __restrict__ corners, int N, int M,
                                                         unsigned long long* __restrict__ keys, int* __restrict__ codes,
                                                         int* __restrict__ order, int* __restrict__ left,
                                                         int* __restrict__ right, int* __restrict__ complete,
                                                         int* __restrict__ reach, int* __restrict__ live)
{
    __shared__ float s_red[16];
    __shared__ int s_changed;
    const int tid = threadIdx.x, nt = blockDim.x;
    if (N == 1) { if (tid == 0) live[0] = 1; return; }
    // centroids (:570) - mean of the four corners - and their bounding box (:427-428)
    float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
    for (int k = tid; k < N; k += nt)
        for (int ax = 0; ax < 3; ++ax) {
            const float* c = corners + 16 * (int64_t)k + ax;
            const float m = (((c[0] + c[4]) + c[8]) + c[12]) / 4.0f;
            lo[ax] = fminf(lo[ax], m); hi[ax] = fmaxf(hi[ax], m);
        }
    for (int ax = 0; ax < 3; ++ax) { lo[ax] = -block_reduce(-lo[ax], s_red, true); hi[ax] = block_reduce(hi[ax], s_red, true); }
    const float span = fmaxf(fmaxf(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
    const float scale = 1023.0f / (span + 1e-6f);                                            // :430-431
    for (int k = tid; k < M; k += nt) {
        unsigned long long key = ~0ull;
        if (k < N) {
            unsigned q[3];
            for (int ax = 0; ax < 3; ++ax) {
                const float* c = corners + 16 * (int64_t)k + ax;
                const float m = (((c[0] + c[4]) + c[8]) + c[12]) / 4.0f;
                q[ax] = (unsigned)(int)((m - lo[ax]) * scale);
            }
            const unsigned code = (expand_bits10(q[1]) << 2) | (expand_bits10(q[0]) << 1) | expand_bits10(q[2]);   // :437-443
            key = ((unsigned long long)code << 32) | (unsigned)k;      // unique keys: ties keep the index order
        }
        keys[k] = key;
    }
    __syncthreads();
    for (int k2 = 2; k2 <= M; k2 <<= 1)                                                      // bitonic sort (:573)
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < M; i += nt) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long x = keys[i], y = keys[ixj];
                    const bool up = (i & k2) == 0;
                    if ((x > y) == up) { keys[i] = y; keys[ixj] = x; }
                }
            }
            __syncthreads();
        }
    for (int k = tid; k < N; k += nt) { codes[k] = (int)(keys[k] >> 32); order[k] = (int)(keys[k] & 0xFFFFFFFFu); complete[k] = 0; }
    for (int k = tid; k < 2 * N; k += nt) reach[k] = 0;
    __syncthreads();
    const int leaf_offset = N - 1;
    for (int i = tid; i < N - 1; i += nt) {                                                  // :578-699, per internal node
        const int lr = lcp30(codes, N, i, i + 1), ll = lcp30(codes, N, i, i - 1);
        const int d = lr > ll ? 1 : -1;
        const int dmin = min(lr, ll);
        int lmax = 2;
        while (lcp30(codes, N, i, i + lmax * d) > dmin) lmax *= 2;
        int l = 0;
        for (int t = lmax / 2; t >= 1; t /= 2)
            if (lcp30(codes, N, i, i + (l + t) * d) > dmin) l += t;
        const int j = i + l * d;
        const int dnode = lcp30(codes, N, i, j);
        int split = 0;
        for (int t = (l + 1) / 2; t >= 1; t /= 2)                                            // floor halving, as :640-650
            if (lcp30(codes, N, i, i + (split + t) * d) > dnode) split += t;
        const int gamma = i + split * d + min(d, 0);
        const int lo_i = min(i, j), hi_i = max(i, j);
        left[i] = lo_i == gamma ? leaf_offset + gamma : gamma;
        right[i] = hi_i == gamma + 1 ? leaf_offset + gamma + 1 : gamma + 1;
    }
    __syncthreads();
    // bottom-up box pass (:705-733): a node completes one round after both of its children
    for (int round = 0; round < 2 * (N - 1); ++round) {
        if (tid == 0) s_changed = 0;
        __syncthreads();
        for (int i = tid; i < N - 1; i += nt) {
            if (complete[i]) continue;
            const int L = left[i], Rr = right[i];
            const bool l_ok = L >= leaf_offset || (L >= 0 && complete[L] == 1);
            const bool r_ok = Rr >= leaf_offset || (Rr >= 0 && complete[Rr] == 1);
            if (l_ok && r_ok) { complete[i] = 2; s_changed = 1; }
        }
        __syncthreads();
        for (int i = tid; i < N - 1; i += nt)
            if (complete[i] == 2) complete[i] = 1;
        const bool again = s_changed != 0;
        __syncthreads();
        if (!again) break;
    }
    // top-down reachability through complete nodes (an incomplete node keeps an all-zero box)
    if (tid == 0 && complete[0] == 1) reach[0] = 1;
    __syncthreads();
    for (int round = 0; round < 2 * N; ++round) {
        if (tid == 0) s_changed = 0;
        __syncthreads();
        for (int i = tid; i < N - 1; i += nt) {
            if (!reach[i]) continue;
            const int ch[2] = {left[i], right[i]};
            for (int c = 0; c < 2; ++c) {
                const int node = ch[c];
                if (node < 0 || node >= 2 * N - 1 || reach[node]) continue;
                if (node >= leaf_offset || complete[node] == 1) { reach[node] = 1; s_changed = 1; }
            }
        }
        __syncthreads();
        const bool again = s_changed != 0;
        __syncthreads();
        if (!again) break;
    }
    for (int k = tid; k < N; k += nt) live[order[k]] = reach[leaf_offset + k];
}

__global__ void fill_int_kernel(int* p, int n, int v)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ---------------------------------------------------------------------------------------------------
// filter: thread <-> point, loop over a chunk of samples; every ray tests the boxes of the candidates that are
// still undecided.  grid.x = H * n_ptiles * n_rchunks.
// ---------------------------------------------------------------------------------------------------
struct FilterTables { float box[kMaxCand][6]; float sph[kMaxCand][4]; int id[kMaxCand]; int done[kMaxCand]; int n, left; };

// One batch of at most kMaxCand candidates of heliostat h (entries c0 ... c0 + nb - 1 of its list): the tables and the lanes'
// masks hold that many.  The returns before the last barrier are workgroup-uniform.
template <bool INTERLEAVED>
__device__ __forceinline__ void blocking_filter_batch(const TraceArgs& a, const float* __restrict__ corners, const int* __restrict__ owner,
                                                      const int* __restrict__ cand, int Cmax, const int* __restrict__ live,
                                                      int* __restrict__ flags, FilterTables& tab, int h, int ptile, int rchunk, int c0, int nb)
{
    auto& s_box = tab.box; auto& s_sph = tab.sph; auto& s_id = tab.id; auto& s_done = tab.done;
    int& s_n = tab.n; int& s_left = tab.left;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const int own = owner[h];
    for (int c = threadIdx.x; c < nb; c += blockDim.x) {
        const int k = cand[(int64_t)h * Cmax + c0 + c];
        // undecided = foreign (:944-947), reachable in the reference tree, not flagged by an earlier workgroup
        if (k != own && live[k] && !*(volatile const int*)(flags + k)) {
            const int slot = atomicAdd(&s_n, 1);
            s_id[slot] = k; s_done[slot] = 0;
            const float* cr = corners + 16 * (int64_t)k;
            for (int ax = 0; ax < 3; ++ax) {
                const float v0 = cr[ax], v1 = cr[4 + ax], v2 = cr[8 + ax], v3 = cr[12 + ax];
                s_box[slot][ax] = fminf(fminf(v0, v1), fminf(v2, v3));            // :568-569
                s_box[slot][3 + ax] = fmaxf(fmaxf(v0, v1), fmaxf(v2, v3));
                s_sph[slot][ax] = 0.5f * (s_box[slot][ax] + s_box[slot][3 + ax]);
            }
            const float ex = s_box[slot][3] - s_box[slot][0], ey = s_box[slot][4] - s_box[slot][1],
                        ez = s_box[slot][5] - s_box[slot][2];
            s_sph[slot][3] = 0.5f * sqrtf(ex * ex + ey * ey + ez * ez) * 1.001f + 1e-4f;
        }
    }
    __syncthreads();
    const int n = s_n;
    if (n == 0) return;
    if (threadIdx.x == 0) s_left = n;
    __syncthreads();

    const int p = ptile * kFilterBlock + threadIdx.x;
    const bool active = p < a.P;
    const int t = a.target_idx[h];
    if ((unsigned)t >= (unsigned)(a.T + a.Tc)) return;     // workgroup-uniform; art_trace_fwd reports the bad index (ART_ETARGET)
    const bool is_cyl = t >= a.T;
    Plane pl; Cyl cy;
    if (is_cyl) cy = load_cyl(a.cyl_centers, a.cyl_normals, a.cyl_axes, a.cyl_radii, a.cyl_heights, a.cyl_opening,
                              t - a.T, a.W, a.Hh, a.mag, a.k_ext, a.k_refl);
    else pl = load_plane(a.centers, a.pnormals, a.dims, t, a.W, a.Hh, a.mag, a.k_ext, a.k_refl);
    float4 o = make_float4(0.f, 0.f, 0.f, 1.f), d = make_float4(0.f, 0.f, -1.f, 0.f);
    float numer = 0.0f; CylPoint cp = {0.f, 0.f, 0.f, 0.f};
    if (active) {
        o = a.origins[(int64_t)h * a.P + p];
        float s;
        reflect(a.incident[h], a.normals[(int64_t)h * a.P + p], d, s);
        if (is_cyl) cp = cyl_point(cy, o); else numer = plane_numer(pl, o);
    }
    // boxes a ray of this point can reach at all: bounding sphere against the cone of the point's scattered rays
    unsigned pmask = 0u;
    if (active) {
        const float il = rsqrtf(fmaxf(d.x * d.x + d.y * d.y + d.z * d.z, 1e-30f));
        const float dx = d.x * il, dy = d.y * il, dz = d.z * il;
        for (int c = 0; c < n; ++c) {
            const float wx = s_sph[c][0] - o.x, wy = s_sph[c][1] - o.y, wz = s_sph[c][2] - o.z;
            const float l2 = wx * wx + wy * wy + wz * wz, tt = wx * dx + wy * dy + wz * dz;
            const float perp = sqrtf(fmaxf(l2 - tt * tt, 0.0f)), rho = s_sph[c][3];
            if (!(perp * a.cone_cos - tt * a.cone_sin <= rho || l2 <= rho * rho)) continue;
            // The sphere is a loose hull of a flat rectangle's box: a candidate that no ray hits would keep every
            // workgroup of this heliostat tracing all its samples (2.5 ms of the exact mode at the metric size).  A ray
            // within theta of the chief ray that meets the box at X (|X - o| <= |w| + rho) passes within
            // delta = (|w| + rho) sin(theta) of the chief ray, so the chief ray must hit the box grown by delta.
            if (a.cone_cos > 0.0f) {
                const float delta = (sqrtf(l2) + rho) * a.cone_sin * 1.001f + 1e-4f;
                const float jx = 1.0f / (fabsf(dx) > 1e-12f ? dx : 1e-12f), jy = 1.0f / (fabsf(dy) > 1e-12f ? dy : 1e-12f),
                            jz = 1.0f / (fabsf(dz) > 1e-12f ? dz : 1e-12f);
                const float x0 = (s_box[c][0] - delta - o.x) * jx, x1 = (s_box[c][3] + delta - o.x) * jx;
                const float y0 = (s_box[c][1] - delta - o.y) * jy, y1 = (s_box[c][4] + delta - o.y) * jy;
                const float z0 = (s_box[c][2] - delta - o.z) * jz, z1 = (s_box[c][5] + delta - o.z) * jz;
                const float entry = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
                const float exit_ = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
                if (!(exit_ >= entry && exit_ >= 0.0f)) continue;
            }
            pmask |= 1u << c;
        }
    }
    const unsigned wmask = wave_or_mask(pmask, n);
    if (wmask == 0u) return;                                      // (no barrier below this point)
    const int r0 = rchunk * a.r_chunk;
    const int r1 = min(r0 + a.r_chunk, a.R);
    int64_t off = (int64_t)h * a.sh + (int64_t)r0 * a.sr + (int64_t)p * a.sp;
    for (int r = r0; r < r1; ++r, off += a.sr) {
        if (*(volatile int*)&s_left <= 0) break;                  // every candidate of this heliostat is decided
        // ... also by OTHER workgroups: a rectangle this workgroup's rays only graze is usually hit squarely by somebody
        // else's, and without a look at the global flags these samples would all be traced for nothing.
        // (every wave looks for itself: the others may have left already)
        const int lane = threadIdx.x & 63;
        if (((r - r0) & 3) == 3 && lane < n && !*(volatile int*)&s_done[lane] &&
            *(volatile const int*)(flags + s_id[lane]) && atomicExch(&s_done[lane], 1) == 0)
            atomicSub(&s_left, 1);
        float rx = 0.f, ry = 0.f, rz = -1.f, tt = 0.f;
        if (active) {
            float u, e;
            load_dist<INTERLEAVED>(a, off, u, e);
            const Rot m = make_rot(e, u);
            scatter(m, d, rx, ry, rz);
            if (is_cyl) { const CylHit ch = cyl_hit(cy, cp, rx, ry, rz); tt = ch.ok ? ch.t : 0.0f; }     // geometry.py:430-432
            else { const Hit hit = intersect(pl, o, numer, rx, ry, rz); tt = hit.valid ? hit.t : 0.0f; }  // :186-190
        }
        const float ix = 1.0f / (rx + 1e-12f), iy = 1.0f / (ry + 1e-12f), iz = 1.0f / (rz + 1e-12f);     // blocking.py:912
        for (unsigned mm = wmask; mm != 0u; mm &= mm - 1u) {
            const int c = __builtin_ctz(mm);
            if (*(volatile int*)&s_done[c]) continue;
            // slab test, :787-791, and the hit condition :928-932
            const float x0 = (s_box[c][0] - o.x) * ix, x1 = (s_box[c][3] - o.x) * ix;
            const float y0 = (s_box[c][1] - o.y) * iy, y1 = (s_box[c][4] - o.y) * iy;
            const float z0 = (s_box[c][2] - o.z) * iz, z1 = (s_box[c][5] - o.z) * iz;
            const float entry = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
            const float exit_ = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
            const bool hit = ((pmask >> c) & 1u) && exit_ >= entry && exit_ > 1e-6f && entry <= tt;
            if (wave_any(hit)) {
                if ((threadIdx.x & 63) == 0 && atomicExch(&s_done[c], 1) == 0) {
                    flags[s_id[c]] = 1;
                    atomicSub(&s_left, 1);
                }
            }
        }
    }
}

// The first kMaxCand candidates of every list (REST = false: for every field met so far, the whole list), and - a second
// launch, made when the rows are wider than the tables - the rest of the longer lists in batches of kMaxCand (the samples are
// traced once per batch; a workgroup whose list fits the tables leaves at once).  Two instantiations rather than one loop: as a
// loop the kernel took 97 instead of 66 VGPRs and ran 2.4 times slower at the metric size.
template <bool INTERLEAVED, bool REST>
__global__ __launch_bounds__(kFilterBlock) void blocking_filter_kernel(TraceArgs a, const float* __restrict__ corners,
                                                                       const int* __restrict__ owner,
                                                                       const int* __restrict__ cand,
                                                                       const int* __restrict__ cand_count, int Cmax,
                                                                       const int* __restrict__ live,
                                                                       int* __restrict__ flags)
{
    __shared__ FilterTables tab;
    const int bid = blockIdx.x;
    const int ptile = bid % a.n_ptiles;
    const int rchunk = (bid / a.n_ptiles) % a.n_rchunks;
    const int h = bid / (a.n_ptiles * a.n_rchunks);
    const int nc = min(cand_count[h], Cmax);
    if constexpr (!REST) {
        blocking_filter_batch<INTERLEAVED>(a, corners, owner, cand, Cmax, live, flags, tab, h, ptile, rchunk, 0, min(nc, kMaxCand));
    } else {
        for (int c0 = kMaxCand; c0 < nc; c0 += kMaxCand) {
            blocking_filter_batch<INTERLEAVED>(a, corners, owner, cand, Cmax, live, flags, tab, h, ptile, rchunk, c0, min(nc - c0, kMaxCand));
            __syncthreads();                 // every wave is done with this batch's tables
        }
    }
}

// cand[h] <- its flagged entries (the heliostat's own rectangle included when foreign rays flagged it: the
// reference's mask is evaluated against every filtered primitive for every ray).
__global__ void compact_kernel(int* __restrict__ cand, int* __restrict__ cand_count, int H, int Cmax,
                               const int* __restrict__ flags, unsigned* __restrict__ status)
{
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= H) return;
    const int nc = min(cand_count[h], Cmax);
    int m = 0;
    for (int c = 0; c < nc; ++c) {
        const int k = cand[(int64_t)h * Cmax + c];
        if (flags[k]) cand[(int64_t)h * Cmax + m++] = k;
    }
    // a count above Cmax (overflow of the cull) is kept so that the host can see it, and reported through the device
    // status word (art_async_status / the next trace call: ART_ECANDIDATES) - no host read of the counts per call
    if (cand_count[h] <= Cmax) cand_count[h] = m;
    else if (status != nullptr) atomicOr(status, 2u);
}

static int next_pow2(int64_t n)
{
    int m = 1;
    while (m < n) m <<= 1;
    return m;
}

}  // namespace art

using namespace art;

// ws layout (bytes): beams[H] | scatter bits | live[N] | keys[M] | codes,order,left,right,complete[N] | reach[2N]
extern "C" int64_t art_blocking_workspace_bytes(int64_t H, int64_t N)
{
    if (H < 0 || N < 0) return -1;
    const int64_t M = next_pow2(N > 0 ? N : 1);
    return (int64_t)sizeof(Beam) * H + 16 + 4 * N + 8 * M + 4 * 5 * N + 4 * 2 * N + 64;
}

extern "C" int art_blocking_filter(const float* origins, const float* normals, const float* incident,
                                   const float* dist_u, const float* dist_e, int64_t dist_sh, int64_t dist_sr,
                                   int64_t dist_sp, const int32_t* target_idx, const float* plane_centers,
                                   const float* plane_normals, const float* plane_dims, const float* cyl_centers,
                                   const float* cyl_normals, const float* cyl_axes, const float* cyl_radii,
                                   const float* cyl_heights, const float* cyl_opening, double ray_magnitude,
                                   int64_t H, int64_t R, int64_t P, int64_t T, int64_t Tc, int64_t W, int64_t Hh,
                                   const float* prim_corners, const int32_t* owner, int64_t N,
                                   double max_scatter_angle, int lbvh_compat, int64_t Cmax, int32_t* flags,
                                   int32_t* cand, int32_t* cand_count, void* workspace, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (H == 0) return ART_OK;
    TraceArgs a;
    if (!prim_corners || !owner || !flags || !cand || !cand_count || !workspace || N <= 0 || N > (1 << 22) ||
        Cmax < 1 || Cmax > (1 << 22) ||
        !fill_args(a, origins, normals, incident, dist_u, dist_e, dist_sh, dist_sr, dist_sp, target_idx, plane_centers,
                   plane_normals, plane_dims, cyl_centers, cyl_normals, cyl_axes, cyl_radii, cyl_heights, cyl_opening,
                   ray_magnitude, 0.0, 1.0, H, R, P, T, Tc, W, Hh, 0))
        return ART_EINVAL;
    set_cone(a, max_scatter_angle);
    char* ws = static_cast<char*>(workspace);
    Beam* beams = reinterpret_cast<Beam*>(ws); ws += sizeof(Beam) * H;
    unsigned* scatter_bits = reinterpret_cast<unsigned*>(ws); ws += 16;
    int* live = reinterpret_cast<int*>(ws); ws += 4 * N;
    ws = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(ws) + 7) & ~uintptr_t(7));
    const int M = next_pow2(N);
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(ws); ws += 8 * (int64_t)M;
    int* codes = reinterpret_cast<int*>(ws); ws += 4 * N;
    int* order = reinterpret_cast<int*>(ws); ws += 4 * N;
    int* left = reinterpret_cast<int*>(ws); ws += 4 * N;
    int* right = reinterpret_cast<int*>(ws); ws += 4 * N;
    int* complete = reinterpret_cast<int*>(ws); ws += 4 * N;
    int* reach = reinterpret_cast<int*>(ws);

    ART_HIP(hipMemsetAsync(flags, 0, sizeof(int32_t) * N, stream));
    if (max_scatter_angle < 0.0) {
        ART_HIP(hipMemsetAsync(scatter_bits, 0, 4, stream));
        hipLaunchKernelGGL(max_abs_kernel, dim3(1024), dim3(256), 0, stream, a.dist_u, a.dist_e, a.sh, a.sr, a.sp, a.H,
                           a.R, a.P, scatter_bits);
    }
    hipLaunchKernelGGL(beam_kernel, dim3((unsigned)H), dim3(256), 0, stream, a, (float)max_scatter_angle, scatter_bits, beams);
    hipLaunchKernelGGL(cull_kernel, dim3((unsigned)H), dim3(256), 0, stream, beams, prim_corners, (int)N, (int)Cmax, cand,
                       cand_count);
    if (lbvh_compat)
        hipLaunchKernelGGL(lbvh_live_kernel, dim3(1), dim3(1024), 0, stream, prim_corners, (int)N, M, keys, codes, order,
                           left, right, complete, reach, live);
    else
        hipLaunchKernelGGL(fill_int_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream, live, (int)N, 1);
    ART_HIP(hipGetLastError());
    // samples are chunked so that a few thousand workgroups share the work; most exit at once (no candidates)
    a.n_ptiles = (int)((P + kFilterBlock - 1) / kFilterBlock);
    {
        const int64_t base = (int64_t)a.H * a.n_ptiles;
        int64_t want = (4096 + base - 1) / base;
        if (want < 1) want = 1;
        int chunk = (int)((a.R + want - 1) / want);
        if (chunk < 8) chunk = 8;
        if (chunk > a.R) chunk = a.R;
        a.r_chunk = chunk;
        a.n_rchunks = (a.R + chunk - 1) / chunk;
    }
    const int64_t blocks = (int64_t)a.H * a.n_ptiles * a.n_rchunks;
    if (blocks > 2147483647LL) return ART_EINVAL;
    if (interleaved_layout(a))
        hipLaunchKernelGGL((blocking_filter_kernel<true, false>), dim3((unsigned)blocks), dim3(kFilterBlock), 0, stream, a,
                           prim_corners, owner, cand, cand_count, (int)Cmax, live, flags);
    else
        hipLaunchKernelGGL((blocking_filter_kernel<false, false>), dim3((unsigned)blocks), dim3(kFilterBlock), 0, stream, a,
                           prim_corners, owner, cand, cand_count, (int)Cmax, live, flags);
    if (Cmax > kMaxCand) {             // the rest of the lists that are longer than the tables (usually none: the workgroups leave at once)
        if (interleaved_layout(a))
            hipLaunchKernelGGL((blocking_filter_kernel<true, true>), dim3((unsigned)blocks), dim3(kFilterBlock), 0, stream, a,
                               prim_corners, owner, cand, cand_count, (int)Cmax, live, flags);
        else
            hipLaunchKernelGGL((blocking_filter_kernel<false, true>), dim3((unsigned)blocks), dim3(kFilterBlock), 0, stream, a,
                               prim_corners, owner, cand, cand_count, (int)Cmax, live, flags);
    }
    hipLaunchKernelGGL(compact_kernel, dim3((unsigned)((H + 255) / 256)), dim3(256), 0, stream, cand, cand_count, (int)H,
                       (int)Cmax, flags, status_word().dev);
    ART_HIP(hipGetLastError());
    return ART_OK;
}
