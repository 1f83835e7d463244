// trace_common.hpp - argument block and distortion fetch shared by the trace and blocking kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>
#include <stdint.h>
#include <math.h>
#include <stdlib.h>

#include "ray_math.hpp"
#include "launch_common.hpp"

namespace art {

constexpr int kBlock = 256;     // 4 waves; one wave per SIMD, several blocks per CU
constexpr int kMaxCand = 32;    // candidates of a heliostat held in LDS (116 B per rectangle) and in the lanes' 32-bit masks; the rest of a
                                // longer list is read from the caller's tables (trace_kernels.hip: "wide" heliostats)

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
    const float* cyl_centers;   // [Tc,4]  TowerTargetAreasCylindrical tensors (NULL when Tc == 0)
    const float* cyl_normals;   // [Tc,4]
    const float* cyl_axes;      // [Tc,4]
    const float* cyl_radii;     // [Tc]
    const float* cyl_heights;   // [Tc]
    const float* cyl_opening;   // [Tc]
    const float* prim_corners;  // [N,4,4] blocking rectangles (NULL: blocking off), artist/raytracing/blocking.py:123-209
    const float* prim_spans;    // [N,2,4]
    const float* prim_normals;  // [N,4]
    const int32_t* cand;        // [H,Cmax] rectangles each heliostat's rays are tested against (art_blocking_filter)
    const int32_t* cand_count;  // [H]
    int Cmax;                   // row width of `cand` (any size: the first kMaxCand entries of a row go to LDS, the others are "wide")
    double* wide_grad;          // backward: [H, Cmax - (kMaxCand - 1), 12] gradient sums of the listed candidates of wide heliostats (or NULL)
    float cone_cos, cone_sin;   // of the largest angle between a ray and its point's chief ray (0, 0: unknown)
    int slab_cull;              // per-point rectangle culling also by the three slabs of cone_mask (ray_math.hpp)
    float mag, k_ext, k_refl;
    int H, R, P, T, Tc, W, Hh;  // target index t < T: planar area t; T <= t < T + Tc: cylinder t - T
    int mode;                 // 0: bitmap per heliostat, 1: bitmap per target
    int r_chunk;              // samples per block
    int n_rchunks;            // ceil(R / r_chunk)
    int reverse_bwd;          // backward queue order (experiment knob; the list order measured best)
    int reverse_items;        // forward queue order: 0 first-to-last, 1 last-to-first, -1 decided on the device (the end
                              // of the list whose heliostat is farther from its target goes first)
    int n_ptiles;             // ceil(P / kBlock)
    int p_block;              // points per workgroup (LDS-window kernel)
    int n_pblocks;            // point blocks per heliostat
    int facet_points;         // blocks subdivide runs of this many consecutive points (a facet, several facets, or P)
    int blocks_per_facet;     // ceil(facet_points / p_block)
    int tile_cap;             // LDS bitmap-window capacity in pixels
    int split;                // 0, or which heliostats this launch owns when blocking is on: 1 unblocked (lean kernels), 2 blocked
    int pack_edge;            // lean backward kernel: 0, or the edge margin in 1/64 of the scatter pad (edge points are packed)
    int multipass_ratio;      // footprints above ratio x capacity are swept in several passes
    int h_group, n_groups;    // forward, mode 1, few samples per point: an item is a group of h_group consecutive heliostats
                              // (trace_fwd_item_field); 1: one heliostat per item row
    int tail_h, tail_bpf, tail_pblock, tail_npb;   // the queue's finer-grained end: see work_item_count / set_queue_tail
    int win_sample;           // window phase on one evenly spaced point per thread (1) or on every point of the block (0)
    // Forward accumulation (windowed kernels): every bitmap pixel has a 64-bit FIXED-POINT accumulator in `accum`
    // ([n_maps,Hh,W], all zero on entry).  Window flushes, cell carries and stray rays add integers to it - integer
    // addition is associative, so the flux is bit-reproducible whatever the order of the workgroups - and
    // accum_to_flux_kernel turns it into the fp32 bitmap (one rounding per pixel) and leaves it zero again.
    // One unit = sign(k) 2^(ex_g - 28) with 2^ex_g > |mag k_ext k_refl|: the largest single contribution is < 2^28 |d||m| units.
    unsigned long long* accum;
    // The windows of this launch's work items, by queue position, made by a kernel of their own before the trace kernel starts
    // (trace_kernels.hip: window_table_kernel) - or NULL: every item works its window out itself, as its first phase.
    const void* win_table;
    int ex_g;
    float scale_g;            // 2^(28 - ex_g)
    unsigned int* status;     // device status word (mapped host memory): bit 0 = a target index was out of range
};

// Target indices come from the caller: a stale one must not index the target tables.  The heliostat is skipped (its
// bitmap and factors stay zero) and bit 0 of the status word - mapped host memory, read by art_async_status - is set.
__device__ __forceinline__ bool target_in_range(const TraceArgs& a, int t)
{
    if ((unsigned)t < (unsigned)(a.T + a.Tc)) return true;
    if (threadIdx.x == 0 && a.status != nullptr) __hip_atomic_fetch_or(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return false;
}

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

// Same fetch with the address split into a wave-uniform row pointer (heliostat, sample -> SGPRs) and a 32-bit
// per-lane offset (point): compiles to global_load ... v_off, s[base:base+1] with no per-ray VALU address math.
template <bool INTERLEAVED>
__device__ __forceinline__ void load_dist_row(const float* __restrict__ row_u, const float* __restrict__ row_e,
                                              int lane_off, float& u, float& e)
{
    if constexpr (INTERLEAVED) {
        const float2 v = *reinterpret_cast<const float2*>(row_u + lane_off);
        u = v.x; e = v.y;
    } else {
        u = row_u[lane_off]; e = row_e[lane_off];
    }
}

// The same fetch for the hot sample loops: non-temporal (the 8 B/ray stream is read exactly once; MI355X_MICROARCH.md
// "nt-weights": issued -> landed ~18 % sooner), which matters because these kernels live on the latency of this stream.
template <bool INTERLEAVED>
__device__ __forceinline__ void load_dist_stream(const float* __restrict__ row_u, const float* __restrict__ row_e,
                                                 int lane_off, float& u, float& e)
{
    typedef float v2f __attribute__((ext_vector_type(2)));
    // scalar base + zero-extended 32-bit byte offset: the form `global_load ... v_off, s[base:base+1]` takes without any
    // 64-bit vector arithmetic (lane_off >= 0: an element offset inside one heliostat's row)
    const unsigned byte_off = (unsigned)lane_off * 4u;
    const char* pu = reinterpret_cast<const char*>(row_u) + byte_off;
    if constexpr (INTERLEAVED) {
        const v2f v = __builtin_nontemporal_load(reinterpret_cast<const v2f*>(pu));
        u = v.x; e = v.y;
    } else {
        const char* pe = reinterpret_cast<const char*>(row_e) + byte_off;
        u = __builtin_nontemporal_load(reinterpret_cast<const float*>(pu)); e = __builtin_nontemporal_load(reinterpret_cast<const float*>(pe));
    }
}

static inline bool fill_args(TraceArgs& a, const float* origins, const float* normals, const float* incident,
                      const float* dist_u, const float* dist_e, int64_t sh, int64_t sr, int64_t sp,
                      const int32_t* target_idx, const float* centers, const float* pnormals, const float* dims,
                      const float* cyl_centers, const float* cyl_normals, const float* cyl_axes, const float* cyl_radii,
                      const float* cyl_heights, const float* cyl_opening,
                      double mag, double ext, double refl, int64_t H, int64_t R, int64_t P, int64_t T, int64_t Tc,
                      int64_t W, int64_t Hh, int mode)
{
    if (!origins || !normals || !incident || !dist_u || !dist_e || !target_idx) return false;
    if (T < 0 || Tc < 0 || T + Tc <= 0 || T + Tc > (1 << 24)) return false;
    if (T > 0 && (!centers || !pnormals || !dims)) return false;
    if (Tc > 0 && (!cyl_centers || !cyl_normals || !cyl_axes || !cyl_radii || !cyl_heights || !cyl_opening)) return false;
    if (H < 0 || R <= 0 || P <= 0 || W < 2 || Hh < 2 || (mode != 0 && mode != 1)) return false;
    if (H > (1 << 24) || R > (1 << 24) || P > (1 << 26) || W > 32768 || Hh > 32768) return false;
    if ((double)R * (double)P >= 4294967296.0) return false;   // uint32 ray counters
    if (sp < 0 || (double)P * (double)sp >= 1073741824.0) return false;   // 32-bit per-lane distortion offsets
    a.origins = reinterpret_cast<const float4*>(origins);
    a.normals = reinterpret_cast<const float4*>(normals);
    a.incident = reinterpret_cast<const float4*>(incident);
    a.dist_u = dist_u; a.dist_e = dist_e; a.sh = sh; a.sr = sr; a.sp = sp;
    a.target_idx = target_idx; a.centers = centers; a.pnormals = pnormals; a.dims = dims;
    a.cyl_centers = cyl_centers; a.cyl_normals = cyl_normals; a.cyl_axes = cyl_axes; a.cyl_radii = cyl_radii;
    a.cyl_heights = cyl_heights; a.cyl_opening = cyl_opening; a.Tc = (int)Tc;
    a.prim_corners = a.prim_spans = a.prim_normals = nullptr; a.cand = a.cand_count = nullptr; a.Cmax = 0; a.wide_grad = nullptr; a.win_table = nullptr;
    a.cone_cos = a.cone_sin = 0.0f; a.slab_cull = 1;
    a.mag = (float)mag; a.k_ext = (float)(1.0 - ext); a.k_refl = (float)refl;
    a.H = (int)H; a.R = (int)R; a.P = (int)P; a.T = (int)T; a.W = (int)W; a.Hh = (int)Hh; a.mode = mode;
    a.n_ptiles = (int)((P + kBlock - 1) / kBlock);
    a.facet_points = (int)P; a.blocks_per_facet = 1; a.pack_edge = 0; a.split = 0;
    a.accum = nullptr; a.ex_g = 0; a.scale_g = 1.0f; a.status = nullptr;
    a.win_sample = 0; a.h_group = 1; a.n_groups = 0;
    a.tail_h = 0; a.tail_bpf = 1; a.tail_pblock = 0; a.tail_npb = 0;
    return true;
}


// Largest angle between a scattered ray and its point's chief ray: the two rotations of rotate_distortions compose
// to at most sqrt(2) x the larger angle (+ slack).  A negative bound = unknown: the per-point cone test accepts all.
static inline void set_cone(TraceArgs& a, double max_scatter_angle)
{
    a.slab_cull = debug_env_int("ARTIST_HIP_BLOCK_SLABS", 1) != 0;     // 0: sphere test only (diagnostic; same results)
    if (max_scatter_angle < 0.0) { a.cone_cos = a.cone_sin = 0.0f; return; }
    double theta = 1.4143 * max_scatter_angle * 1.001 + 1e-4;
    if (theta > 1.5) { a.cone_cos = a.cone_sin = 0.0f; return; }
    a.cone_cos = (float)cos(theta); a.cone_sin = (float)sin(theta);
}

static inline bool interleaved_layout(const TraceArgs& a)
{
    return a.dist_e == a.dist_u + 1 && a.sp == 2 && (a.sr % 2) == 0 && (a.sh % 2) == 0 &&
           (reinterpret_cast<uintptr_t>(a.dist_u) % 8) == 0;
}




// Device status word: 4 bytes of mapped host memory per GPU, allocated on the first trace call and kept.  Kernels set
// bit 0 when they meet a target index outside the tables (the heliostat is skipped), bit 1 when a heliostat has more
// candidate rectangles inside its ray cone than the tables hold (art_blocking_filter; the surplus is not evaluated); the host reads it without a
// synchronisation at the start of every trace call (a failure of an EARLIER launch then surfaces as ART_ETARGET / ART_ECANDIDATES) and,
// synchronised, in art_async_status.
struct StatusWord { unsigned* host; unsigned* dev; };
inline StatusWord status_word()
{
    constexpr int kMaxDevices = 64;
    static StatusWord words[kMaxDevices] = {};
    static std::mutex lock;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return {nullptr, nullptr};
    std::lock_guard<std::mutex> guard(lock);
    if (words[dev].host == nullptr) {
        unsigned* h = nullptr; unsigned* d = nullptr;
        if (hipHostMalloc(reinterpret_cast<void**>(&h), 64, hipHostMallocMapped) != hipSuccess) return {nullptr, nullptr};
        *h = 0u;
        if (hipHostGetDevicePointer(reinterpret_cast<void**>(&d), h, 0) != hipSuccess) { (void)hipHostFree(h); return {nullptr, nullptr}; }
        words[dev] = {h, d};
    }
    return words[dev];
}


}  // namespace art
