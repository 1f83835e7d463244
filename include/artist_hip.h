/*
 * artist_hip.h - C ABI of libartist_hip.so: the MI355X (gfx950) implementation of ARTIST's
 * heliostat ray-tracing hot path.
 *
 * ARTIST (v2.0.0) is pure Python/PyTorch and has no FFI of its own; the boundary its callers
 * see is the Python method surface quoted below.  Each entry point here replaces the chain of
 * ATen ops behind one of those methods, and is what a binding inside ARTIST would call
 * (INTEGRATION.md shows the ctypes stub).  Reference paths are relative to the ARTIST repo.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; all tensors fp32, row-major, contiguous unless
 *     an explicit element stride is passed; all pointers are DEVICE pointers;
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous on that stream.  Calls on DIFFERENT
 *     streams (from one host thread or several) may overlap freely - each brings its own output, scratch and
 *     accumulator buffers; calls on the SAME stream must come from one host thread at a time, like any HIP work;
 *   - process-wide state, all of it created on first use and kept: per GPU one sticky status word in mapped host
 *     memory (art_async_status below - shared by every stream: an error found by a kernel of one stream makes the
 *     trace entry points refuse on all of them until somebody clears it), per (GPU, stream) eight 4-byte work
 *     counters in device memory (zero whenever no launch is using them: the launch's last fetch resets them, so
 *     there is no bound on the number of queued launches), per (host thread, GPU) one side stream + two events
 *     for the second launch of a split call (blocking on / mixed towers), per (GPU, stream) 40 KB of device memory
 *     for the part sums of art_flux_crop_pixel_loss_fwd/_bwd on small batches (allocated by the first such call
 *     on a stream - which must therefore not be made while the stream is being captured into a graph; the values
 *     are rewritten by every call that reads them); nothing else survives a call;
 *   - return 0 on success, a negative ART_E* code otherwise (never throws across the ABI);
 *     art_strerror() maps a code to text;
 *   - outputs are fully written by the callee (zero-filled first where they are accumulators).
 */
#ifndef ARTIST_HIP_H
#define ARTIST_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ART_OK 0
#define ART_EINVAL -1      /* bad size / null pointer / unsupported degree */
#define ART_ETARGET -2     /* a target index was outside [0, T + Tc): found on the DEVICE (art_async_status) */
#define ART_ELAUNCH -3     /* HIP launch or runtime error (see art_last_hip_error) */
#define ART_EUNSUPPORTED -4
#define ART_EQUEUE -6      /* a work counter of this stream was not zero at the start of a call: an earlier launch on it ended abnormally (art_async_status) */
#define ART_ECANDIDATES -5 /* workspace exhausted: a heliostat has more blocking rectangles inside its ray cone than its candidate ROW (Cmax entries, the caller's choice up to N) holds: found on the DEVICE (art_async_status) */

/* Library / ABI version (bumped when a signature changes). */
int art_abi_version(void);
const char *art_strerror(int code);
/* hipError_t of the last failing HIP call made by this library on the calling thread (0 = none). */
int art_last_hip_error(void);

/* ---------------------------------------------------------------------------------------------
 * art_trace_fwd - HeliostatRayTracer.trace_rays, planar and cylindrical targets, blocking on or off:
 *   artist/raytracing/heliostat_ray_tracer.py:220-508 =
 *     geometry.reflect                (artist/raytracing/geometry.py:11-41)
 *   + scatter_rays/rotate_distortions (heliostat_ray_tracer.py:510-561, artist/geometry/transforms.py:7-83)
 *   + line_plane_intersections        (artist/raytracing/geometry.py:44-204)      planar target areas
 *   + line_cylinder_intersections     (artist/raytracing/geometry.py:207-445)     cylindrical target areas
 *   + intensity product               (heliostat_ray_tracer.py:482-487)
 *   + bilinear_splatting              (heliostat_ray_tracer.py:610-778)
 *   + the three diagnostic factors    (heliostat_ray_tracer.py:498-506)
 *   + (mode 1) get_bitmaps_per_target (heliostat_ray_tracer.py:563-608) fused into the splat.
 *
 *   origins, normals  [H,P,4]  heliostat_group.active_surface_points / _normals (aligned)
 *   incident          [H,4]    incident_ray_directions
 *   dist_u, dist_e    distortion angles; element (h,r,p) at base[h*dist_sh + r*dist_sr + p*dist_sp]
 *                     (element strides: Sun.get_distortions returns stride-2 views of one
 *                     interleaved [H,R,P,2] buffer, artist/scene/sun.py:227-234)
 *   target_idx        [H] int32 in [0, T + Tc): t < T is planar area t, t >= T is cylinder t - T
 *                     (the reference keeps two index lists, indices.planar_target_areas /
 *                     indices.cylindrical_target_areas, heliostat_ray_tracer.py:337-429; the binding
 *                     concatenates them in that order)
 *   plane_centers / plane_normals [T,4], plane_dims [T,2]  TowerTargetAreasPlanar tensors (NULL if T == 0)
 *   cyl_centers / cyl_normals / cyl_axes [Tc,4], cyl_radii / cyl_heights / cyl_opening [Tc]
 *                     TowerTargetAreasCylindrical tensors (NULL if Tc == 0)
 *   prim_corners [N,4,4], prim_spans [N,2,4], prim_normals [N,4]   blocking rectangles of ALL heliostats
 *                     (create_blocking_primitives_rectangles_by_index, artist/raytracing/blocking.py:123-209);
 *                     prim_corners == NULL <=> blocking_active=False
 *   cand [H,Cmax], cand_count [H]   per heliostat the rectangles its rays are tested against, as written by
 *                     art_blocking_filter (the filtered set of lbvh_filter_blocking_planes, :832-995); the soft
 *                     mask (soft_ray_blocking_mask, :212-354) is evaluated in the kernel for every ray.  Cmax is the
 *                     ROW WIDTH of `cand` - workspace, any size up to N, not a limit of the kernels: a heliostat's
 *                     first 32 rectangles live in LDS, the rest of a longer list is read from the list itself (the
 *                     reference has no such number: every ray meets every filtered rectangle)
 *   max_scatter_angle bound on |distortion angle| in radians (blocking only; < 0 = unknown): lets a surface point
 *                     skip the rectangles that none of its scattered rays can reach - speed only, never results
 *   ray_magnitude     Rays.ray_magnitudes fill value (heliostat_ray_tracer.py:185-203)
 *   extinction, reflectivity  trace_rays(ray_extinction_factor, mirror_reflectivity)
 *   facet_points      0, or the number M of consecutive surface points that form one facet (ARTIST's surface tensors are
 *                     [H, F * M, 4], facet-major: artist/field/surface.py, heliostat_group.py) - P must be a multiple.
 *                     A layout hint for speed, never for results: the kernels cut a heliostat's points into blocks that
 *                     share an LDS window, and a block that straddles two facets sees two separate images
 *   W, Hh             bitmap_resolution[0] (east / angle), bitmap_resolution[1] (up)
 *   mode              0: flux is [H,Hh,W] (one bitmap per active heliostat)
 *                     1: flux is [T+Tc,Hh,W] (summed per target area)
 *   flux              output, zero-filled then accumulated; rows already up-down flipped
 *   factors           output [3,H]: intercept, on_target, blocking fractions
 *   accum             [n_maps,Hh,W] uint64, 16-byte aligned, n_maps = H (mode 0) or T + Tc (mode 1): the pixels'
 *                     fixed-point accumulators.  ALL ZERO on entry, all zero again when the call's work has finished
 *                     (the caller keeps one buffer per stream and never clears it).  Window flushes, cell carries and
 *                     stray rays add integers to it and a last kernel converts it to `flux` with one rounding per
 *                     pixel: the flux does not depend on the order in which workgroups finish - two calls with the
 *                     same inputs give the same bits (the reference needs torch.use_deterministic_algorithms for
 *                     that, tests/conftest.py:109).  `flux` itself needs no initialisation.
 *   moments           NULL, or output [n_maps,4,3] fp64: per bitmap and quarter of its rows (sum f, sum x f, sum y f) with
 *                     x, y = torch.linspace(-1, 1, W / Hh) - the sums crop_flux_distributions_around_center starts from
 *                     (artist/flux/bitmap.py:165-182), left behind by the conversion pass, which streams every pixel
 *                     anyway; art_flux_crop_pixel_loss_fwd takes them.  Formed when Hh >= 4, Hh * W is even and
 *                     n_maps <= 65535; all-NaN otherwise (and for a heliostat whose blocking candidates overflowed).
 * Device memory: every buffer is the caller's, except 4 KB pages that the library allocates on first use and keeps
 * (work counters of its persistent workgroups: 32 bytes per stream that has made a trace call, see Conventions; the
 * first trace call on a stream must therefore not be made while that stream is being captured into a graph)
 * and 64 bytes of mapped host memory per GPU (the status word of art_async_status).
 * ------------------------------------------------------------------------------------------- */
int art_trace_fwd(const float *origins, const float *normals, const float *incident,
                  const float *dist_u, const float *dist_e, int64_t dist_sh, int64_t dist_sr, int64_t dist_sp,
                  const int32_t *target_idx, const float *plane_centers, const float *plane_normals,
                  const float *plane_dims, const float *cyl_centers, const float *cyl_normals,
                  const float *cyl_axes, const float *cyl_radii, const float *cyl_heights,
                  const float *cyl_opening, const float *prim_corners, const float *prim_spans,
                  const float *prim_normals, const int32_t *cand, const int32_t *cand_count, int64_t Cmax,
                  double max_scatter_angle, double ray_magnitude, double extinction, double reflectivity,
                  int64_t H, int64_t R, int64_t P, int64_t facet_points, int64_t T, int64_t Tc, int64_t W, int64_t Hh,
                  int mode, float *flux, float *factors, uint64_t *accum, double *moments, void *stream);

/* ---------------------------------------------------------------------------------------------
 * art_async_status - the entry points are asynchronous, so what only the DEVICE can find out is reported here:
 * synchronises `stream` and returns ART_ETARGET if a kernel launched through this library met a target index
 * outside [0, T + Tc) since the status was last cleared (the heliostat was skipped - no table is indexed out of
 * bounds - its bitmap and factors are zero, and so are its gradients in art_trace_bwd), ART_ECANDIDATES if
 * art_blocking_filter found more rectangles inside a heliostat's ray cone than its candidate row holds (Cmax < N was the
 * caller's choice: with Cmax = N it cannot happen.  The surplus is not evaluated:
 * that heliostat's blocking is incomplete; cand_count[h] holds the number found, and art_trace_fwd writes NaN
 * into that heliostat's bitmap - its target's bitmap in mode 1 - and factors, so the call that overflowed cannot
 * be mistaken for a result), ART_OK otherwise.  `clear` != 0 resets the status.  Until it is cleared, every later
 * art_trace_fwd / art_trace_bwd call returns the same code at once (checked without a synchronisation).
 * SCOPE: one status word per GPU, not per stream or call.  Work of OTHER streams is not waited for here - a
 * caller with several streams synchronises the device first (artist_amd.ops.check_async_errors does) - and a status
 * raised by one stream's kernel stops the trace calls of every stream until it is cleared (by any of them).
 * The reference fails in the same situation with an IndexError from its target-area gather
 * (artist/raytracing/geometry.py:104-105).
 * ------------------------------------------------------------------------------------------- */
int art_async_status(void *stream, int clear);

/* ---------------------------------------------------------------------------------------------
 * art_trace_bwd - what torch.autograd derives for the op chain of art_trace_fwd (mode 0 or 1):
 * indices and masks are constants, gradients flow through the bilinear weights, the Lambert
 * intensity and the hit point (heliostat_ray_tracer.py:285-290, 328-335, 390-429, 482-494,
 * 610-778; geometry.py:287-445 for cylinders).  Same inputs as the forward plus
 *   grad_flux     [H,Hh,W] (mode 0) or [T+Tc,Hh,W] (mode 1)
 *   grad_origins, grad_normals   outputs [H,P,4] (w components 0 / as autograd gives them)
 *   grad_scratch, grad_scratch_floats   see art_trace_bwd_scratch_floats
 *   grad_prim_corners [N,4,4], grad_prim_spans [N,2,4], grad_prim_normals [N,4]   (blocking only; fully
 *                 written, in a fixed summation order) DIRECT gradients of the soft mask w.r.t. the rectangle tables - corner 0, both spans,
 *                 the normal; the caller chains them through whatever built the tables (blocking.py:170-207)
 * ------------------------------------------------------------------------------------------- */
int art_trace_bwd(const float *origins, const float *normals, const float *incident,
                  const float *dist_u, const float *dist_e, int64_t dist_sh, int64_t dist_sr, int64_t dist_sp,
                  const int32_t *target_idx, const float *plane_centers, const float *plane_normals,
                  const float *plane_dims, const float *cyl_centers, const float *cyl_normals,
                  const float *cyl_axes, const float *cyl_radii, const float *cyl_heights,
                  const float *cyl_opening, const float *prim_corners, const float *prim_spans,
                  const float *prim_normals, const int32_t *cand, const int32_t *cand_count, int64_t Cmax, int64_t N,
                  double max_scatter_angle, double ray_magnitude, double extinction, double reflectivity,
                  int64_t H, int64_t R, int64_t P, int64_t facet_points, int64_t T, int64_t Tc, int64_t W, int64_t Hh,
                  int mode, const float *grad_flux, float *grad_origins, float *grad_normals, float *grad_prim_corners,
                  float *grad_prim_spans, float *grad_prim_normals, float *grad_scratch, int64_t grad_scratch_floats,
                  void *stream);

/* Scratch of art_trace_bwd in floats.  Without blocking (Cmax = 0): 0 for fields that fill the chip without cutting a
 * point's samples into chunks; with a 16-byte aligned buffer of at least this size the chunks' partial gradients are
 * written to slabs and added in chunk order - bit-reproducible gradients; with NULL (or less) the samples of a point stay
 * in one work item, which is reproducible too but leaves most of the chip idle on a field of a few heliostats.
 * With blocking (Cmax = the row width passed to art_blocking_filter) the buffer is REQUIRED: every work item leaves its
 * rectangle gradients in a [min(Cmax, 32),12] slab and a last kernel adds the slabs in item order - the rectangle gradients
 * are bit-reproducible as well (no float atomics).  Rows wider than 32: one fp64 row [Cmax - 31, 12] per heliostat more, for
 * the listed candidates of a heliostat with more than 32 (fp64 atomics, rounded to fp32 once - the only float atomics of
 * the library; reproducible to an fp64 rounding).
 * The size covers every receiver configuration art_trace_bwd can meet for these sizes (planar, cylindrical, mixed: the
 * launch geometries differ and the caller allocates before the tables' types matter); art_trace_bwd never rejects a
 * buffer of this size (tests/test_host_logic.py sweeps it). */
int64_t art_trace_bwd_scratch_floats(int64_t H, int64_t R, int64_t P, int64_t facet_points, int64_t Cmax);

/* ... and what art_trace_bwd uses of it for ONE receiver configuration (T planar and Tc cylindrical target areas): the
 * floats of the launch geometry it takes when given at least that much.  Host arithmetic only, no GPU call. */
int64_t art_trace_bwd_scratch_need(int64_t H, int64_t R, int64_t P, int64_t facet_points, int64_t T, int64_t Tc, int64_t Cmax);

/* ---------------------------------------------------------------------------------------------
 * art_blocking_filter - lbvh_filter_blocking_planes (artist/raytracing/blocking.py:832-995, with the tree of
 * :514-749) for ONE batch holding every traced heliostat (heliostat_ray_tracer.py:444-461): which rectangles can
 * block at all, and which of them each heliostat's rays have to be tested against.
 *   geometry arguments as art_trace_fwd (the rays and their target distances are recomputed, not stored);
 *   prim_corners [N,4,4]; owner [H] = index of heliostat h's own rectangle (self hits are ignored, :944-947);
 *   max_scatter_angle  bound on |distortion angle| of the dataset in radians (< 0: measured here);
 *   lbvh_compat != 0   reproduce the reference tree's reachability: its split search (:640-650) leaves most
 *                      leaves of a larger tree unreachable from the root, and an unreachable rectangle is never
 *                      returned; 0 = every rectangle whose box is hit (what the method documents);
 *   Cmax               row width of `cand`: 1 ... N.  Lists of up to 32 rectangles cost what they always did; longer ones
 *                      are filtered in batches of 32 and traced with the rest read from the list (slower, correct);
 *   flags [N]          out: 1 = in the filtered set;
 *   cand [H,Cmax], cand_count [H]   out: filtered rectangles inside heliostat h's ray cone, in ascending order;
 *                      cand_count[h] > Cmax reports a row that was too narrow (the list is then truncated: raise Cmax,
 *                      at most to N) - and so does the device status
 *                      word: ART_ECANDIDATES from art_async_status and from the next art_trace_fwd / art_trace_bwd,
 *                      so a caller need not read the counts back;
 *   workspace          art_blocking_workspace_bytes(H, N) bytes of device memory, 256-byte aligned.
 * ------------------------------------------------------------------------------------------- */
int64_t art_blocking_workspace_bytes(int64_t H, int64_t N);
int art_blocking_filter(const float *origins, const float *normals, const float *incident,
                        const float *dist_u, const float *dist_e, int64_t dist_sh, int64_t dist_sr, int64_t dist_sp,
                        const int32_t *target_idx, const float *plane_centers, const float *plane_normals,
                        const float *plane_dims, const float *cyl_centers, const float *cyl_normals,
                        const float *cyl_axes, const float *cyl_radii, const float *cyl_heights,
                        const float *cyl_opening, double ray_magnitude,
                        int64_t H, int64_t R, int64_t P, int64_t T, int64_t Tc, int64_t W, int64_t Hh,
                        const float *prim_corners, const int32_t *owner, int64_t N, double max_scatter_angle,
                        int lbvh_compat, int64_t Cmax, int32_t *flags, int32_t *cand, int32_t *cand_count,
                        void *workspace, void *stream);

/* ---------------------------------------------------------------------------------------------
 * art_per_target_sum - HeliostatRayTracer.get_bitmaps_per_target
 * (heliostat_ray_tracer.py:563-608): out[t] = sum of bitmaps[h] with target_idx[h] == t.
 *   bitmaps [H,npix], target_idx [H], out [T,npix] (fully written).
 * ------------------------------------------------------------------------------------------- */
int art_per_target_sum(const float *bitmaps, const int32_t *target_idx, int64_t H, int64_t T, int64_t npix,
                       float *out, void *stream);

/* ---------------------------------------------------------------------------------------------
 * art_nurbs_fwd - NURBSSurfaces.calculate_surface_points_and_normals
 * (artist/nurbs/surfaces.py:475-689: find_spans :157-245, basis_functions_and_derivatives
 * :247-417, A3.6 accumulation :578-613, normals :615-672, canting + translation :674-687 via
 * artist/geometry/transforms.py:276-347).
 *   control_points [H,F,nu,nv,3]; eval_points element (h,f,m,c) at
 *   base[h*uv_sh + f*uv_sf + m*2 + c] (strides 0 broadcast one [M,2] grid to every facet);
 *   knots_u [H,F,nu+p+1], knots_v [H,F,nv+q+1];
 *   uniform != 0: span = floor(x*(n_unique-1)) + degree (surfaces.py:198-207), else the
 *   linear search of :209-243;  canting [H,F,2,4] or NULL; translations [H,F,4] (used only
 *   with canting);  outputs points, normals [H,F,M,4].   Degrees 1..7.
 *   orientation [H,4,4] or NULL: when given, the alignment of art_align_fwd (points @ M^T, normals @ M^T,
 *   artist/field/heliostat_group_rigid_body.py:217-222) is applied in the kernel's epilogue - what
 *   SurfaceReconstructor's epoch does right after the evaluation (surface_reconstructor.py:516-546) - with the same
 *   arithmetic, so the outputs equal art_nurbs_fwd + art_align_fwd bit for bit without the [H,P,4] x 2 round trip.
 * ------------------------------------------------------------------------------------------- */
int art_nurbs_fwd(const float *control_points, const float *eval_points, int64_t uv_sh, int64_t uv_sf,
                  const float *knots_u, const float *knots_v, const float *canting, const float *translations,
                  int p, int q, int uniform, int64_t n_unique_u, int64_t n_unique_v,
                  int64_t H, int64_t F, int64_t M, int64_t nu, int64_t nv,
                  const float *orientation, float *points, float *normals, void *stream);

/* art_nurbs_bwd - autograd of art_nurbs_fwd w.r.t. the control points.
 *   grad_points, grad_normals [H,F,M,4] -> grad_control_points [H,F,nu,nv,3] (fully written);
 *   orientation as in the forward call (the gradients are then those of the ALIGNED points / normals; the
 *   orientation itself gets no gradient through this entry point: use art_align_bwd when the kinematics learns). */
int art_nurbs_bwd(const float *control_points, const float *eval_points, int64_t uv_sh, int64_t uv_sf,
                  const float *knots_u, const float *knots_v, const float *canting,
                  int p, int q, int uniform, int64_t n_unique_u, int64_t n_unique_v,
                  int64_t H, int64_t F, int64_t M, int64_t nu, int64_t nv, const float *orientation,
                  const float *grad_points, const float *grad_normals, float *grad_control_points, void *stream);

/* ---------------------------------------------------------------------------------------------
 * art_reflect - geometry.reflect (artist/raytracing/geometry.py:11-41) as a tensor: out[h,p,:] = i[h] - 2 (i[h].n[h,p]) n[h,p]
 * over all four components, reference operation order.  The trace kernels reflect in registers; this entry point serves
 * `heliostat_group.preferred_reflection_directions`, which HeliostatRayTracer.trace_rays publishes on every call
 * (artist/raytracing/heliostat_ray_tracer.py:285-290).
 *   incident [H,4], normals [H,P,4], out [H,P,4]
 * ------------------------------------------------------------------------------------------- */
int art_reflect(const float *incident, const float *normals, int64_t H, int64_t P, float *out, void *stream);

/* ---------------------------------------------------------------------------------------------
 * art_adam_step - the optimiser step of the reconstruction epochs: torch.optim.Adam's update rule
 * (torch/optim/adam.py, _single_tensor_adam; no amsgrad) on ONE contiguous fp32 tensor of n elements, as the reference
 * steps the control points (artist/optim/surface_reconstructor.py:452-455 creates the optimiser, :779 steps it) and the
 * kinematics deviations (artist/optim/kinematics_reconstructor.py).  `step` = number of this step (1 for the first): the
 * bias corrections are computed on the host from it, in double.  param, exp_avg, exp_avg_sq are updated in place.
 *   lock_nu, lock_nv > 0: the tensor is a batch of [nu,nv,3] control nets and the first two components of the gradient of
 *   every net's outer-edge control points count as zero - SurfaceReconstructor.lock_control_points_on_outer_edges
 *   (surface_reconstructor.py:1155-1224: the outline is kept, z stays free) without a pass over the gradient; 0, 0: plain Adam.
 * ------------------------------------------------------------------------------------------- */
int art_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, double lr, double beta1,
                  double beta2, double eps, double weight_decay, int64_t step, int maximize, int64_t lock_nu, int64_t lock_nv,
                  void *stream);

/* ---------------------------------------------------------------------------------------------
 * art_align_fwd - the alignment apply of HeliostatGroupRigidBody.align_surfaces_with_incident_ray_directions /
 * align_surfaces_with_motor_positions (artist/field/heliostat_group_rigid_body.py:217-222, 265-270):
 *   out_points = points @ orientation^T, out_normals = normals @ orientation^T, one pass over both.
 *   points, normals [H,P,4]; orientation [H,4,4]; outputs [H,P,4].
 * art_align_bwd - its autograd: grad_points = g @ orientation (same for normals) and, when
 *   grad_orientation != NULL, grad_orientation[h] = sum_p (g_points^T x_points + g_normals^T x_normals)
 *   ([H,4,4], fully written) - the path by which kinematic parameters receive gradients.
 * ------------------------------------------------------------------------------------------- */
int art_align_fwd(const float *points, const float *normals, const float *orientation, int64_t H, int64_t P,
                  float *out_points, float *out_normals, void *stream);
int art_align_bwd(const float *points, const float *normals, const float *orientation,
                  const float *grad_out_points, const float *grad_out_normals, int64_t H, int64_t P,
                  float *grad_points, float *grad_normals, float *grad_orientation, void *stream);

/* ---------------------------------------------------------------------------------------------
 * art_flux_crop_fwd - crop_flux_distributions_around_center (artist/flux/bitmap.py:121-246): centre of mass of each
 * bitmap (:165-182) -> affine grid scaled by crop size / target size and centred on it (:218-237) -> bilinear
 * grid_sample, align_corners=True, zeros padding (:239-246).
 *   flux [B,Hh,W]; target_dims [B,2] = (width, height) in metres of each bitmap's target area - planar
 *   dimensions, or radius x opening angle and height for cylinders (the gather of :183-216 is host logic);
 *   out [B,Hh,W]; centers [B,3] out = (x centre, y centre, sum + 1e-8), needed by the backward.
 * art_flux_crop_bwd - its autograd w.r.t. flux (what torch derives for :165-246), both paths: the sampled values -
 *   grid_sample's input gradient, here a GATHER over the output pixels that sampled an input pixel (the map is
 *   axis-aligned and monotone), so bit-reproducible where torch's CUDA backward scatters with float atomics
 *   (tests/conftest.py:93-97) - and the path through the centre of mass (grid_sample's grid gradient, reduced to
 *   the two centre coordinates and chained through x centre = sum(x flux) / (sum flux + 1e-8), :165-182).
 *   centers [B,3] as written by the forward call; grad_out [B,Hh,W]; grad_flux [B,Hh,W] out (fully written);
 *   workspace: at least 2 * B floats of device memory (per bitmap the gradients of the two centre coordinates).
 * ------------------------------------------------------------------------------------------- */
int art_flux_crop_fwd(const float *flux, const float *target_dims, int64_t B, int64_t Hh, int64_t W,
                      double crop_width, double crop_height, float *out, float *centers, void *stream);
int art_flux_crop_bwd(const float *flux, const float *target_dims, const float *centers, int64_t B, int64_t Hh,
                      int64_t W, double crop_width, double crop_height, const float *grad_out, float *grad_flux,
                      float *workspace, void *stream);

/* ---------------------------------------------------------------------------------------------
 * art_flux_loss - PixelLoss (kind 0, artist/optim/loss.py:251-318: sum (p-g)^2 / sum g) or KLDivergenceLoss
 * (kind 1, :321-410: L1-normalise both, KLDivLoss(log_target) on log(. + 1e-12)), reduced over the bitmap.
 *   prediction, ground_truth [B,npix]; loss [B] out (may be NULL in a backward-only call);
 *   grad_loss [B] + grad_prediction [B,npix] out: both NULL for forward only.
 * ------------------------------------------------------------------------------------------- */
int art_flux_loss(const float *prediction, const float *ground_truth, int64_t B, int64_t npix, int kind,
                  float *loss, const float *grad_loss, float *grad_prediction, void *stream);

/* ---------------------------------------------------------------------------------------------
 * art_flux_crop_pixel_loss_fwd / _bwd - the two calls every surface-reconstruction epoch makes on the tracer's
 * bitmaps, crop_flux_distributions_around_center (artist/flux/bitmap.py:121-246) followed by PixelLoss
 * (artist/optim/loss.py:251-318; surface_reconstructor.py:575-590, 664-676), fused: the cropped bitmaps never reach
 * HBM, and the backward call is ONE pass over the residual the forward call kept.
 * The same numbers as art_flux_crop_fwd + art_flux_loss(kind 0) and their backward calls up to the rounding of the sums
 * (a bitmap's rows are always summed in four parts, whatever the batch size, and the parts added in order: < 1e-6 relative)
 * - and the same BITS for every batch size and from run to run: small batches give a bitmap two or four workgroups, large
 * ones a single workgroup, the arithmetic is the same (a rank's share of a field and the whole field agree bit for bit).
 *   flux [B,Hh,W], target_dims [B,2], ground_truth [B,Hh,W] (the measured, already cropped flux)
 *   loss [B] out; centers4 [B,4] out (centre of mass x, y, bitmap sum + 1e-8, sum of the measured flux)
 *   residual [B,Hh,W] out, center_grad_unit [B,2] out: crop - ground_truth and the gradient of the two centre coordinates
 *   per unit of 2 grad_loss / sum(ground_truth) - what the backward call needs; both NULL for a forward-only call.
 *   moments [B,4,3] fp64 or NULL: the bitmaps' centre-of-mass sums as art_trace_fwd leaves them (see there); with them the
 *   call does not read the bitmaps a second time for the centre (same sums, same bits, as when it forms them itself).
 *   _bwd: grad_loss [B] -> grad_flux [B,Hh,W] (fully written).
 * ------------------------------------------------------------------------------------------- */
int art_flux_crop_pixel_loss_fwd(const float *flux, const float *target_dims, const float *ground_truth, int64_t B,
                                 int64_t Hh, int64_t W, double crop_width, double crop_height, float *loss,
                                 float *centers4, float *residual, float *center_grad_unit, const double *moments,
                                 void *stream);
int art_flux_crop_pixel_loss_bwd(const float *target_dims, const float *centers4, const float *grad_loss,
                                 int64_t grad_loss_stride, const float *residual, const float *center_grad_unit, int64_t B,
                                 int64_t Hh, int64_t W, double crop_width, double crop_height, float *grad_flux, void *stream);
/* (grad_loss_stride: 1 = grad_loss [B]; 0 = one value for every bitmap - the gradient of `loss.sum()` as autograd hands it
 *  over, an expanded scalar: no copy to [B] first) */

/* ---------------------------------------------------------------------------------------------
 * art_flux_crop_kl_loss_fwd / _bwd - crop_flux_distributions_around_center (artist/flux/bitmap.py:121-246) followed by
 * KLDivergenceLoss (artist/optim/loss.py:321-410: L1-normalise both bitmaps, KLDivLoss(log_target) on log(. + 1e-12),
 * summed over the bitmap) as ONE pass per direction, like the PixelLoss pair above; results equal art_flux_crop_fwd +
 * art_flux_loss(kind 1) and their backward calls (same arithmetic, same order).
 *   flux [B,Hh,W], target_dims [B,2], ground_truth [B,Hh,W] (already cropped); loss [B] out;
 *   record8 [B,8] out (centre of mass x, y, bitmap sum + 1e-8, |crop|_1, |truth|_1, the normalisation's dot product,
 *   2 spare): pass it back to the backward call.  grad_loss [B]; grad_flux [B,Hh,W] out; workspace B*Hh*W + 5*B floats.
 * ------------------------------------------------------------------------------------------- */
int art_flux_crop_kl_loss_fwd(const float *flux, const float *target_dims, const float *ground_truth, int64_t B,
                              int64_t Hh, int64_t W, double crop_width, double crop_height, float *loss,
                              float *record8, void *stream);
int art_flux_crop_kl_loss_bwd(const float *flux, const float *target_dims, const float *ground_truth,
                              const float *record8, const float *grad_loss, int64_t B, int64_t Hh, int64_t W,
                              double crop_width, double crop_height, float *grad_flux, float *workspace,
                              void *stream);

/* ---------------------------------------------------------------------------------------------
 * art_flux_center_of_mass - get_center_of_mass (artist/flux/bitmap.py:12-71): bitmap (pixel) coordinates of each
 * bitmap's centre of mass, e = sum_ij j f_ij / (sum f + 1e-8), u likewise with i; (0, 0) for an empty bitmap.  It is
 * what FocalSpotLoss (artist/optim/loss.py:124-250) and the kinematics reconstructor's validation
 * (artist/optim/kinematics_reconstructor.py:120) compute from the tracer's bitmaps.
 *   flux [B,Hh,W]; com [B,3] out = (e pixel, u pixel, sum + 1e-8).
 * art_flux_center_of_mass_bwd - its autograd: grad_flux[b,i,j] = (g_e (j - e) + g_u (i - u)) / (sum + 1e-8).
 *   com [B,3] as written by the forward call; grad_com [B,2]; grad_flux [B,Hh,W] out (fully written).
 * ------------------------------------------------------------------------------------------- */
int art_flux_center_of_mass(const float *flux, int64_t B, int64_t Hh, int64_t W, float *com, void *stream);
int art_flux_center_of_mass_bwd(const float *com, const float *grad_com, int64_t B, int64_t Hh, int64_t W,
                                float *grad_flux, void *stream);

/* ---------------------------------------------------------------------------------------------
 * art_rigid_body_fwd - RigidBody kinematics of H heliostats in one launch
 * (artist/field/kinematics_rigid_body.py:194-634 with artist/field/actuators_ideal.py:66-111 and
 * actuators_linear.py:79-370).
 *   mode 0 = motor_positions_to_orientations (:510-538): orientations from the given motor_positions [H,2];
 *   mode 1 = incident_ray_directions_to_orientations (:540-634): the fixed-point iteration from motor positions
 *            0, at most max_iter evaluations, stopped when EVERY heliostat's loss moved by <= min_eps;
 *            motor_positions [H,2] receives the final motor positions (RigidBody.active_motor_positions).
 *   positions [H,4]; rot_dev [H,4]; trans_dev [H,9]; act_nonopt [H,act_rows,2] with act_rows 4 (ideal) or
 *   7 (linear: + increment, offset, pivot radius); act_opt [H,2,2] (linear; NULL for ideal);
 *   offsets [4,4] = RigidBody.initial_orientation_offsets; incident, aim [H,4] (mode 1; may be NULL in mode 0).
 *   orientations [H,4,4] out (already multiplied by offsets); scratch [H + max_iter + 1] 4-byte words;
 *   evaluations [1] int32 out =
 *   number of forward-kinematics evaluations made (what the backward replays).
 * art_rigid_body_bwd - its autograd w.r.t. rot_dev, trans_dev and act_opt by forward-mode differentiation of the
 *   same chain (the gradients torch.autograd gives the reference): grad_orientations [H,4,4] in;
 *   grad_rot_dev [H,4], grad_trans_dev [H,9], grad_act_opt [H,2,2] (NULL for ideal actuators) out, fully written.
 *   motor_positions is the forward's input in mode 0 and ignored in mode 1; evaluations as written by the forward.
 *   grad_motor_positions [H,2] out (mode 0 only, may be NULL): the gradient the aim-point optimiser needs
 *   (artist/optim/aim_point_optimizer.py:384-405 learns motor positions through align_surfaces_with_motor_positions).
 * ------------------------------------------------------------------------------------------- */
int art_rigid_body_fwd(int mode, const float *positions, const float *rot_dev, const float *trans_dev,
                       const float *act_nonopt, int64_t act_rows, const float *act_opt, const float *offsets,
                       const float *incident, const float *aim, int64_t H, int max_iter, double min_eps,
                       float *motor_positions, float *orientations, float *scratch, int32_t *evaluations,
                       void *stream);
int art_rigid_body_bwd(int mode, const float *positions, const float *rot_dev, const float *trans_dev,
                       const float *act_nonopt, int64_t act_rows, const float *act_opt, const float *offsets,
                       const float *incident, const float *aim, int64_t H, const float *motor_positions,
                       const int32_t *evaluations, const float *grad_orientations, float *grad_rot_dev,
                       float *grad_trans_dev, float *grad_act_opt, float *grad_motor_positions, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ARTIST_HIP_H */
