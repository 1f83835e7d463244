"""``torch.autograd.Function`` wrappers around the C ABI of ``libartist_hip.so``.

PyTorch is plumbing here (device memory, streams, autograd tape); every number is produced by
the hand-written gfx950 kernels in ``artist_amd/csrc``.  No CPU path exists: non-CUDA inputs
raise.
"""
from __future__ import annotations

import collections
import contextlib
import math

import weakref

import torch

from . import _lib

__all__ = ["trace_rays", "nurbs_surface_points_and_normals", "per_target_sum", "align_surfaces", "TraceRays",
           "NurbsEval", "AlignSurfaces", "check_async_errors", "record_launch_events"]


def _stream(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


# Measurement hook (bench.py's roofline leg, tools/): when switched on, the two trace calls are bracketed by a pair of HIP events
# on the stream they are launched on, so a caller can read the launch durations of a region it times WITHOUT changing that region
# (no synchronisation; two event records per call).  Off by default: None.
_LAUNCH_EVENTS = None


def record_launch_events(on: bool = True):
    """Start (``True``: returns the dict that fills up, ``{"art_trace_fwd": [(start, end), ...], "art_trace_bwd": [...]}``
    of ``torch.cuda.Event`` pairs) or stop (``False``: returns the dict collected so far) recording.  Read the pairs with
    ``start.elapsed_time(end)`` after a synchronisation."""
    global _LAUNCH_EVENTS
    if on:
        _LAUNCH_EVENTS = {}
        return _LAUNCH_EVENTS
    out, _LAUNCH_EVENTS = _LAUNCH_EVENTS, None
    return out


@contextlib.contextmanager
def _launch(name: str, device: torch.device):
    rec = _LAUNCH_EVENTS
    with torch.cuda.device(device):
        if rec is None:
            yield
            return
        stream = torch.cuda.current_stream(device)
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record(stream)
        yield
        end.record(stream)
        rec.setdefault(name, []).append((start, end))


def _require_cuda(*tensors: torch.Tensor) -> torch.device:
    dev = tensors[0].device
    if dev.type != "cuda":
        raise _lib.ArtistHipError(
            f"artist_amd ops run on the GPU only (got a tensor on {dev}); there is no CPU fallback")
    for t in tensors:
        if t is not None and t.device != dev:
            raise ValueError(f"all tensors must be on {dev}, found one on {t.device}")
    return dev


def _f32c(t: torch.Tensor) -> torch.Tensor:
    """float32 + contiguous (no copy when already so)."""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


# Small per-call conversions that an optimisation loop repeats with the SAME tensors every epoch - the int32 copy of the
# target indices, the [H,F,K] materialisation of an expanded knot vector - are kept, keyed by the source's address, shape,
# strides and version counter (an in-place write makes a new entry).  Each is a 5 us kernel that a 1 ms epoch of a rank's
# share of a field would pay several times over.  An entry holds the source tensor, so its address cannot be handed to another
# tensor while the entry lives; sixteen entries of a few KB, the oldest goes first.
_CONVERTED: "collections.OrderedDict" = collections.OrderedDict()


def _converted(t: torch.Tensor, what: str, make):
    try:
        version = t._version
    except (RuntimeError, AttributeError):          # inference-mode tensors have no version counter: no caching
        return make(t)
    key = (what, t.data_ptr(), tuple(t.shape), tuple(t.stride()), t.dtype, version, str(t.device))
    hit = _CONVERTED.get(key)
    if hit is not None:
        _CONVERTED.move_to_end(key)
        return hit[1]
    out = make(t)
    if out is not t and t.numel() <= (1 << 20):                                  # (nothing to keep when no copy was made)
        _CONVERTED[key] = (t, out)
        while len(_CONVERTED) > 16:
            _CONVERTED.popitem(last=False)
    return out


def _int32c(t: torch.Tensor) -> torch.Tensor:
    return t if (t.dtype == torch.int32 and t.is_contiguous()) else _converted(t, "i32", lambda x: x.to(torch.int32).contiguous())


def _dist_views(dist_u: torch.Tensor, dist_e: torch.Tensor, shape):
    """Validate the two [H,R,P] distortion views and return (u, e, element strides).

    ``Sun.get_distortions`` (artist/scene/sun.py:227-234) returns stride-(2RP,2P,2) views of one
    interleaved buffer; they are passed through as-is (the kernel does one 8-byte load per ray)."""
    if dist_u.shape != dist_e.shape:
        # same condition / message as transforms.rotate_distortions (artist/geometry/transforms.py:47-50)
        raise ValueError("The two tensors containing angles for the east and up rotation must have the same shape.")
    if tuple(dist_u.shape) != tuple(shape):
        raise ValueError(f"distortions must have shape {tuple(shape)}, got {tuple(dist_u.shape)}")
    if dist_u.dtype != torch.float32 or dist_e.dtype != torch.float32:
        dist_u, dist_e = dist_u.float(), dist_e.float()
    if dist_u.stride() != dist_e.stride():
        dist_u, dist_e = dist_u.contiguous(), dist_e.contiguous()
    return dist_u, dist_e, dist_u.stride()


def _cyl_tables(cyl, dev):
    """(centers[Tc,4], normals[Tc,4], axes[Tc,4], radii[Tc], heights[Tc], opening[Tc]) -> fp32 contiguous
    tensors on ``dev`` + their data pointers (NULL x 6 when there are no cylindrical target areas)."""
    if cyl is None or cyl[0].shape[0] == 0:
        return (), (None,) * 6, 0
    if len(cyl) != 6:
        raise ValueError("cylindrical target areas: (centers, normals, axes, radii, heights, opening_angles)")
    tabs = tuple(_f32c(t.to(dev)) for t in cyl)
    Tc = tabs[0].shape[0]
    for t, shape in zip(tabs, ((Tc, 4), (Tc, 4), (Tc, 4), (Tc,), (Tc,), (Tc,))):
        if tuple(t.shape) != shape:
            raise ValueError(f"cylindrical target table has shape {tuple(t.shape)}, expected {shape}")
    return tabs, tuple(t.data_ptr() for t in tabs), Tc


def _planar_ptrs(centers, plane_normals, dims):
    if centers.shape[0] == 0:
        return (None, None, None)
    return (centers.data_ptr(), plane_normals.data_ptr(), dims.data_ptr())


#: Row width of a heliostat's list of candidate blocking rectangles (``Cmax`` of art_blocking_filter) - workspace, not a limit of
#: the kernels: they keep the first 32 rectangles of a list in LDS and read any others from the list itself (slower, never
#: refused).  A list that does not fit its ROW is reported (ART_ECANDIDATES: workspace exhausted - raise this number; ``None`` =
#: the number of rectangles, which no list can exceed; 4 B per heliostat and entry, 96 B more in the backward scratch).
BLOCKING_CANDIDATES = 256
_LAST_BLOCKING = None

# Pixel accumulators of art_trace_fwd ([n_maps,Hh,W] uint64, zero on entry and zero again afterwards): one buffer per
# (device, stream), grown on demand, so that calls on different streams never share one.
_ACCUM: dict = {}


def _accumulators(dev: torch.device, n: int) -> torch.Tensor:
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), _stream(dev))
    buf = _ACCUM.get(key)
    if buf is None or buf.numel() < n:
        buf = torch.zeros((n,), dtype=torch.int64, device=dev)
        _ACCUM[key] = buf
    return buf


def reflect_directions(incident: torch.Tensor, normals: torch.Tensor) -> torch.Tensor:
    """``incident[h] - 2 (incident[h] . normals[h,p]) normals[h,p]`` for ``incident [H,4]``, ``normals [H,P,4]``
    (``art_reflect``; artist/raytracing/geometry.py:11-41).  Not differentiable."""
    dev = _require_cuda(incident, normals)
    incident, normals = _f32c(incident.detach()), _f32c(normals.detach())
    H, P = int(normals.shape[0]), int(normals.shape[1])
    if incident.shape != (H, 4) or normals.shape != (H, P, 4):
        raise ValueError("incident must be [H,4] and normals [H,P,4]")
    out = torch.empty_like(normals)
    with torch.cuda.device(dev):
        rc = _lib.lib().art_reflect(incident.data_ptr(), normals.data_ptr(), H, P, out.data_ptr(), _stream(dev))
    _lib.check(rc, "art_reflect")
    return out


def check_async_errors(device=None, clear: bool = True) -> None:
    """Synchronise ``device`` and raise if a kernel met a target index outside the target tables or the blocking filter
    found more candidate rectangles for a heliostat than the kernels hold (``art_async_status``): the entry points are
    asynchronous, so the device-side checks report here - or at the next trace call, which refuses to start while the
    status is set.  The status is ONE sticky word per GPU, shared by every stream and tracer of the process: whoever
    calls this clears it for all of them (``clear=False`` only looks).  A heliostat that was skipped or traced with an
    incomplete rectangle list is also marked in the results themselves: zero bitmap and factors for a bad target index,
    NaN bitmap and factors for a candidate overflow."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    with torch.cuda.device(dev):
        # the status word is one per GPU (include/artist_hip.h): a kernel on ANY stream of this process may have set it,
        # so wait for the whole device, not only for the current stream
        torch.cuda.synchronize(dev)
        rc = _lib.lib().art_async_status(_stream(dev), 1 if clear else 0)
    if rc == -2:
        _ACCUM.clear()              # a skipped heliostat leaves nothing behind, but take no chances with the invariant
        raise IndexError("target_area_indices out of range (found by the kernels: art_async_status)")
    if rc == -5:
        raise _lib.ArtistHipError(f"a heliostat has more blocking rectangles inside its ray cone than its candidate row holds "
                                  f"(artist_amd.ops.BLOCKING_CANDIDATES = {BLOCKING_CANDIDATES}: raise it, or None for no bound; "
                                  "found by art_blocking_filter: art_async_status); its blocking would be incomplete")
    if rc == -6:
        _ACCUM.clear()              # (a launch that ended abnormally may have left pixel accumulators behind)
    _lib.check(rc, "art_async_status")


# Centre-of-mass sums that came with a traced bitmap tensor: data pointer -> (weak reference to the tensor object, sums).  An
# entry serves exactly the tensor OBJECT the trace returned, at the version it returned it with (an in-place edit, a copy or a
# slice of the bitmaps is another tensor or another version: the crop then forms the sums itself).
_MOMENTS: dict = {}


def bitmap_moments(flux: torch.Tensor):
    """The sums ``art_trace_fwd`` left for ``flux`` (``[n_maps,4,3]`` float64) if ``flux`` is an untouched output of
    :func:`trace_rays` / ``HeliostatRayTracer.trace_rays``, else None."""
    entry = _MOMENTS.get(flux.data_ptr())
    if entry is None or entry[0]() is not flux or flux._version != 0:
        return None
    return entry[1]


class TraceRays(torch.autograd.Function):
    """reflect -> scatter -> plane / cylinder intersection -> (blocking mask) -> bilinear splat (+ factors), fused.

    Replaces the body of ``HeliostatRayTracer.trace_rays``
    (artist/raytracing/heliostat_ray_tracer.py:285-506).  Target index ``t < T`` is planar area ``t``; ``t >= T``
    is cylindrical area ``t - T`` of the ``cyl`` tables.  With ``prim_corners/spans/normals`` (the rectangles of
    ``create_blocking_primitives_rectangles_by_index``) blocking is on: ``art_blocking_filter`` picks the
    rectangles (``lbvh_filter_blocking_planes``) and the kernels evaluate ``soft_ray_blocking_mask`` per ray.
    Differentiable w.r.t. ``origins``, ``normals`` and the three rectangle tables exactly like the eager chain
    (indices, masks and the filter are constants).  Returns ``(flux, factors, filter_flags)``.
    """

    @staticmethod
    def forward(ctx, origins, normals, incident, dist_u, dist_e, target_idx, centers, plane_normals, dims,
                ray_magnitude, extinction, reflectivity, width, height, per_target, cyl=None,
                prim_corners=None, prim_spans=None, prim_normals=None, owner=None, max_scatter_angle=-1.0,
                lbvh_compat=True, points_per_facet=0):
        dev = _require_cuda(origins, normals, incident, dist_u, dist_e, target_idx, centers, plane_normals, dims)
        origins, normals, incident = _f32c(origins), _f32c(normals), _f32c(incident)
        H, P = origins.shape[0], origins.shape[1]
        if origins.shape != (H, P, 4) or normals.shape != (H, P, 4) or incident.shape != (H, 4):
            raise ValueError("origins/normals must be [H,P,4] and incident [H,4]")
        # a performance hint, never a semantic one: the kernels cut a heliostat's points into blocks that share an LDS
        # window, and blocks that do not straddle two facets (two separate images) fit their windows better
        points_per_facet = int(points_per_facet or 0)
        if points_per_facet < 0 or (points_per_facet and P % points_per_facet):
            raise ValueError("points_per_facet must divide the number of surface points per heliostat")
        R = dist_u.shape[1] if dist_u.dim() == 3 else -1
        dist_u, dist_e, (sh, sr, sp) = _dist_views(dist_u, dist_e, (H, R, P))
        target_idx = _int32c(target_idx)
        centers, plane_normals, dims = _f32c(centers), _f32c(plane_normals), _f32c(dims)
        T = centers.shape[0]
        cyl_tabs, cyl_ptrs, Tc = _cyl_tables(cyl, dev)
        n_maps = T + Tc if per_target else H
        geometry = (origins.data_ptr(), normals.data_ptr(), incident.data_ptr(), dist_u.data_ptr(), dist_e.data_ptr(),
                    sh, sr, sp, target_idx.data_ptr(), *_planar_ptrs(centers, plane_normals, dims), *cyl_ptrs)

        blocking = prim_corners is not None
        block_tabs, block_ptrs, Cmax, N = (), (None,) * 5, 0, 0
        flags = torch.empty((0,), dtype=torch.int32, device=dev)
        if blocking and H > 0:
            _require_cuda(prim_corners, prim_spans, prim_normals, owner)
            if max_scatter_angle < 0:    # the kernels cull per surface point with this bound: measure it once
                max_scatter_angle = float(torch.maximum(dist_u.abs().max(), dist_e.abs().max()))
            prim_corners, prim_spans, prim_normals = _f32c(prim_corners), _f32c(prim_spans), _f32c(prim_normals)
            N = prim_corners.shape[0]
            if prim_corners.shape != (N, 4, 4) or prim_spans.shape != (N, 2, 4) or prim_normals.shape != (N, 4):
                raise ValueError("blocking primitives must be corners [N,4,4], spans [N,2,4], normals [N,4]")
            owner = owner.to(torch.int32).contiguous()
            if owner.shape != (H,):
                raise ValueError("owner must hold one primitive index per traced heliostat")
            Cmax = N if BLOCKING_CANDIDATES is None else max(1, min(N, int(BLOCKING_CANDIDATES)))
            flags = torch.empty((N,), dtype=torch.int32, device=dev)
            cand = torch.empty((H, Cmax), dtype=torch.int32, device=dev)
            cand_count = torch.empty((H,), dtype=torch.int32, device=dev)
            workspace = torch.empty((int(_lib.lib().art_blocking_workspace_bytes(H, N)),), dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                rc = _lib.lib().art_blocking_filter(
                    *geometry, float(ray_magnitude), H, R, P, T, Tc, width, height, prim_corners.data_ptr(),
                    owner.data_ptr(), N, float(max_scatter_angle), 1 if lbvh_compat else 0, Cmax, flags.data_ptr(),
                    cand.data_ptr(), cand_count.data_ptr(), workspace.data_ptr(), _stream(dev))
            _lib.check(rc, "art_blocking_filter")
            # (more than Cmax rectangles inside a heliostat's ray cone: the device reports it - ART_ECANDIDATES from
            #  check_async_errors() or from the next trace call - instead of a host read of the counts in every call)
            block_tabs = (prim_corners, prim_spans, prim_normals, cand, cand_count)
            global _LAST_BLOCKING
            _LAST_BLOCKING = (cand, cand_count)            # (diagnostics and tests: the last call's candidate lists, still on the device)
            block_ptrs = tuple(t.data_ptr() for t in block_tabs)

        flux = torch.empty((n_maps, height, width), dtype=torch.float32, device=dev)
        factors = torch.empty((3, H), dtype=torch.float32, device=dev)
        accum = _accumulators(dev, n_maps * height * width)
        # the bitmaps' centre-of-mass sums, left behind by the pass that converts the accumulators (include/artist_hip.h): the
        # crop + loss op that follows in a reconstruction epoch starts from them (artist_amd.flux: bitmap_moments)
        moments = None
        if height >= 4 and (height * width) % 2 == 0 and 0 < n_maps <= 65535:
            moments = torch.empty((n_maps, 4, 3), dtype=torch.float64, device=dev)
        with _launch("art_trace_fwd", dev):
            rc = _lib.lib().art_trace_fwd(
                *geometry, *block_ptrs, Cmax, float(max_scatter_angle), float(ray_magnitude), float(extinction),
                float(reflectivity),
                H, R, P, points_per_facet, T, Tc, width, height, 1 if per_target else 0, flux.data_ptr(), factors.data_ptr(),
                accum.data_ptr(), None if moments is None else moments.data_ptr(), _stream(dev))
        if rc != 0:
            _ACCUM.clear()
        if rc == -2:
            raise IndexError("target_area_indices out of range (found by the kernels of an earlier call; "
                             "artist_amd.ops.check_async_errors() clears the status)")
        if rc == -5:
            raise _lib.ArtistHipError(
                f"a heliostat has more blocking rectangles inside its ray cone than its candidate row holds "
                f"(artist_amd.ops.BLOCKING_CANDIDATES = {BLOCKING_CANDIDATES}: raise it, or set it to None; found by "
                "art_blocking_filter in an earlier call, whose bitmaps and factors for that heliostat are NaN; "
                "artist_amd.ops.check_async_errors() clears the status)")
        _lib.check(rc, "art_trace_fwd")
        ctx.save_for_backward(origins, normals, incident, dist_u, dist_e, target_idx, centers, plane_normals, dims,
                              *cyl_tabs, *block_tabs)
        ctx.n_cyl = len(cyl_tabs)
        ctx.scalars = (float(ray_magnitude), float(extinction), float(reflectivity), width, height, bool(per_target),
                       Cmax, N, float(max_scatter_angle), points_per_facet)
        ctx.mark_non_differentiable(factors, flags)
        ctx.set_materialize_grads(False)            # no zero tensors (one fill kernel each) for outputs nobody differentiates
        if moments is not None:
            _MOMENTS[flux.data_ptr()] = (weakref.ref(flux), moments)
            if len(_MOMENTS) > 64:
                for key in [k for k, v in _MOMENTS.items() if v[0]() is None]:
                    del _MOMENTS[key]
        return flux, factors, flags

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_flux, _grad_factors, _grad_flags):
        origins, normals, incident, dist_u, dist_e, target_idx, centers, plane_normals, dims = ctx.saved_tensors[:9]
        cyl_tabs = ctx.saved_tensors[9:9 + ctx.n_cyl]
        block_tabs = ctx.saved_tensors[9 + ctx.n_cyl:]
        cyl_ptrs = tuple(t.data_ptr() for t in cyl_tabs) if cyl_tabs else (None,) * 6
        block_ptrs = tuple(t.data_ptr() for t in block_tabs) if block_tabs else (None,) * 5
        Tc = cyl_tabs[0].shape[0] if cyl_tabs else 0
        mag, ext, refl, width, height, per_target, Cmax, N, max_scatter, points_per_facet = ctx.scalars
        dev = origins.device
        if grad_flux is None:                       # the bitmaps were not used downstream
            return (None,) * 23
        H, P = origins.shape[0], origins.shape[1]
        R = dist_u.shape[1]
        sh, sr, sp = dist_u.stride()
        grad_flux = _f32c(grad_flux)
        g_o = torch.empty_like(origins)
        g_n = torch.empty_like(normals)
        g_pc = g_ps = g_pn = None
        if block_tabs:
            g_pc, g_ps, g_pn = (torch.empty_like(t) for t in block_tabs[:3])
        n_scratch = int(_lib.lib().art_trace_bwd_scratch_floats(H, R, P, points_per_facet, Cmax if block_tabs else 0))
        scratch = torch.empty((n_scratch,), dtype=torch.float32, device=dev) if n_scratch else None
        with _launch("art_trace_bwd", dev):
            rc = _lib.lib().art_trace_bwd(
                origins.data_ptr(), normals.data_ptr(), incident.data_ptr(), dist_u.data_ptr(), dist_e.data_ptr(),
                sh, sr, sp, target_idx.data_ptr(), *_planar_ptrs(centers, plane_normals, dims), *cyl_ptrs,
                *block_ptrs, Cmax, N, max_scatter, mag, ext, refl, H, R, P, points_per_facet, centers.shape[0], Tc, width, height,
                1 if per_target else 0, grad_flux.data_ptr(), g_o.data_ptr(), g_n.data_ptr(),
                *(t.data_ptr() if t is not None else None for t in (g_pc, g_ps, g_pn)),
                None if scratch is None else scratch.data_ptr(), n_scratch, _stream(dev))
        if rc == -2:
            raise IndexError("target_area_indices out of range (found by the kernels of an earlier call; "
                             "artist_amd.ops.check_async_errors() clears the status)")
        _lib.check(rc, "art_trace_bwd")
        return (g_o, g_n) + (None,) * 14 + (g_pc, g_ps, g_pn, None, None, None, None)


def trace_rays(origins, normals, incident, dist_u, dist_e, target_idx, centers, plane_normals, dims,
               ray_magnitude=1.0, extinction=0.0, reflectivity=0.935, resolution=(256, 256), per_target=False,
               cyl=None, blocking=None, points_per_facet=0):
    """Functional form.  Returns ``(flux, factors)`` with ``flux`` ``[H,Hh,W]`` (or ``[T+Tc,Hh,W]`` when
    ``per_target``) and ``factors`` ``[3,H]`` = intercept, on-target, blocking fractions.  ``cyl`` = the six
    ``TowerTargetAreasCylindrical`` tensors (centers, normals, axes, radii, heights, opening_angles) or None.
    ``blocking`` = None or a dict with ``corners [N,4,4]``, ``spans [N,2,4]``, ``normals [N,4]``, ``owner [H]`` and
    optionally ``max_scatter_angle`` / ``lbvh_compat``; the filtered set is then returned as a third value.
    ``points_per_facet`` (optional, performance only): the surface points of a heliostat are F runs of that many points,
    one run per facet (ARTIST's ``[H, F * M, 4]`` layout) - results do not depend on it."""
    if blocking is None:
        flux, factors, _ = TraceRays.apply(origins, normals, incident, dist_u, dist_e, target_idx, centers,
                                           plane_normals, dims, ray_magnitude, extinction, reflectivity,
                                           int(resolution[0]), int(resolution[1]), bool(per_target), cyl,
                                           None, None, None, None, -1.0, True, points_per_facet)
        return flux, factors
    return TraceRays.apply(origins, normals, incident, dist_u, dist_e, target_idx, centers, plane_normals, dims,
                           ray_magnitude, extinction, reflectivity, int(resolution[0]), int(resolution[1]),
                           bool(per_target), cyl, blocking["corners"], blocking["spans"], blocking["normals"],
                           blocking["owner"], float(blocking.get("max_scatter_angle", -1.0)),
                           bool(blocking.get("lbvh_compat", True)), points_per_facet)


def per_target_sum(bitmaps: torch.Tensor, target_idx: torch.Tensor, n_targets: int) -> torch.Tensor:
    """``get_bitmaps_per_target`` (heliostat_ray_tracer.py:563-608) as one kernel; differentiable
    through a gather in torch (the backward of a masked sum is an index_select)."""
    _require_cuda(bitmaps, target_idx)
    return _PerTargetSum.apply(bitmaps, _int32c(target_idx), int(n_targets))


class _PerTargetSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, bitmaps, target_idx, n_targets):
        dev = bitmaps.device
        b = _f32c(bitmaps)
        H = b.shape[0]
        npix = int(math.prod(b.shape[1:]))
        out = torch.empty((n_targets,) + tuple(b.shape[1:]), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.lib().art_per_target_sum(b.data_ptr(), target_idx.data_ptr(), H, n_targets, npix,
                                               out.data_ptr(), _stream(dev))
        _lib.check(rc, "art_per_target_sum")
        ctx.save_for_backward(target_idx)
        ctx.n_targets = n_targets
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (target_idx,) = ctx.saved_tensors
        idx = target_idx.long()
        ok = (idx >= 0) & (idx < ctx.n_targets)
        g = grad_out.index_select(0, idx.clamp(0, ctx.n_targets - 1))
        return g * ok.view(-1, *([1] * (g.dim() - 1))), None, None


class NurbsEval(torch.autograd.Function):
    """``NURBSSurfaces.calculate_surface_points_and_normals`` (artist/nurbs/surfaces.py:475-689),
    differentiable w.r.t. the control points."""

    @staticmethod
    def forward(ctx, control_points, eval_points, knots_u, knots_v, canting, translations, p, q, uniform,
                n_unique_u, n_unique_v, orientation=None):
        dev = _require_cuda(control_points, eval_points, knots_u, knots_v)
        cp = _f32c(control_points)
        H, F, nu, nv, three = cp.shape
        if three != 3:
            raise ValueError("control_points must be [H,F,nu,nv,3]")
        uv = eval_points if eval_points.dtype == torch.float32 else eval_points.float()
        if uv.dim() != 4 or uv.shape[0] != H or uv.shape[1] != F or uv.shape[3] != 2:
            raise ValueError("evaluation_points must be [H,F,M,2]")
        M = uv.shape[2]
        if uv.stride(3) != 1 or uv.stride(2) != 2 or (uv.data_ptr() % 8) != 0:
            uv = uv.contiguous()   # expanded (stride-0) heliostat/facet dims are passed through
        ku = _converted(knots_u.expand(H, F, nu + p + 1), "knots", _f32c)
        kv = _converted(knots_v.expand(H, F, nv + q + 1), "knots", _f32c)
        cant = None if canting is None else _f32c(canting)
        tr = None if canting is None else _f32c(translations.reshape(H, F, 4))
        points = torch.empty((H, F, M, 4), dtype=torch.float32, device=dev)
        normals = torch.empty((H, F, M, 4), dtype=torch.float32, device=dev)
        ori = None
        if orientation is not None:      # fused alignment (a constant here: a learning kinematics takes align_surfaces)
            ori = _f32c(orientation.detach())
            if ori.shape != (H, 4, 4) or ori.device != dev:
                raise ValueError("orientation must be [H,4,4] on the control points' device")
        with torch.cuda.device(dev):
            rc = _lib.lib().art_nurbs_fwd(
                cp.data_ptr(), uv.data_ptr(), uv.stride(0), uv.stride(1), ku.data_ptr(), kv.data_ptr(),
                None if cant is None else cant.data_ptr(), None if tr is None else tr.data_ptr(),
                p, q, 1 if uniform else 0, n_unique_u, n_unique_v, H, F, M, nu, nv,
                None if ori is None else ori.data_ptr(), points.data_ptr(), normals.data_ptr(), _stream(dev))
        _lib.check(rc, "art_nurbs_fwd")
        ctx.save_for_backward(cp, uv, ku, kv, cant if cant is not None else cp.new_empty(0),
                              ori if ori is not None else cp.new_empty(0))
        ctx.meta = (p, q, bool(uniform), n_unique_u, n_unique_v, cant is not None, ori is not None)
        return points, normals

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_points, g_normals):
        cp, uv, ku, kv, cant, ori = ctx.saved_tensors
        p, q, uniform, nuq_u, nuq_v, has_cant, has_ori = ctx.meta
        dev = cp.device
        H, F, nu, nv, _ = cp.shape
        M = uv.shape[2]
        g_points, g_normals = _f32c(g_points), _f32c(g_normals)
        g_cp = torch.empty_like(cp)
        with torch.cuda.device(dev):
            rc = _lib.lib().art_nurbs_bwd(
                cp.data_ptr(), uv.data_ptr(), uv.stride(0), uv.stride(1), ku.data_ptr(), kv.data_ptr(),
                cant.data_ptr() if has_cant else None, p, q, 1 if uniform else 0, nuq_u, nuq_v, H, F, M, nu, nv,
                ori.data_ptr() if has_ori else None, g_points.data_ptr(), g_normals.data_ptr(), g_cp.data_ptr(),
                _stream(dev))
        _lib.check(rc, "art_nurbs_bwd")
        return (g_cp,) + (None,) * 11


def nurbs_surface_points_and_normals(control_points, eval_points, knots_u, knots_v, degrees, canting=None,
                                     translations=None, uniform=True, n_unique=None, orientation=None):
    p, q = int(degrees[0]), int(degrees[1])
    nu, nv = control_points.shape[2], control_points.shape[3]
    if n_unique is None:
        n_unique = (nu - p + 1, nv - q + 1)
    return NurbsEval.apply(control_points, eval_points, knots_u, knots_v, canting, translations, p, q,
                           bool(uniform), int(n_unique[0]), int(n_unique[1]), orientation)


class AlignSurfaces(torch.autograd.Function):
    """``points @ orientation^T`` and ``normals @ orientation^T`` in one pass
    (artist/field/heliostat_group_rigid_body.py:217-222, 265-270), differentiable w.r.t. all three inputs."""

    @staticmethod
    def forward(ctx, points, normals, orientation):
        dev = _require_cuda(points, normals, orientation)
        points, normals, orientation = _f32c(points), _f32c(normals), _f32c(orientation)
        H, P = points.shape[0], points.shape[1]
        if points.shape != (H, P, 4) or normals.shape != (H, P, 4) or orientation.shape != (H, 4, 4):
            raise ValueError("points/normals must be [H,P,4] and orientation [H,4,4]")
        out_p, out_n = torch.empty_like(points), torch.empty_like(normals)
        with torch.cuda.device(dev):
            rc = _lib.lib().art_align_fwd(points.data_ptr(), normals.data_ptr(), orientation.data_ptr(), H, P,
                                          out_p.data_ptr(), out_n.data_ptr(), _stream(dev))
        _lib.check(rc, "art_align_fwd")
        ctx.save_for_backward(points, normals, orientation)
        return out_p, out_n

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_out_p, g_out_n):
        points, normals, orientation = ctx.saved_tensors
        dev = points.device
        H, P = points.shape[0], points.shape[1]
        g_out_p, g_out_n = _f32c(g_out_p), _f32c(g_out_n)
        g_p, g_n = torch.empty_like(points), torch.empty_like(normals)
        g_m = torch.empty_like(orientation) if ctx.needs_input_grad[2] else None
        with torch.cuda.device(dev):
            rc = _lib.lib().art_align_bwd(points.data_ptr(), normals.data_ptr(), orientation.data_ptr(),
                                          g_out_p.data_ptr(), g_out_n.data_ptr(), H, P, g_p.data_ptr(), g_n.data_ptr(),
                                          None if g_m is None else g_m.data_ptr(), _stream(dev))
        _lib.check(rc, "art_align_bwd")
        return g_p, g_n, g_m


def align_surfaces(points: torch.Tensor, normals: torch.Tensor, orientation: torch.Tensor):
    """Aligned ``(points, normals)`` = ``(points @ orientation^T, normals @ orientation^T)``."""
    return AlignSurfaces.apply(points, normals, orientation)
