"""The reference-SHAPED CPU baseline: ARTIST's hot loop as eager PyTorch-CPU tensor ops.

What BASELINE.md section 4 (item 1) / SURVEY.md 8(d) call "the reference CPU path": the op sequence of
``HeliostatRayTracer.trace_rays`` (artist/raytracing/heliostat_ray_tracer.py:316-506) - a Python loop over batches of
``batch_size`` heliostats, each batch a chain of whole-tensor ATen ops with every per-ray intermediate materialised
(the ``[B,R,P,4,4]`` scatter matrices of ``rotate_distortions``, artist/geometry/transforms.py:52-83; the masks and
temporaries of ``line_plane_intersections``, artist/raytracing/geometry.py:116-197; the int64 index tensors and four
``scatter_add_`` calls of ``bilinear_splatting``, heliostat_ray_tracer.py:674-778).  ARTIST itself does not travel to
the GPU box, so this restatement - pinned to the fixtures the imported reference generated
(tests/test_host_logic.py::test_torch_eager_baseline_equals_reference_fixtures) - is what bench.py times on the host
cores next to the C oracle.  Blocking off, planar receivers: the setting of the metric config.

Yardstick only: nothing under ``artist_amd/`` imports this file.
"""
from __future__ import annotations

import time

import torch


def scatter_matrices(e: torch.Tensor, u: torch.Tensor) -> torch.Tensor:
    """``[B,R,P,4,4]`` rotation matrices from the two distortion angles (transforms.py:52-83): zeros + nine fills."""
    ce, se, cu, su = torch.cos(e), torch.sin(e), torch.cos(u), torch.sin(u)
    m = torch.zeros(e.shape + (4, 4), dtype=e.dtype)
    m[..., 0, 0] = cu
    m[..., 0, 1] = -su
    m[..., 1, 0] = ce * su
    m[..., 1, 1] = ce * cu
    m[..., 1, 2] = -se
    m[..., 2, 0] = se * su
    m[..., 2, 1] = se * cu
    m[..., 2, 2] = ce
    m[..., 3, 3] = torch.ones_like(e)
    return m


def trace_batch(points, normals, incident, dist_u, dist_e, centers, plane_normals, dims, resolution, ray_magnitude,
                extinction, reflectivity):
    """One batch of heliostats: returns ``(flux [B,Hh,W], intercept [B], on_target [B])``."""
    B, R, P = dist_u.shape
    W, Hh = int(resolution[0]), int(resolution[1])
    # reflect (geometry.py:32-41)
    inc = incident.unsqueeze(1)
    reflected = inc - 2 * torch.sum(inc * normals, dim=-1, keepdim=True) * normals
    # scatter_rays (heliostat_ray_tracer.py:543-560)
    directions = (scatter_matrices(dist_e, dist_u) @ reflected.unsqueeze(1).unsqueeze(-1)).squeeze(-1)
    magnitudes = torch.full(directions.shape[:3], ray_magnitude, dtype=points.dtype)
    # line_plane_intersections (geometry.py:101-204)
    d3, o3 = directions[..., :3], points[..., :3]
    n3, c3 = plane_normals[..., :3], centers[..., :3]
    cosines = (d3 * n3[:, None, None, :]).sum(dim=-1)
    front = cosines < 0.0
    numerator = ((c3[:, None, :] - o3) * n3[:, None, :]).sum(dim=-1)[:, None, :]
    distances = (numerator / torch.where(front, cosines, 1.0)) * front
    hits = o3[:, None, :, :] + d3 * distances[:, :, :, None]
    intensities = magnitudes * -cosines
    on_plane_e = hits[..., 0] + (dims[:, 0] / 2)[:, None, None] - centers[:, 0][:, None, None]
    on_plane_u = hits[..., 2] + (dims[:, 1] / 2)[:, None, None] - centers[:, 2][:, None, None]
    px_e = on_plane_e / dims[:, 0, None, None] * (W - 1)
    px_u = on_plane_u / dims[:, 1, None, None] * (Hh - 1)
    valid = (0 <= px_e) & (px_e <= W - 1) & (0 <= px_u) & (px_u <= Hh - 1) & front
    px_e = px_e * valid
    px_u = px_u * valid
    distances = distances * valid
    intensities = intensities * valid
    px_e = (W - 1) - px_e
    # intensity product (heliostat_ray_tracer.py:482-487), blocking off
    blocked = torch.zeros_like(intensities)
    absolute = intensities * (1 - blocked) * (1 - extinction) * reflectivity
    # bilinear_splatting (heliostat_ray_tracer.py:674-778)
    lo_e, lo_u = px_e.long(), px_u.long()
    w_lo_e = lo_e + 1 - px_e
    w_lo_u = lo_u + 1 - px_u
    w_hi_e = px_e - lo_e
    w_hi_u = px_u - lo_u
    v1 = w_lo_e * w_hi_u * absolute
    v2 = w_hi_e * w_hi_u * absolute
    v3 = w_hi_e * w_lo_u * absolute
    v4 = w_lo_e * w_lo_u * absolute
    on = (0 <= lo_e) & (lo_e + 1 < W) & (0 <= lo_u) & (lo_u + 1 < Hh)
    flat = torch.zeros((B, Hh * W), dtype=points.dtype)
    i1 = (lo_u + 1) * W + lo_e
    i2 = (lo_u + 1) * W + lo_e + 1
    i3 = lo_u * W + lo_e + 1
    i4 = lo_u * W + lo_e
    for idx in (i1, i2, i3, i4):
        idx[~on] = 0
    flat.scatter_add_(1, i1.reshape(B, -1), (v1 * on).reshape(B, -1))
    flat.scatter_add_(1, i2.reshape(B, -1), (v2 * on).reshape(B, -1))
    flat.scatter_add_(1, i3.reshape(B, -1), (v3 * on).reshape(B, -1))
    flat.scatter_add_(1, i4.reshape(B, -1), (v4 * on).reshape(B, -1))
    flux = torch.flip(flat.view(B, Hh, W), [1])
    rays = R * P
    return flux, (absolute > 0).sum((1, 2)) / rays, (intensities > 0).sum((1, 2)) / rays


def trace_rays(points, normals, incident, dist_u, dist_e, target_idx, centers, plane_normals, dims, resolution=(256, 256),
               ray_magnitude=1.0, extinction=0.0, reflectivity=0.935, batch_size=100):
    """The Python batch loop (heliostat_ray_tracer.py:316-506).  CPU float32 tensors in, ``(flux, intercept, on_target)``
    out.  Works under autograd (points / normals may require grad), like the reference."""
    H = points.shape[0]
    flux = torch.empty((H, int(resolution[1]), int(resolution[0])), dtype=points.dtype)
    intercept, on_target = torch.empty(H), torch.empty(H)
    tix = target_idx.long()
    for start in range(0, H, batch_size):
        sl = slice(start, min(start + batch_size, H))
        t = tix[sl]
        # the DataLoader's collate copies each batch of distortion views to contiguous tensors (:316)
        f, a, b = trace_batch(points[sl], normals[sl], incident[sl], dist_u[sl].contiguous(), dist_e[sl].contiguous(),
                              centers[t], plane_normals[t], dims[t], resolution, ray_magnitude, extinction, reflectivity)
        flux[sl], intercept[sl], on_target[sl] = f, a, b
    return flux, intercept, on_target


def time_epoch(points, normals, incident, dist_u, dist_e, target_idx, centers, plane_normals, dims, batch_size, backward,
               resolution=(256, 256)):
    """Seconds for one forward (+ backward to points / normals: what autograd does in a reconstruction epoch)."""
    p = points.clone().requires_grad_(backward)
    n = normals.clone().requires_grad_(backward)
    t0 = time.perf_counter()
    with torch.set_grad_enabled(backward):
        flux, _, _ = trace_rays(p, n, incident, dist_u, dist_e, target_idx, centers, plane_normals, dims, resolution,
                                batch_size=batch_size)
        if backward:
            (flux * flux).sum().backward()
    return time.perf_counter() - t0
