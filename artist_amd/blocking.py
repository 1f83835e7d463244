"""Blocking primitives - the host-side half of ``artist/raytracing/blocking.py``.

A heliostat blocks as the rectangle spanned by four of its aligned surface points.  Building those
rectangles is a gather of ``4 N`` points plus one cross product per heliostat: it stays in torch so that
the corner points remain in the autograd graph (the reference differentiates through them as well), and
the per-ray work - which primitives matter, the soft blocking mask, its adjoint - runs inside the HIP
trace kernels (``artist_amd/csrc/trace_kernels.hip``), which receive the tables built here.

The reference narrows the primitives with an LBVH (``lbvh_filter_blocking_planes``, blocking.py:832-995)
because its mask is a dense ``[rays x primitives]`` tensor; a leaf of that tree is reached exactly when the
ray passes the leaf's own box test, so the set it returns is the set of primitives whose box is hit by at
least one foreign ray.  The HIP path computes that same set directly (per-heliostat beam cull + per-ray box
test, ``art_blocking_filter``), no tree needed.
"""
from __future__ import annotations

import math

import torch


def _spans_and_normals(corners: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """spans = (corner1 - corner0, corner3 - corner0); normal = normalize(span_u x span_v)  (blocking.py:190-207)."""
    spans = torch.stack((corners[:, 1] - corners[:, 0], corners[:, 3] - corners[:, 0]), dim=1)
    cross = torch.linalg.cross(spans[:, 0, :3], spans[:, 1, :3], dim=-1)
    normals = torch.cat((torch.nn.functional.normalize(cross, dim=-1), torch.zeros_like(cross[:, :1])), dim=-1)
    return spans, normals


_CORNER_INDEX: dict = {}


def create_blocking_primitives_rectangles_by_index(blocking_heliostats_active_surface_points: torch.Tensor,
                                                   device: torch.device | None = None
                                                   ) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Rectangles from the known indices of the corner points (blocking.py:123-209).

    Four facets in two rows and two columns, each with ``sqrt(P/4)`` x ``sqrt(P/4)`` row-major points.
    Corner order ``1 | 2 / 0 | 3`` (0 = lower left).  Returns ``corners [N,4,4]``, ``spans [N,2,4]``,
    ``normals [N,4]``.
    """
    pts = blocking_heliostats_active_surface_points
    P = pts.shape[1]
    side = math.sqrt(P / 4)
    key = (P, pts.device)
    index = _CORNER_INDEX.get(key)
    if index is None:      # (a host list -> device tensor copy waits for the stream: once per shape, not once per trace call)
        index = _CORNER_INDEX[key] = torch.tensor([int(P / 2), int(side - 1), int((P / 2) - 1), int(P - side)], device=pts.device)
    corners = pts.index_select(1, index)
    spans, normals = _spans_and_normals(corners)
    return corners, spans, normals


def create_blocking_primitives_rectangle(blocking_heliostats_surface_points: torch.Tensor,
                                         blocking_heliostats_active_surface_points: torch.Tensor,
                                         device: torch.device | None = None
                                         ) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Rectangles from the surface points closest to the east/north bounding box of the UNALIGNED surface
    (blocking.py:13-120); positions are then read from the aligned surface."""
    unaligned, aligned = blocking_heliostats_surface_points, blocking_heliostats_active_surface_points
    e, n = unaligned[:, :, 0], unaligned[:, :, 1]
    lo_e, hi_e, lo_n, hi_n = e.amin(1), e.amax(1), n.amin(1), n.amax(1)
    wanted = torch.stack((torch.stack((lo_e, lo_n), 1), torch.stack((lo_e, hi_n), 1), torch.stack((hi_e, hi_n), 1),
                          torch.stack((hi_e, lo_n), 1)), dim=1)                                  # [N,4,2]
    distance = torch.linalg.vector_norm(unaligned[:, :, None, :2] - wanted[:, None], dim=-1)     # [N,P,4]
    index = distance.argmin(dim=1)                                                               # [N,4]
    corners = torch.gather(aligned, 1, index[:, :, None].expand(-1, -1, 4))
    spans, normals = _spans_and_normals(corners)
    return corners, spans, normals
