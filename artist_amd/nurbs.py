"""Drop-in for ``artist.nurbs.NURBSSurfaces`` backed by the gfx950 kernels.

Same constructor, attributes and method names as ``artist/nurbs/surfaces.py:8-727``; the body of
``calculate_surface_points_and_normals`` (:475-689) is one fused HIP kernel (plus its backward),
not a chain of ATen ops.  Helper constructors follow ``artist/nurbs/utils.py``.
"""
from __future__ import annotations

import weakref

import torch

from . import ops


_KNOT_CACHE: dict = {}
_DEGREE_CACHE: dict = {}


def _host_degrees(degrees: torch.Tensor) -> tuple[int, int]:
    """``(p, q)`` as Python ints.  Reading a device tensor is a stream synchronisation, which would stop the host
    from running ahead of the GPU once per epoch: the values are remembered for the SAME tensor object (weak
    reference) at the same version - a new tensor, even at a recycled address, is read again."""
    if degrees.device.type == "cpu":
        return int(degrees[0]), int(degrees[1])
    entry = _DEGREE_CACHE.get(id(degrees))
    if entry is not None and entry[0]() is degrees and entry[1] == degrees._version:
        return entry[2]
    if len(_DEGREE_CACHE) > 256:
        _DEGREE_CACHE.clear()
    value = (int(degrees[0]), int(degrees[1]))
    _DEGREE_CACHE[id(degrees)] = (weakref.ref(degrees), degrees._version, value)
    return value


class NURBSSurfaces(torch.nn.Module):
    """See ``artist/nurbs/surfaces.py:8-96`` for the attribute documentation.

    Parameters are those of the reference: ``degrees [2]``, ``control_points [H,F,nu,nv,3]``,
    ``uniform`` and ``device``.
    """

    def __init__(self, degrees: torch.Tensor, control_points: torch.Tensor, uniform: bool = True,
                 device: torch.device | None = None) -> None:
        super().__init__()
        device = control_points.device if device is None else torch.device(device)
        self.degrees = degrees
        self.control_points = control_points
        self.uniform = uniform
        self.number_of_surfaces = self.control_points.shape[0]
        self.number_of_facets_per_surface = self.control_points.shape[1]
        self._degrees_host = _host_degrees(degrees)
        self.knot_vectors_u = self.calculate_uniform_knot_vectors(direction=0, device=device)
        self.knot_vectors_v = self.calculate_uniform_knot_vectors(direction=1, device=device)
        # torch.unique(knot_vectors, dim=2) of the reference (surfaces.py:199) counted analytically for
        # the knots built here; recounted (one sync) only if a caller swaps the knot tensors.
        self._built_knots = (self.knot_vectors_u, self.knot_vectors_v)
        self._n_unique = tuple(self.control_points.shape[2 + d] - self._degrees_host[d] + 1 for d in (0, 1))

    def calculate_uniform_knot_vectors(self, direction: int, device: torch.device | None = None) -> torch.Tensor:
        """Clamped uniform knots ``[0]*deg + linspace(0,1,n-deg+1) + [1]*deg`` replicated ``[H,F,K]``
        (artist/nurbs/surfaces.py:98-155)."""
        degree = self._degrees_host[direction]
        n = self.control_points.shape[2 + direction]
        key = (n, degree, str(device))
        knot_vector = _KNOT_CACHE.get(key)
        if knot_vector is None:
            # a reconstruction loop builds a NURBSSurfaces per epoch (surface_reconstructor.py:516-531): the knots
            # depend on the sizes only, so they are built once per (size, degree, device) - five tiny kernels less
            # per direction and epoch.  The cached tensor is shared: replace the attribute, never write into it.
            knot_vector = torch.zeros(n + degree + 1, device=device)
            knot_vector[degree:-degree] = torch.linspace(0, 1, n - degree + 1, device=device)
            knot_vector[-degree:] = 1
            _KNOT_CACHE[key] = knot_vector
        # expand() instead of repeat(): identical values, no H*F copies in HBM
        return knot_vector.unsqueeze(0).unsqueeze(0).expand(
            self.number_of_surfaces, self.number_of_facets_per_surface, -1)

    def _unique_counts(self) -> tuple[int, int]:
        if self.knot_vectors_u is self._built_knots[0] and self.knot_vectors_v is self._built_knots[1]:
            return self._n_unique
        return (int(torch.unique(self.knot_vectors_u, dim=2).shape[2]),
                int(torch.unique(self.knot_vectors_v, dim=2).shape[2]))

    def find_spans(self, direction: int, evaluation_points: torch.Tensor, knot_vectors: torch.Tensor,
                   device: torch.device | None = None) -> torch.Tensor:
        """Knot spans in one direction (artist/nurbs/surfaces.py:157-245).  Not on the hot path (the
        fused kernel computes spans itself); kept for API parity and used by the tests."""
        degree = self._degrees_host[direction]
        x = evaluation_points[:, :, :, direction]
        if self.uniform:
            n_unique = torch.unique(knot_vectors, dim=2).shape[2]
            return torch.floor(x * (n_unique - 1)).long() + degree
        number_of_knots = knot_vectors.shape[2] - degree - 1
        lefts = knot_vectors[:, :, degree:number_of_knots]
        rights = knot_vectors[:, :, degree + 1:number_of_knots + 1]
        in_span = (x.unsqueeze(-1) >= lefts.unsqueeze(2)) & (x.unsqueeze(-1) < rights.unsqueeze(2))
        is_last = torch.isclose(x, knot_vectors[:, :, number_of_knots].unsqueeze(-1).expand_as(x), atol=1e-5, rtol=1e-5)
        spans = in_span.int().argmax(dim=-1) + degree
        return torch.where(is_last, torch.full_like(spans, number_of_knots - 1), spans)

    def calculate_surface_points_and_normals(self, evaluation_points: torch.Tensor, canting: torch.Tensor | None,
                                             facet_translations: torch.Tensor | None,
                                             device: torch.device | None = None,
                                             orientations: torch.Tensor | None = None) -> tuple[torch.Tensor, torch.Tensor]:
        """Surface points and unit normals ``[H,F,M,4]`` (artist/nurbs/surfaces.py:475-689).

        ``orientations`` (an addition to the reference signature): ``[H,4,4]`` alignment matrices applied in the same
        kernel - ``points @ M^T``, ``normals @ M^T`` as in ``heliostat_group_rigid_body.py:217-222`` - for the epoch of
        the surface reconstructor, which aligns right after evaluating (surface_reconstructor.py:516-546).  The result
        equals evaluation + ``artist_amd.align_surfaces`` bit for bit; the matrices are constants to autograd (a
        kinematics that learns passes through ``align_surfaces``)."""
        p, q = self._degrees_host
        nuq = self._unique_counts()
        if orientations is not None and orientations.requires_grad:
            raise ValueError("orientations that require grad go through artist_amd.align_surfaces, not the fused evaluation")
        return ops.NurbsEval.apply(self.control_points, evaluation_points, self.knot_vectors_u, self.knot_vectors_v,
                                   canting, facet_translations, p, q, bool(self.uniform), nuq[0], nuq[1], orientations)

    def forward(self, evaluation_points, canting, facet_translations, device=None):
        """Alias of :meth:`calculate_surface_points_and_normals` (artist/nurbs/surfaces.py:691-727)."""
        return self.calculate_surface_points_and_normals(evaluation_points, canting, facet_translations, device)


def create_nurbs_evaluation_grid(number_of_evaluation_points: torch.Tensor, epsilon: float = 1e-7,
                                 device: torch.device | None = None) -> torch.Tensor:
    """``[Ne*Nn, 2]`` cartesian grid on ``[eps, 1-eps]^2`` (artist/nurbs/utils.py:7-49)."""
    e = torch.linspace(epsilon, 1 - epsilon, int(number_of_evaluation_points[0]), device=device)
    n = torch.linspace(epsilon, 1 - epsilon, int(number_of_evaluation_points[1]), device=device)
    return torch.cartesian_prod(e, n)


def create_planar_nurbs_control_points(number_of_control_points: torch.Tensor, canting: torch.Tensor,
                                       device: torch.device | None = None) -> torch.Tensor:
    """Flat equidistant control nets ``[F,nu,nv,3]`` sized by the canting-vector norms
    (artist/nurbs/utils.py:52-121)."""
    canting = canting.to(device)
    n_u, n_v = int(number_of_control_points[0]), int(number_of_control_points[1])
    control_points = torch.zeros((canting.shape[0], n_u, n_v, 3), device=device, dtype=canting.dtype)
    u_lin = torch.linspace(0, 1, n_u, device=device, dtype=canting.dtype)
    v_lin = torch.linspace(0, 1, n_v, device=device, dtype=canting.dtype)
    facet_dimensions = torch.norm(canting, dim=2)
    u_coordinates = -facet_dimensions[:, 0, None] + 2 * facet_dimensions[:, 0, None] * u_lin
    v_coordinates = -facet_dimensions[:, 1, None] + 2 * facet_dimensions[:, 1, None] * v_lin
    control_points[..., 0] = u_coordinates[:, :, None]
    control_points[..., 1] = v_coordinates[:, None, :]
    return control_points
