"""Drop-in for ``artist.raytracing.heliostat_ray_tracer.HeliostatRayTracer`` on MI355X.

Same constructor arguments, method names, return values and error behaviour as
``artist/raytracing/heliostat_ray_tracer.py:19-778``; the per-batch Python loop of eager ATen
ops inside ``trace_rays`` (:316-506) is ONE fused HIP kernel launch (``art_trace_fwd``), and its
autograd is ``art_trace_bwd``.  ``scenario`` / ``heliostat_group`` are duck-typed: ARTIST's own
``Scenario`` / ``HeliostatGroup`` objects work unchanged, and so do the light stand-ins in
``artist_amd.scene`` used by the tests and the benchmark.

Multi-rank contract (the reference's is broken for world_size > 1): a rank
returns rows for the heliostat samples it OWNS only, in ``get_sampler_indices()`` order, so that
``get_bitmaps_per_target(flux_local, target_area_indices[sampler_indices])`` is well defined and
the sum over ranks equals the single-rank result.
"""
from __future__ import annotations

import os

import logging

import weakref

import torch

from . import ops
from .blocking import create_blocking_primitives_rectangles_by_index
from .sampling import DistortionsDataset, RestrictedDistributedSampler

log = logging.getLogger(__name__)

_DEFAULT_RESOLUTION = torch.tensor([256, 256])   # artist/util/indices.py:307


def reflect(incident_ray_directions: torch.Tensor, reflection_surface_normals: torch.Tensor) -> torch.Tensor:
    """``i - 2 (i.n) n`` (artist/raytracing/geometry.py:11-41); only used to publish
    ``heliostat_group.preferred_reflection_directions`` - the kernel reflects in registers.  ``[H,1,4]`` directions and
    ``[H,P,4]`` normals on the GPU go through ``art_reflect`` (one pass, the reference's operation order) instead of
    five elementwise torch kernels (0.41 -> 0.07 ms per call at the metric size)."""
    i, nrm = incident_ray_directions, reflection_surface_normals
    if (nrm.is_cuda and nrm.dim() == 3 and nrm.shape[-1] == 4 and i.dim() == 3 and i.shape == (nrm.shape[0], 1, 4)
            and nrm.dtype == torch.float32 and i.dtype == torch.float32 and not torch.is_grad_enabled()):
        return ops.reflect_directions(i[:, 0], nrm)
    return i - 2 * torch.sum(i * nrm, dim=-1, keepdim=True) * nrm


def _version_of(t: torch.Tensor):
    """``t._version`` or None where there is no version counter (inference-mode tensors): no caching then.  The per-call
    caches of this module key on tensor identity + version, so index tensors and masks have to be changed with tracked
    in-place ops (or replaced): a write through ``.data`` or by a raw kernel is invisible to them."""
    try:
        return t._version
    except (RuntimeError, AttributeError):
        return None


def target_area_counts(tower) -> tuple[int, int]:
    """(planar, cylindrical) target areas of a tower, from the HOST side of its tables (``number_of_target_areas`` is
    ``len(names)``, artist/field/tower_target_areas.py:62): ``solar_tower.number_of_target_areas_per_type``
    is a device tensor, and reading it costs a host-device synchronisation in every call."""
    counts = []
    for k in range(2):
        if k >= len(tower.target_areas):
            counts.append(0)
            continue
        area = tower.target_areas[k]
        n = getattr(area, "number_of_target_areas", None)
        if not isinstance(n, int):
            table = getattr(area, "centers", None)
            n = int(table.shape[0]) if table is not None else 0
        counts.append(n)
    return counts[0], counts[1]


def _planar_tables(tower, device):
    """(centers [T,4], normals [T,4], dimensions [T,2]) of ``solar_tower.target_areas[planar]``; empty tables when
    the tower has cylindrical receivers only."""
    planar = tower.target_areas[0]
    if target_area_counts(tower)[0] == 0 or not hasattr(planar, "centers"):
        z = torch.zeros((0, 4), device=device)
        return z, z, torch.zeros((0, 2), device=device)
    return planar.centers, planar.normals, planar.dimensions


def _cylinder_tables(tower):
    """The six ``TowerTargetAreasCylindrical`` tensors (artist/field/tower_target_areas_cylindrical.py:52-102) or
    None when the tower has none."""
    if target_area_counts(tower)[1] == 0:
        return None
    c = tower.target_areas[1]
    return (c.centers, c.normals, c.axes, c.radii, c.heights, c.opening_angles)


class HeliostatRayTracer:
    """See ``artist/raytracing/heliostat_ray_tracer.py:19-69`` for the attribute documentation."""

    #: publish ``heliostat_group.preferred_reflection_directions`` like the reference does (:285-290)
    publish_reflection_directions = True

    def __init__(self, scenario, heliostat_group, blocking_active: bool = True, world_size: int = 1, rank: int = 0,
                 batch_size: int = 100, random_seed: int = 7, bitmap_resolution: torch.Tensor = _DEFAULT_RESOLUTION,
                 dni: float | None = None) -> None:
        self.scenario = scenario
        self.heliostat_group = heliostat_group
        self.blocking_active = blocking_active
        self.world_size = world_size
        self.rank = rank
        self.batch_size = batch_size   # accepted for API parity; the fused kernel has no per-ray intermediates

        self.light_source = scenario.light_sources.light_source_list[0]
        number_of_samples = int(self.heliostat_group.number_of_active_heliostats)
        self.distortions_sampler = RestrictedDistributedSampler(
            number_of_samples=number_of_samples,
            number_of_active_heliostats=int((self.heliostat_group.active_heliostats_mask > 0).sum()),
            world_size=self.world_size,
            rank=self.rank,
        )
        # A rank keeps the distortions of the heliostat samples it owns only (the reference samples all [H,R,P] on every
        # rank, sampling.py:49-53): 8 B per ray of this rank instead of 8 B per ray of the field.
        self.distortions_dataset = DistortionsDataset(
            light_source=self.light_source,
            number_of_points_per_heliostat=self.heliostat_group.active_surface_points.shape[1],
            number_of_active_heliostats=number_of_samples,
            random_seed=random_seed,
            rows=self.distortions_sampler.rank_indices if self.world_size > 1 else None,
        )
        self.bitmap_resolution = bitmap_resolution
        self._resolution_host = (int(bitmap_resolution[0]), int(bitmap_resolution[1]))

        if self.blocking_active:
            # heliostat_ray_tracer.py:159-183: every heliostat of every group can block; aligned surfaces where a
            # group has been aligned, horizontal ones at their positions otherwise
            surfaces = []
            for group in self.scenario.heliostat_field.heliostat_groups:
                points = group.surface_points + group.positions.unsqueeze(1)
                mask = group.active_heliostats_mask.bool()
                if mask.any():
                    points = points.index_put((torch.nonzero(mask, as_tuple=True)[0],), group.active_surface_points)
                else:
                    log.warning("Not all heliostat groups have been aligned yet. "
                                "Using horizontal heliostats as blocking planes.")
                surfaces.append(points)
            self.blocking_heliostat_surfaces = torch.cat(
                [group.surface_points for group in self.scenario.heliostat_field.heliostat_groups])
            self.blocking_heliostat_surfaces_active = torch.cat(surfaces)
            self._scatter_angle_cache = (None, 0.0)
        #: reproduce which rectangles the reference's LBVH can reach (see artist_amd/blocking.py); False = every
        #: rectangle whose box is hit, as ``lbvh_filter_blocking_planes`` documents
        self._checked_targets = None
        self.lbvh_compat = True
        #: indices of the rectangles the last ``trace_rays`` call filtered (blocking only)
        self._filter_flags = self._filtered = None
        self._owner_cache = None

        if dni is not None:
            # heliostat_ray_tracer.py:185-201
            canting_norm = (torch.norm(self.heliostat_group.canting[0], dim=1)[0])[:2]
            dimensions = (canting_norm * 4) + 0.02
            heliostat_surface_area = dimensions[0] * dimensions[1]
            power_single_heliostat = dni * heliostat_surface_area
            rays_per_heliostat = self.heliostat_group.surface_points.shape[1] * self.light_source.number_of_rays
            self.ray_magnitude = power_single_heliostat / rays_per_heliostat
        else:
            self.ray_magnitude = 1.0
        self._local_cache = None

    # ------------------------------------------------------------------------------------------
    def get_sampler_indices(self) -> torch.Tensor:
        """Indices of the heliostat samples assigned to this rank (:205-218)."""
        return torch.tensor(self.distortions_sampler.rank_indices, device=self.distortions_dataset.distortions_u.device)

    def _local_rows(self, device):
        """(index tensor of the owned heliostat samples or None when this rank owns all, dist_u, dist_e of those rows)."""
        if self._local_cache is not None and self._local_cache[0] == device:
            return self._local_cache[1:]
        ds = self.distortions_dataset
        du, de = ds.distortions_u, ds.distortions_e
        idx_list = self.distortions_sampler.rank_indices
        n_total = int(self.heliostat_group.number_of_active_heliostats)
        owns_all = len(idx_list) == n_total and idx_list == list(range(n_total))
        if len(idx_list) != du.shape[0]:            # a dataset someone swapped in: select the owned rows here
            sel = torch.tensor(idx_list, dtype=torch.long, device=du.device)
            du, de = du.index_select(0, sel), de.index_select(0, sel)
        if du.device != device:
            both = torch.stack((du, de), dim=-1).to(device)          # one interleaved buffer on the device
            du, de = both[..., 0], both[..., 1]
        idx = None if owns_all else torch.tensor(idx_list, dtype=torch.long, device=device)
        self._local_cache = (device, idx, du, de)
        self._owner_cache = None                       # (keyed on `idx`: rebuilt with it)
        return idx, du, de

    def _trace(self, incident_ray_directions, active_heliostats_mask, target_area_indices, ray_extinction_factor,
               mirror_reflectivity, device, per_target: bool):
        """The one place that assembles a trace call: alignment check, the rank's rows of every per-heliostat tensor, the
        tower's tables, the blocking rectangles.  ``per_target``: splat into the targets' bitmaps instead of the heliostats'."""
        group = self.heliostat_group
        # (the same tensor object needs no device comparison: one host-device synchronisation less per epoch)
        assert group.active_heliostats_mask is active_heliostats_mask or \
            torch.equal(group.active_heliostats_mask, active_heliostats_mask), \
            "Some heliostats were not aligned and cannot be raytraced."

        points, normals = group.active_surface_points, group.active_surface_normals
        device = points.device if device is None else torch.device(device)
        if self.publish_reflection_directions and not per_target:
            with torch.no_grad():
                group.preferred_reflection_directions = reflect(incident_ray_directions.unsqueeze(1), normals)

        tower = self.scenario.solar_tower
        self._validate_targets(target_area_indices, tower)
        idx, dist_u, dist_e = self._local_rows(device)
        if idx is not None:
            points, normals = points.index_select(0, idx), normals.index_select(0, idx)
            incident_ray_directions = incident_ray_directions.index_select(0, idx)
            target_area_indices = target_area_indices.index_select(0, idx)

        width, height = self._resolution_host
        flux, factors, flags = ops.TraceRays.apply(
            points, normals, incident_ray_directions, dist_u, dist_e, target_area_indices, *_planar_tables(tower, points.device),
            float(self.ray_magnitude), float(ray_extinction_factor), float(mirror_reflectivity), width, height, bool(per_target),
            _cylinder_tables(tower), *(self._blocking_arguments(idx, active_heliostats_mask) or (None, None, None, None, -1.0, True)),
            self._points_per_facet(points))
        if self.blocking_active:
            self._filter_flags, self._filtered = flags, None       # (indices on demand: nonzero() waits for the device)
        return flux, factors[0], factors[1], factors[2]

    def trace_rays(self, incident_ray_directions: torch.Tensor, active_heliostats_mask: torch.Tensor,
                   target_area_indices: torch.Tensor, ray_extinction_factor: float = 0.0,
                   mirror_reflectivity: float = 0.935, device: torch.device | None = None
                   ) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        """Heliostat ray tracing (:220-508).  Returns ``(flux [H,res_u,res_e], intercept_factor [H],
        on_target_factor [H], blocking_factor [H])`` for the heliostat samples owned by this rank."""
        return self._trace(incident_ray_directions, active_heliostats_mask, target_area_indices, ray_extinction_factor,
                           mirror_reflectivity, device, per_target=False)

    @property
    def filtered_blocking_primitive_indices(self):
        """Indices of the blocking rectangles the last ``trace_rays`` call kept (heliostat_ray_tracer.py:445-461) or None.
        The call leaves one flag per rectangle on the device; the index list is made when somebody reads it, so an epoch
        that never looks at it has no host-device synchronisation in its ray tracing."""
        if self._filtered is None and self._filter_flags is not None:
            self._filtered = torch.nonzero(self._filter_flags, as_tuple=True)[0]
            # this read waited for the device anyway: the moment to learn that the filter met more candidate rectangles
            # than the kernels hold (the device-side report of the asynchronous call, ops.check_async_errors)
            ops.check_async_errors(self._filter_flags.device, clear=False)
        return self._filtered

    def finish(self, device: torch.device | None = None) -> None:
        """Wait for the ray tracing queued so far and raise what only the device could find out (a target index outside
        the tables, more blocking rectangles inside one heliostat's ray cone than the kernels hold) -
        ``ops.check_async_errors``.  ``trace_rays`` itself never waits; call this after a single prediction or the last
        epoch (an overflowed heliostat's bitmap and factors are NaN either way)."""
        points = self.heliostat_group.active_surface_points
        ops.check_async_errors(points.device if device is None else device)

    @filtered_blocking_primitive_indices.setter
    def filtered_blocking_primitive_indices(self, value) -> None:
        self._filter_flags, self._filtered = None, value

    def _points_per_facet(self, points) -> int:
        """The group's surface tensors are facet-major ``[H, F * M, 4]`` (heliostat_group.py:26-63): tell the kernels M, so
        that a block of points never holds two facets' images (a layout hint - results do not depend on it)."""
        if os.environ.get("ARTIST_HIP_DEBUG") == "1" and os.environ.get("ARTIST_AMD_FACET_HINT", "1") == "0":      # A/B switch (debug only)
            return 0
        facets = int(getattr(self.heliostat_group, "number_of_facets_per_heliostat", 0) or 0)
        n_points = int(points.shape[1])
        return n_points // facets if facets > 1 and n_points % facets == 0 else 0

    def _validate_targets(self, target_area_indices, tower) -> None:
        """The kernels index the target tables with these: validated on the host, once per tensor object and version
        (one host-device synchronisation, not one per epoch).  The C ABI checks them again on the device and reports
        ``ART_ETARGET`` instead of reading out of bounds (include/artist_hip.h)."""
        checked = self._checked_targets
        version = _version_of(target_area_indices)
        if target_area_indices.numel() > 0 and not (version is not None and checked is not None and
                                                     checked[0]() is target_area_indices and checked[1] == version):
            lo, hi = int(target_area_indices.min()), int(target_area_indices.max())
            if lo < 0 or hi >= sum(target_area_counts(tower)):
                raise IndexError("target_area_indices out of range")
            self._checked_targets = (weakref.ref(target_area_indices), version)

    def _blocking_arguments(self, idx, active_heliostats_mask):
        """Rectangles of all heliostats (:291-301) + the rectangle index of each traced heliostat (:445-448)."""
        if not self.blocking_active:
            return ()
        corners, spans, normals = create_blocking_primitives_rectangles_by_index(self.blocking_heliostat_surfaces_active)
        # rectangle index of each active heliostat: once per mask tensor object and version (nonzero() waits for the device)
        cached = self._owner_cache
        version = _version_of(active_heliostats_mask)
        key = (version, None if idx is None else id(idx), str(active_heliostats_mask.device))
        if version is None or cached is None or cached[0]() is not active_heliostats_mask or cached[1] != key:
            owner = torch.nonzero(active_heliostats_mask, as_tuple=True)[0]
            if idx is not None:
                owner = owner.index_select(0, idx.to(owner.device))
            # (`idx` is kept alive by the entry, so its id cannot be handed to another tensor while the entry lives)
            self._owner_cache = cached = (weakref.ref(active_heliostats_mask), key, owner, idx)
        owner = cached[2]
        return corners, spans, normals, owner, self._max_scatter_angle(), self.lbvh_compat

    def _max_scatter_angle(self) -> float:
        """Largest scatter angle of the distortion dataset: bounds every heliostat's ray cone for the blocking cull.
        Computed once per dataset (one reduction + host read), again if a caller swaps the dataset's tensors."""
        ds = self.distortions_dataset
        u, e = ds.distortions_u, ds.distortions_e
        cached = self._scatter_angle_cache[0]
        # valid for the SAME tensor objects at the same versions only: a bound that is too small would change results
        vu, ve = _version_of(u), _version_of(e)
        if vu is None or ve is None or cached is None or cached[0]() is not u or cached[1]() is not e or cached[2:] != (vu, ve):
            value = float(torch.maximum(u.abs().max(), e.abs().max()))
            self._scatter_angle_cache = ((weakref.ref(u), weakref.ref(e), vu, ve), value)
        return self._scatter_angle_cache[1]

    def trace_rays_per_target(self, incident_ray_directions, active_heliostats_mask, target_area_indices,
                              ray_extinction_factor: float = 0.0, mirror_reflectivity: float = 0.935,
                              device: torch.device | None = None):
        """``trace_rays`` + ``get_bitmaps_per_target`` without materialising ``[H,res,res]``: every
        heliostat's rays are splatted straight into its target's bitmap (``[T,res_u,res_e]``, planar areas
        first, cylindrical second).  Extension of the reference API for field-scale flux prediction
        (configs 3 and 5)."""
        return self._trace(incident_ray_directions, active_heliostats_mask, target_area_indices, ray_extinction_factor,
                           mirror_reflectivity, device, per_target=True)

    def get_bitmaps_per_target(self, bitmaps_per_heliostat: torch.Tensor, target_area_indices: torch.Tensor,
                               device: torch.device | None = None) -> torch.Tensor:
        """Bitmaps per heliostat -> bitmaps per target area ``[T,res_u,res_e]`` (:563-608)."""
        n_targets = sum(target_area_counts(self.scenario.solar_tower))
        return ops.per_target_sum(bitmaps_per_heliostat, target_area_indices, n_targets)
