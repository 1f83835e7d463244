"""Light-weight holders for the tensors the hot path reads, plus the synthetic-field generator.

On a machine with ARTIST installed its own ``Scenario`` / ``HeliostatGroupRigidBody`` / ``SolarTower``
/ ``Sun`` objects are passed to :class:`artist_amd.raytracing.HeliostatRayTracer` directly.  These
stand-ins carry the same attribute names ("state read from other objects")
so that the tests and ``bench.py`` can build a scene on the GPU box, where ARTIST is absent.
They are NOT a re-implementation of ARTIST's field / kinematics packages: alignment here is an
ideal two-axis mount (no deviations, no actuators).

Reference attribute sources: ``artist/field/heliostat_group.py:133-222, 225-315``,
``artist/field/tower_target_areas_planar.py:45-80``, ``artist/field/solar_tower.py:50-180``,
``artist/scene/sun.py:39-119, 199-234``.
"""
from __future__ import annotations

import torch

from .nurbs import NURBSSurfaces, create_nurbs_evaluation_grid, create_planar_nurbs_control_points


class Sun:
    """Gaussian sun shape; ``get_distortions`` = seeded ``MultivariateNormal`` sample permuted to
    ``(u, e)`` views of one interleaved buffer (artist/scene/sun.py:96-119, 199-234)."""

    def __init__(self, number_of_rays: int, distribution_parameters: dict | None = None,
                 device: torch.device | None = None) -> None:
        params = dict(distribution_type="normal", mean=0.0, covariance=4.3681e-06)
        params.update(distribution_parameters or {})
        if params["distribution_type"] != "normal":
            raise ValueError("Unknown sunlight distribution type.")
        self.distribution_parameters = params
        self.number_of_rays = number_of_rays
        mean = torch.tensor([params["mean"], params["mean"]], dtype=torch.float, device=device)
        cov = torch.tensor([[params["covariance"], 0], [0, params["covariance"]]], dtype=torch.float, device=device)
        self.distribution = torch.distributions.MultivariateNormal(mean, cov)

    @classmethod
    def from_hdf5(cls, config_file, light_source_name: str | None = None, device: torch.device | None = None) -> "Sun":
        """artist/scene/sun.py:121-197 (``config_file`` = the light source's own group)."""
        from . import scenario
        t = scenario.read_light_source(config_file, light_source_name)
        return cls(number_of_rays=t["number_of_rays"], distribution_parameters=t["distribution_parameters"], device=device)

    def get_distortions(self, number_of_points: int, number_of_active_heliostats: int, random_seed: int = 7):
        loc = self.distribution.loc
        if loc.device.type == "cpu":
            torch.manual_seed(random_seed)
            sample = self.distribution.sample((number_of_active_heliostats, self.number_of_rays, number_of_points))
            distortions_u, distortions_e = sample.permute(3, 0, 1, 2)
            return distortions_u, distortions_e
        return self.get_distortions_rows(range(number_of_active_heliostats), number_of_points, number_of_active_heliostats,
                                         random_seed)

    def get_distortions_rows(self, rows, number_of_points: int, number_of_active_heliostats: int, random_seed: int = 7):
        """Rows ``rows`` of the ``[H,R,P]`` distortion views when the light source lives on the GPU (None on the CPU: the
        caller then slices the reference's one seeded stream, ``artist_amd.sampling.DistortionsDataset``).

        On the device every heliostat sample has a Philox stream of its own, keyed by (seed, row): a rank that owns
        some of the heliostats draws exactly its rows, bit-identical to the rows of an unsharded draw,
        and nothing of size ``[H,R,P]`` exists anywhere.  Same law as ``MultivariateNormal.sample``
        (``loc + scale_tril @ eps``), written element-wise: the batched 2x2 matrix-vector product of torch's
        ``MultivariateNormal.sample`` faulted on ROCm for ~1e7 and more samples."""
        loc, tril = self.distribution.loc, self.distribution.scale_tril
        if loc.device.type == "cpu":
            return None
        rows = [int(r) for r in rows]
        out = torch.empty((len(rows), self.number_of_rays, number_of_points, 2), dtype=loc.dtype, device=loc.device)
        gen = torch.Generator(device=loc.device)
        for k, row in enumerate(rows):
            gen.manual_seed((int(random_seed) * 1000003 + row) & 0x7FFFFFFFFFFFFFFF)
            torch.randn(out[k].shape, generator=gen, dtype=loc.dtype, device=loc.device, out=out[k])
        if float(tril[1, 0]) != 0.0:
            out[..., 1] = tril[1, 0] * out[..., 0] + tril[1, 1] * out[..., 1]
        else:
            out[..., 1] *= tril[1, 1]
        out[..., 0] *= tril[0, 0]
        if float(loc.abs().max()) != 0.0:
            out += loc
        distortions_u, distortions_e = out.permute(3, 0, 1, 2)
        return distortions_u, distortions_e


class LightSourceArray:
    def __init__(self, light_source_list) -> None:
        self.light_source_list = light_source_list

    @classmethod
    def from_hdf5(cls, config_file, device: torch.device | None = None) -> "LightSourceArray":
        """artist/scene/light_source_array.py:48-98."""
        from . import scenario
        return cls([Sun(number_of_rays=s["number_of_rays"], distribution_parameters=s["distribution_parameters"], device=device)
                    for s in scenario.read_light_sources(config_file)])


class TowerTargetAreasPlanar:
    def __init__(self, names, centers, normals, dimensions) -> None:
        self.names = names
        self.centers = centers
        self.normals = normals
        self.dimensions = dimensions
        self.number_of_target_areas = len(names)

    @classmethod
    def from_hdf5(cls, config_file, device: torch.device | None = None) -> "TowerTargetAreasPlanar":
        """artist/field/tower_target_areas_planar.py:75-143."""
        from . import scenario
        t = scenario.read_planar_target_areas(config_file)
        to = lambda a: scenario._to_device(a, device)  # noqa: E731
        return cls(names=t["names"], centers=to(t["centers"]), normals=to(t["normals"]), dimensions=to(t["dimensions"]))


class TowerTargetAreasCylindrical:
    """artist/field/tower_target_areas_cylindrical.py:52-102 (centre = midpoint of the axis; ``normals`` points to
    the middle of the opening sector)."""

    def __init__(self, names, centers, normals, axes, radii, heights, opening_angles) -> None:
        self.names = names
        self.centers = centers
        self.normals = normals
        self.axes = axes
        self.radii = radii
        self.heights = heights
        self.opening_angles = opening_angles
        self.number_of_target_areas = len(names)

    @classmethod
    def from_hdf5(cls, config_file, device: torch.device | None = None) -> "TowerTargetAreasCylindrical":
        """artist/field/tower_target_areas_cylindrical.py:103-193."""
        from . import scenario
        t = scenario.read_cylindrical_target_areas(config_file)
        to = lambda a: scenario._to_device(a, device)  # noqa: E731
        return cls(names=t["names"], centers=to(t["centers"]), normals=to(t["normals"]), axes=to(t["axes"]),
                   radii=to(t["radii"]), heights=to(t["heights"]), opening_angles=to(t["opening_angles"]))


class _NoCylinders:
    names: list = []
    number_of_target_areas = 0


class SolarTower:
    """Planar target areas first, cylindrical second (artist/field/solar_tower.py:50-100)."""

    def __init__(self, target_areas, device: torch.device | None = None) -> None:
        self.target_areas = list(target_areas)
        if len(self.target_areas) == 1:
            self.target_areas.append(_NoCylinders())
        self.number_of_target_area_types = len(self.target_areas)
        self.number_of_target_areas_per_type = torch.tensor(
            [t.number_of_target_areas for t in self.target_areas], device=device)
        names = [n for t in self.target_areas for n in t.names]
        self.target_name_to_index = {n: i for i, n in enumerate(names)}

    @classmethod
    def from_hdf5(cls, config_file, device: torch.device | None = None) -> "SolarTower":
        """artist/field/solar_tower.py:93-127: planar areas first, cylindrical second."""
        return cls(target_areas=[TowerTargetAreasPlanar.from_hdf5(config_file, device),
                                 TowerTargetAreasCylindrical.from_hdf5(config_file, device)], device=device)

    def get_centers_of_target_areas(self, target_area_indices: torch.Tensor, device=None) -> torch.Tensor:
        """Aim points by GLOBAL target index, planar first, cylindrical second: a planar area's centre; for a cylinder
        the point of its mantle in the middle of the opening sector, centre + radius * normal
        (artist/field/solar_tower.py:129-188)."""
        tables = []
        if self.target_areas[0].number_of_target_areas > 0:
            tables.append(self.target_areas[0].centers)
        cyl = self.target_areas[1]
        if cyl.number_of_target_areas > 0:
            tables.append(cyl.centers + cyl.radii.reshape(-1, 1) * cyl.normals)
        centers = torch.cat(tables)[target_area_indices.long()].clone()
        centers[:, 3] = 1.0
        return centers


def ideal_orientations(positions: torch.Tensor, aim_points: torch.Tensor, incident: torch.Tensor) -> torch.Tensor:
    """``[H,4,4]`` rigid transforms of an ideal two-axis mount: the mirror frame (x east-ish, y along the
    mirror, z = normal) is rotated so that the normal bisects ``-incident`` and the direction to the aim
    point, then translated to the heliostat position.  Stand-in for
    ``RigidBody.incident_ray_directions_to_orientations`` (artist/field/kinematics_rigid_body.py:540-634)
    with zero deviations; O(H) host-side geometry, not part of the hot path."""
    to_aim = torch.nn.functional.normalize(aim_points[:, :3] - positions[:, :3], dim=1)
    n = torch.nn.functional.normalize(-incident[:, :3] + to_aim, dim=1)
    up = torch.tensor([0.0, 0.0, 1.0], device=positions.device, dtype=positions.dtype).expand_as(n)
    x = torch.nn.functional.normalize(torch.linalg.cross(up, n), dim=1)
    y = torch.linalg.cross(n, x)
    m = torch.zeros(positions.shape[0], 4, 4, device=positions.device, dtype=positions.dtype)
    m[:, :3, 0], m[:, :3, 1], m[:, :3, 2] = x, y, n
    m[:, :3, 3] = positions[:, :3]
    m[:, 3, 3] = 1.0
    return m


class HeliostatGroup:
    """SoA tensors of one heliostat group with the attribute names the ray tracer reads
    (artist/field/heliostat_group.py:133-222)."""

    def __init__(self, names, positions, surface_points, surface_normals, canting, facet_translations,
                 nurbs_control_points, nurbs_degrees, device: torch.device | None = None, kinematics=None) -> None:
        self.names = names
        self.kinematics = kinematics        # artist_amd.kinematics.RigidBody, or None = the ideal mount above
        self.number_of_heliostats = len(names)
        self.number_of_facets_per_heliostat = canting.shape[1]
        self.positions = positions
        self.surface_points = surface_points
        self.surface_normals = surface_normals
        self.canting = canting
        self.facet_translations = facet_translations
        self.nurbs_control_points = nurbs_control_points
        self.nurbs_degrees = nurbs_degrees
        self.number_of_active_heliostats = 0
        self.active_heliostats_mask = torch.zeros(self.number_of_heliostats, dtype=torch.int32, device=device)
        self.active_surface_points = torch.empty_like(surface_points)
        self.active_surface_normals = torch.empty_like(surface_normals)
        self.preferred_reflection_directions = torch.empty_like(surface_normals)

    def activate_heliostats(self, active_heliostats_mask: torch.Tensor | None = None, device=None) -> None:
        """artist/field/heliostat_group.py:225-315 (tensor part)."""
        if active_heliostats_mask is None:
            active_heliostats_mask = torch.ones(self.number_of_heliostats, dtype=torch.int32,
                                                device=self.positions.device)
        self.number_of_active_heliostats = int(active_heliostats_mask.sum())
        self.active_heliostats_mask = active_heliostats_mask
        rep = lambda t: t.repeat_interleave(active_heliostats_mask, dim=0)  # noqa: E731
        self.active_surface_points = rep(self.surface_points)
        self.active_surface_normals = rep(self.surface_normals)
        self.active_canting = rep(self.canting)
        self.active_facet_translations = rep(self.facet_translations)
        self.active_nurbs_control_points = rep(self.nurbs_control_points)
        self.active_positions = rep(self.positions)
        kin = self.kinematics
        if kin is not None:                                              # heliostat_group.py:273-315
            kin.number_of_active_heliostats = self.number_of_active_heliostats
            kin.active_heliostat_positions = rep(kin.heliostat_positions)
            kin.active_initial_orientations = rep(kin.initial_orientations)
            kin.active_translation_deviation_parameters = rep(kin.translation_deviation_parameters)
            kin.active_rotation_deviation_parameters = rep(kin.rotation_deviation_parameters)
            kin.active_motor_positions = rep(kin.motor_positions)
            kin.actuators.active_non_optimizable_parameters = rep(kin.actuators.non_optimizable_parameters)
            if kin.actuators.optimizable_parameters.numel() > 0:
                kin.actuators.active_optimizable_parameters = rep(kin.actuators.optimizable_parameters)
            else:
                kin.actuators.active_optimizable_parameters = torch.tensor([], requires_grad=True)

    def _align(self, orientations) -> None:
        self.active_orientations = orientations
        from .ops import align_surfaces
        self.active_surface_points, self.active_surface_normals = align_surfaces(
            self.active_surface_points, self.active_surface_normals, orientations)

    def align_surfaces_with_incident_ray_directions(self, aim_points, incident_ray_directions,
                                                    active_heliostats_mask, device=None) -> None:
        """``points @ orientation^T`` (artist/field/heliostat_group_rigid_body.py:169-222): orientations from the
        rigid-body kinematics (``artist_amd.kinematics.RigidBody``) when the group has one, else the ideal mount."""
        assert torch.equal(self.active_heliostats_mask, active_heliostats_mask), \
            "Some heliostats were not activated and cannot be aligned."
        if self.kinematics is not None:
            orientations = self.kinematics.incident_ray_directions_to_orientations(
                incident_ray_directions=incident_ray_directions, aim_points=aim_points, device=device)
        else:
            orientations = ideal_orientations(self.active_positions, aim_points, incident_ray_directions)
        self._align(orientations)

    def align_surfaces_with_motor_positions(self, motor_positions, active_heliostats_mask, device=None) -> None:
        """artist/field/heliostat_group_rigid_body.py:224-270 - the calibration path."""
        assert torch.equal(self.active_heliostats_mask, active_heliostats_mask), \
            "Some heliostats were not activated and cannot be aligned."
        if self.kinematics is None:
            raise ValueError("aligning with motor positions needs a group with rigid-body kinematics")
        self._align(self.kinematics.motor_positions_to_orientations(motor_positions=motor_positions, device=device))


class HeliostatField:
    def __init__(self, heliostat_groups, device=None) -> None:
        self.heliostat_groups = list(heliostat_groups)

    @classmethod
    def from_hdf5(cls, config_file, prototype_surface=None, prototype_kinematics=None, prototype_actuators=None,
                  number_of_surface_points_per_facet: torch.Tensor = torch.tensor([50, 50]),
                  change_number_of_control_points_per_facet: torch.Tensor | None = None,
                  device: torch.device | None = None) -> "HeliostatField":
        """artist/field/heliostat_field.py:80-435.  The prototypes are the tables ``artist_amd.scenario.read_prototypes``
        returns (``surface``, ``kinematics``, ``actuators``); when none is given they are read from ``config_file``."""
        from . import scenario
        if prototype_surface is None and prototype_kinematics is None and prototype_actuators is None and \
                "prototypes" in config_file.keys():
            prototypes = scenario.read_prototypes(config_file)
            prototype_surface, prototype_kinematics, prototype_actuators = (prototypes["surface"], prototypes["kinematics"],
                                                                            prototypes["actuators"])
        heliostats = scenario.read_heliostats(config_file, prototype_surface, prototype_kinematics, prototype_actuators)
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        return scenario.build_heliostat_field(heliostats, number_of_surface_points_per_facet,
                                              change_number_of_control_points_per_facet, dev)


class Scenario:
    def __init__(self, power_plant_position, solar_tower, light_sources, heliostat_field) -> None:
        self.power_plant_position = power_plant_position
        self.solar_tower = solar_tower
        self.light_sources = light_sources
        self.heliostat_field = heliostat_field


# ----------------------------------------------------------------------------------------------
# Synthetic field: 2x2 facets of 1.605 m x 1.275 m, planar control nets with
# 1e-3 N(0,1) z-noise, 50x50 evaluation points per facet, heliostats on a deterministic fan,
# one 8 m x 8 m planar receiver at (0,0,55) facing north.
# ----------------------------------------------------------------------------------------------
CANTING = [[0.8025, 0.0, 0.0, 0.0], [0.0, 0.6375, 0.0, 0.0]]
FACET_TRANSLATIONS = [[-0.8075, 0.6425, 0.0, 0.0], [0.8075, 0.6425, 0.0, 0.0],
                      [-0.8075, -0.6425, 0.0, 0.0], [0.8075, -0.6425, 0.0, 0.0]]


def fan_positions(n: int, device=None) -> torch.Tensor:
    i = torch.arange(n, dtype=torch.float32, device=device)
    if n == 1:
        e, nn = torch.zeros(1, device=device), torch.full((1,), 60.0, device=device)
    else:
        e = -60.0 + 120.0 * ((i * 0.61803398875) % 1.0)
        nn = 30.0 + 120.0 * i / (n - 1)
    return torch.stack([e, nn, torch.zeros_like(e), torch.ones_like(e)], dim=1)


def synthetic_control_points(n_heliostats: int, n_cp=(10, 10), z_noise: float = 1e-3, seed: int = 7,
                             device=None) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """(control_points [H,4,nu,nv,3], canting [H,4,2,4], facet_translations [H,4,4])."""
    canting = torch.tensor(CANTING, device=device).unsqueeze(0).repeat(4, 1, 1)
    cp = create_planar_nurbs_control_points(torch.tensor(n_cp), canting, device=device)
    cp = cp.unsqueeze(0).repeat(n_heliostats, 1, 1, 1, 1)
    if z_noise:
        g = torch.Generator(device="cpu").manual_seed(seed)
        noise = torch.randn(cp[..., 2].shape, generator=g, dtype=torch.float32)
        cp[..., 2] += z_noise * noise.to(cp.device)
    transl = torch.tensor(FACET_TRANSLATIONS, device=device).unsqueeze(0).repeat(n_heliostats, 1, 1)
    return cp, canting.unsqueeze(0).repeat(n_heliostats, 1, 1, 1), transl


def build_synthetic_scenario(n_heliostats: int, n_rays: int, n_cp=(10, 10), degrees=(3, 3), n_eval: int = 50,
                             z_noise: float = 1e-3, covariance: float = 4.3681e-06, device=None,
                             target_centers=((0.0, 0.0, 55.0, 1.0),), target_normals=((0.0, 1.0, 0.0, 0.0),),
                             target_dims=((8.0, 8.0),)):
    """Scenario stand-in + evaluation grid.  Surface points/normals are evaluated once with the HIP
    NURBS kernel, like ``HeliostatField.from_hdf5`` does at load time
    (artist/field/heliostat_field.py:328 -> artist/field/surface.py:61)."""
    device = torch.device("cuda") if device is None else torch.device(device)
    cp, canting, transl = synthetic_control_points(n_heliostats, n_cp, z_noise, device=device)
    deg = torch.tensor(degrees)
    uv = create_nurbs_evaluation_grid(torch.tensor([n_eval, n_eval]), device=device)
    uv_full = uv[None, None].expand(n_heliostats, 4, -1, -1)
    with torch.no_grad():
        pts, nrm = NURBSSurfaces(deg, cp, device=device).calculate_surface_points_and_normals(uv_full, canting, transl)
    P = 4 * uv.shape[0]
    group = HeliostatGroup(
        names=[f"h{i}" for i in range(n_heliostats)], positions=fan_positions(n_heliostats, device),
        surface_points=pts.reshape(n_heliostats, P, 4), surface_normals=nrm.reshape(n_heliostats, P, 4),
        canting=canting, facet_translations=transl, nurbs_control_points=cp, nurbs_degrees=deg, device=device)
    planar = TowerTargetAreasPlanar(
        names=[f"receiver_{i}" for i in range(len(target_centers))],
        centers=torch.tensor(target_centers, device=device), normals=torch.tensor(target_normals, device=device),
        dimensions=torch.tensor(target_dims, device=device))
    scenario = Scenario(
        power_plant_position=torch.tensor([50.91, 6.39, 87.0]), solar_tower=SolarTower([planar], device=device),
        light_sources=LightSourceArray([Sun(n_rays, dict(covariance=covariance), device=device)]),
        heliostat_field=HeliostatField([group], device=device))
    return scenario, uv_full
