"""ARTIST scenario files (HDF5) -> a heliostat field resident on the GPU.

Mirrors ``Scenario.load_scenario_from_hdf5`` (artist/scenario/scenario.py:104-259) with the parsers it calls
(artist/io/h5_scenario_parser.py:13-723, ``HeliostatField.from_hdf5`` artist/field/heliostat_field.py:80-435,
``SolarTower.from_hdf5`` artist/field/solar_tower.py:93-127, target areas, ``LightSourceArray.from_hdf5``,
``Sun.from_hdf5``): same file layout, same defaults for missing entries, same grouping of heliostats by
(kinematics type, actuator type), same tensor shapes and attribute names on the result.  The file is read on the
host once (``read_scenario_tables`` - pure numpy, no GPU needed); the surfaces of a whole group are then sampled
in one NURBS kernel launch (the reference evaluates one heliostat at a time, twice).

``scenario_file`` may be an ``h5py.File`` or the built-in reader's ``artist_amd.h5lite.File``
(``open_scenario_file`` picks h5py when it is installed).
"""
from __future__ import annotations

import logging
from collections import defaultdict

import numpy as np
import torch

from . import h5lite, scene
from .kinematics import RigidBody
from .nurbs import NURBSSurfaces, create_nurbs_evaluation_grid, create_planar_nurbs_control_points

log = logging.getLogger(__name__)

__all__ = ["Scenario", "open_scenario_file", "read_scenario_tables", "load_scenario_from_hdf5"]

# artist/util/indices.py:21-46 - slot of each named deviation in the [9] translation / [4] rotation tables
TRANSLATION_DEVIATIONS = ["first_joint_translation_e", "first_joint_translation_n", "first_joint_translation_u",
                          "second_joint_translation_e", "second_joint_translation_n", "second_joint_translation_u",
                          "concentrator_translation_e", "concentrator_translation_n", "concentrator_translation_u"]
ROTATION_DEVIATIONS = ["first_joint_tilt_n", "first_joint_tilt_u", "second_joint_tilt_e", "second_joint_tilt_n"]
LINEAR_ACTUATOR_INT, IDEAL_ACTUATOR_INT = 0, 1          # artist/util/constants.py:124-128
RIGID_BODY_ACTUATORS = 2


def open_scenario_file(path):
    """``h5py.File(path, "r")`` when h5py is installed, else the built-in reader."""
    try:
        import h5py
        return h5py.File(path, "r")
    except ImportError:
        return h5lite.File(str(path), "r")


def _text(dataset) -> str:
    value = dataset[()]
    return value.decode("utf-8") if isinstance(value, bytes) else str(value)


def _f32(dataset) -> np.ndarray:
    return np.asarray(dataset[()], dtype=np.float32)


def _optional(group, path, what, owner):
    """``group.get(path)`` as float32, 0 with the reference's warning when it is missing
    (h5_scenario_parser.py:223-301, 540-575)."""
    node = group.get(path)
    if node is None:
        log.warning(f"No individual {what} for {owner} set. Using default values!")
        return np.float32(0.0)
    return np.float32(node[()])


def _surface(facets) -> dict:
    """h5_scenario_parser.surface_config (:13-75): per-facet control points, degrees, translation, canting."""
    keys = list(facets.keys())
    return dict(control_points=np.stack([_f32(facets[k]["control_points"]) for k in keys]),
                degrees=np.stack([np.asarray(facets[k]["degrees"][()], dtype=np.int32)[:2] for k in keys]),
                translations=np.stack([_f32(facets[k]["position"]) for k in keys]),
                canting=np.stack([_f32(facets[k]["canting"]) for k in keys]))


def _kinematics(config, owner) -> dict:
    """Initial orientation, type and rigid-body deviations (h5_scenario_parser.py:78-393)."""
    kind = _text(config["type"])
    if kind != "rigid_body":
        raise ValueError(f"The kinematics type: {kind} is not yet implemented!")
    return dict(type=kind, initial_orientation=_f32(config["initial_orientation"]),
                translation=np.array([_optional(config, f"deviations/{k}", f"kinematics {k}", owner)
                                      for k in TRANSLATION_DEVIATIONS], dtype=np.float32),
                rotation=np.array([_optional(config, f"deviations/{k}", f"kinematics {k}", owner)
                                   for k in ROTATION_DEVIATIONS], dtype=np.float32))


def _actuators(config, owner) -> dict:
    """Type + the [4,2] / [7,2] non-optimisable and [2,2] optimisable tables (h5_scenario_parser.py:396-723)."""
    keys = list(config.keys())
    types = [_text(config[k]["type"]) for k in keys]
    if not types:
        raise ValueError("Prototype actuator type list is empty.")
    if len(set(types)) > 1:
        raise ValueError("When using the rigid body kinematics, all actuators for a given heliostat must have the same type.")
    kind = types[0]
    if kind not in ("linear", "ideal"):
        raise ValueError(f"The actuator type: {kind} is not yet implemented!")
    if len(keys) != RIGID_BODY_ACTUATORS:
        raise ValueError("This scenario file contains the wrong amount of actuators for this heliostat and its kinematics type."
                         f" Expected {RIGID_BODY_ACTUATORS} actuators, found {len(keys)} actuator(s).")
    linear = kind == "linear"
    nonopt = np.zeros((7 if linear else 4, len(keys)), dtype=np.float32)
    opt = np.zeros((2, len(keys)), dtype=np.float32) if linear else np.zeros((0,), dtype=np.float32)
    for i, key in enumerate(keys):
        limits = config[key]["min_max_motor_positions"][()]
        nonopt[0, i] = LINEAR_ACTUATOR_INT if linear else IDEAL_ACTUATOR_INT
        nonopt[1, i] = 1 if bool(config[key]["clockwise_axis_movement"][()]) else 0
        nonopt[2, i], nonopt[3, i] = float(limits[0]), float(limits[1])
        if linear:
            get = lambda name: _optional(config, f"{key}/parameters/{name}", f"{name} set for {key} on", owner)  # noqa: E731
            nonopt[4, i], nonopt[5, i], nonopt[6, i] = get("increment"), get("offset"), get("pivot_radius")
            opt[0, i], opt[1, i] = get("initial_angle"), get("initial_stroke_length")
    if linear:
        # the stored initial angle refers to a surface facing up; the kinematics' reference pose faces south: + the east
        # component of the axis-angle rotation (0,-1,0) -> (0,0,1), i.e. -acos(0) in fp32 (:697-708, rotations.py:67-119)
        opt[0, 0] += np.float32(-1.0) * torch.acos(torch.zeros((), dtype=torch.float32)).numpy()
    return dict(type=kind, non_optimizable=nonopt, optimizable=opt)


def read_planar_target_areas(config_file) -> dict:
    """artist/field/tower_target_areas_planar.py:75-143 (names in sorted order)."""
    planar = config_file["target_areas_planar"]
    names = sorted(planar.keys())
    return dict(names=names,
                centers=np.stack([_f32(planar[k]["position_center"]).reshape(4) for k in names]).reshape(-1, 4),
                normals=np.stack([_f32(planar[k]["normal_vector"]).reshape(4) for k in names]).reshape(-1, 4),
                dimensions=np.array([[float(planar[k]["plane_e"][()]), float(planar[k]["plane_u"][()])] for k in names],
                                    dtype=np.float32).reshape(-1, 2))


def read_cylindrical_target_areas(config_file) -> dict:
    """artist/field/tower_target_areas_cylindrical.py:103-193."""
    cyl = config_file["target_areas_cylindrical"]
    names = sorted(cyl.keys())
    vec = lambda key: (np.stack([_f32(cyl[k][key]).reshape(4) for k in names]).reshape(-1, 4) if names  # noqa: E731
                       else np.zeros((0, 4), np.float32))
    sca = lambda key: np.array([np.float32(cyl[k][key][()]) for k in names], dtype=np.float32)  # noqa: E731
    return dict(names=names, centers=vec("cylinder_center"), normals=vec("cylinder_normal"), axes=vec("cylinder_axis"),
                radii=sca("cylinder_radius"), heights=sca("cylinder_height"), opening_angles=sca("cylinder_opening_angle"))


def read_light_source(config, name=None) -> dict:
    """One light source group (artist/scene/sun.py:121-197)."""
    kind = _text(config["type"])
    if kind != "sun":
        raise KeyError(f"Currently the selected light source: {kind} is not supported.")
    params = dict(distribution_type=_text(config["distribution_parameters"]["distribution_type"]))
    for key in ("mean", "covariance"):
        if key in config["distribution_parameters"].keys():
            params[key] = float(config["distribution_parameters"][key][()])
    return dict(name=name, number_of_rays=int(config["number_of_rays"][()]), distribution_parameters=params)


def read_light_sources(config_file) -> list:
    """artist/scene/light_source_array.py:48-98 (sorted by name)."""
    return [read_light_source(config_file["lightsources"][key], key) for key in sorted(config_file["lightsources"].keys())]


def read_prototypes(config_file) -> dict:
    """The prototype surface, kinematics and actuators (scenario.py:159-236)."""
    proto = config_file["prototypes"]
    return dict(surface=_surface(proto["surface"]["facets"]), kinematics=_kinematics(proto["kinematics"], None),
                actuators=_actuators(proto["actuator"], None))


def read_heliostats(config_file, prototype_surface=None, prototype_kinematics=None, prototype_actuators=None) -> list:
    """Per-heliostat tables, prototypes filled in where a heliostat has no entry of its own
    (artist/field/heliostat_field.py:137-262, same errors)."""
    heliostats = []
    for name in config_file["heliostats"].keys():
        cfg = config_file["heliostats"][name]
        keys = list(cfg.keys())
        if "surface" not in keys and prototype_surface is None:
            raise ValueError("If the heliostat does not have individual surface parameters, a surface prototype must be provided!")
        if "kinematics" not in keys and prototype_kinematics is None:
            raise ValueError("If the heliostat does not have an individual kinematics, a kinematics prototype must be provided!")
        if "actuator" not in keys and prototype_actuators is None:
            raise ValueError("If the heliostat does not have individual actuators, an actuator prototype must be provided!")
        heliostats.append(dict(
            name=name, position=_f32(cfg["position"]),
            surface=_surface(cfg["surface"]["facets"]) if "surface" in keys else prototype_surface,
            kinematics=_kinematics(cfg["kinematics"], name) if "kinematics" in keys else prototype_kinematics,
            actuators=_actuators(cfg["actuator"], name) if "actuator" in keys else prototype_actuators))
    return heliostats


def read_scenario_tables(scenario_file) -> dict:
    """Everything a scenario file holds, as numpy arrays on the host (no GPU needed)."""
    out = dict(version=scenario_file.attrs.get("version") if hasattr(scenario_file, "attrs") else None)
    out["power_plant_position"] = np.asarray(scenario_file["power_plant"]["position"][()], dtype=np.float64)
    out["planar"] = read_planar_target_areas(scenario_file)
    out["cylindrical"] = read_cylindrical_target_areas(scenario_file)
    out["light_sources"] = read_light_sources(scenario_file)
    prototypes = read_prototypes(scenario_file)
    out["heliostats"] = read_heliostats(scenario_file, prototypes["surface"], prototypes["kinematics"], prototypes["actuators"])
    if "number_of_heliostat_groups" in scenario_file.keys():
        out["number_of_heliostat_groups"] = int(scenario_file["number_of_heliostat_groups"][()])
    return out


def _to_device(array, device) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(array)).to(device)


def build_solar_tower(planar: dict, cylindrical: dict, device) -> "scene.SolarTower":
    t = lambda a: _to_device(a, device)  # noqa: E731
    return scene.SolarTower(target_areas=[
        scene.TowerTargetAreasPlanar(names=planar["names"], centers=t(planar["centers"]), normals=t(planar["normals"]),
                                     dimensions=t(planar["dimensions"])),
        scene.TowerTargetAreasCylindrical(names=cylindrical["names"], centers=t(cylindrical["centers"]),
                                          normals=t(cylindrical["normals"]), axes=t(cylindrical["axes"]),
                                          radii=t(cylindrical["radii"]), heights=t(cylindrical["heights"]),
                                          opening_angles=t(cylindrical["opening_angles"]))], device=device)


def build_heliostat_field(heliostats: list, number_of_surface_points_per_facet, change_number_of_control_points_per_facet,
                          device) -> "scene.HeliostatField":
    """Group by (kinematics, actuator) type in order of first appearance and put each group on the GPU
    (artist/field/heliostat_field.py:263-435)."""
    t = lambda a: _to_device(a, device)  # noqa: E731
    grouped = defaultdict(list)
    for h in heliostats:
        grouped[f"{h['kinematics']['type']}_{h['actuators']['type']}"].append(h)
    groups = []
    for key, members in grouped.items():
        points, normals, canting, translations, control_points, degrees = _sample_surfaces(
            members, number_of_surface_points_per_facet, change_number_of_control_points_per_facet, device)
        positions = t(np.stack([m["position"] for m in members]))
        initial_orientations = t(np.stack([m["kinematics"]["initial_orientation"] for m in members]))
        kinematics = RigidBody(
            number_of_heliostats=len(members), heliostat_positions=positions, initial_orientations=initial_orientations,
            translation_deviation_parameters=t(np.stack([m["kinematics"]["translation"] for m in members])),
            rotation_deviation_parameters=t(np.stack([m["kinematics"]["rotation"] for m in members])),
            actuator_parameters_non_optimizable=t(np.stack([m["actuators"]["non_optimizable"] for m in members])),
            actuator_parameters_optimizable=t(np.stack([m["actuators"]["optimizable"] for m in members])), device=device)
        group = scene.HeliostatGroup(names=[m["name"] for m in members], positions=positions, surface_points=points,
                                     surface_normals=normals, canting=canting, facet_translations=translations,
                                     nurbs_control_points=control_points, nurbs_degrees=degrees, device=device,
                                     kinematics=kinematics)
        group.initial_orientations = initial_orientations
        group.group_type = key
        groups.append(group)
        log.info(f"Added a heliostat group with kinematics type: {members[0]['kinematics']['type']}, and actuator type: "
                 f"{members[0]['actuators']['type']}, to the heliostat field.")
    return scene.HeliostatField(heliostat_groups=groups, device=device)


def _sample_surfaces(members, number_of_surface_points_per_facet, change_control_points, device):
    """Surface points / normals ``[H, F*M, 4]`` of the heliostats of one group.  Heliostats whose control nets are
    flat (all z = 0) are canted and translated by the NURBS kernel, fitted ones carry both already
    (artist/field/surface.py:88-121); each class is one launch."""
    dev = torch.device(device)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    canting = torch.stack([t(m["surface"]["canting"]) for m in members])
    translations = torch.stack([t(m["surface"]["translations"]) for m in members])
    if change_control_points is not None:
        control_points = torch.stack([create_planar_nurbs_control_points(
            number_of_control_points=change_control_points, canting=c, device=dev) for c in canting])
    else:
        control_points = torch.stack([t(m["surface"]["control_points"]) for m in members])
    grid = create_nurbs_evaluation_grid(number_of_evaluation_points=number_of_surface_points_per_facet, device=dev)
    H, F = control_points.shape[:2]
    points = torch.empty((H, F, grid.shape[0], 4), dtype=torch.float32, device=dev)
    normals = torch.empty_like(points)
    flat = (control_points[..., 2] == 0).flatten(1).all(dim=1)
    first_degrees = torch.stack([t(m["surface"]["degrees"][0]) for m in members])
    for is_flat in (True, False):
        for degrees in torch.unique(first_degrees, dim=0):
            sel = torch.nonzero((flat == is_flat) & (first_degrees == degrees).all(dim=1)).flatten()
            if sel.numel() == 0:
                continue
            surfaces = NURBSSurfaces(degrees=degrees, control_points=control_points[sel], device=dev)
            evaluation_points = grid[None, None].expand(sel.numel(), F, -1, -1)
            p, n = surfaces.calculate_surface_points_and_normals(
                evaluation_points=evaluation_points, canting=canting[sel] if is_flat else None,
                facet_translations=translations[sel] if is_flat else None, device=dev)
            points[sel], normals[sel] = p.detach(), n.detach()
    # the group's degrees are those of the LAST facet read (heliostat_field.py:264-270)
    group_degrees = t(members[-1]["surface"]["degrees"][-1])
    return points.reshape(H, -1, 4), normals.reshape(H, -1, 4), canting, translations, control_points, group_degrees


class Scenario(scene.Scenario):
    """``artist.scenario.scenario.Scenario``: ``load_scenario_from_hdf5``, ``index_mapping``, ``set_number_of_rays``."""

    @staticmethod
    def get_number_of_heliostat_groups_from_hdf5(scenario_path) -> int:
        """scenario.py:83-101."""
        with open_scenario_file(scenario_path) as scenario_file:
            return int(scenario_file["number_of_heliostat_groups"][()])

    @classmethod
    def load_scenario_from_hdf5(cls, scenario_file, number_of_surface_points_per_facet: torch.Tensor = torch.tensor([50, 50]),
                                change_number_of_control_points_per_facet: torch.Tensor | None = None,
                                device: torch.device | None = None) -> "Scenario":
        """scenario.py:104-259.  ``device`` must be a GPU: the surfaces are sampled by the NURBS kernel."""
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        tables = read_scenario_tables(scenario_file)
        log.info(f"Loading an ARTIST scenario HDF5 file. This scenario file is version {tables['version']}.")
        solar_tower = build_solar_tower(tables["planar"], tables["cylindrical"], dev)
        light_sources = scene.LightSourceArray([scene.Sun(number_of_rays=s["number_of_rays"],
                                                          distribution_parameters=s["distribution_parameters"], device=dev)
                                                for s in tables["light_sources"]])
        heliostat_field = build_heliostat_field(tables["heliostats"], number_of_surface_points_per_facet,
                                                change_number_of_control_points_per_facet, dev)
        return cls(power_plant_position=torch.tensor(tables["power_plant_position"], dtype=torch.float64, device=dev),
                   solar_tower=solar_tower, light_sources=light_sources, heliostat_field=heliostat_field)

    def index_mapping(self, heliostat_group, string_mapping=None, single_incident_ray_direction: torch.Tensor | None = None,
                      single_target_area_index: int = 0, device: torch.device | None = None):
        """(active_heliostats_mask, target_area_indices, incident_ray_directions) of one group from a list of
        (heliostat name, target name, incident direction) - scenario.py:261-418, same validation and messages."""
        dev = heliostat_group.positions.device if device is None else torch.device(device)
        if single_incident_ray_direction is None:
            single_incident_ray_direction = torch.tensor([0.0, 1.0, 0.0, 0.0], device=dev)
        else:
            single_incident_ray_direction = single_incident_ray_direction.to(dev)
        names = list(heliostat_group.names)
        number_of_target_areas = len(self.solar_tower.target_name_to_index)

        def unit_direction(v, tol_w, tol_n, three_only):
            if tuple(v.shape) != (4,):
                return False
            norm = torch.norm(v[:3] if three_only else v)
            return abs(float(v[3])) <= tol_w[0] and abs(float(norm) - 1.0) <= tol_n[0] + tol_n[1]

        if string_mapping is None:
            if not unit_direction(single_incident_ray_direction, (1e-8,), (1e-8, 1e-5), True):
                raise ValueError("The specified single incident ray direction is invalid. Please provide a normalized 4D "
                                 "tensor with last element 0.0.")
            if single_target_area_index >= number_of_target_areas:
                raise ValueError(f"The specified single target area index is invalid. Only {number_of_target_areas} target "
                                 "areas exist in this scenario.")
            mask = torch.ones(len(names), dtype=torch.int32, device=dev)
            targets = torch.full((len(names),), single_target_area_index, dtype=torch.int32, device=dev)
            return mask, targets, single_incident_ray_direction.expand(len(names), -1).to(dev)

        filtered = [m for m in string_mapping if m[0] in names]
        errors = []
        for i, (_, target_name, direction) in enumerate(filtered):
            if target_name not in self.solar_tower.target_name_to_index:
                errors.append(f"Invalid target '{target_name}' (Found at index {i} of provided mapping) not found in this scenario.")
            if not unit_direction(direction, (1e-2,), (1e-4, 1e-4), False):
                errors.append(f"Invalid incident ray direction (Found at index {i} of provided mapping). This must be a "
                              "normalized 4D tensor with last element 0.0.")
        if errors:
            raise ValueError(" ".join(errors))
        per_heliostat = defaultdict(list)
        mask = torch.zeros(len(names), dtype=torch.int32)
        for heliostat_name, target_name, direction in filtered:
            mask[names.index(heliostat_name)] += 1
            per_heliostat[heliostat_name].append((self.solar_tower.target_name_to_index[target_name], direction))
        targets, directions = [], []
        for name in names:                                    # rows in group order, repeats adjacent (repeat_interleave)
            for target_index, direction in per_heliostat.get(name, []):
                targets.append(target_index)
                directions.append(direction.to(torch.float32).cpu())
        incident = torch.stack(directions).to(dev) if directions else torch.empty((0, 4), device=dev)
        return mask.to(dev), torch.tensor(targets, dtype=torch.int32, device=dev), incident

    def set_number_of_rays(self, number_of_rays: int) -> None:
        """scenario.py:420-430."""
        self.light_sources.light_source_list[0].number_of_rays = number_of_rays

    def __repr__(self) -> str:
        return (f"ARTIST Scenario containing:\n\tA Power Plant located at: {self.power_plant_position.tolist()}"
                f" with {len(self.solar_tower.target_name_to_index)} Target Area(s),"
                f" {len(self.light_sources.light_source_list)} Light Source(s),"
                f" and {sum(len(group.names) for group in self.heliostat_field.heliostat_groups)} Heliostat(s).")


load_scenario_from_hdf5 = Scenario.load_scenario_from_hdf5
