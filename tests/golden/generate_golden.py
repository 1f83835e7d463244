#!/usr/bin/env python3
"""Golden-vector generator for the ARTIST hot path (runs ONLY in the build container).

TEST INFRASTRUCTURE - not part of the product.  This script imports the *reference*
(ARTIST v2.0.0, read-only at /root/reference), drives it through its public constructors
on small synthetic scenarios, and writes the inputs + outputs of every stage of the hot
path as ``.npz`` fixtures next to this file.  The fixtures (pure data) travel to the GPU
box; the reference does not.

Import recipe: four packages that the reference imports at module
scope but that are absent from this image (colorlog, h5py, torchvision, paint) are
replaced by empty module objects - they are only used for logging colours, file IO and
type annotations, none of which the hot path touches.  ``artist.field`` must be imported
first (circular import otherwise) and the CWD must be a scratch dir because importing
``artist.util`` creates ./runtime_log.txt.

Usage:  python tests/golden/generate_golden.py            (writes tests/golden/*.npz)

Stages captured per scenario (reference file:line each array comes from):
  nurbs points/normals     artist/nurbs/surfaces.py:475-689
  orientation [H,4,4]      artist/field/kinematics_rigid_body.py:540-634
  aligned points/normals   artist/field/heliostat_group_rigid_body.py:217-222
  reflected                artist/raytracing/geometry.py:11-41
  scattered directions     artist/raytracing/heliostat_ray_tracer.py:510-561
  e_px,u_px,t,intensity    artist/raytracing/geometry.py:44-204
  bitmaps                  artist/raytracing/heliostat_ray_tracer.py:610-778
  flux + 3 factors         artist/raytracing/heliostat_ray_tracer.py:220-508
  per-target sums          artist/raytracing/heliostat_ray_tracer.py:563-608
  autograd gradients       torch.autograd through all of the above
"""
from __future__ import annotations

import os
import pathlib
import sys
import tempfile
import types

OUT_DIR = pathlib.Path(__file__).resolve().parent
REFERENCE = "/root/reference"


def _import_reference():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Fmt:  # colorlog.ColoredFormatter stand-in (logging colours only)
        def __init__(self, *a, **k):
            pass

    stub("colorlog", ColoredFormatter=_Fmt)
    # h5py is absent: artist_amd/h5lite.py reads the reference's own scenario files (superblock-0 HDF5)
    sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
    from artist_amd import h5lite
    stub("h5py", File=h5lite.File, Group=h5lite.Group, Dataset=h5lite.Dataset)
    stub("torchvision")
    stub("torchvision.transforms")
    stub("paint")
    stub("paint.util")
    pm = stub("paint.util.paint_mappings")
    pm.__getattr__ = lambda n: n
    sys.path.insert(0, REFERENCE)
    os.chdir(tempfile.mkdtemp(prefix="artist_ref_cwd_"))
    import artist.field  # noqa: F401  (must come first)


_import_reference()

import numpy as np  # noqa: E402
import torch  # noqa: E402
from artist.field import (  # noqa: E402
    HeliostatField,
    HeliostatGroupRigidBody,
    SolarTower,
    TowerTargetAreasCylindrical,
    TowerTargetAreasPlanar,
)
from artist.geometry import transforms  # noqa: E402
from artist.nurbs import NURBSSurfaces  # noqa: E402
from artist.nurbs.utils import (  # noqa: E402
    create_nurbs_evaluation_grid,
    create_planar_nurbs_control_points,
)
from artist.raytracing import blocking as ref_blocking  # noqa: E402
from artist.raytracing import geometry  # noqa: E402
from artist.raytracing.heliostat_ray_tracer import HeliostatRayTracer  # noqa: E402
from artist.raytracing.sampling import RestrictedDistributedSampler  # noqa: E402
from artist.scenario.scenario import Scenario  # noqa: E402
from artist.scene import LightSourceArray, Rays, Sun  # noqa: E402

CPU = torch.device("cpu")


def npy(t):
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().numpy()
    return np.asarray(t)


# --------------------------------------------------------------------------------------
# Synthetic scenario through the reference's public constructors.
# --------------------------------------------------------------------------------------
CANTING = [[0.8025, 0.0, 0.0, 0.0], [0.0, 0.6375, 0.0, 0.0]]
FACET_TRANSLATIONS = [
    [-0.8075, 0.6425, 0.0, 0.0],
    [0.8075, 0.6425, 0.0, 0.0],
    [-0.8075, -0.6425, 0.0, 0.0],
    [0.8075, -0.6425, 0.0, 0.0],
]


def fan_positions(n, dtype):
    """Deterministic fan: e in [-60,60] m, n in [30,150] m, u = 0."""
    i = torch.arange(n, dtype=dtype)
    if n == 1:
        e = torch.zeros(1, dtype=dtype)
        nn = torch.full((1,), 60.0, dtype=dtype)
    else:
        e = -60.0 + 120.0 * ((i * 0.61803398875) % 1.0)
        nn = 30.0 + 120.0 * i / (n - 1)
    return torch.stack([e, nn, torch.zeros_like(e), torch.ones_like(e)], dim=1)


def build(case, dtype=torch.float32):
    """Build reference objects for a case description dict; return everything needed."""
    torch.set_default_dtype(dtype)
    n_hel = case["n_heliostats"]
    F = 4
    nu, nv = case["n_cp"]
    p, q = case["degrees"]
    npts = case["n_eval"]
    canting = torch.tensor(CANTING, dtype=dtype).unsqueeze(0).repeat(F, 1, 1)
    translations = torch.tensor(FACET_TRANSLATIONS, dtype=dtype)

    cp = create_planar_nurbs_control_points(
        torch.tensor([nu, nv]), canting, device=CPU
    )  # [F,nu,nv,3]
    cp = cp.unsqueeze(0).repeat(n_hel, 1, 1, 1, 1).clone()
    torch.manual_seed(7)
    cp[..., 2] += case.get("z_noise", 1e-3) * torch.randn(cp[..., 2].shape, dtype=torch.float32).to(dtype)
    if case.get("curvature", 0.0):
        # mild paraboloid so that normals vary smoothly (focusing mirror)
        cp[..., 2] += case["curvature"] * (cp[..., 0] ** 2 + cp[..., 1] ** 2)

    uv = create_nurbs_evaluation_grid(torch.tensor([npts, npts]), device=CPU)  # [M,2]
    uv_full = uv[None, None].expand(n_hel, F, -1, -1).contiguous()
    canting_h = canting.unsqueeze(0).repeat(n_hel, 1, 1, 1)
    transl_h = translations.unsqueeze(0).repeat(n_hel, 1, 1)
    degrees = torch.tensor([p, q])

    nurbs = NURBSSurfaces(degrees=degrees, control_points=cp, device=CPU)
    pts, nrm = nurbs.calculate_surface_points_and_normals(
        evaluation_points=uv_full, canting=canting_h, facet_translations=transl_h, device=CPU
    )
    P = F * uv.shape[0]
    surface_points = pts.reshape(n_hel, P, 4).detach()
    surface_normals = nrm.reshape(n_hel, P, 4).detach()

    positions = fan_positions(n_hel, dtype)
    if "positions" in case:
        positions = torch.tensor(case["positions"], dtype=dtype)
    group = HeliostatGroupRigidBody(
        names=[f"h{i}" for i in range(n_hel)],
        positions=positions,
        surface_points=surface_points,
        surface_normals=surface_normals,
        canting=canting_h,
        facet_translations=transl_h,
        initial_orientations=torch.tensor([[0.0, -1.0, 0.0, 0.0]], dtype=dtype).repeat(n_hel, 1),
        nurbs_control_points=cp,
        nurbs_degrees=degrees,
        kinematics_translation_deviation_parameters=torch.zeros(n_hel, 9, dtype=dtype),
        kinematics_rotation_deviation_parameters=torch.zeros(n_hel, 4, dtype=dtype),
        actuator_parameters_non_optimizable=torch.tensor(
            [[1.0, 1.0], [0.0, 0.0], [-10.0, -10.0], [10.0, 10.0]], dtype=dtype
        ).unsqueeze(0).repeat(n_hel, 1, 1),
        device=CPU,
    )
    planar = TowerTargetAreasPlanar(
        names=[f"t{i}" for i in range(len(case["target_centers"]))],
        centers=torch.tensor(case["target_centers"], dtype=dtype),
        normals=torch.tensor(case["target_normals"], dtype=dtype),
        dimensions=torch.tensor(case["target_dims"], dtype=dtype),
    )
    cc = case.get("cyl_centers", [])
    cyl = TowerTargetAreasCylindrical(
        names=[f"c{i}" for i in range(len(cc))],
        centers=torch.tensor(cc, dtype=dtype).reshape(-1, 4),
        normals=torch.tensor(case.get("cyl_normals", []), dtype=dtype).reshape(-1, 4),
        axes=torch.tensor(case.get("cyl_axes", []), dtype=dtype).reshape(-1, 4),
        radii=torch.tensor(case.get("cyl_radii", []), dtype=dtype),
        heights=torch.tensor(case.get("cyl_heights", []), dtype=dtype),
        opening_angles=torch.tensor(case.get("cyl_opening", []), dtype=dtype),
    )
    tower = SolarTower([planar, cyl], device=CPU)
    sun = Sun(
        number_of_rays=case["n_rays"],
        distribution_parameters=dict(
            distribution_type="normal", mean=0.0, covariance=case.get("covariance", 4.3681e-06)
        ),
        device=CPU,
    )
    scenario = Scenario(
        power_plant_position=torch.tensor([50.91, 6.39, 87.0], dtype=dtype),
        solar_tower=tower,
        light_sources=LightSourceArray([sun]),
        heliostat_field=HeliostatField([group], device=CPU),
    )
    return dict(
        scenario=scenario, group=group, nurbs_inputs=(degrees, cp, uv_full, canting_h, transl_h),
        planar=planar, cyl=cyl, sun=sun, P=P,
    )


def run_case(name, case, with_grads=True, store_rays=True, dtype=torch.float32, distortions_f32=None):
    """Run the reference on one case; return dict of arrays."""
    b = build(case, dtype)
    scenario, group = b["scenario"], b["group"]
    degrees, cp, uv_full, canting_h, transl_h = b["nurbs_inputs"]
    out = {}
    mask = torch.tensor(case.get("active_mask", [1] * case["n_heliostats"]), dtype=torch.int32)
    H = int(mask.sum())
    target_idx = torch.tensor(case.get("target_idx", [0] * H), dtype=torch.int64)
    incident = torch.tensor(case.get("incident", [[0.0, 1.0, 0.0, 0.0]] * H), dtype=dtype)
    incident = torch.nn.functional.normalize(incident, dim=1)
    res = torch.tensor(case.get("resolution", [256, 256]))

    group.activate_heliostats(active_heliostats_mask=mask, device=CPU)
    # --- K1 in graph (surface_reconstructor.py:510-543 pattern) ---
    cp_active = group.active_nurbs_control_points.detach().clone().requires_grad_(with_grads)
    nurbs = NURBSSurfaces(degrees=degrees, control_points=cp_active, device=CPU)
    uv_a = uv_full.repeat_interleave(mask, dim=0)
    cant_a = group.active_canting
    tr_a = group.active_facet_translations
    pts, nrm = nurbs.calculate_surface_points_and_normals(
        evaluation_points=uv_a, canting=cant_a, facet_translations=tr_a, device=CPU
    )
    out.update(
        degrees=npy(degrees), control_points=npy(cp_active), eval_points=npy(uv_a),
        canting=npy(cant_a), facet_translations=npy(tr_a),
        nurbs_points=npy(pts), nurbs_normals=npy(nrm),
        knots_u=npy(nurbs.knot_vectors_u[0, 0]), knots_v=npy(nurbs.knot_vectors_v[0, 0]),
    )
    P = b["P"]
    sp = pts.reshape(H, P, 4)
    sn = nrm.reshape(H, P, 4)
    if with_grads:
        sp.retain_grad()
        sn.retain_grad()

    # --- K2 alignment: same ops as heliostat_group_rigid_body.py:210-222, kept explicit so that
    #     the orientation matrices can be captured and differentiated ---
    aim = scenario.solar_tower.get_centers_of_target_areas(target_area_indices=target_idx, device=CPU)
    if "aim_offset" in case:
        aim = aim + torch.tensor(case["aim_offset"], dtype=dtype)
    with torch.no_grad():
        orientation = group.kinematics.incident_ray_directions_to_orientations(
            incident_ray_directions=incident, aim_points=aim, device=CPU
        )
    orientation = orientation.detach().clone().requires_grad_(with_grads)
    if with_grads:
        # kinematic parameters: d(orientation)/d(rotation / translation deviations) of the reference's rigid-body
        # kinematics (artist/field/kinematics_rigid_body.py:194-634), per heliostat - the chain by which the
        # orientation gradient reaches the parameters the kinematics reconstructor optimises
        kin = group.kinematics
        rot0 = kin.active_rotation_deviation_parameters.detach().clone()
        trans0 = kin.active_translation_deviation_parameters.detach().clone()

        def _orientations(rot, trans):
            kin.active_rotation_deviation_parameters, kin.active_translation_deviation_parameters = rot, trans
            return kin.incident_ray_directions_to_orientations(incident_ray_directions=incident, aim_points=aim, device=CPU)

        jr, jt = torch.autograd.functional.jacobian(_orientations, (rot0, trans0))      # [H,4,4,H,4], [H,4,4,H,9]
        hh = torch.arange(H)
        out.update(kin_rot_params=npy(rot0), kin_trans_params=npy(trans0),
                   kin_jac_rot=npy(jr[hh, :, :, hh]), kin_jac_trans=npy(jt[hh, :, :, hh]))
        kin.active_rotation_deviation_parameters, kin.active_translation_deviation_parameters = rot0, trans0
    apts = sp @ orientation.transpose(1, 2)
    anrm = sn @ orientation.transpose(1, 2)
    if with_grads:
        apts.retain_grad()
        anrm.retain_grad()
    group.active_surface_points = apts
    group.active_surface_normals = anrm
    out.update(orientation=npy(orientation), aim_points=npy(aim), incident=npy(incident),
               aligned_points=npy(apts), aligned_normals=npy(anrm),
               active_mask=npy(mask), target_idx=npy(target_idx))

    use_blocking = bool(case.get("blocking", False))
    captured_prims = []
    if use_blocking:
        # trace_rays builds the primitives internally: wrap the builder so that the tensors it returned can be
        # stored and their gradients retained (they are non-leaf tensors of the autograd graph)
        _orig_builder = ref_blocking.create_blocking_primitives_rectangles_by_index

        def _capturing_builder(*a, **k):
            prims = _orig_builder(*a, **k)
            if with_grads and prims[0].requires_grad:
                for t_ in prims:
                    t_.retain_grad()
            captured_prims.append(prims)
            return prims

        ref_blocking.create_blocking_primitives_rectangles_by_index = _capturing_builder
    rt = HeliostatRayTracer(
        scenario=scenario, heliostat_group=group, blocking_active=use_blocking,
        batch_size=case.get("batch_size", 100), random_seed=case.get("seed", 7),
        bitmap_resolution=res, dni=case.get("dni", None),
    )
    du32, de32 = rt.distortions_dataset.distortions_u, rt.distortions_dataset.distortions_e
    if distortions_f32 is not None:
        du32, de32 = distortions_f32
    if dtype != torch.float32:
        rt.distortions_dataset.distortions_u = du32.to(dtype)
        rt.distortions_dataset.distortions_e = de32.to(dtype)
    du, de = rt.distortions_dataset.distortions_u, rt.distortions_dataset.distortions_e
    ext = case.get("extinction", 0.0)
    refl = case.get("reflectivity", 0.935)
    out.update(
        ray_magnitude=np.float64(float(rt.ray_magnitude)), extinction=np.float64(ext), reflectivity=np.float64(refl),
        resolution=npy(res), target_centers=npy(b["planar"].centers), target_normals=npy(b["planar"].normals),
        target_dims=npy(b["planar"].dimensions), n_rays=np.int64(case["n_rays"]), seed=np.int64(case.get("seed", 7)),
        cyl_centers=npy(b["cyl"].centers), cyl_normals=npy(b["cyl"].normals), cyl_axes=npy(b["cyl"].axes),
        cyl_radii=npy(b["cyl"].radii), cyl_heights=npy(b["cyl"].heights), cyl_opening=npy(b["cyl"].opening_angles),
        covariance=np.float64(case.get("covariance", 4.3681e-06)),
    )
    if store_rays:
        out.update(distortions_u=npy(du), distortions_e=npy(de))

    # --- stage-by-stage (same calls trace_rays makes, heliostat_ray_tracer.py:285-494) ---
    if store_rays:
        with torch.no_grad():
            refl_dirs = geometry.reflect(incident.unsqueeze(1), anrm)
            rays = rt.scatter_rays(distortion_u=du.contiguous(), distortion_e=de.contiguous(),
                                   original_ray_direction=refl_dirs, device=CPU)
            n_planar = b["planar"].centers.shape[0]
            e_px = torch.zeros(rays.ray_magnitudes.shape, dtype=dtype)
            u_px, t, inten = torch.zeros_like(e_px), torch.zeros_like(e_px), torch.zeros_like(e_px)
            pm = target_idx < n_planar
            if pm.any():
                e_px[pm], u_px[pm], t[pm], inten[pm] = geometry.line_plane_intersections(
                    rays=Rays(rays.ray_directions[pm], rays.ray_magnitudes[pm]), points_at_ray_origins=apts[pm],
                    target_areas=b["planar"], target_area_indices=target_idx[pm], bitmap_resolution=res, device=CPU)
            if (~pm).any():
                e_px[~pm], u_px[~pm], t[~pm], inten[~pm] = geometry.line_cylinder_intersections(
                    rays=Rays(rays.ray_directions[~pm], rays.ray_magnitudes[~pm]), points_at_ray_origins=apts[~pm],
                    target_areas=b["cyl"], target_area_indices=target_idx[~pm] - n_planar, bitmap_resolution=res,
                    device=CPU)
            blocked = torch.zeros_like(inten)
            if use_blocking:
                # the calls of heliostat_ray_tracer.py:444-480 for ONE batch holding every heliostat
                corners, spans, pnormals = _orig_builder(
                    blocking_heliostats_active_surface_points=rt.blocking_heliostat_surfaces_active, device=CPU)
                owner = torch.nonzero(mask, as_tuple=True)[0].repeat_interleave(e_px.shape[1] * e_px.shape[2])
                filt = ref_blocking.lbvh_filter_blocking_planes(
                    points_at_ray_origins=apts, ray_directions=rays.ray_directions,
                    blocking_primitives_corners=corners, ray_to_heliostat_mapping=owner,
                    intersection_distances_target=t, device=CPU)
                if filt.numel() > 0:
                    blocked = ref_blocking.soft_ray_blocking_mask(
                        ray_origins=apts, ray_directions=rays.ray_directions,
                        blocking_primitives_corners=corners[filt], blocking_primitives_spans=spans[filt],
                        blocking_primitives_normals=pnormals[filt], epsilon=1e-12, softness=1000.0)
                out.update(prim_corners=npy(corners), prim_spans=npy(spans), prim_normals=npy(pnormals),
                           filter_indices=npy(filt), blocked=npy(blocked),
                           blocking_surfaces=npy(rt.blocking_heliostat_surfaces_active))
            inten_abs = inten * (1 - blocked) * (1 - ext) * refl
            bitmaps = rt.bilinear_splatting(e_px, u_px, inten_abs, device=CPU)
        out.update(reflected=npy(refl_dirs), scattered=npy(rays.ray_directions), e_px=npy(e_px), u_px=npy(u_px),
                   distances=npy(t), intensities=npy(inten), stage_bitmaps=npy(bitmaps))

    flux, intercept, on_target, blocking = rt.trace_rays(
        incident_ray_directions=incident, active_heliostats_mask=mask, target_area_indices=target_idx,
        ray_extinction_factor=ext, mirror_reflectivity=refl, device=CPU,
    )
    per_target = rt.get_bitmaps_per_target(flux, target_idx, device=CPU)
    out.update(flux=npy(flux), intercept=npy(intercept), on_target=npy(on_target), blocking=npy(blocking),
               per_target=npy(per_target), sampler_indices=npy(rt.get_sampler_indices()))

    if with_grads:
        g = torch.Generator().manual_seed(1234)
        weights = torch.rand(flux.shape, generator=g, dtype=torch.float32).to(dtype)
        loss = (flux * weights).sum()
        loss.backward()
        out.update(loss_weights=npy(weights), loss=npy(loss),
                   grad_aligned_points=npy(apts.grad), grad_aligned_normals=npy(anrm.grad),
                   grad_nurbs_points=npy(sp.grad), grad_nurbs_normals=npy(sn.grad),
                   grad_orientation=npy(orientation.grad), grad_control_points=npy(cp_active.grad))
        # gradients of the loss w.r.t. the kinematic deviation parameters = orientation gradient through the Jacobians
        out.update(grad_kin_rot=np.einsum("hij,hijk->hk", out["grad_orientation"], out["kin_jac_rot"]),
                   grad_kin_trans=np.einsum("hij,hijk->hk", out["grad_orientation"], out["kin_jac_trans"]))
        if use_blocking and captured_prims and captured_prims[-1][0].grad is not None:
            c_, s_, n_ = captured_prims[-1]
            out.update(grad_prim_corners=npy(c_.grad), grad_prim_spans=npy(s_.grad), grad_prim_normals=npy(n_.grad))
    if use_blocking:
        ref_blocking.create_blocking_primitives_rectangles_by_index = _orig_builder
    torch.set_default_dtype(torch.float32)
    return out, (du32, de32)


def run_scenario_file(filename, mapping, n_rays, points_per_facet, blocking, resolution, dtype=torch.float32,
                      distortions_f32=None):
    """The reference's own scenario files (tests/data/scenarios/*.h5) through its own loader, kinematics and ray
    tracer - the path of tests/raytracing/test_blocking.py:336-426 and tests/field/test_integration_alignment.py -
    captured at the boundary of the hot path: aligned surface points / normals in, bitmaps / factors / gradients out.
    ``mapping`` = [(heliostat name, target name, incident direction)]."""
    import h5py     # the stand-in installed by _import_reference()
    torch.set_default_dtype(dtype)
    torch.manual_seed(7)
    with h5py.File(pathlib.Path(REFERENCE) / "tests/data/scenarios" / filename, "r") as scenario_file:
        scenario = Scenario.load_scenario_from_hdf5(
            scenario_file=scenario_file, number_of_surface_points_per_facet=torch.tensor(points_per_facet), device=CPU)
    group = scenario.heliostat_field.heliostat_groups[0]
    string_mapping = [(h, t_, torch.nn.functional.normalize(torch.tensor(d, dtype=dtype), dim=-1)) for h, t_, d in mapping]
    mask, target_idx, incident = scenario.index_mapping(heliostat_group=group, string_mapping=string_mapping, device=CPU)
    group.activate_heliostats(active_heliostats_mask=mask, device=CPU)
    # the learnable kinematic parameters become leaves, so that the backward below also yields the END-TO-END gradients
    # flux -> aligned surfaces -> orientations -> kinematics (what the kinematics reconstructor optimises)
    kin = group.kinematics
    kin_rot = kin.active_rotation_deviation_parameters.detach().clone().requires_grad_(True)
    kin_trans = kin.active_translation_deviation_parameters.detach().clone().requires_grad_(True)
    kin_opt = kin.actuators.active_optimizable_parameters.detach().clone()
    if kin_opt.numel():
        kin_opt.requires_grad_(True)
    kin.active_rotation_deviation_parameters, kin.active_translation_deviation_parameters = kin_rot, kin_trans
    kin.actuators.active_optimizable_parameters = kin_opt
    surface_points, surface_normals = group.active_surface_points.detach().clone(), group.active_surface_normals.detach().clone()
    aim_points = scenario.solar_tower.get_centers_of_target_areas(target_area_indices=target_idx, device=CPU)
    group.align_surfaces_with_incident_ray_directions(
        aim_points=aim_points, incident_ray_directions=incident, active_heliostats_mask=mask, device=CPU)
    apts = group.active_surface_points.to(dtype)
    anrm = group.active_surface_normals.to(dtype)
    apts.retain_grad()
    anrm.retain_grad()
    group.active_surface_points, group.active_surface_normals = apts, anrm
    scenario.set_number_of_rays(number_of_rays=n_rays)
    captured = []
    _orig_builder = ref_blocking.create_blocking_primitives_rectangles_by_index

    def _capturing_builder(*a, **k):
        prims = _orig_builder(*a, **k)
        if prims[0].requires_grad:
            for t_ in prims:
                t_.retain_grad()
        captured.append(prims)
        return prims

    ref_blocking.create_blocking_primitives_rectangles_by_index = _capturing_builder
    res = torch.tensor(resolution)
    rt = HeliostatRayTracer(scenario=scenario, heliostat_group=group, blocking_active=blocking, batch_size=100,
                            bitmap_resolution=res)
    du32, de32 = rt.distortions_dataset.distortions_u, rt.distortions_dataset.distortions_e
    if distortions_f32 is not None:
        du32, de32 = distortions_f32
    rt.distortions_dataset.distortions_u, rt.distortions_dataset.distortions_e = du32.to(dtype), de32.to(dtype)
    flux, intercept, on_target, unblocked = rt.trace_rays(
        incident_ray_directions=incident, active_heliostats_mask=mask, target_area_indices=target_idx, device=CPU)
    g = torch.Generator().manual_seed(1234)
    weights = torch.rand(flux.shape, generator=g, dtype=torch.float32).to(dtype)
    (flux * weights).sum().backward()
    tower = scenario.solar_tower
    planar, cyl = tower.target_areas[0], tower.target_areas[1]
    out = dict(
        aligned_points=npy(apts), aligned_normals=npy(anrm), incident=npy(incident), target_idx=npy(target_idx),
        active_mask=npy(mask), distortions_u=npy(rt.distortions_dataset.distortions_u),
        distortions_e=npy(rt.distortions_dataset.distortions_e), resolution=npy(res),
        target_centers=npy(planar.centers), target_normals=npy(planar.normals), target_dims=npy(planar.dimensions),
        cyl_centers=npy(cyl.centers), cyl_normals=npy(cyl.normals), cyl_axes=npy(cyl.axes), cyl_radii=npy(cyl.radii),
        cyl_heights=npy(cyl.heights), cyl_opening=npy(cyl.opening_angles),
        ray_magnitude=np.float64(float(rt.ray_magnitude)), extinction=np.float64(0.0), reflectivity=np.float64(0.935),
        flux=npy(flux), intercept=npy(intercept), on_target=npy(on_target), blocking=npy(unblocked),
        loss_weights=npy(weights), grad_aligned_points=npy(apts.grad), grad_aligned_normals=npy(anrm.grad),
        control_points=npy(group.nurbs_control_points), positions=npy(group.positions),
        surface_points=npy(surface_points), surface_normals=npy(surface_normals), aim_points=npy(aim_points),
        kin_positions=npy(kin.active_heliostat_positions), kin_rot_dev=npy(kin_rot), kin_trans_dev=npy(kin_trans),
        kin_act_nonopt=npy(kin.actuators.active_non_optimizable_parameters), kin_act_opt=npy(kin_opt),
        grad_kin_rot_dev=npy(kin_rot.grad), grad_kin_trans_dev=npy(kin_trans.grad))
    if kin_opt.numel():
        out.update(grad_kin_act_opt=npy(kin_opt.grad))
    if blocking:
        c_, s_, n_ = captured[-1]
        out.update(blocking_surfaces=npy(rt.blocking_heliostat_surfaces_active), prim_corners=npy(c_), prim_spans=npy(s_),
                   prim_normals=npy(n_), owner=npy(torch.nonzero(mask, as_tuple=True)[0]))
        if c_.grad is not None:
            out.update(grad_prim_corners=npy(c_.grad), grad_prim_spans=npy(s_.grad), grad_prim_normals=npy(n_.grad))
        # the filtered set, from a second (gradient-free) pass through the same stages
        with torch.no_grad():
            refl = geometry.reflect(incident.unsqueeze(1), anrm)
            rays = rt.scatter_rays(distortion_u=rt.distortions_dataset.distortions_u.contiguous(),
                                   distortion_e=rt.distortions_dataset.distortions_e.contiguous(),
                                   original_ray_direction=refl, device=CPU)
            n_planar = planar.centers.shape[0]
            t = torch.zeros(rays.ray_magnitudes.shape, dtype=dtype)
            pm = target_idx < n_planar
            if pm.any():
                t[pm] = geometry.line_plane_intersections(
                    rays=Rays(rays.ray_directions[pm], rays.ray_magnitudes[pm]), points_at_ray_origins=apts[pm],
                    target_areas=planar, target_area_indices=target_idx[pm], bitmap_resolution=res, device=CPU)[2]
            if (~pm).any():
                t[~pm] = geometry.line_cylinder_intersections(
                    rays=Rays(rays.ray_directions[~pm], rays.ray_magnitudes[~pm]), points_at_ray_origins=apts[~pm],
                    target_areas=cyl, target_area_indices=target_idx[~pm] - n_planar, bitmap_resolution=res, device=CPU)[2]
            owner = torch.nonzero(mask, as_tuple=True)[0].repeat_interleave(t.shape[1] * t.shape[2])
            filt = ref_blocking.lbvh_filter_blocking_planes(
                points_at_ray_origins=apts, ray_directions=rays.ray_directions, blocking_primitives_corners=c_.detach(),
                ray_to_heliostat_mapping=owner, intersection_distances_target=t, device=CPU)
            out.update(filter_indices=npy(filt))
    ref_blocking.create_blocking_primitives_rectangles_by_index = _orig_builder
    torch.set_default_dtype(torch.float32)
    return out, (du32, de32)


# The reference's own scenario files.  "real_blocking" is the setting of tests/raytracing/test_blocking.py:336-426 (six
# heliostats, five of them in a cluster north of the tower, planar target_0) with fewer rays and surface points so that
# the fixture stays small; "real_paint_mixed" traces the four PAINT heliostats onto the cylindrical receiver and the
# planar multi-focus tower with a slanted sun.
REAL_CASES = {
    "real_blocking": dict(
        filename="test_blocking.h5", n_rays=6, points_per_facet=[10, 10], blocking=True, resolution=[64, 64],
        mapping=[(f"heliostat_{i}", "target_0", [0.0, 1.0, 0.0, 0.0]) for i in range(6)]),
    "real_paint_mixed": dict(
        filename="test_scenario_paint_four_heliostats.h5", n_rays=5, points_per_facet=[8, 8], blocking=False,
        resolution=[96, 64],
        mapping=[("AA28", "receiver", [0.3, 0.8, -0.52, 0.0]), ("AA31", "multi_focus_tower", [0.3, 0.8, -0.52, 0.0]),
                 ("AA39", "receiver", [0.3, 0.8, -0.52, 0.0]), ("AC43", "solar_tower_juelich_upper", [0.3, 0.8, -0.52, 0.0])]),
    # tests/field/test_integration_alignment.py:13-22: ONE heliostat built from the prototypes (STRAL surface, ideal
    # actuators), activated four times with the sun in the south, west, east and zenith
    "real_stral_single": dict(
        filename="test_scenario_stral_single_heliostat.h5", n_rays=6, points_per_facet=[12, 12], blocking=False,
        resolution=[64, 64],
        mapping=[("heliostat_1", "receiver", [0.0, 1.0, 0.0, 0.0]), ("heliostat_1", "receiver", [-1.0, 0.0, 0.0, 0.0]),
                 ("heliostat_1", "receiver", [1.0, 0.0, 0.0, 0.0]), ("heliostat_1", "receiver", [0.0, 0.0, -1.0, 0.0])]),
}


def kinematics_fixture():
    """Rigid-body kinematics of the reference (kinematics_rigid_body.py:194-634) on its own scenario files - linear
    actuators with fitted parameters, real deviation parameters: orientation matrices, motor positions and the
    Jacobians of the orientations w.r.t. the rotation / translation deviations and the optimisable actuator parameters
    (torch.autograd.functional.jacobian), in fp32 and fp64; plus the calibration path (orientations from given motor
    positions)."""
    import h5py
    out = {}
    cases = [("blocking", "test_blocking.h5", [f"heliostat_{i}" for i in range(6)], "target_0", [0.0, 1.0, 0.0, 0.0]),
             ("paint", "test_scenario_paint_four_heliostats.h5", ["AA28", "AA31", "AA39", "AC43"], "receiver", [0.2, 0.9, -0.39, 0.0])]
    for tag, filename, names, target, sun in cases:
        for dtype, sfx in ((torch.float32, "f32"), (torch.float64, "f64")):
            torch.set_default_dtype(dtype)
            with h5py.File(pathlib.Path(REFERENCE) / "tests/data/scenarios" / filename, "r") as scenario_file:
                scenario = Scenario.load_scenario_from_hdf5(
                    scenario_file=scenario_file, number_of_surface_points_per_facet=torch.tensor([4, 4]), device=CPU)
            group = scenario.heliostat_field.heliostat_groups[0]
            kin = group.kinematics
            mapping = [(n_, target, torch.nn.functional.normalize(torch.tensor(sun, dtype=dtype), dim=0)) for n_ in names]
            mask, target_idx, incident = scenario.index_mapping(heliostat_group=group, string_mapping=mapping, device=CPU)
            group.activate_heliostats(active_heliostats_mask=mask, device=CPU)
            aim = scenario.solar_tower.get_centers_of_target_areas(target_area_indices=target_idx, device=CPU).to(dtype)
            incident = incident.to(dtype)
            # cast every parameter tensor of the kinematics to the run's dtype (the loader builds float32 tensors)
            for obj, attrs in ((kin, ["active_heliostat_positions", "active_rotation_deviation_parameters",
                                      "active_translation_deviation_parameters", "initial_orientation_offsets",
                                      "kinematics_standard_orientation", "homogeneous_origin"]),
                               (kin.actuators, ["active_non_optimizable_parameters", "active_optimizable_parameters"])):
                for a_ in attrs:
                    setattr(obj, a_, getattr(obj, a_).detach().to(dtype))
            rot0 = kin.active_rotation_deviation_parameters.clone()
            trans0 = kin.active_translation_deviation_parameters.clone()
            opt0 = kin.actuators.active_optimizable_parameters.clone()

            def _ori(rot, trans, opt):
                kin.active_rotation_deviation_parameters, kin.active_translation_deviation_parameters = rot, trans
                kin.actuators.active_optimizable_parameters = opt
                return kin.incident_ray_directions_to_orientations(incident_ray_directions=incident, aim_points=aim, device=CPU)

            ori = _ori(rot0, trans0, opt0)
            motor = kin.active_motor_positions.detach().clone()
            has_opt = opt0.numel() > 0
            if has_opt:
                jr, jt, jo = torch.autograd.functional.jacobian(_ori, (rot0, trans0, opt0))
            else:
                jr, jt = torch.autograd.functional.jacobian(lambda r_, t_: _ori(r_, t_, opt0), (rot0, trans0))
                jo = None
            Hk = ori.shape[0]
            hh = torch.arange(Hk)
            kin.active_rotation_deviation_parameters, kin.active_translation_deviation_parameters = rot0, trans0
            kin.actuators.active_optimizable_parameters = opt0
            motor_given = motor + torch.arange(2 * Hk, dtype=dtype).reshape(Hk, 2) * (500.0 if has_opt else 0.01)
            ori_motor = kin.motor_positions_to_orientations(motor_positions=motor_given, device=CPU)
            pre = f"kin_{tag}_{sfx}_"
            out.update({pre + "positions": npy(kin.active_heliostat_positions), pre + "rot_dev": npy(rot0),
                        pre + "trans_dev": npy(trans0), pre + "act_nonopt": npy(kin.actuators.active_non_optimizable_parameters),
                        pre + "act_opt": npy(opt0), pre + "offsets": npy(kin.initial_orientation_offsets[0]),
                        pre + "incident": npy(incident), pre + "aim": npy(aim), pre + "orientation": npy(ori.detach()),
                        pre + "motor": npy(motor), pre + "jac_rot": npy(jr[hh, :, :, hh]), pre + "jac_trans": npy(jt[hh, :, :, hh]),
                        pre + "motor_given": npy(motor_given), pre + "orientation_from_motor": npy(ori_motor.detach())})
            if jo is not None:
                out[pre + "jac_opt"] = npy(jo[hh, :, :, hh])
            torch.set_default_dtype(torch.float32)
    save("kinematics", out)


def save_interop_check():
    """Duck-typing check of the drop-in classes against the REFERENCE's own objects (no kernel runs: there is no GPU
    here): ``artist_amd.HeliostatRayTracer`` is constructed on ARTIST's ``Scenario`` / ``HeliostatGroupRigidBody``
    loaded from its own HDF5 file, and every host-side helper that reads ARTIST attributes is called on them.  The
    result (what was checked, sizes seen) is stored as tests/golden/interop_check.json."""
    import json

    import h5py
    sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
    import artist_amd
    from artist_amd import flux as amd_flux
    from artist_amd import raytracing as amd_rt
    from artist_amd.blocking import create_blocking_primitives_rectangles_by_index

    torch.manual_seed(7)
    with h5py.File(pathlib.Path(REFERENCE) / "tests/data/scenarios/test_blocking.h5", "r") as scenario_file:
        scenario = Scenario.load_scenario_from_hdf5(scenario_file=scenario_file,
                                                    number_of_surface_points_per_facet=torch.tensor([10, 10]), device=CPU)
    group = scenario.heliostat_field.heliostat_groups[0]
    mapping = [(f"heliostat_{i}", "target_0", torch.tensor([0.0, 1.0, 0.0, 0.0])) for i in range(6)]
    mask, target_idx, incident = scenario.index_mapping(heliostat_group=group, string_mapping=mapping, device=CPU)
    group.activate_heliostats(active_heliostats_mask=mask, device=CPU)
    group.align_surfaces_with_incident_ray_directions(
        aim_points=scenario.solar_tower.get_centers_of_target_areas(target_area_indices=target_idx, device=CPU),
        incident_ray_directions=incident, active_heliostats_mask=mask, device=CPU)
    scenario.set_number_of_rays(number_of_rays=3)
    ours = artist_amd.HeliostatRayTracer(scenario=scenario, heliostat_group=group, blocking_active=True, batch_size=10,
                                         bitmap_resolution=torch.tensor([64, 64]))
    theirs = HeliostatRayTracer(scenario=scenario, heliostat_group=group, blocking_active=True, batch_size=10,
                                bitmap_resolution=torch.tensor([64, 64]))
    checks = {}
    checks["blocking_surfaces_equal"] = bool(torch.equal(ours.blocking_heliostat_surfaces_active,
                                                         theirs.blocking_heliostat_surfaces_active))
    checks["distortions_equal"] = bool(torch.equal(ours.distortions_dataset.distortions_u, theirs.distortions_dataset.distortions_u)
                                       and torch.equal(ours.distortions_dataset.distortions_e, theirs.distortions_dataset.distortions_e))
    checks["sampler_indices_equal"] = ours.get_sampler_indices().tolist() == theirs.get_sampler_indices().tolist()
    checks["ray_magnitude_equal"] = float(ours.ray_magnitude) == float(theirs.ray_magnitude)
    planar = amd_rt._planar_tables(scenario.solar_tower, CPU)
    cyl = amd_rt._cylinder_tables(scenario.solar_tower)
    checks["planar_tables"] = [list(t.shape) for t in planar]
    checks["cylinder_tables"] = [list(t.shape) for t in cyl]
    ref_prims = ref_blocking.create_blocking_primitives_rectangles_by_index(theirs.blocking_heliostat_surfaces_active, device=CPU)
    our_prims = create_blocking_primitives_rectangles_by_index(ours.blocking_heliostat_surfaces_active)
    checks["primitives_max_abs_diff"] = float(max((a - b).abs().max() for a, b in zip(ref_prims, our_prims)))
    tix_all = torch.tensor([0, 4, 5, 10])           # planar 0 and 4, cylinders 0 and 5
    dims = amd_flux.target_dimensions(scenario.solar_tower, tix_all)
    checks["target_dimensions"] = dims.tolist()
    corners, _, _, owner, max_angle, compat = ours._blocking_arguments(None, mask)
    checks["owner"] = owner.tolist()
    checks["max_scatter_angle"] = max_angle
    with torch.no_grad():
        try:
            ours.trace_rays(incident, mask, target_idx)
            checks["cpu_trace"] = "unexpectedly ran"
        except artist_amd.ArtistHipError as exc:       # the product has no CPU path: must fail loudly
            checks["cpu_trace"] = f"ArtistHipError: {exc}"[:120]
    assert checks["blocking_surfaces_equal"] and checks["distortions_equal"] and checks["sampler_indices_equal"]
    assert checks["primitives_max_abs_diff"] < 1e-6 and "no CPU fallback" in checks["cpu_trace"]
    path = OUT_DIR / "interop_check.json"
    path.write_text(json.dumps(checks, indent=1))
    print(f"wrote {path.name}: {checks}")


def surface_reconstructor_epochs(n_epochs=3, dtype=torch.float32):
    """ONE real optimiser run of the reference: ``SurfaceReconstructor.reconstruct_surfaces`` (artist/optim/surface_reconstructor.py:
    842-1152, unmodified) for a few epochs on a small synthetic scenario - three heliostats with two samples each (one training,
    one test sample per heliostat after the reference's own train/test split), flat model surfaces against measured flux from
    deflected ones.  The only stand-in is the calibration-data parser (``_parse_group_calibration_data`` reads PNG / JSON files
    through PAINT's parser; it is handed the synthetic measurements instead) and the regulariser weights are zero.  Captured per
    epoch, by wrapping the instance's own methods: control points at the start, orientation matrices of the training samples,
    cropped predicted flux, per-sample flux loss, total loss, the control-point gradient after
    ``_synchronize_and_lock_gradients``, the learning rate of the step and the control points after ``optimizer.step()``.
    ``dtype=float64``: the same run in double precision on the SAME (fp32-sampled) ray distortions - the yardstick for the fp32
    results of the first epoch (later epochs start from control points that differ in the last bits)."""
    from artist.flux import bitmap
    from artist.optim import SurfaceReconstructor
    from artist.optim.loss import PixelLoss
    from artist.util import constants

    torch.manual_seed(7)
    case = dict(n_heliostats=3, n_cp=(6, 6), degrees=(3, 3), n_eval=20, n_rays=6, resolution=[64, 64], z_noise=0.0,
                positions=[[-12.0, 45.0, 0.0, 1.0], [4.0, 70.0, 0.0, 1.0], [20.0, 100.0, 0.0, 1.0]], **RECEIVER)
    b = build(case, dtype)
    scenario, group = b["scenario"], b["group"]
    degrees, cp_flat, uv_full, canting_h, transl_h = b["nurbs_inputs"]
    n_hel, res = 3, torch.tensor([64, 64])
    if dtype != torch.float32:
        sun = scenario.light_sources.light_source_list[0]

        def f32_distortions(number_of_points, number_of_facets=4, number_of_active_heliostats=1, random_seed=7):
            torch.manual_seed(random_seed)                              # artist/scene/sun.py:224-233 with float32 parameters
            mvn = torch.distributions.MultivariateNormal(torch.zeros(2, dtype=torch.float32),
                                                         torch.tensor([[4.3681e-06, 0.0], [0.0, 4.3681e-06]], dtype=torch.float32))
            du, de = mvn.sample((number_of_active_heliostats, sun.number_of_rays, number_of_points)).permute(3, 0, 1, 2)
            return du.to(dtype), de.to(dtype)

        sun.get_distortions = f32_distortions
    mask = torch.tensor([2, 2, 2], dtype=torch.int32)                  # two samples per heliostat
    n_samples = int(mask.sum())
    tix = torch.zeros(n_samples, dtype=torch.int64)
    incident = torch.nn.functional.normalize(torch.tensor(
        [[0.0, 1.0, 0.0, 0.0], [0.25, 0.9, -0.2, 0.0], [-0.1, 0.95, -0.1, 0.0], [0.0, 0.8, -0.6, 0.0], [0.15, 0.97, 0.0, 0.0],
         [-0.3, 0.85, -0.3, 0.0]], dtype=torch.float32), dim=1).to(dtype)

    # measured flux: the reference's own chain on the TRUE surfaces (flat nets + a smooth bump and a tilt per heliostat)
    cp_true = cp_flat.clone()
    x, y = cp_true[..., 0], cp_true[..., 1]
    amp = torch.tensor([1.5e-3, -1.0e-3, 2.0e-3], dtype=torch.float32).to(dtype).view(3, 1, 1, 1)
    tilt = torch.tensor([4e-4, -4e-4, 2e-4], dtype=torch.float32).to(dtype).view(3, 1, 1, 1)
    cp_true[..., 2] = amp * (x ** 2 - 0.5 * y ** 2) + tilt * x

    def measure(cp):
        group.nurbs_control_points = cp
        group.activate_heliostats(active_heliostats_mask=mask, device=CPU)
        surf = NURBSSurfaces(degrees=degrees, control_points=group.active_nurbs_control_points, device=CPU)
        pts, nrm = surf.calculate_surface_points_and_normals(
            evaluation_points=uv_full.repeat_interleave(mask, dim=0), canting=group.active_canting,
            facet_translations=group.active_facet_translations, device=CPU)
        group.active_surface_points = pts.reshape(n_samples, -1, 4)
        group.active_surface_normals = nrm.reshape(n_samples, -1, 4)
        group.align_surfaces_with_incident_ray_directions(
            aim_points=scenario.solar_tower.get_centers_of_target_areas(target_area_indices=tix, device=CPU),
            incident_ray_directions=incident, active_heliostats_mask=mask, device=CPU)
        rt = HeliostatRayTracer(scenario=scenario, heliostat_group=group, blocking_active=False, batch_size=100, random_seed=3,
                                bitmap_resolution=res)
        flux, _, _, _ = rt.trace_rays(incident_ray_directions=incident, active_heliostats_mask=mask, target_area_indices=tix, device=CPU)
        return bitmap.crop_flux_distributions_around_center(flux_distributions=flux, solar_tower=scenario.solar_tower,
                                                            target_area_indices=tix, device=CPU)

    with torch.no_grad():
        flux_measured = measure(cp_true).clone()
    group.nurbs_control_points = cp_flat.clone()

    optimizer_dict = {constants.initial_learning_rate: 2e-5, constants.tolerance: 0.0, constants.max_epoch: n_epochs - 1,
                      constants.batch_size: 30, constants.log_step: 0, constants.early_stopping_delta: 1e-9,
                      constants.early_stopping_patience: 100, constants.early_stopping_window: 10}
    scheduler_dict = {constants.scheduler_type: constants.exponential, constants.gamma: 0.9}
    constraint_dict = {constants.rho_flux_integral: 1.0, constants.energy_tolerance: 0.01, constants.weight_smoothness: 0.0,
                       constants.weight_ideal_surface: 0.0}
    config = {constants.optimization: optimizer_dict, constants.scheduler: scheduler_dict, constants.constraints: constraint_dict}
    ddp = dict(device=CPU, is_distributed=False, is_nested=False, rank=0, world_size=1, process_subgroup=None,
               groups_to_ranks_mapping={0: [0]}, heliostat_group_rank=0, heliostat_group_world_size=1, ranks_to_groups_mapping={0: [0]})
    rec = SurfaceReconstructor(ddp_setup=ddp, scenario=scenario, data={constants.data_parser: None, constants.heliostat_data_mapping: []},
                               optimization_configuration=config, number_of_surface_points=torch.tensor([20, 20]),
                               bitmap_resolution=res, device=CPU)
    rec._parse_group_calibration_data = lambda heliostat_group, device: (
        flux_measured, torch.zeros(n_samples, 4), incident, torch.zeros(n_samples, 2), mask, tix)

    log = dict(cp_start=[], cropped=[], orientation=[], loss_per_sample=[], grad_locked=[], lr=[], cp_after=[], train_idx=None)
    predict = rec._predict_flux

    def predict_wrapped(**kw):
        log["cp_start"].append(npy(kw["heliostat_group"].nurbs_control_points).copy())
        out = predict(**kw)
        log["cropped"].append(npy(out[0]).copy())
        log["train_idx"] = npy(kw["data_split"].train_indices)
        # the orientation matrices the alignment used (heliostat_group_rigid_body.py:210-222: recomputed here, no_grad)
        ds = kw["data_split"]
        with torch.no_grad():
            ori = kw["heliostat_group"].kinematics.incident_ray_directions_to_orientations(
                incident_ray_directions=ds.incident_ray_directions_train,
                aim_points=scenario.solar_tower.get_centers_of_target_areas(target_area_indices=ds.target_area_indices_train, device=CPU),
                device=CPU)
        log["orientation"].append(npy(ori).copy())
        return out

    rec._predict_flux = predict_wrapped
    lock = rec._synchronize_and_lock_gradients

    def lock_wrapped(optimizer, device):
        lock(optimizer=optimizer, device=device)
        prm = optimizer.param_groups[0]["params"][0]
        log["grad_locked"].append(npy(prm.grad).copy())
        log["lr"].append(float(optimizer.param_groups[0]["lr"]))

    rec._synchronize_and_lock_gradients = lock_wrapped
    setup = rec._setup_optimizer_scheduler_early_stopping

    def setup_wrapped(heliostat_group):
        optimizer, scheduler, stopper = setup(heliostat_group=heliostat_group)
        step = optimizer.step

        def step_wrapped(*a, **k):
            r = step(*a, **k)
            log["cp_after"].append(npy(optimizer.param_groups[0]["params"][0]).copy())
            return r

        optimizer.step = step_wrapped
        return optimizer, scheduler, stopper

    rec._setup_optimizer_scheduler_early_stopping = setup_wrapped

    class RecordingLoss(PixelLoss):
        def __call__(self, *a, **k):
            out = super().__call__(*a, **k)
            if out.requires_grad:                                       # (the validation calls run under no_grad)
                log["loss_per_sample"].append(npy(out).copy())
            return out

    _, history = rec.reconstruct_surfaces(loss_definition=RecordingLoss(), device=CPU)
    hist = history[0][0]
    E = len(log["cp_after"])
    assert E == n_epochs and len(log["cropped"]) == E and len(log["grad_locked"]) == E, (E, len(log["cropped"]))
    train_idx = log["train_idx"]
    out = dict(
        degrees=npy(degrees), eval_points=npy(uv_full[:1]), canting=npy(canting_h), facet_translations=npy(transl_h),
        positions=npy(group.positions), incident_train=npy(incident[train_idx]), target_idx_train=npy(tix[train_idx]),
        train_indices=train_idx, flux_measured_train=npy(flux_measured[train_idx]), target_centers=npy(b["planar"].centers),
        target_normals=npy(b["planar"].normals), target_dims=npy(b["planar"].dimensions), resolution=npy(res),
        n_rays=np.int64(case["n_rays"]), seed=np.int64(0), covariance=np.float64(4.3681e-06), reflectivity=np.float64(0.935),
        extinction=np.float64(0.0), ray_magnitude=np.float64(1.0), number_of_train_samples=np.int64(1),
        rho_flux_integral=np.float64(1.0), energy_tolerance=np.float64(0.01), epsilon=np.float64(1e-12),
        cp_start=np.stack(log["cp_start"]), orientation=np.stack(log["orientation"]), cropped_flux=np.stack(log["cropped"]),
        flux_loss_per_sample=np.stack(log["loss_per_sample"][:E]), grad_locked=np.stack(log["grad_locked"]), lr=np.asarray(log["lr"]),
        cp_after=np.stack(log["cp_after"]), total_loss=np.asarray(hist["total_loss"][:E], dtype=np.float64),
        flux_loss=np.asarray(hist["flux_loss"][:E], dtype=np.float64))
    print("  total loss per epoch:", out["total_loss"], " |grad| per epoch:", [float(np.linalg.norm(g)) for g in out["grad_locked"]])
    return out


def kinematics_reconstructor_epochs(n_epochs=3, dtype=torch.float32):
    """ONE real run of the reference's ``KinematicsReconstructor`` in its flux-driven mode
    (artist/optim/kinematics_reconstructor.py:886-1065, unmodified): the reference's own scenario file test_blocking.h5 (six
    heliostats with rigid-body kinematics and linear actuators; heliostat_3 stands behind the target and is left out), two
    calibration samples per heliostat (one training, one test sample after the reference's split), measured flux = the reference's
    own chain with the TRUE rotation deviations and the motor positions the true kinematics drives to; the model starts from
    perturbed deviations and Adam steps them through kinematics -> alignment -> trace -> FocalSpotLoss -> median per heliostat ->
    mean.  The calibration-data parser is the only stand-in.  Captured per epoch: the rotation deviations at the start, per-sample
    loss, total loss, their gradient, the learning rate, the deviations after the step."""
    import h5py     # the stand-in installed by _import_reference()
    from artist.optim import KinematicsReconstructor
    from artist.optim.loss import FocalSpotLoss
    from artist.util import constants

    torch.set_default_dtype(dtype)
    torch.manual_seed(7)
    with h5py.File(pathlib.Path(REFERENCE) / "tests/data/scenarios" / "test_blocking.h5", "r") as scenario_file:
        scenario = Scenario.load_scenario_from_hdf5(
            scenario_file=scenario_file, number_of_surface_points_per_facet=torch.tensor([12, 12]), device=CPU)
    scenario.set_number_of_rays(number_of_rays=10)
    group = scenario.heliostat_field.heliostat_groups[0]
    kin = group.kinematics
    res = torch.tensor([64, 64])
    names = [n_ for n_ in group.names if n_ != "heliostat_3"]
    suns = [[0.1, 1.0, -0.2, 0.0], [-0.25, 0.9, -0.35, 0.0]]
    mapping = [(n_, "target_3", torch.nn.functional.normalize(torch.tensor(d_, dtype=torch.float32), dim=0).to(dtype))
               for n_ in names for d_ in suns]
    mask, tix, incident = scenario.index_mapping(heliostat_group=group, string_mapping=mapping, device=CPU)
    n_samples = int(mask.sum())
    if dtype != torch.float32:
        sun = scenario.light_sources.light_source_list[0]

        def f32_distortions(number_of_points, number_of_facets=4, number_of_active_heliostats=1, random_seed=7):
            torch.manual_seed(random_seed)
            mvn = torch.distributions.MultivariateNormal(torch.zeros(2, dtype=torch.float32),
                                                         torch.tensor([[4.3681e-06, 0.0], [0.0, 4.3681e-06]], dtype=torch.float32))
            du, de = mvn.sample((number_of_active_heliostats, sun.number_of_rays, number_of_points)).permute(3, 0, 1, 2)
            return du.to(dtype), de.to(dtype)

        sun.get_distortions = f32_distortions

    # calibration data with the TRUE kinematics: motor positions it drives to, flux it produces
    true_rotation = kin.rotation_deviation_parameters.detach().clone()
    with torch.no_grad():
        group.activate_heliostats(active_heliostats_mask=mask, device=CPU)
        group.align_surfaces_with_incident_ray_directions(
            aim_points=scenario.solar_tower.get_centers_of_target_areas(target_area_indices=tix, device=CPU),
            incident_ray_directions=incident, active_heliostats_mask=mask, device=CPU)
        motors = kin.active_motor_positions.detach().clone()
        rt = HeliostatRayTracer(scenario=scenario, heliostat_group=group, blocking_active=False, batch_size=100, random_seed=3,
                                bitmap_resolution=res)
        flux_measured = rt.trace_rays(incident_ray_directions=incident, active_heliostats_mask=mask, target_area_indices=tix,
                                      device=CPU)[0].clone()
    g = torch.Generator().manual_seed(5)
    start = true_rotation + (3e-3 * torch.randn(true_rotation.shape, generator=g, dtype=torch.float32)).to(dtype)
    kin.rotation_deviation_parameters = start.clone()

    optimizer_dict = {constants.initial_learning_rate_rotation_deviation: 2e-4, constants.tolerance: 0.0, constants.max_epoch: n_epochs - 1,
                      constants.batch_size: 30, constants.log_step: 0, constants.early_stopping_delta: 1e-9,
                      constants.early_stopping_patience: 100, constants.early_stopping_window: 10}
    scheduler_dict = {constants.scheduler_type: constants.exponential, constants.gamma: 0.9}
    config = {constants.optimization: optimizer_dict, constants.scheduler: scheduler_dict}
    ddp = dict(device=CPU, is_distributed=False, is_nested=False, rank=0, world_size=1, process_subgroup=None,
               groups_to_ranks_mapping={0: [0]}, heliostat_group_rank=0, heliostat_group_world_size=1, ranks_to_groups_mapping={0: [0]})
    rec = KinematicsReconstructor(ddp_setup=ddp, scenario=scenario, data={constants.data_parser: None, constants.heliostat_data_mapping: []},
                                  optimization_configuration=config, reconstruction_method=constants.kinematics_reconstruction_raytracing,
                                  bitmap_resolution=res)
    rec._parse_group_calibration_data = lambda heliostat_group, device: (
        flux_measured, torch.zeros(n_samples, 4), incident, motors, mask, tix)

    log = dict(start=[], loss_per_sample=[], grad=[], lr=[], after=[], train_idx=None, flux=[])
    loss_fn = rec._compute_raytracing_loss

    def loss_wrapped(**kw):
        log["start"].append(npy(kw["heliostat_group"].kinematics.rotation_deviation_parameters).copy())
        log["train_idx"] = npy(kw["data_split"].train_indices)
        return loss_fn(**kw)

    rec._compute_raytracing_loss = loss_wrapped
    setup = rec._setup_optimizer_scheduler_early_stopping

    def setup_wrapped(heliostat_group):
        optimizer, scheduler, stopper = setup(heliostat_group=heliostat_group)
        step = optimizer.step

        def step_wrapped(*a, **k):
            prm = optimizer.param_groups[0]["params"][0]
            log["grad"].append(npy(prm.grad).copy())
            log["lr"].append(float(optimizer.param_groups[0]["lr"]))
            r = step(*a, **k)
            log["after"].append(npy(prm).copy())
            return r

        optimizer.step = step_wrapped
        return optimizer, scheduler, stopper

    rec._setup_optimizer_scheduler_early_stopping = setup_wrapped

    class RecordingLoss(FocalSpotLoss):
        def __call__(self, *a, **k):
            out = super().__call__(*a, **k)
            if out.requires_grad:
                log["loss_per_sample"].append(npy(out).copy())
                log["flux"].append(npy(k["prediction"]).copy())
            return out

    _, history = rec.reconstruct_kinematics(loss_definition=RecordingLoss(scenario=scenario), device=CPU)
    E = len(log["after"])
    assert E == n_epochs and len(log["loss_per_sample"]) == E, (E, len(log["loss_per_sample"]))
    tr = log["train_idx"]
    out = dict(
        heliostats=np.asarray([group.names.index(n_) for n_ in names]), suns=np.asarray(suns, dtype=np.float32), target=np.asarray("target_3"),
        incident_train=npy(incident[tr]), target_idx_train=npy(tix[tr]), motor_positions_train=npy(motors[tr]), train_indices=tr,
        flux_measured_train=npy(flux_measured[tr]), resolution=npy(res), n_rays=np.int64(10), seed=np.int64(0),
        points_per_facet=np.asarray([12, 12]), true_rotation=npy(true_rotation),
        rotation_start=np.stack(log["start"]), flux_predicted=np.stack(log["flux"][:E]), loss_per_sample=np.stack(log["loss_per_sample"][:E]),
        total_loss=np.asarray(history[0][0]["total_loss"][:E], dtype=np.float64), grad=np.stack(log["grad"]), lr=np.asarray(log["lr"]),
        rotation_after=np.stack(log["after"]))
    print("  kinematics reconstructor, total loss per epoch:", out["total_loss"], " |grad|:", [float(np.linalg.norm(g_)) for g_ in out["grad"]])
    return out


def save(name, arrays):
    path = OUT_DIR / f"{name}.npz"
    np.savez_compressed(path, **arrays)
    print(f"wrote {path.name}: {path.stat().st_size/1024:.1f} KiB, keys={len(arrays)}")


# --------------------------------------------------------------------------------------
# Cases
# --------------------------------------------------------------------------------------
RECEIVER = dict(target_centers=[[0.0, 0.0, 55.0, 1.0]], target_normals=[[0.0, 1.0, 0.0, 0.0]], target_dims=[[8.0, 8.0]])

CASES = {
    # H=4 active (each heliostat twice - the reference's sampler needs a uniform replica count,
    # sampling.py:133-135), degree 3, 10x10 cps, 64x64 bitmap; every stage stored.
    "small_deg3": dict(n_heliostats=2, active_mask=[2, 2], n_cp=(10, 10), degrees=(3, 3), n_eval=8, n_rays=4,
                       resolution=[64, 64], **RECEIVER),
    # degree 2, 7x7 cps, two tilted target planes, non-square bitmap, non-default scalars, dni.
    "small_deg2_tilted": dict(
        n_heliostats=2, n_cp=(7, 7), degrees=(2, 2), n_eval=6, n_rays=5, resolution=[96, 48],
        target_centers=[[0.0, 0.0, 55.0, 1.0], [3.0, -2.0, 40.0, 1.0]],
        target_normals=[[0.0, 1.0, 0.0, 0.0], [0.2182, 0.7071, 0.7071, 0.0]],
        target_dims=[[8.0, 8.0], [6.0, 5.0]], target_idx=[1, 0],
        incident=[[0.1, 0.9, -0.3, 0.0], [-0.2, 0.8, -0.5, 0.0]],
        extinction=0.1, reflectivity=0.9, dni=800.0, curvature=2e-3),
    # tiny target: most rays fall off the 1 m x 1 m receiver; one heliostat aims 30 m off -> all rays miss.
    "small_offtarget": dict(n_heliostats=2, n_cp=(6, 6), degrees=(3, 3), n_eval=6, n_rays=3, resolution=[32, 32],
                            target_centers=[[0.0, 0.0, 55.0, 1.0], [30.0, 0.0, 55.0, 1.0]],
                            target_normals=[[0.0, 1.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0]],
                            target_dims=[[1.0, 1.0], [1.0, 1.0]], target_idx=[0, 0],
                            aim_offset=[[0.0, 0.0, 0.0, 0.0], [30.0, 0.0, 0.0, 0.0]]),
    # mid-size: 2 heliostats, 20x20 points per facet, 8 rays, full 256x256 bitmap.
    "mid_256": dict(n_heliostats=2, n_cp=(6, 6), degrees=(3, 3), n_eval=20, n_rays=8, resolution=[256, 256],
                    curvature=1e-3, **RECEIVER),
}

# Cylindrical receivers (artist/raytracing/geometry.py:207-445) mixed with a planar one: heliostat 0 -> planar,
# 1 -> half cylinder facing north, 2 -> full cylinder (the -pi/2 seam of the unwrapped angle included), 3 -> a
# narrow sector most rays miss.
CASES["small_cyl_mixed"] = dict(
    n_heliostats=4, n_cp=(6, 6), degrees=(3, 3), n_eval=7, n_rays=5, resolution=[64, 48], curvature=1e-3,
    target_centers=[[0.0, 0.0, 55.0, 1.0]], target_normals=[[0.0, 1.0, 0.0, 0.0]], target_dims=[[8.0, 8.0]],
    cyl_centers=[[0.0, 0.0, 55.0, 1.0], [0.0, 0.0, 50.0, 1.0], [1.0, 0.5, 52.0, 1.0]],
    cyl_normals=[[0.0, 1.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0], [0.6, 0.8, 0.0, 0.0]],
    cyl_axes=[[0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 1.0, 0.0]],
    cyl_radii=[3.0, 2.5, 4.0], cyl_heights=[8.0, 6.0, 5.0], cyl_opening=[3.141592653589793, 6.283185307179586, 0.6],
    target_idx=[0, 1, 2, 3], extinction=0.05, reflectivity=0.9)
CASES["mid_cyl"] = dict(
    n_heliostats=2, n_cp=(6, 6), degrees=(3, 3), n_eval=20, n_rays=8, resolution=[256, 256], curvature=1e-3,
    target_centers=[[0.0, 0.0, 55.0, 1.0]], target_normals=[[0.0, 1.0, 0.0, 0.0]], target_dims=[[8.0, 8.0]],
    cyl_centers=[[0.0, 0.0, 55.0, 1.0]], cyl_normals=[[0.0, 1.0, 0.0, 0.0]], cyl_axes=[[0.0, 0.0, 1.0, 0.0]],
    cyl_radii=[3.5], cyl_heights=[9.0], cyl_opening=[3.141592653589793], target_idx=[1, 1])

# Blocking (artist/raytracing/blocking.py): a column of heliostats far north of the tower with the sun low in the
# south, so that the nearly vertical mirrors in front cut into the beams of the ones behind them; the third one
# is shifted east (partial overlap -> rays inside the sigmoid edge band), the fourth stands clear of the others.
CASES["small_blocking"] = dict(
    n_heliostats=4, n_cp=(6, 6), degrees=(3, 3), n_eval=8, n_rays=6, resolution=[64, 64], curvature=1e-3, blocking=True,
    positions=[[0.0, 140.0, 0.0, 1.0], [0.0, 137.0, 0.0, 1.0], [1.2, 134.0, 0.0, 1.0], [30.0, 120.0, 0.0, 1.0]],
    extinction=0.02, reflectivity=0.9, **RECEIVER)
CASES["mid_blocking"] = dict(
    n_heliostats=5, n_cp=(6, 6), degrees=(3, 3), n_eval=16, n_rays=8, resolution=[128, 128], curvature=1e-3,
    blocking=True, incident=[[0.1, 0.95, -0.1, 0.0]] * 5,
    positions=[[0.0, 150.0, 0.0, 1.0], [0.3, 147.2, 0.0, 1.0], [-0.8, 144.0, 0.0, 1.0], [2.0, 141.0, 0.0, 1.0],
               [-25.0, 100.0, 0.0, 1.0]], **RECEIVER)

# Config 1 of BASELINE.json: 1 heliostat, 4 planar facets, point sun, 10k rays.
CONFIG1 = dict(n_heliostats=1, n_cp=(10, 10), degrees=(3, 3), n_eval=50, n_rays=1, z_noise=0.0, covariance=1e-12,
               resolution=[256, 256], **RECEIVER)
# Config 2: 1 NURBS heliostat, Gaussian sun, 1M rays.
CONFIG2 = dict(n_heliostats=1, n_cp=(10, 10), degrees=(3, 3), n_eval=50, n_rays=100, resolution=[256, 256], **RECEIVER)


# More 1e6-ray pins of the north-star bound (one seed was a one-sample claim): another seed of the same heliostat, and an
# off-axis heliostat under a slanted sun (oblique incidence on the receiver, asymmetric image).
CONFIG2_SEED11 = dict(CONFIG2, seed=11)
CONFIG2_OFFAXIS = dict(CONFIG2, seed=23, positions=[[40.0, 75.0, 0.0, 1.0]], incident=[[0.25, 0.9, -0.35, 0.0]])

# A WELL-CONDITIONED cylinder through the reference (radius 25 m, mirrors ~40 m from the mantle: the quadratic cancels
# ~10-fold, not the 400-fold of the 3 m test cylinders above), so that a tight tolerance on the cylinder arithmetic can be
# asserted against reference output and not only against the restatement.
CASES["wide_cyl"] = dict(
    n_heliostats=3, n_cp=(6, 6), degrees=(3, 3), n_eval=16, n_rays=8, resolution=[192, 64], curvature=1e-3,
    positions=[[-30.0, 60.0, 0.0, 1.0], [5.0, 75.0, 0.0, 1.0], [40.0, 55.0, 0.0, 1.0]],
    target_centers=[[0.0, 0.0, 55.0, 1.0]], target_normals=[[0.0, 1.0, 0.0, 0.0]], target_dims=[[8.0, 8.0]],
    cyl_centers=[[0.0, 0.0, 40.0, 1.0]], cyl_normals=[[0.0, 1.0, 0.0, 0.0]], cyl_axes=[[0.0, 0.0, 1.0, 0.0]],
    cyl_radii=[25.0], cyl_heights=[12.0], cyl_opening=[2.6], target_idx=[1, 1, 1], extinction=0.1, reflectivity=0.9)


def summarize_large(arrs, keep):
    return {k: v for k, v in arrs.items() if k in keep}


def known_answers():
    """Inputs/expected values of the reference's own inline known-answer tests, read from the
    pytest parametrisation of the test modules (data only), plus the values the reference
    computes for them here."""
    sys.path.insert(0, REFERENCE)
    import importlib
    out = {}
    tg = importlib.import_module("tests.raytracing.test_geometry")
    # reflect: tests/raytracing/test_geometry.py:13-51
    for i, (inc, nrm, exp) in enumerate(tg.test_reflect_function.pytestmark[0].args[1]):
        out[f"reflect{i}_incident"] = npy(inc)
        out[f"reflect{i}_normals"] = npy(nrm)
        out[f"reflect{i}_expected"] = npy(exp)
        out[f"reflect{i}_reference"] = npy(geometry.reflect(inc, nrm))
    # line-plane: tests/raytracing/test_geometry.py:195-259 (targets from fixtures :110-166)
    targets = {
        "target_area_1_planar": ([[0.0, 0.0, 0.0, 1.0]], [[0.0, 1.0, 0.0, 0.0]], [[2.0, 2.0]]),
        "target_area_2_planar": ([[0.0, 0.0, 0.0, 1.0]], [[0.2182, 0.7071, 0.7071, 0.0]], [[3.0, 3.0]]),
    }
    for i, (rays, tname, origins, ee, eu, ed, ei) in enumerate(tg.test_line_plane_intersection.pytestmark[0].args[1]):
        c, n, d = (torch.tensor(x) for x in targets[tname])
        ta = TowerTargetAreasPlanar(names=["p"], centers=c, normals=n, dimensions=d)
        r = Rays(ray_directions=rays[0], ray_magnitudes=rays[1])
        e_px, u_px, t, inten = geometry.line_plane_intersections(r, origins, ta, torch.tensor([0]), device=CPU)
        out.update({
            f"plane{i}_dirs": npy(rays[0]), f"plane{i}_mags": npy(rays[1]), f"plane{i}_origins": npy(origins),
            f"plane{i}_center": npy(c), f"plane{i}_normal": npy(n), f"plane{i}_dims": npy(d),
            f"plane{i}_expected_e": npy(ee), f"plane{i}_expected_u": npy(eu), f"plane{i}_expected_t": npy(ed),
            f"plane{i}_expected_i": npy(ei), f"plane{i}_reference_e": npy(e_px), f"plane{i}_reference_u": npy(u_px),
            f"plane{i}_reference_t": npy(t), f"plane{i}_reference_i": npy(inten)})
    # line-cylinder: tests/raytracing/test_geometry.py:411-551 (targets from fixtures :345-408)
    import math
    cyl_targets = {"target_area_1_cylindrical": 2 * math.pi, "target_area_2_cylindrical": math.pi / 2}
    for i, (rays, tname, origins, ee, eu, ed, ei) in enumerate(tg.test_line_cylinder_intersection.pytestmark[0].args[1]):
        ta = TowerTargetAreasCylindrical(
            names=["c"], centers=torch.tensor([[0.0, 0.0, 0.0, 1.0]]), normals=torch.tensor([[0.0, 1.0, 0.0, 0.0]]),
            axes=torch.tensor([[0.0, 0.0, 1.0, 0.0]]), radii=torch.tensor([1.0]), heights=torch.tensor([2.0]),
            opening_angles=torch.tensor([cyl_targets[tname]]))
        r = Rays(ray_directions=rays[0], ray_magnitudes=rays[1])
        e_px, u_px, t, inten = geometry.line_cylinder_intersections(r, origins, ta, torch.tensor([0]), device=CPU)
        out.update({
            f"cyl{i}_dirs": npy(rays[0]), f"cyl{i}_mags": npy(rays[1]), f"cyl{i}_origins": npy(origins),
            f"cyl{i}_opening": np.float32(cyl_targets[tname]),
            f"cyl{i}_expected_e": npy(ee), f"cyl{i}_expected_u": npy(eu), f"cyl{i}_expected_t": npy(ed),
            f"cyl{i}_expected_i": npy(ei), f"cyl{i}_reference_e": npy(e_px), f"cyl{i}_reference_u": npy(u_px),
            f"cyl{i}_reference_t": npy(t), f"cyl{i}_reference_i": npy(inten)})
    out["cyl_count"] = np.int64(i + 1)

    # tests/raytracing/test_blocking.py:170-333 create_blocking_primitives_rectangles_by_index known answers:
    # four 5x5-point facets (origins (0,0), (2,0), (2,1), (0,2), size 2x2), flat and rotated/translated.
    facets = []
    for x, y in [(0.0, 0.0), (2.0, 0.0), (2.0, 1.0), (0.0, 2.0)]:
        gx, gy = torch.meshgrid(torch.linspace(x, x + 2.0, 5), torch.linspace(y, y + 2.0, 5), indexing="ij")
        facets.append(torch.stack([gx.flatten(), gy.flatten(), torch.zeros(25)], dim=-1))
    flat = torch.cat(facets, dim=0)
    flat = torch.cat([flat, torch.ones(flat.shape[0], 1)], dim=-1)[None]
    translation = torch.tensor([[1.0, 0.0, 0.0, 2.0], [0.0, 1.0, 0.0, 3.0], [0.0, 0.0, 1.0, 1.5], [0.0, 0.0, 0.0, 1.0]])
    moved = flat @ (translation.T @ transforms.rotate_n(n=torch.tensor([0.2]), device=CPU)
                    @ transforms.rotate_e(e=torch.tensor([0.5]), device=CPU))
    expected = [
        ([[[2.0, 1.0, 0.0, 1.0], [0.0, 2.0, 0.0, 1.0], [4.0, 2.0, 0.0, 1.0], [2.0, 2.0, 0.0, 1.0]]],
         [[[-2.0, 1.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0]]], [[0.0, 0.0, -1.0, 0.0]]),
        ([[[4.2183, 3.8341, -1.3250, 1.0], [2.2581, 4.9022, -1.4557, 1.0], [6.1784, 4.5212, -2.1531, 1.0],
           [4.2183, 4.7117, -1.8044, 1.0]]],
         [[[-1.9601, 1.0681, -0.1307, 0.0], [0.0, 0.8776, -0.4794, 0.0]]], [[-0.1987, -0.4699, -0.8601, 0.0]]),
    ]
    for i, (surf, exp) in enumerate(zip((flat, moved.reshape(1, -1, 4)), expected)):
        c_, s_, n_ = ref_blocking.create_blocking_primitives_rectangles_by_index(surf, device=CPU)
        out.update({f"prim{i}_surface": npy(surf), f"prim{i}_expected_corners": np.asarray(exp[0], np.float32),
                    f"prim{i}_expected_spans": np.asarray(exp[1], np.float32),
                    f"prim{i}_expected_normals": np.asarray(exp[2], np.float32),
                    f"prim{i}_reference_corners": npy(c_), f"prim{i}_reference_spans": npy(s_),
                    f"prim{i}_reference_normals": npy(n_)})

    # lbvh_filter_blocking_planes (blocking.py:832-995) on random rectangles and rays: the output of the reference's
    # tree build + traversal, which the brute-force box test of the restatement has to reproduce exactly.
    # Also soft_ray_blocking_mask (:212-354) of the same rays against ALL rectangles.
    g = torch.Generator().manual_seed(99)
    for i, (n_prim, n_ray) in enumerate([(1, 60), (7, 400), (40, 1500)]):
        centre = (torch.rand((n_prim, 1, 3), generator=g) - 0.5) * torch.tensor([40.0, 40.0, 6.0])
        a = torch.nn.functional.normalize(torch.randn((n_prim, 3), generator=g), dim=-1)
        b = torch.nn.functional.normalize(torch.linalg.cross(a, torch.randn((n_prim, 3), generator=g)), dim=-1)
        ha, hb = 1.0 + torch.rand((n_prim, 1), generator=g), 0.8 + torch.rand((n_prim, 1), generator=g)
        c0 = centre[:, 0] - ha * a - hb * b
        corners = torch.stack([c0, c0 + 2 * ha * a, c0 + 2 * ha * a + 2 * hb * b, c0 + 2 * hb * b], dim=1)
        corners = torch.cat([corners, torch.ones(n_prim, 4, 1)], dim=-1)
        spans = torch.zeros(n_prim, 2, 4)
        spans[:, 0], spans[:, 1] = corners[:, 1] - corners[:, 0], corners[:, 3] - corners[:, 0]
        pn = torch.cat([torch.nn.functional.normalize(torch.linalg.cross(spans[:, 0, :3], spans[:, 1, :3]), dim=-1),
                        torch.zeros(n_prim, 1)], dim=-1)
        ro = (torch.rand((1, 1, n_ray, 3), generator=g) - 0.5) * torch.tensor([50.0, 50.0, 8.0])
        aim = centre[torch.randint(0, n_prim, (n_ray,), generator=g), 0] + 1.5 * torch.randn((n_ray, 3), generator=g)
        rd = torch.nn.functional.normalize(aim[None, None] - ro, dim=-1)
        rd[0, 0, ::7, 2] = 0.0                                   # axis-parallel components (1 / (0 + 1e-12))
        tt = torch.rand((1, 1, n_ray), generator=g) * 60.0
        tt[0, 0, ::5] = 0.0                                      # rays that missed the target carry distance 0
        ro4 = torch.cat([ro, torch.ones(1, 1, n_ray, 1)], -1)
        rd4 = torch.cat([rd, torch.zeros(1, 1, n_ray, 1)], -1)
        owner = torch.randint(0, n_prim, (n_ray,), generator=g)
        filt = ref_blocking.lbvh_filter_blocking_planes(
            points_at_ray_origins=ro4[0], ray_directions=rd4, blocking_primitives_corners=corners,
            ray_to_heliostat_mapping=owner, intersection_distances_target=tt, device=CPU)
        soft = ref_blocking.soft_ray_blocking_mask(ro4[0], rd4, corners, spans, pn)
        out.update({f"lbvh{i}_corners": npy(corners), f"lbvh{i}_spans": npy(spans), f"lbvh{i}_normals": npy(pn),
                    f"lbvh{i}_origins": npy(ro[0, 0]), f"lbvh{i}_dirs": npy(rd[0, 0]), f"lbvh{i}_t": npy(tt[0, 0]),
                    f"lbvh{i}_owner": npy(owner), f"lbvh{i}_filtered": npy(filt), f"lbvh{i}_soft": npy(soft[0, 0])})
    out["lbvh_count"] = np.int64(i + 1)

    # Structure of the reference's tree on field-like layouts (rows of near-vertical rectangles at similar height):
    # which primitives hang off a path from the root (build_linear_bounding_volume_hierarchies, blocking.py:514-749).
    for i, (rows, cols) in enumerate([(2, 3), (5, 9), (17, 23), (40, 50)]):
        n_prim = rows * cols
        gy, gx = torch.meshgrid(torch.arange(rows, dtype=torch.float32), torch.arange(cols, dtype=torch.float32), indexing="ij")
        centre = torch.stack([(gx.flatten() - cols / 2) * 6.3 + 0.7 * torch.randn(n_prim, generator=g),
                              60.0 + gy.flatten() * 7.1 + 0.7 * torch.randn(n_prim, generator=g),
                              1.5 + 0.2 * torch.randn(n_prim, generator=g)], dim=-1)
        a = torch.tensor([[1.0, 0.0, 0.0]]).repeat(n_prim, 1)
        b = torch.nn.functional.normalize(torch.tensor([[0.0, 0.3, 1.0]]) + 0.05 * torch.randn((n_prim, 3), generator=g), dim=-1)
        c0 = centre - 1.6 * a - 1.3 * b
        corners = torch.stack([c0, c0 + 2.6 * b, c0 + 3.2 * a + 2.6 * b, c0 + 3.2 * a], dim=1)
        corners = torch.cat([corners, torch.ones(n_prim, 4, 1)], dim=-1)
        tree = ref_blocking.build_linear_bounding_volume_hierarchies(corners, device=CPU)
        left, right, prim = tree["left"].tolist(), tree["right"].tolist(), tree["primitive_index"].tolist()
        seen, stack, reachable = set(), [0], []
        while stack:
            node = stack.pop()
            if node < 0 or node in seen:
                continue
            seen.add(node)
            if prim[node] >= 0:
                reachable.append(prim[node])
            stack += [left[node], right[node]]
        out.update({f"tree{i}_corners": npy(corners), f"tree{i}_reachable": np.asarray(sorted(reachable), np.int64)})
    out["tree_count"] = np.int64(i + 1)

    # tests/flux/test_bitmap.py:66-173 crop_flux_distributions_around_center known answers (planar area 3 m x 3 m at
    # index 0, cylinder radius 1 m x opening 3 rad, height 3 m at index 1) + the reference's own output, and a random
    # case with autograd gradients in fp32 and fp64.
    from artist.flux import bitmap as ref_bitmap
    from artist.optim.loss import KLDivergenceLoss, PixelLoss

    def crop_tower(dtype):
        planar = TowerTargetAreasPlanar(names=["multi_focus_tower"], centers=torch.zeros(1, 4, dtype=dtype),
                                        normals=torch.zeros(1, 4, dtype=dtype),
                                        dimensions=torch.tensor([[3.0, 3.0]], dtype=dtype))
        cyl = TowerTargetAreasCylindrical(names=["receiver"], centers=torch.zeros(1, 4, dtype=dtype),
                                          normals=torch.zeros(1, 4, dtype=dtype), axes=torch.zeros(1, 4, dtype=dtype),
                                          radii=torch.tensor([1.0], dtype=dtype), heights=torch.tensor([3.0], dtype=dtype),
                                          opening_angles=torch.tensor([3.0], dtype=dtype))
        return SolarTower([planar, cyl], device=CPU)

    crop_cases = [
        ([[[1.0, 2.0, 1.0], [2.0, 3.0, 2.0], [1.0, 2.0, 1.0]], [[0.5, 0.0, 0.5], [0.5, 1.0, 0.5], [0.5, 0.0, 0.5]]], 3.0, [0, 1],
         [[[1.0, 2.0, 1.0], [2.0, 3.0, 2.0], [1.0, 2.0, 1.0]], [[0.5, 0.0, 0.5], [0.5, 1.0, 0.5], [0.5, 0.0, 0.5]]]),
        ([[[1.0, 2.0, 2.0, 1.0], [2.0, 3.0, 3.0, 2.0], [2.0, 3.0, 3.0, 2.0], [1.0, 2.0, 2.0, 1.0]]] * 2, 5.0, [0, 1],
         [[[0.0, 0.0, 0.0, 0.0], [0.0, 2.3333, 2.3333, 0.0], [0.0, 2.3333, 2.3333, 0.0], [0.0, 0.0, 0.0, 0.0]]] * 2),
        ([[[1.0, 0.0, 0.0], [0.0, 0.0, 0.0], [0.0, 0.0, 0.0]], [[0.0, 0.0, 0.0], [0.0, 0.0, 0.0], [1.0, 0.0, 0.0]]], 3.0, [0, 1],
         [[[0.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 0.0]]] * 2),
        ([[[1.0, 1.0, 1.0, 0.0, 0.0, 0.0], [1.0, 2.0, 1.0, 0.0, 0.0, 0.0], [1.0, 1.0, 1.0, 0.0, 0.0, 0.0],
           [0.0] * 6, [0.0] * 6, [0.0] * 6]], 2.0, [0],
         [[[0.1111, 0.3333, 0.3333, 0.3333, 0.3333, 0.1111], [0.3333, 1.0, 1.0, 1.0, 1.0, 0.3333],
           [0.3333, 1.0, 1.4444, 1.4444, 1.0, 0.3333], [0.3333, 1.0, 1.4444, 1.4444, 1.0, 0.3333],
           [0.3333, 1.0, 1.0, 1.0, 1.0, 0.3333], [0.1111, 0.3333, 0.3333, 0.3333, 0.3333, 0.1111]]]),
    ]
    for i, (image, crop, tix, expected) in enumerate(crop_cases):
        img = torch.tensor(image)
        got = ref_bitmap.crop_flux_distributions_around_center(img, crop_tower(torch.float32), torch.tensor(tix), crop, crop, device=CPU)
        out.update({f"crop{i}_image": npy(img), f"crop{i}_size": np.float64(crop), f"crop{i}_target_idx": np.asarray(tix),
                    f"crop{i}_dims": np.asarray([[3.0, 3.0] for _ in tix], np.float32),
                    f"crop{i}_expected": np.asarray(expected, np.float32), f"crop{i}_reference": npy(got)})
    out["crop_count"] = np.int64(i + 1)
    g2 = torch.Generator().manual_seed(4242)
    img32 = torch.rand((4, 48, 64), generator=g2) ** 4
    img32[1, :, :30] = 0.0
    img32[2] = 0.0                                       # an empty bitmap: the 1e-8 in the normalisation matters
    img32[2, 5, 7] = 1e-9
    ground_truth32 = torch.rand((4, 48, 64), generator=g2) + 0.1
    weights32 = torch.rand((4, 48, 64), generator=g2)
    for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        torch.set_default_dtype(dtype)
        img = img32.to(dtype).clone().requires_grad_(True)
        tower = crop_tower(dtype)
        tix = torch.tensor([0, 1, 0, 1])
        cropped = ref_bitmap.crop_flux_distributions_around_center(img, tower, tix, device=CPU)     # default 6 m x 6 m
        (cropped * weights32.to(dtype)).sum().backward()
        out.update({f"cropgrad_{tag}_out": npy(cropped), f"cropgrad_{tag}_grad": npy(img.grad)})
        # losses on the cropped prediction (artist/optim/loss.py:251-410), gradients w.r.t. the prediction
        for name, loss_cls in (("pixel", PixelLoss), ("kl", KLDivergenceLoss)):
            pred = (img32.to(dtype) + 0.05).clone().requires_grad_(True)
            per_sample = loss_cls()(pred, ground_truth32.to(dtype), reduction_dimensions=(1, 2))
            (per_sample * torch.tensor([1.0, 2.0, 3.0, 4.0], dtype=dtype)).sum().backward()
            out.update({f"loss_{name}_{tag}": npy(per_sample), f"loss_{name}_{tag}_grad": npy(pred.grad)})
        torch.set_default_dtype(torch.float32)
    out.update(cropgrad_image=npy(img32), cropgrad_weights=npy(weights32), cropgrad_target_idx=np.asarray([0, 1, 0, 1]),
               cropgrad_dims=np.asarray([[3.0, 3.0]] * 4, np.float32), loss_ground_truth=npy(ground_truth32),
               loss_sample_weights=np.asarray([1.0, 2.0, 3.0, 4.0], np.float32))
    # crop -> KLDivergenceLoss as ONE autograd chain (what the fused crop + KL pass replaces): loss per sample and the
    # gradient w.r.t. the uncropped bitmaps, fp32 and fp64
    for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        torch.set_default_dtype(dtype)
        img = (img32.to(dtype) + 0.01).clone().requires_grad_(True)
        cropped = ref_bitmap.crop_flux_distributions_around_center(img, crop_tower(dtype), torch.tensor([0, 1, 0, 1]), device=CPU)
        per_sample = KLDivergenceLoss()(cropped, ground_truth32.to(dtype), reduction_dimensions=(1, 2))
        (per_sample * torch.tensor([1.0, 2.0, 3.0, 4.0], dtype=dtype)).sum().backward()
        out.update({f"cropkl_{tag}_loss": npy(per_sample), f"cropkl_{tag}_grad": npy(img.grad)})
        torch.set_default_dtype(torch.float32)
    out["cropkl_offset"] = np.float64(0.01)
    # get_center_of_mass (artist/flux/bitmap.py:12-71) + FocalSpotLoss (artist/optim/loss.py:124-250):
    #  * the inline known answers of tests/optim/test_loss_functions.py:130-170 (planar area of 2 m x 2 m, index 0),
    #  * tests/geometry/test_coordinates.py:126-163 (bitmap -> target coordinates, two planar areas and a half cylinder),
    #  * a random case on a tower with a planar area and a cylinder, with autograd gradients, fp32 and fp64.
    from artist.optim.loss import FocalSpotLoss

    class _Scn:          # FocalSpotLoss reads scenario.solar_tower only
        def __init__(self, tower):
            self.solar_tower = tower

    def focal_tower(dtype, center):
        planar = TowerTargetAreasPlanar(names=["multi_focus_tower"], centers=torch.tensor(center, dtype=dtype),
                                        normals=torch.tensor([[0.0, 1.0, 0.0, 0.0]], dtype=dtype),
                                        dimensions=torch.tensor([[2.0, 2.0]], dtype=dtype))
        cyl = TowerTargetAreasCylindrical(names=["receiver"], centers=torch.zeros(1, 4, dtype=dtype),
                                          normals=torch.tensor([[0.0, 1.0, 0.0, 0.0]], dtype=dtype),
                                          axes=torch.tensor([[0.0, 0.0, 1.0, 0.0]], dtype=dtype),
                                          radii=torch.tensor([1.0], dtype=dtype), heights=torch.tensor([3.0], dtype=dtype),
                                          opening_angles=torch.tensor([3.0], dtype=dtype))
        return SolarTower([planar, cyl], device=CPU)

    focal_cases = [
        (torch.ones((1, 2, 2)), [[0.0, 0.0, 0.0, 1.0]], torch.ones((1, 2, 2)), [0.0]),
        (torch.ones((1, 2, 2)), [[1.5, 0.0, 0.0, 1.0]], torch.ones((1, 2, 2)), [0.0]),
        (torch.ones((1, 2, 2)), [[0.0, 0.0, 0.0, 1.0]], torch.tensor([[[0.0, 1.0], [0.0, 0.0]]]), [0.7071]),
    ]
    for i, (pred, center, truth, expected) in enumerate(focal_cases):
        got = FocalSpotLoss(_Scn(focal_tower(torch.float32, center)))(pred, truth, target_area_indices=torch.tensor([0]), device=CPU)
        torch.testing.assert_close(got, torch.tensor(expected), atol=1e-5, rtol=1e-6)
        out.update({f"focal{i}_prediction": npy(pred), f"focal{i}_center": np.asarray(center, np.float32),
                    f"focal{i}_ground_truth": npy(truth), f"focal{i}_expected": np.asarray(expected, np.float32),
                    f"focal{i}_reference": npy(got)})
    out["focal_count"] = np.int64(i + 1)
    from artist.geometry import coordinates as ref_coordinates

    def coord_tower(dtype):
        planar = TowerTargetAreasPlanar(names=["planar1", "planar2"],
                                        centers=torch.tensor([[0.0, 0.0, 0.0, 1.0], [1.0, 0.0, 2.0, 1.0]], dtype=dtype),
                                        normals=torch.tensor([[0.0, 1.0, 0.0, 0.0]] * 2, dtype=dtype),
                                        dimensions=torch.tensor([[6.0, 6.0], [2.0, 4.0]], dtype=dtype))
        cyl = TowerTargetAreasCylindrical(names=["cylinder1"], centers=torch.tensor([[0.0, 0.0, 0.0, 1.0]], dtype=dtype),
                                          normals=torch.tensor([[0.0, 1.0, 0.0, 0.0]], dtype=dtype),
                                          axes=torch.tensor([[0.0, 0.0, 1.0, 0.0]], dtype=dtype),
                                          radii=torch.tensor([2.0], dtype=dtype), heights=torch.tensor([6.0], dtype=dtype),
                                          opening_angles=torch.tensor([3.141592653589793], dtype=dtype))
        return SolarTower([planar, cyl], device=CPU)

    coord_cases = [
        ([[127.5, 127.5], [63.75, 255.0], [0.0, 0.0]], [0, 0, 1],
         [[0.0, 0.0, 0.0, 1.0], [1.4941, 0.0, -2.9883, 1.0], [1.9961, 0.0, 3.9922, 1.0]]),
        ([[127.5, 127.5], [127.5, 255.0], [0.0, 63.75]], [2, 2, 2],
         [[0.0, 2.0, 0.0, 1.0], [0.0, 2.0, -2.9883, 1.0], [2.0, 0.0123, 1.4941, 1.0]]),
        ([[255.0, 191.25], [255.0, 255.0]], [2, 0], [[-2.0, 0.0123, -1.4941, 1.0], [-2.9883, 0.0, -2.9883, 1.0]]),
    ]
    for i, (bc, tix, expected) in enumerate(coord_cases):
        got = ref_coordinates.bitmap_coordinates_to_target_coordinates(
            bitmap_coordinates=torch.tensor(bc), bitmap_resolution=torch.tensor([256, 256]), solar_tower=coord_tower(torch.float32),
            target_area_indices=torch.tensor(tix), device=CPU)
        torch.testing.assert_close(got, torch.tensor(expected), rtol=1e-4, atol=1e-4)
        out.update({f"coord{i}_bitmap_coordinates": np.asarray(bc, np.float32), f"coord{i}_target_idx": np.asarray(tix),
                    f"coord{i}_expected": np.asarray(expected, np.float32), f"coord{i}_reference": npy(got)})
    out["coord_count"] = np.int64(i + 1)
    g3 = torch.Generator().manual_seed(777)
    com_img32 = torch.rand((5, 40, 56), generator=g3) ** 3
    com_img32[1, :, 20:] = 0.0
    com_img32[3] = 0.0                                    # empty bitmap: (0, 0) by the 1e-8 in the normalisation
    focal_truth32 = torch.rand((5, 40, 56), generator=g3) ** 2
    com_w32 = torch.rand((5, 2), generator=g3)
    focal_tix = torch.tensor([0, 1, 2, 2, 1])
    for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        torch.set_default_dtype(dtype)
        img = com_img32.to(dtype).clone().requires_grad_(True)
        com = ref_bitmap.get_center_of_mass(img, device=CPU)
        (com * com_w32.to(dtype)).sum().backward()
        out.update({f"com_{tag}": npy(com), f"com_{tag}_grad": npy(img.grad)})
        img2 = (com_img32.to(dtype) + 0.02).clone().requires_grad_(True)
        loss = FocalSpotLoss(_Scn(coord_tower(dtype)))(img2, focal_truth32.to(dtype), target_area_indices=focal_tix, device=CPU)
        (loss * torch.tensor([1.0, 2.0, 3.0, 4.0, 5.0], dtype=dtype)).sum().backward()
        out.update({f"focalrand_{tag}_loss": npy(loss), f"focalrand_{tag}_grad": npy(img2.grad)})
        torch.set_default_dtype(torch.float32)
    out.update(com_image=npy(com_img32), com_weights=npy(com_w32), focalrand_ground_truth=npy(focal_truth32),
               focalrand_target_idx=npy(focal_tix), focalrand_offset=np.float64(0.02),
               focalrand_sample_weights=np.asarray([1.0, 2.0, 3.0, 4.0, 5.0], np.float32))
    # rotate_distortions: tests/geometry/test_transforms.py (test_distortion_rotations)
    tt = importlib.import_module("tests.geometry.test_transforms")
    k = 0
    for (e, u, rays, exp) in tt.test_distortion_rotations.pytestmark[0].args[1]:
        if exp is None:
            continue
        got = (transforms.rotate_distortions(e=e, u=u, device=CPU) @ rays.unsqueeze(-1)).squeeze(-1)
        out.update({f"rot{k}_e": npy(e), f"rot{k}_u": npy(u), f"rot{k}_rays": npy(rays),
                    f"rot{k}_expected": npy(exp), f"rot{k}_reference": npy(got)})
        k += 1
    out["rot_count"] = np.int64(k)
    # sampler partition table: tests/raytracing/test_sampling.py:8-15
    ts = importlib.import_module("tests.raytracing.test_sampling")
    rows = []
    for (ns, nh, ws, per_rank) in ts.test_distributed_sampler.pytestmark[0].args[1]:
        for rank in range(ws):
            got = list(RestrictedDistributedSampler(ns, nh, ws, rank))
            assert got == per_rank[rank]
            rows.append([ns, nh, ws, rank] + got + [-1] * (16 - len(got)))
    out["sampler_table"] = np.asarray(rows, dtype=np.int64)
    # NURBS forward known answer: tests/nurbs/test_surfaces.py:202-300
    canting = torch.tensor([[[[8.0249e-01, -0.0, -4.7736e-03, 0.0], [1.7949e-05, 6.3749e-01, 3.0172e-03, 0.0]]]])
    transl = torch.tensor([[[1.0, 0.0, 0.0, 0.0]]])
    uv = torch.cartesian_prod(torch.linspace(1e-5, 1 - 1e-5, 2), torch.linspace(1e-5, 1 - 1e-5, 2))[None, None]
    cp = create_planar_nurbs_control_points(torch.tensor([4, 4]), canting[0], device=CPU)[None]
    nurbs = NURBSSurfaces(degrees=torch.tensor([2, 2]), control_points=cp, device=CPU)
    pts, nrm = nurbs(uv, canting, transl, CPU)
    exp_p = torch.tensor([[[[1.975133419037e-01, -6.374730467796e-01, 1.756353536621e-03, 1.0],
                            [1.975492835045e-01, 6.374730467796e-01, 7.790592499077e-03, 1.0],
                            [1.802450656891e00, -6.374730467796e-01, -7.790592499077e-03, 1.0],
                            [1.802486538887e00, 6.374729871750e-01, -1.756352838129e-03, 1.0]]]])
    exp_n = torch.tensor([[[[0.005948313046, -0.004732967820, 0.999971091747, 0.0]] * 4]])
    torch.testing.assert_close(pts, exp_p)
    torch.testing.assert_close(nrm, exp_n)
    out.update(nurbsfwd_canting=npy(canting), nurbsfwd_transl=npy(transl), nurbsfwd_uv=npy(uv), nurbsfwd_cp=npy(cp),
               nurbsfwd_degrees=np.asarray([2, 2]), nurbsfwd_expected_points=npy(exp_p),
               nurbsfwd_expected_normals=npy(exp_n), nurbsfwd_reference_points=npy(pts),
               nurbsfwd_reference_normals=npy(nrm))
    # span search: tests/nurbs/test_surfaces.py:150-199
    from artist.geometry import coordinates
    ev = coordinates.normalize_points(create_nurbs_evaluation_grid(torch.tensor([4, 5]), device=CPU))
    x, y = torch.meshgrid(torch.linspace(1e-2, 1 - 1e-2, 6), torch.linspace(1e-2, 1 - 1e-2, 6), indexing="ij")
    knots = torch.tensor([0.0, 0.0, 0.0, 0.0, 0.1, 0.7, 1.0, 1.0, 1.0, 1.0])
    ns = NURBSSurfaces(degrees=torch.tensor([3, 3]), control_points=torch.stack([x, y], -1)[None, None], device=CPU)
    span = ns.find_spans(direction=0, evaluation_points=ev[None, None], knot_vectors=knots[None, None], device=CPU)
    exp_span = [3, 3, 3, 3, 3, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 5, 5, 5, 5, 5]
    assert span.flatten().tolist() == exp_span
    ns2 = NURBSSurfaces(degrees=torch.tensor([3, 3]), control_points=torch.stack([x, y], -1)[None, None],
                        uniform=False, device=CPU)
    span_nu = ns2.find_spans(direction=0, evaluation_points=ev[None, None], knot_vectors=knots[None, None], device=CPU)
    out.update(span_eval=npy(ev), span_knots=npy(knots), span_expected=np.asarray(exp_span),
               span_nonuniform_reference=npy(span_nu.flatten()))
    # sun distortions layout: tests/scene/test_sun.py:10-57 semantics (seeded MVN sample, permute)
    sun = Sun(number_of_rays=3, device=CPU)
    du, de = sun.get_distortions(number_of_points=5, number_of_active_heliostats=2, random_seed=7)
    out.update(sun_u=npy(du), sun_e=npy(de), sun_strides=np.asarray(du.stride()))
    return out


def main():
    only = None
    if len(sys.argv) > 2 and sys.argv[1] == "--only":     # regenerate a subset (every case is deterministic)
        only = set(sys.argv[2].split(","))
    if only is None or "known_answers" in only:
        save("known_answers", known_answers())
    for name, case in CASES.items():
        if only is not None and name not in only:
            continue
        arrs32, dist = run_case(name, case, dtype=torch.float32)
        save(name, arrs32)
        arrs64, _ = run_case(name, case, dtype=torch.float64, distortions_f32=dist)
        drop = {"reflected", "scattered", "distances", "stage_bitmaps", "aligned_points", "aligned_normals",
                "grad_nurbs_points", "grad_nurbs_normals", "knots_u", "knots_v"}
        save(name + "_f64", {k: v for k, v in arrs64.items() if k not in drop})
    for name, case in REAL_CASES.items():
        if only is not None and name not in only:
            continue
        arrs32, dist = run_scenario_file(dtype=torch.float32, **case)
        save(name, arrs32)
        arrs64, _ = run_scenario_file(dtype=torch.float64, distortions_f32=dist, **case)
        save(name + "_f64", arrs64)
    if only is None or "interop" in only:
        save_interop_check()
    if only is None or "kinematics" in only:
        kinematics_fixture()
    if only is None or "kinematics_reconstructor_epochs" in only:
        k32 = kinematics_reconstructor_epochs(dtype=torch.float32)
        k64 = kinematics_reconstructor_epochs(dtype=torch.float64)
        for key in ("loss_per_sample", "grad", "rotation_after", "total_loss", "flux_predicted"):
            k32[key + "_f64_epoch0"] = k64[key][0]
        torch.set_default_dtype(torch.float32)
        save("kinematics_reconstructor_epochs", k32)
    if only is None or "surface_reconstructor_epochs" in only:
        a32 = surface_reconstructor_epochs(dtype=torch.float32)
        a64 = surface_reconstructor_epochs(dtype=torch.float64)
        # fp64 run, first epoch (same start, same rays): the yardstick; the measured flux of the fp64 run differs from the fp32
        # run's in the last digits, so its loss and gradient are those of (almost) the same problem
        for key in ("cropped_flux", "flux_loss_per_sample", "grad_locked", "cp_after", "total_loss"):
            a32[key + "_f64_epoch0"] = a64[key][0]
        torch.set_default_dtype(torch.float32)
        save("surface_reconstructor_epochs", a32)
    # config 1 / config 2: inputs regenerate from the recipe (seeded torch CPU RNG); store outputs only.
    keep = {"flux", "intercept", "on_target", "blocking", "per_target", "control_points", "orientation",
            "aligned_points", "aligned_normals", "incident", "target_idx", "target_centers", "target_normals",
            "target_dims", "resolution", "ray_magnitude", "extinction", "reflectivity", "n_rays", "seed", "covariance",
            "degrees", "eval_points_grid", "canting", "facet_translations", "nurbs_points", "nurbs_normals"}
    for name, case in (("config1", CONFIG1), ("config2", CONFIG2), ("config2_seed11", CONFIG2_SEED11),
                       ("config2_offaxis", CONFIG2_OFFAXIS)):
        if only is not None and name not in only:
            continue
        arrs32, dist = run_case(name, case, with_grads=False, store_rays=False, dtype=torch.float32)
        a = summarize_large(arrs32, keep)
        a["eval_points_grid"] = arrs32["eval_points"][0, 0]
        arrs64, _ = run_case(name, case, with_grads=False, store_rays=False, dtype=torch.float64, distortions_f32=dist)
        a["flux_f64"] = arrs64["flux"]
        # aligned points are 160 KB each - keep (they are the op's inputs); drop the per-facet duplicates
        a.pop("nurbs_points"), a.pop("nurbs_normals")
        save(name, a)


if __name__ == "__main__":
    main()
