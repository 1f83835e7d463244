"""GPU parity suite (``-m gpu``): the HIP path, called through the C ABI, against
  (1) the CPU oracle on the same seeded inputs,
  (2) the golden fixtures generated from the imported reference, and
  (3) size-independent properties at BASELINE.json's full sizes.
Tolerances are fp32 tolerances and are written next to each assertion.  Nothing here reads
/root/reference.
"""
import numpy as np
import pytest
import torch

import oracle
from conftest import (BLOCKING_CASES, CYL_CASES, KINEMATICS_CASES, REAL_CASES, ROOT, STAGE_CASES, kinematics_case, rel_l2,
                      sun_distortions)

pytestmark = pytest.mark.gpu

DEV = torch.device("cuda:0")


def t(x, dtype=None):
    out = torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
    return out if dtype is None else out.to(dtype)


def n(x):
    return x.detach().cpu().numpy()


def interleave(du, de):
    """Return (u, e) as stride-2 views of ONE device buffer, like Sun.get_distortions."""
    both = torch.stack((t(du), t(de)), dim=-1).contiguous()
    return both[..., 0], both[..., 1]


def trace_inputs(d, interleaved=True):
    du, de = (interleave(d["distortions_u"], d["distortions_e"]) if interleaved
              else (t(d["distortions_u"]).contiguous(), t(d["distortions_e"]).contiguous()))
    return dict(origins=t(d["aligned_points"]), normals=t(d["aligned_normals"]), incident=t(d["incident"]),
                dist_u=du, dist_e=de, target_idx=t(d["target_idx"]), centers=t(d["target_centers"]),
                plane_normals=t(d["target_normals"]), dims=t(d["target_dims"]),
                ray_magnitude=float(d["ray_magnitude"]), extinction=float(d["extinction"]),
                reflectivity=float(d["reflectivity"]), resolution=tuple(int(v) for v in d["resolution"]))


def oracle_fwd(d, **kw):
    return oracle.trace_fwd(d["aligned_points"], d["aligned_normals"], d["incident"], d["distortions_u"],
                            d["distortions_e"], d["target_idx"], d["target_centers"], d["target_normals"],
                            d["target_dims"], d["resolution"], float(d["ray_magnitude"]), float(d["extinction"]),
                            float(d["reflectivity"]), **kw)


def test_library_is_the_hip_one():
    from artist_amd import _lib
    assert _lib.lib().art_abi_version() == _lib.ABI_VERSION
    assert len(_lib.loaded_hip_runtimes()) == 1, _lib.loaded_hip_runtimes()


@pytest.mark.parametrize("name", STAGE_CASES)
@pytest.mark.parametrize("interleaved", [True, False])
def test_trace_forward(golden, name, interleaved):
    from artist_amd import trace_rays
    d = golden(name)
    flux, fac = trace_rays(**trace_inputs(d, interleaved))
    o_flux, o_fac = oracle_fwd(d)
    # vs oracle: same op order; only sinf/cosf (<=1 ULP) and the atomic summation order differ.
    # Tiny cases have O(1) rays per pixel so per-ray 1e-4 px noise is not averaged: 2e-4 relative L2.
    assert rel_l2(n(flux), o_flux) < 2e-4, rel_l2(n(flux), o_flux)
    assert rel_l2(n(flux), d["flux"]) < 2e-4, rel_l2(n(flux), d["flux"])
    np.testing.assert_array_equal(n(fac), o_fac)          # ray counts are integers: exact
    np.testing.assert_array_equal(n(fac[0]), d["intercept"])
    np.testing.assert_array_equal(n(fac[1]), d["on_target"])
    np.testing.assert_array_equal(n(fac[2]), d["blocking"])


@pytest.mark.parametrize("name", STAGE_CASES)
def test_trace_per_target_mode(golden, name):
    from artist_amd import per_target_sum, trace_rays
    d = golden(name)
    T = d["target_centers"].shape[0]
    inp = trace_inputs(d)
    flux_h, _ = trace_rays(**inp)
    flux_t, fac = trace_rays(**inp, per_target=True)
    summed = per_target_sum(flux_h, inp["target_idx"], T)
    assert flux_t.shape == (T, int(d["resolution"][1]), int(d["resolution"][0]))
    scale = float(np.abs(d["per_target"]).max()) + 1e-30
    np.testing.assert_allclose(n(flux_t), n(summed), rtol=0, atol=2e-6 * scale)    # summation order only
    assert rel_l2(n(summed), d["per_target"]) < 2e-4
    np.testing.assert_allclose(n(summed), oracle.per_target(n(flux_h), d["target_idx"], T), rtol=1e-6, atol=1e-6 * scale)


@pytest.mark.parametrize("name", STAGE_CASES)
def test_trace_backward(golden, name):
    from artist_amd import trace_rays
    d, d64 = golden(name), golden(name + "_f64")
    inp = trace_inputs(d)
    inp["origins"].requires_grad_(True)
    inp["normals"].requires_grad_(True)
    flux, _ = trace_rays(**inp)
    (flux * t(d["loss_weights"])).sum().backward()
    go, gn = oracle.trace_bwd(d["aligned_points"], d["aligned_normals"], d["incident"], d["distortions_u"],
                              d["distortions_e"], d["target_idx"], d["target_centers"], d["target_normals"],
                              d["target_dims"], d["resolution"], d["loss_weights"], float(d["ray_magnitude"]),
                              float(d["extinction"]), float(d["reflectivity"]))
    for got, orc, key in ((inp["origins"].grad, go, "grad_aligned_points"), (inp["normals"].grad, gn, "grad_aligned_normals")):
        ref = d[key]
        if np.linalg.norm(ref) == 0:
            assert float(got.abs().sum()) == 0
            continue
        # yardstick: the reference's own fp32-vs-fp64 gradient error (cell flips at pixel borders)
        ref_err = rel_l2(ref, d64[key])
        assert rel_l2(n(got), orc) < max(ref_err, 2e-4), (key, rel_l2(n(got), orc))
        assert rel_l2(n(got), ref) < max(3 * ref_err, 2e-4), (key, rel_l2(n(got), ref), ref_err)
        assert float(got[..., 3].abs().max()) == 0 or key == "grad_aligned_normals"


@pytest.mark.parametrize("name", STAGE_CASES + ["known"])
def test_nurbs_forward_backward(golden, name):
    from artist_amd import NURBSSurfaces
    if name == "known":   # tests/nurbs/test_surfaces.py:202-300 known answer, through the mirror class
        ka = golden("known_answers")
        surf = NURBSSurfaces(torch.tensor([2, 2]), t(ka["nurbsfwd_cp"]), device=DEV)
        pts, nrm = surf(t(ka["nurbsfwd_uv"]), t(ka["nurbsfwd_canting"]), t(ka["nurbsfwd_transl"]), DEV)
        torch.testing.assert_close(pts.cpu(), torch.from_numpy(ka["nurbsfwd_expected_points"]))
        torch.testing.assert_close(nrm.cpu(), torch.from_numpy(ka["nurbsfwd_expected_normals"]))
        return
    d = golden(name)
    cp = t(d["control_points"]).requires_grad_(True)
    surf = NURBSSurfaces(torch.from_numpy(d["degrees"]), cp, device=DEV)
    assert np.array_equal(n(surf.knot_vectors_u[0, 0]), d["knots_u"])
    pts, nrm = surf.calculate_surface_points_and_normals(t(d["eval_points"]), t(d["canting"]), t(d["facet_translations"]))
    o_pts, o_nrm = oracle.nurbs_fwd(d["control_points"], d["eval_points"], d["degrees"], d["canting"], d["facet_translations"])
    np.testing.assert_array_equal(n(pts), o_pts)                              # same op order: bit-exact
    np.testing.assert_array_equal(n(pts), d["nurbs_points"])                  # ... and equal to the reference
    np.testing.assert_allclose(n(nrm), d["nurbs_normals"], rtol=0, atol=1.2e-7)   # 1 ULP (sqrt / vector_norm)
    gp, gn = d["grad_nurbs_points"].reshape(d["nurbs_points"].shape), d["grad_nurbs_normals"].reshape(d["nurbs_normals"].shape)
    torch.autograd.backward([pts, nrm], [t(gp), t(gn)])
    o_g = oracle.nurbs_bwd(d["control_points"], d["eval_points"], d["degrees"], gp, gn, d["canting"])
    if np.linalg.norm(o_g) > 0:
        assert rel_l2(n(cp.grad), o_g) < 1e-5, rel_l2(n(cp.grad), o_g)               # LDS-atomic summation order
        assert rel_l2(n(cp.grad), d["grad_control_points"]) < 2e-5


@pytest.mark.parametrize("name", STAGE_CASES)
def test_nurbs_with_fused_alignment(golden, name):
    """``calculate_surface_points_and_normals(..., orientations=M)`` = evaluation + ``align_surfaces`` in one kernel:
    the same bits forward, the same control-point gradients (surfaces.py:475-689 + heliostat_group_rigid_body.py:217-222)."""
    from artist_amd import NURBSSurfaces, align_surfaces
    d = golden(name)
    ori = t(d["orientation"])
    H = ori.shape[0]
    args = (t(d["eval_points"]), t(d["canting"]), t(d["facet_translations"]))
    cp_a = t(d["control_points"]).requires_grad_(True)
    pts, nrm = NURBSSurfaces(torch.from_numpy(d["degrees"]), cp_a, device=DEV)(*args)
    ap, an = align_surfaces(pts.reshape(H, -1, 4), nrm.reshape(H, -1, 4), ori)
    cp_b = t(d["control_points"]).requires_grad_(True)
    fp, fn = NURBSSurfaces(torch.from_numpy(d["degrees"]), cp_b, device=DEV).calculate_surface_points_and_normals(
        *args, orientations=ori)
    np.testing.assert_array_equal(n(fp.reshape(H, -1, 4)), n(ap))
    np.testing.assert_array_equal(n(fn.reshape(H, -1, 4)), n(an))
    np.testing.assert_allclose(n(ap), d["aligned_points"], rtol=0, atol=2e-5)          # the reference's aligned surface
    gen = torch.Generator(device=DEV).manual_seed(2)
    wp, wn = torch.rand(ap.shape, generator=gen, device=DEV), torch.rand(an.shape, generator=gen, device=DEV)
    ((ap * wp).sum() + (an * wn).sum()).backward()
    ((fp.reshape(H, -1, 4) * wp).sum() + (fn.reshape(H, -1, 4) * wn).sum()).backward()
    assert rel_l2(n(cp_b.grad), n(cp_a.grad)) < 1e-6          # (fp64 LDS sums in a different order: a last-bit difference)


def test_nurbs_broadcast_grid_and_no_canting(golden):
    """Expanded (stride-0) evaluation grid, canting=None branch (surfaces.py:689), mixed degrees (generic kernel)."""
    from artist_amd import NURBSSurfaces
    g = torch.Generator().manual_seed(3)
    cp = torch.rand(2, 3, 6, 5, 3, generator=g)
    uv = torch.rand(37, 2, generator=g) * 0.98 + 0.01
    for deg in ([3, 3], [2, 3], [5, 2], [1, 1]):
        surf = NURBSSurfaces(torch.tensor(deg), cp.to(DEV), device=DEV)
        uvx = uv.to(DEV)[None, None].expand(2, 3, -1, -1)
        pts, nrm = surf(uvx, None, None)
        o_pts, o_nrm = oracle.nurbs_fwd(cp.numpy(), uvx.cpu().contiguous().numpy(), deg)
        np.testing.assert_allclose(n(pts), o_pts, rtol=0, atol=1e-6)
        np.testing.assert_allclose(n(nrm), o_nrm, rtol=0, atol=2e-6)


@pytest.mark.parametrize("name", STAGE_CASES)
def test_end_to_end_autograd_to_control_points(golden, name):
    """control points -> NURBS (HIP) -> alignment (torch bmm) -> trace (HIP) -> loss; d loss / d control points and
    d loss / d orientation vs torch.autograd of the reference (north_star: grads w.r.t. NURBS control points and
    kinematic parameters - the kinematic chain enters through the orientation matrices)."""
    from artist_amd import NURBSSurfaces, trace_rays
    d, d64 = golden(name), golden(name + "_f64")
    cp = t(d["control_points"]).requires_grad_(True)
    ori = t(d["orientation"]).requires_grad_(True)
    H = ori.shape[0]
    pts, nrm = NURBSSurfaces(torch.from_numpy(d["degrees"]), cp, device=DEV)(
        t(d["eval_points"]), t(d["canting"]), t(d["facet_translations"]))
    ap = pts.reshape(H, -1, 4) @ ori.transpose(1, 2)
    an = nrm.reshape(H, -1, 4) @ ori.transpose(1, 2)
    inp = trace_inputs(d)
    inp.update(origins=ap, normals=an)
    flux, _ = trace_rays(**inp)
    (flux * t(d["loss_weights"])).sum().backward()
    for got, key in ((cp.grad, "grad_control_points"), (ori.grad, "grad_orientation")):
        ref = d[key]
        if np.linalg.norm(ref) == 0:
            assert float(got.abs().sum()) == 0
            continue
        ref_err = rel_l2(ref, d64[key])
        assert rel_l2(n(got), ref) < max(3 * ref_err, 5e-4), (key, rel_l2(n(got), ref), ref_err)
    # ... and on to the kinematic parameters: d(orientation)/d(rotation, translation deviations) of the reference's
    # rigid-body kinematics is stored with the fixture (torch.autograd.functional.jacobian of
    # kinematics_rigid_body.py:540-634), the chain rule does the rest
    for jac, key in (("kin_jac_rot", "grad_kin_rot"), ("kin_jac_trans", "grad_kin_trans")):
        got = torch.einsum("hij,hijk->hk", ori.grad, t(d[jac]))
        ref, ref64 = d[key], d64[key]
        if np.linalg.norm(ref64) == 0:
            assert float(got.abs().sum()) == 0
            continue
        ref_err = rel_l2(ref, ref64)
        assert rel_l2(n(got), ref) < max(3 * ref_err, 5e-4), (key, rel_l2(n(got), ref), ref_err)
        assert rel_l2(n(got), ref64) < max(4 * ref_err, 5e-4), (key, rel_l2(n(got), ref64), ref_err)


@pytest.mark.parametrize("name,tol", [("config1", 1e-6), ("config2", 1e-5)])
def test_baseline_configs_flux_l2(golden, name, tol):
    """BASELINE.json configs 1 and 2 on the GPU vs the reference PyTorch-CPU flux (north_star: < 1e-5)."""
    from artist_amd import trace_rays
    d = golden(name)
    H, P = d["aligned_points"].shape[:2]
    du, de = sun_distortions(H, int(d["n_rays"]), P, float(d["covariance"]), seed=int(d["seed"]))
    dd = dict(d, distortions_u=du.numpy(), distortions_e=de.numpy())
    flux, fac = trace_rays(**trace_inputs(dd))
    err = rel_l2(n(flux), d["flux"])
    print(f"{name}: flux rel L2 vs reference = {err:.3e}; reference fp32-vs-fp64 = {rel_l2(d['flux'], d['flux_f64']):.3e}")
    assert err < tol, err
    np.testing.assert_array_equal(n(fac[0]), d["intercept"])
    np.testing.assert_array_equal(n(fac[1]), d["on_target"])


# ---- properties at full size (H=8 heliostats x 10^6 rays each; same kernel paths as H=1000) ----------------
@pytest.fixture(scope="module")
def field():
    from artist_amd.scene import build_synthetic_scenario
    H = 8
    scenario, uv = build_synthetic_scenario(H, n_rays=100, device=DEV)
    group = scenario.heliostat_field.heliostat_groups[0]
    mask = torch.ones(H, dtype=torch.int32, device=DEV)
    group.activate_heliostats(mask)
    tix = torch.zeros(H, dtype=torch.long, device=DEV)
    inc = torch.tensor([[0.0, 1.0, 0.0, 0.0]], device=DEV).repeat(H, 1)
    group.align_surfaces_with_incident_ray_directions(scenario.solar_tower.get_centers_of_target_areas(tix), inc, mask)
    return scenario, group, mask, tix, inc


def test_full_size_properties(field):
    from artist_amd import HeliostatRayTracer
    scenario, group, mask, tix, inc = field
    rt = HeliostatRayTracer(scenario, group, blocking_active=False)
    flux, intercept, on_target, blocking = rt.trace_rays(inc, mask, tix)
    H = flux.shape[0]
    assert flux.shape == (H, 256, 256) and bool(torch.isfinite(flux).all()) and float(flux.min()) >= 0
    # energy: bilinear weights sum to 1, Lambert factor <= 1  =>  sum(flux) <= reflectivity * splatted rays
    rays = 100 * group.active_surface_points.shape[1]
    total = flux.sum((1, 2))
    assert bool((total <= 0.935 * intercept * rays * (1 + 1e-5)).all())
    assert bool((total >= 0.25 * 0.935 * intercept * rays).all())           # Lambert cosine >= 0.25 on this fan
    assert bool((intercept <= on_target).all()) and bool((blocking == 1).all())
    # linearity in reflectivity / extinction (pure scaling of every ray)
    flux2, *_ = rt.trace_rays(inc, mask, tix, ray_extinction_factor=0.5, mirror_reflectivity=0.935)
    assert rel_l2(n(flux2), 0.5 * n(flux)) < 1e-6
    # per-target fused mode == sum of per-heliostat bitmaps
    pt, *_ = rt.trace_rays_per_target(inc, mask, tix)
    assert rel_l2(n(pt[0]), n(flux.sum(0))) < 1e-6
    assert rel_l2(n(rt.get_bitmaps_per_target(flux, tix)[0]), n(flux.sum(0))) < 1e-6
    # rank sharding: sum over ranks of the per-target bitmaps == single rank (SURVEY 8e invariant)
    acc = torch.zeros_like(pt)
    rows = []
    for rank in range(3):
        rtr = HeliostatRayTracer(scenario, group, blocking_active=False, world_size=3, rank=rank)
        f_local, *_ = rtr.trace_rays(inc, mask, tix)
        idx = rtr.get_sampler_indices()
        rows.append(idx)
        assert f_local.shape[0] == idx.numel()
        np.testing.assert_allclose(n(f_local), n(flux[idx]), rtol=0, atol=2e-6 * float(flux.max()))
        acc += rtr.get_bitmaps_per_target(f_local, tix[idx])
    assert sorted(torch.cat(rows).tolist()) == list(range(H))
    assert rel_l2(n(acc), n(pt)) < 1e-6
    # one heliostat against the oracle at full size (10^6 rays)
    planar = scenario.solar_tower.target_areas[0]
    o_flux, o_fac = oracle.trace_fwd(n(group.active_surface_points[:1]), n(group.active_surface_normals[:1]), n(inc[:1]),
                                     n(rt.distortions_dataset.distortions_u[:1]), n(rt.distortions_dataset.distortions_e[:1]),
                                     n(tix[:1]), n(planar.centers), n(planar.normals), n(planar.dimensions), (256, 256))
    assert rel_l2(n(flux[:1]), o_flux) < 1e-5, rel_l2(n(flux[:1]), o_flux)
    np.testing.assert_array_equal(n(intercept[:1]), o_fac[0])


def test_error_behaviour(field):
    from artist_amd import ArtistHipError, HeliostatRayTracer, trace_rays
    scenario, group, mask, tix, inc = field
    rt = HeliostatRayTracer(scenario, group, blocking_active=False)
    with pytest.raises(AssertionError, match="Some heliostats were not aligned and cannot be raytraced."):
        rt.trace_rays(inc, torch.zeros_like(mask), tix)           # tests/raytracing/test_heliostat_ray_tracer.py:44-104
    with pytest.raises(IndexError):
        rt.trace_rays(inc, mask, tix + 5)
    with pytest.raises(IndexError):                                  # the per-target entry point checks too
        rt.trace_rays_per_target(inc, mask, tix + 5)
    cpu = [torch.zeros(1, 4, 4), torch.zeros(1, 4, 4), torch.zeros(1, 4), torch.zeros(1, 2, 4), torch.zeros(1, 2, 4),
           torch.zeros(1, dtype=torch.long), torch.zeros(1, 4), torch.zeros(1, 4), torch.ones(1, 2)]
    with pytest.raises(ArtistHipError, match="no CPU fallback"):
        trace_rays(*cpu)
    with pytest.raises(ValueError, match="must have the same shape"):   # artist/geometry/transforms.py:47-50
        trace_rays(*[c.to(DEV) for c in cpu[:4]], torch.zeros(1, 3, 4, device=DEV), *[c.to(DEV) for c in cpu[5:]])


# ---- kernel-variant / rare-branch coverage --------------------------------------------------------------
@pytest.mark.parametrize("env", [
    {},                                                     # production geometry
    {"ARTIST_HIP_FWD": "global"},                           # plain global-atomic kernel
    {"ARTIST_HIP_FWD_TILE_KB": "4"},                        # window overflow -> centre in LDS, tails global
    {"ARTIST_HIP_FWD_BLOCK": "256", "ARTIST_HIP_FWD_TILE_KB": "36", "ARTIST_HIP_FWD_BLOCKS": "4096",
     "ARTIST_HIP_FWD_MINCHUNK": "1"},                       # many small workgroups, 1-sample chunks
])
def test_forward_variants_agree(golden, monkeypatch, env):
    """Every launch geometry of the forward kernel must give the same bitmap (the LDS window is a pure
    performance device: rays outside it take the global-atomic path)."""
    from artist_amd import trace_rays
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for name in ("mid_256", "small_deg2_tilted", "small_offtarget"):
        d = golden(name)
        flux, fac = trace_rays(**trace_inputs(d))
        o_flux, o_fac = oracle_fwd(d)
        assert rel_l2(n(flux), o_flux) < 2e-4, (name, env, rel_l2(n(flux), o_flux))
        np.testing.assert_array_equal(n(fac), o_fac)
        # the backward kernels share the launch geometry (LDS-staged gradient window / plain gathers)
        inp = trace_inputs(d)
        inp["origins"].requires_grad_(True)
        inp["normals"].requires_grad_(True)
        (trace_rays(**inp)[0] * t(d["loss_weights"])).sum().backward()
        go, gn = oracle.trace_bwd(d["aligned_points"], d["aligned_normals"], d["incident"], d["distortions_u"],
                                  d["distortions_e"], d["target_idx"], d["target_centers"], d["target_normals"],
                                  d["target_dims"], d["resolution"], d["loss_weights"], float(d["ray_magnitude"]),
                                  float(d["extinction"]), float(d["reflectivity"]))
        if np.linalg.norm(go) > 0:
            d64 = golden(name + "_f64")
            tol = max(rel_l2(d["grad_aligned_points"], d64["grad_aligned_points"]), 2e-4)
            assert rel_l2(n(inp["origins"].grad), go) < tol, (name, env)
            assert rel_l2(n(inp["normals"].grad), gn) < tol, (name, env)


@pytest.mark.parametrize("name", ["mid_256", "small_deg2_tilted", "small_offtarget", "small_deg3"])
def test_bitmap_does_not_depend_on_the_launch_geometry(golden, monkeypatch, name):
    """One cell unit per launch, and a stray ray is rounded to it like a window ray: a ray contributes the same integer
    whichever workgroup traces it and whether or not it meets a window.  Per ray body (the lean one, and the generic one that
    also serves blocking and cylinders - whose products the plain global-atomic formulation repeats) every geometry gives
    the same BITS: full windows, 4 KB windows (most rays stray), many small workgroups, no window at all."""
    from artist_amd import trace_rays
    d = golden(name)

    def run(env):
        with monkeypatch.context() as m:
            for k, v in env.items():
                m.setenv(k, v)
            return n(trace_rays(**trace_inputs(d))[0])

    lean = run({})
    for env in ({"ARTIST_HIP_FWD_TILE_KB": "4"}, {"ARTIST_HIP_FWD_PBLOCK": "192"},
                {"ARTIST_HIP_FWD_TILE_KB": "36", "ARTIST_HIP_FWD_BLOCKS": "4096", "ARTIST_HIP_FWD_MINCHUNK": "1"}):
        np.testing.assert_array_equal(run(env), lean, err_msg=str(env))
    generic = run({"ARTIST_HIP_LEAN": "0"})
    for env in ({"ARTIST_HIP_LEAN": "0", "ARTIST_HIP_FWD_TILE_KB": "4"}, {"ARTIST_HIP_LEAN": "0", "ARTIST_HIP_FWD_BLOCK": "256"},
                {"ARTIST_HIP_FWD": "global"}):
        np.testing.assert_array_equal(run(env), generic, err_msg=str(env))
    assert rel_l2(lean, generic) < 1e-6


def test_large_scatter_angles_take_the_full_range_path():
    """|angle| > 2^-3 rad leaves the small-angle sin/cos kernel (ray_math.hpp: sincos_angle); force it with a
    0.3 rad sun shape so that most rays use the OCML branch, and compare with the oracle ray by ray via the
    bitmap (64x64 on a 400 m x 400 m target so that widely scattered rays still land)."""
    from artist_amd import trace_rays
    g = torch.Generator().manual_seed(11)
    H, R, P = 2, 16, 600
    origins = torch.cat([torch.rand(H, P, 3, generator=g) * 2 - 1 + torch.tensor([0.0, 80.0, 0.0]), torch.ones(H, P, 1)], -1)
    nrm = torch.nn.functional.normalize(torch.tensor([0.0, -0.8, 0.6]) + 0.01 * torch.randn(H, P, 3, generator=g), dim=-1)
    normals = torch.cat([nrm, torch.zeros(H, P, 1)], -1)
    incident = torch.tensor([[0.0, 1.0, 0.0, 0.0]]).repeat(H, 1)
    both = 0.3 * torch.randn(H, R, P, 2, generator=g)
    both[0, 0, :10] = 0.0                                    # a few exact-zero and threshold angles
    both[0, 1, :10, 0] = 0.125
    both[0, 1, :10, 1] = -0.125
    centers = torch.tensor([[0.0, 0.0, 60.0, 1.0]])
    pn = torch.tensor([[0.0, 1.0, 0.0, 0.0]])
    dims = torch.tensor([[400.0, 400.0]])
    tix = torch.zeros(H, dtype=torch.long)
    bd = both.to(DEV)
    flux, fac = trace_rays(origins.to(DEV), normals.to(DEV), incident.to(DEV), bd[..., 0], bd[..., 1], tix.to(DEV),
                           centers.to(DEV), pn.to(DEV), dims.to(DEV), resolution=(64, 64))
    assert float((both.abs() > 0.125).float().mean()) > 0.5
    o_flux, o_fac = oracle.trace_fwd(origins.numpy(), normals.numpy(), incident.numpy(), both[..., 0].numpy(),
                                     both[..., 1].numpy(), tix.numpy(), centers.numpy(), pn.numpy(), dims.numpy(), (64, 64))
    assert o_fac[0].min() > 0.2                              # plenty of rays land
    assert rel_l2(n(flux), o_flux) < 1e-5, rel_l2(n(flux), o_flux)
    np.testing.assert_array_equal(n(fac), o_fac)


def test_fixed_point_cells_wrap_correctly():
    """The LDS window accumulates in 32-bit fixed point (quantum 2^-22 of the largest contribution) and pushes
    a carry to the global pixel when a cell wraps.  Force wraps: 2048 identical mirror points x 64 samples with a
    zero-width sun put 131072 rays on the same four pixels (a cell wraps every ~2^10 full-size contributions)."""
    from artist_amd import trace_rays
    H, R, P = 1, 64, 2048
    origins = torch.tensor([0.013, 50.0, 0.021, 1.0]).repeat(H, P, 1)
    to_target = torch.nn.functional.normalize(torch.tensor([0.3 + 0.41, 0.0, 66.0 - 0.77]) - origins[0, 0, :3], dim=0)
    nvec = torch.nn.functional.normalize(to_target - torch.tensor([0.0, 1.0, 0.0]), dim=0)   # bisects -incident, target
    normals = torch.cat([nvec, torch.zeros(1)]).repeat(H, P, 1)
    incident = torch.tensor([[0.0, 1.0, 0.0, 0.0]])
    zeros = torch.zeros(H, R, P, 2)
    centers = torch.tensor([[0.3, 0.0, 66.0, 1.0]])
    pn = torch.tensor([[0.0, 1.0, 0.0, 0.0]])
    dims = torch.tensor([[7.0, 9.0]])
    tix = torch.zeros(H, dtype=torch.long)
    for mag in (1.0, -2.5, 3e-4):
        z = zeros.to(DEV)
        flux, fac = trace_rays(origins.to(DEV), normals.to(DEV), incident.to(DEV), z[..., 0], z[..., 1], tix.to(DEV),
                               centers.to(DEV), pn.to(DEV), dims.to(DEV), ray_magnitude=mag, resolution=(64, 48))
        # expected: every ray is identical, so the bitmap is N x (one ray's four fp32 contributions), summed exactly
        one, _ = oracle.trace_fwd(origins[:, :1].numpy(), normals[:, :1].numpy(), incident.numpy(),
                                  np.zeros((1, 1, 1), np.float32), np.zeros((1, 1, 1), np.float32), tix.numpy(),
                                  centers.numpy(), pn.numpy(), dims.numpy(), (64, 48), mag)
        assert np.count_nonzero(one) == 4
        expected = one.astype(np.float64) * (R * P)
        # identical rays round identically (no averaging): error <= half a quantum = 2^-23 of the contribution bound each
        np.testing.assert_allclose(n(flux), expected, rtol=1e-6, atol=2.0 ** -22 * abs(mag) * R * P)
        assert float(fac[0]) == (1.0 if mag > 0 else 0.0)


@pytest.mark.parametrize("name", STAGE_CASES)
def test_align_surfaces(golden, name):
    """art_align_fwd/bwd vs the reference's two bmm calls (heliostat_group_rigid_body.py:217-222) and their autograd."""
    from artist_amd import align_surfaces
    d = golden(name)
    H = d["orientation"].shape[0]
    pts = t(d["nurbs_points"].reshape(H, -1, 4)).requires_grad_(True)
    nrm = t(d["nurbs_normals"].reshape(H, -1, 4)).requires_grad_(True)
    ori = t(d["orientation"]).requires_grad_(True)
    ap, an = align_surfaces(pts, nrm, ori)
    # reference used a BLAS bmm (unspecified accumulation order): a few ULP of ~100 m coordinates
    np.testing.assert_allclose(n(ap), d["aligned_points"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(n(an), d["aligned_normals"], rtol=0, atol=3e-7)
    gp, gn = t(d["grad_aligned_points"]), t(d["grad_aligned_normals"])
    torch.autograd.backward([ap, an], [gp, gn])
    p2, n2, o2 = (x.detach().clone().requires_grad_(True) for x in (pts, nrm, ori))
    torch.autograd.backward([p2 @ o2.transpose(1, 2), n2 @ o2.transpose(1, 2)], [gp, gn])
    for got, want in ((pts.grad, p2.grad), (nrm.grad, n2.grad), (ori.grad, o2.grad)):
        scale = float(want.abs().max()) + 1e-30
        np.testing.assert_allclose(n(got), n(want), rtol=0, atol=2e-5 * scale)
    if np.linalg.norm(d["grad_orientation"]) > 0:
        assert rel_l2(n(ori.grad), d["grad_orientation"]) < 1e-4
        assert rel_l2(n(pts.grad), d["grad_nurbs_points"]) < 1e-5


# ---- edge cases the reference's tests exercise: empty, ragged, minimal and non-square shapes -------------------
def _random_scene(H, R, P, W, Hh, seed=0, T=2, spread=2e-3):
    g = torch.Generator().manual_seed(seed)
    origins = torch.cat([torch.rand(H, P, 3, generator=g) * 3 - 1.5 + torch.tensor([0.0, 60.0, 0.0]), torch.ones(H, P, 1)], -1)
    aim = torch.tensor([0.0, 0.0, 40.0])
    to_aim = torch.nn.functional.normalize(aim - origins[..., :3], dim=-1)
    nrm = torch.nn.functional.normalize(to_aim - torch.tensor([0.0, 1.0, 0.0]) + 2e-3 * torch.randn(H, P, 3, generator=g), dim=-1)
    normals = torch.cat([nrm, torch.zeros(H, P, 1)], -1)
    incident = torch.tensor([[0.0, 1.0, 0.0, 0.0]]).repeat(H, 1)
    both = spread * torch.randn(H, R, P, 2, generator=g)
    centers = torch.tensor([[0.0, 0.0, 40.0, 1.0], [0.5, -1.0, 41.0, 1.0]])[:T]
    pn = torch.nn.functional.normalize(torch.tensor([[0.0, 1.0, 0.0, 0.0], [0.1, 1.0, 0.2, 0.0]]), dim=1)[:T]
    dims = torch.tensor([[5.0, 4.0], [3.0, 6.0]])[:T]
    tix = torch.arange(H) % T
    return origins, normals, incident, both, tix, centers, pn, dims, (W, Hh)


@pytest.mark.parametrize("H,R,P,W,Hh", [(1, 1, 1, 2, 2), (3, 1, 63, 5, 3), (2, 7, 65, 17, 300), (5, 3, 1025, 64, 64),
                                        (1, 33, 2049, 256, 8), (4, 2, 4097, 33, 47)])
def test_ragged_shapes(H, R, P, W, Hh):
    """Point counts that are not multiples of the wave / block size, sample counts that are not multiples of the
    prefetch group, 1-ray / 1-point inputs, tiny and non-square bitmaps, several targets - forward and backward."""
    from artist_amd import trace_rays
    o, nrm, inc, both, tix, c, pn, dims, res = _random_scene(H, R, P, W, Hh, seed=H * 1000 + P)
    bd = both.to(DEV)
    od, nd = o.to(DEV).requires_grad_(True), nrm.to(DEV).requires_grad_(True)
    flux, fac = trace_rays(od, nd, inc.to(DEV), bd[..., 0], bd[..., 1], tix.to(DEV), c.to(DEV), pn.to(DEV),
                           dims.to(DEV), ray_magnitude=0.7, extinction=0.05, reflectivity=0.9, resolution=res)
    o_flux, o_fac = oracle.trace_fwd(o.numpy(), nrm.numpy(), inc.numpy(), both[..., 0].numpy(), both[..., 1].numpy(),
                                     tix.numpy(), c.numpy(), pn.numpy(), dims.numpy(), res, 0.7, 0.05, 0.9)
    assert flux.shape == (H, Hh, W)
    np.testing.assert_array_equal(n(fac), o_fac)
    scale = float(np.abs(o_flux).max()) + 1e-30
    np.testing.assert_allclose(n(flux), o_flux, rtol=0, atol=2e-3 * scale)     # per-ray 1e-4 px noise, few rays per pixel
    assert abs(float(flux.detach().sum()) - float(o_flux.sum())) <= 1e-5 * abs(float(o_flux.sum())) + 1e-6
    w = torch.rand(flux.shape, generator=torch.Generator().manual_seed(1)).to(DEV)
    (flux * w).sum().backward()
    go, gn = oracle.trace_bwd(o.numpy(), nrm.numpy(), inc.numpy(), both[..., 0].numpy(), both[..., 1].numpy(),
                              tix.numpy(), c.numpy(), pn.numpy(), dims.numpy(), res, n(w), 0.7, 0.05, 0.9)
    if np.linalg.norm(go) > 0:
        # measured 1e-7 ... 1.1e-6: the forward is recomputed op for op, only the gradient arithmetic (FMAs, v_rcp) differs.
        # (For scale: the fp32 oracle is 1e-5 ... 1.7e-2 from the fp64 oracle on these scenes - cell flips at pixel borders.)
        assert rel_l2(n(od.grad), go) < 1e-5 and rel_l2(n(nd.grad), gn) < 1e-5, (rel_l2(n(od.grad), go), rel_l2(n(nd.grad), gn))


@pytest.mark.parametrize("H,R,F,M", [(3, 16, 4, 300), (2, 9, 4, 2500), (400, 8, 2, 640), (1, 40, 6, 77)])
def test_facet_hint_changes_speed_only(H, R, F, M):
    """``points_per_facet`` (ABI v9: ``facet_points``) tells the kernels where the facets are, so that no block of points
    straddles two of them.  A layout hint: the bitmaps are the same BITS with and without it (integer accumulators), the ray
    counters equal, the gradients equal to rounding (bit-equal when no point's samples are cut into chunks); a hint that does
    not divide P is refused."""
    from artist_amd import trace_rays
    P = F * M
    o, nrm, inc, both, tix, c, pn, dims, res = _random_scene(H, R, P, 64, 48, seed=F * 100 + M)
    bd = both.to(DEV)
    w = torch.rand((H, 48, 64), generator=torch.Generator().manual_seed(3)).to(DEV)

    def run(**kw):
        od, nd = o.to(DEV).requires_grad_(True), nrm.to(DEV).requires_grad_(True)
        flux, fac = trace_rays(od, nd, inc.to(DEV), bd[..., 0], bd[..., 1], tix.to(DEV), c.to(DEV), pn.to(DEV), dims.to(DEV),
                               ray_magnitude=0.7, extinction=0.05, reflectivity=0.9, resolution=res, **kw)
        (flux * w).sum().backward()
        return n(flux), n(fac), n(od.grad), n(nd.grad)

    plain, hinted = run(), run(points_per_facet=M)
    np.testing.assert_array_equal(hinted[0], plain[0])
    np.testing.assert_array_equal(hinted[1], plain[1])
    assert rel_l2(hinted[2], plain[2]) < 1e-6 and rel_l2(hinted[3], plain[3]) < 1e-6
    o_flux, o_fac = oracle.trace_fwd(o.numpy(), nrm.numpy(), inc.numpy(), both[..., 0].numpy(), both[..., 1].numpy(),
                                     tix.numpy(), c.numpy(), pn.numpy(), dims.numpy(), res, 0.7, 0.05, 0.9)
    np.testing.assert_array_equal(hinted[1], o_fac)
    assert rel_l2(hinted[0], o_flux) < 1e-5
    with pytest.raises(ValueError):
        run(points_per_facet=M + 1 if P % (M + 1) else M + 2)


def test_chief_rays_miss_but_scattered_rays_hit():
    """A heliostat aimed just beside the target: no chief ray reaches it (the workgroup's window is EMPTY), but the sun
    shape scatters part of the rays onto the edge.  Those rays must all be handled as strays - forward (accumulators)
    and backward (gathers) - and none may touch the never-staged LDS window (this case produced NaN gradients once)."""
    from artist_amd import trace_rays
    H, R, P, res = 2, 48, 700, (64, 48)
    g = torch.Generator().manual_seed(9)
    origins = torch.cat([torch.rand(H, P, 3, generator=g) * 2 - 1 + torch.tensor([0.0, 50.0, 0.0]), torch.ones(H, P, 1)], -1)
    centers = torch.tensor([[0.0, 0.0, 30.0, 1.0]])
    dims = torch.tensor([[4.0, 3.0]])
    pn = torch.tensor([[0.0, 1.0, 0.0, 0.0]])
    aim = torch.tensor([[2.25, 0.0, 30.0], [0.0, 0.0, 31.7]])            # 0.25 m right of / 0.2 m above the plane's edge
    to_aim = torch.nn.functional.normalize(aim[:, None, :] - origins[..., :3], dim=-1)
    nrm = torch.nn.functional.normalize(to_aim - torch.tensor([0.0, 1.0, 0.0]), dim=-1)
    normals = torch.cat([nrm, torch.zeros(H, P, 1)], -1)
    incident = torch.tensor([[0.0, 1.0, 0.0, 0.0]]).repeat(H, 1)
    both = 3e-3 * torch.randn(H, R, P, 2, generator=g)                     # sigma 0.17 m on the target
    tix = torch.zeros(H, dtype=torch.long)
    bd = both.to(DEV)
    od, nd = origins.to(DEV).requires_grad_(True), normals.to(DEV).requires_grad_(True)
    flux, fac = trace_rays(od, nd, incident.to(DEV), bd[..., 0], bd[..., 1], tix.to(DEV), centers.to(DEV), pn.to(DEV), dims.to(DEV),
                           resolution=res)
    args = (origins.numpy(), normals.numpy(), incident.numpy(), both[..., 0].numpy(), both[..., 1].numpy(), tix.numpy(),
            centers.numpy(), pn.numpy(), dims.numpy(), res)
    o_flux, o_fac = oracle.trace_fwd(*args)
    assert 0.02 < float(fac[0].min()) and float(fac[0].max()) < 0.5       # some rays hit, most miss
    np.testing.assert_array_equal(n(fac), o_fac)
    assert rel_l2(n(flux), o_flux) < 1e-5
    w = torch.rand(flux.shape, generator=torch.Generator().manual_seed(1)).to(DEV)
    (flux * w).sum().backward()
    go, gn = oracle.trace_bwd(*args, n(w))
    assert bool(torch.isfinite(od.grad).all()) and bool(torch.isfinite(nd.grad).all())
    assert rel_l2(n(od.grad), go) < 2e-4 and rel_l2(n(nd.grad), gn) < 2e-4


def test_empty_field_and_all_rays_missing():
    from artist_amd import per_target_sum, trace_rays
    o, nrm, inc, both, tix, c, pn, dims, res = _random_scene(2, 3, 70, 16, 16)
    bd = both.to(DEV)
    # H = 0: empty outputs, no launch
    flux, fac = trace_rays(o[:0].to(DEV), nrm[:0].to(DEV), inc[:0].to(DEV), bd[:0, ..., 0], bd[:0, ..., 1], tix[:0].to(DEV),
                           c.to(DEV), pn.to(DEV), dims.to(DEV), resolution=res)
    assert flux.shape == (0, 16, 16) and fac.shape == (3, 0)
    assert float(per_target_sum(flux, tix[:0].to(DEV), 2).abs().sum()) == 0
    # every ray hits the BACK of the target (plane normals flipped): zero flux, zero factors, zero gradients
    od = o.to(DEV).requires_grad_(True)
    flux, fac = trace_rays(od, nrm.to(DEV), inc.to(DEV), bd[..., 0], bd[..., 1], tix.to(DEV), c.to(DEV), -pn.to(DEV),
                           dims.to(DEV), resolution=res)
    assert float(flux.abs().sum()) == 0 and float(fac[:2].abs().sum()) == 0 and bool((fac[2] == 1).all())
    flux.sum().backward()
    assert float(od.grad.abs().sum()) == 0


def test_nurbs_nonuniform_knots_and_many_points():
    """uniform=False span search with clamped non-uniform knot vectors swapped in (surfaces.py:209-243), M not a
    multiple of the block size, evaluation points on knots and at both ends."""
    from artist_amd import NURBSSurfaces
    g = torch.Generator().manual_seed(5)
    cp = torch.rand(2, 2, 7, 6, 3, generator=g)
    uv = torch.rand(2, 2, 777, 2, generator=g)
    uv[0, 0, :6, 0] = torch.tensor([0.0, 0.1, 0.7, 1.0, 1.0 - 1e-6, 0.4])
    uv[0, 0, :6, 1] = torch.tensor([1.0, 0.0, 0.35, 0.9, 0.5, 0.35])
    ku = torch.tensor([0.0, 0.0, 0.0, 0.0, 0.1, 0.4, 0.7, 1.0, 1.0, 1.0, 1.0])
    kv = torch.tensor([0.0, 0.0, 0.0, 0.35, 0.5, 0.9, 1.0, 1.0, 1.0])
    surf = NURBSSurfaces(torch.tensor([3, 2]), cp.to(DEV).requires_grad_(True), uniform=False, device=DEV)
    surf.knot_vectors_u = ku.to(DEV)[None, None].expand(2, 2, -1)
    surf.knot_vectors_v = kv.to(DEV)[None, None].expand(2, 2, -1)
    pts, nrm = surf(uv.to(DEV), None, None)
    o_pts, o_nrm = oracle.nurbs_fwd(cp.numpy(), uv.numpy(), [3, 2], knots_u=ku.numpy(), knots_v=kv.numpy(), uniform=False)
    np.testing.assert_allclose(n(pts), o_pts, rtol=0, atol=1e-6)
    ok = np.linalg.norm(o_nrm[..., :3], axis=-1) > 0.5          # degenerate normals (zero cross product) excluded
    np.testing.assert_allclose(n(nrm)[ok], o_nrm[ok], rtol=0, atol=5e-5)
    gp = torch.rand(pts.shape, generator=g)
    gn = torch.rand(nrm.shape, generator=g) * torch.from_numpy(ok[..., None].astype(np.float32))
    torch.autograd.backward([pts, nrm], [gp.to(DEV), gn.to(DEV)])
    o_g = oracle.nurbs_bwd(cp.numpy(), uv.numpy(), [3, 2], gp.numpy(), gn.numpy(), knots_u=ku.numpy(), knots_v=kv.numpy(),
                           uniform=False)
    assert rel_l2(n(surf.control_points.grad), o_g) < 1e-4


def _grid_points(us, vs):
    """cartesian_prod(us, vs) as ARTIST builds its evaluation grid (artist/nurbs/utils.py:37-49): point m = i Mv + j at (u_i, v_j)."""
    return torch.cartesian_prod(us, vs)


@pytest.mark.parametrize("deg,ncp,mu,mv,uniform,canted", [
    ([3, 3], (10, 10), 50, 50, True, True),        # the metric config's facet
    ([3, 3], (6, 6), 7, 13, True, True),           # rectangular grid, fewer points than threads
    ([2, 2], (5, 8), 1, 40, True, False),          # one row
    ([1, 1], (4, 4), 33, 1, True, True),           # one column (the row length is found to be 1)
    ([3, 2], (7, 6), 21, 17, False, True),         # mixed degrees (run-time degree kernel), non-uniform knots
    ([4, 4], (9, 8), 64, 30, True, False),         # more rows than one LDS pass holds in the forward's row groups
    ([5, 2], (8, 5), 12, 11, True, True)])
def test_nurbs_tensor_product_scheme_equals_the_scattered_scheme(monkeypatch, deg, ncp, mu, mv, uniform, canted):
    """The tensor-product scheme (cartesian evaluation grid, found by the workgroup itself) against the scattered one on the
    same inputs (ARTIST_HIP_NURBS_GRID=0 under ARTIST_HIP_DEBUG=1) and against the oracle: points and normals are the SAME
    BITS (same sums in the same order), the control-point gradient agrees to fp32 rounding with the scattered scheme's (double
    LDS sums) and the oracle's, and two backward passes of the tensor-product scheme give identical bits (every output element
    has one owner that adds in index order)."""
    from artist_amd import NURBSSurfaces
    g = torch.Generator().manual_seed(mu * 100 + mv)
    H, F = 3, 2
    cp = torch.rand(H, F, ncp[0], ncp[1], 3, generator=g)
    us = torch.sort(torch.rand(mu, generator=g) * 0.98 + 0.01).values
    vs = torch.rand(mv, generator=g) * 0.98 + 0.01                     # columns in no particular order
    if mu > 2:
        us[0], us[-1] = 1e-7, 1.0 - 1e-7
    uv = _grid_points(us, vs)[None, None].expand(H, F, -1, -1)
    cant = tr = None
    if canted:
        cant = torch.tensor([[0.8, 0.05, 0.0, 0.0], [0.02, 0.6, 0.1, 0.0]])[None, None].expand(H, F, -1, -1).contiguous()
        tr = torch.rand(H, F, 4, generator=g)
        tr[..., 3] = 0.0
    kw = {}
    if not uniform:
        ku = torch.cat([torch.zeros(deg[0]), torch.sort(torch.rand(ncp[0] - deg[0] + 1, generator=g)).values, torch.ones(deg[0])])
        kv = torch.cat([torch.zeros(deg[1]), torch.sort(torch.rand(ncp[1] - deg[1] + 1, generator=g)).values, torch.ones(deg[1])])
        ku[deg[0]], ku[-deg[0] - 1], kv[deg[1]], kv[-deg[1] - 1] = 0.0, 1.0, 0.0, 1.0
        kw = dict(knots_u=ku.numpy(), knots_v=kv.numpy(), uniform=False)
    gp = torch.rand(H, F, mu * mv, 4, generator=g)
    gn = torch.rand(H, F, mu * mv, 4, generator=g)
    dv = lambda x: None if x is None else x.to(DEV)

    def run():
        c = cp.to(DEV).requires_grad_(True)
        surf = NURBSSurfaces(torch.tensor(deg), c, uniform=uniform, device=DEV)
        if not uniform:
            surf.knot_vectors_u = torch.from_numpy(kw["knots_u"]).to(DEV)[None, None].expand(H, F, -1)
            surf.knot_vectors_v = torch.from_numpy(kw["knots_v"]).to(DEV)[None, None].expand(H, F, -1)
        pts, nrm = surf(uv.to(DEV), dv(cant), dv(tr))
        (g1,) = torch.autograd.grad([pts, nrm], [c], [gp.to(DEV), gn.to(DEV)], retain_graph=True)
        (g2,) = torch.autograd.grad([pts, nrm], [c], [gp.to(DEV), gn.to(DEV)])
        return n(pts), n(nrm), n(g1), n(g2)

    monkeypatch.setenv("ARTIST_HIP_DEBUG", "1")
    monkeypatch.setenv("ARTIST_HIP_NURBS_GRID", "1")
    pts_t, nrm_t, g_t, g_t2 = run()
    monkeypatch.setenv("ARTIST_HIP_NURBS_GRID", "0")
    pts_s, nrm_s, g_s, _ = run()
    np.testing.assert_array_equal(pts_t, pts_s)
    np.testing.assert_array_equal(nrm_t, nrm_s)
    np.testing.assert_array_equal(g_t, g_t2)                               # bit-reproducible
    assert rel_l2(g_t, g_s) < 2e-6, rel_l2(g_t, g_s)
    o_pts, o_nrm = oracle.nurbs_fwd(cp.numpy(), uv.contiguous().numpy(), deg, None if cant is None else cant.numpy(),
                                    None if tr is None else tr.numpy(), **kw)
    if deg[0] == deg[1] and deg[0] <= 4 and uniform:
        np.testing.assert_array_equal(pts_t, o_pts)                        # the compile-time-degree kernels: the oracle's bits
    else:
        np.testing.assert_allclose(pts_t, o_pts, rtol=0, atol=1e-6)
    o_g = oracle.nurbs_bwd(cp.numpy(), uv.contiguous().numpy(), deg, gp.numpy(), gn.numpy(), None if cant is None else cant.numpy(), **kw)
    ok = np.linalg.norm(o_nrm[..., :3], axis=-1) > 0.5
    if ok.all():
        assert rel_l2(g_t, o_g) < 2e-5, rel_l2(g_t, o_g)


def test_nurbs_points_that_are_almost_a_grid_take_the_scattered_scheme(monkeypatch):
    """One point of a 20 x 30 grid moved by one ULP, a grid whose point count is not a multiple of its first row, a NaN at
    point 0, and a different list per facet (one a grid, one not): the workgroup's own check sends each facet to the scheme
    that fits it - the results equal the scattered scheme's bit for bit in every case."""
    from artist_amd import NURBSSurfaces
    g = torch.Generator().manual_seed(77)
    H, F = 2, 2
    cp = torch.rand(H, F, 6, 7, 3, generator=g).to(DEV)
    base = _grid_points(torch.linspace(1e-7, 1 - 1e-7, 20), torch.linspace(1e-7, 1 - 1e-7, 30))
    cases = {}
    moved = base.clone()
    moved[317, 1] = torch.nextafter(moved[317, 1], torch.tensor(2.0))
    cases["one ulp"] = moved[None, None].expand(H, F, -1, -1)
    cases["ragged"] = base[:-7][None, None].expand(H, F, -1, -1)
    per_facet = base[None, None].repeat(H, F, 1, 1)
    per_facet[1, 0] = torch.rand(600, 2, generator=g)
    cases["per facet"] = per_facet
    monkeypatch.setenv("ARTIST_HIP_DEBUG", "1")
    for name, uv in cases.items():
        got = {}
        for flag in ("1", "0"):
            monkeypatch.setenv("ARTIST_HIP_NURBS_GRID", flag)
            c = cp.clone().requires_grad_(True)
            pts, nrm = NURBSSurfaces(torch.tensor([3, 3]), c, device=DEV)(uv.to(DEV), None, None)
            (pts.sum() + nrm[..., 0].sum()).backward()
            got[flag] = (n(pts), n(nrm), n(c.grad))
        np.testing.assert_array_equal(got["1"][0], got["0"][0], err_msg=name)
        np.testing.assert_array_equal(got["1"][1], got["0"][1], err_msg=name)
        assert rel_l2(got["1"][2], got["0"][2]) < 2e-6, (name, rel_l2(got["1"][2], got["0"][2]))


# ---------------------------------------------------------------------------------------------
# Cylindrical receivers (geometry.line_cylinder_intersections, artist/raytracing/geometry.py:207-445).
# In fp32 the hit is ill-conditioned at the fixtures' geometry (b^2 - 4ac cancels ~400x): the yardstick
# is the reference's own fp32-vs-fp64 error, as in tests/test_oracle_golden.py::test_cylinder_stages_fp32.
# ---------------------------------------------------------------------------------------------
def cyl_inputs(d):
    return tuple(t(d[k]) for k in ("cyl_centers", "cyl_normals", "cyl_axes", "cyl_radii", "cyl_heights", "cyl_opening"))


@pytest.mark.parametrize("name", CYL_CASES)
@pytest.mark.parametrize("interleaved", [True, False])
def test_trace_forward_cylinders(golden, name, interleaved):
    from artist_amd import trace_rays
    d, d64 = golden(name), golden(name + "_f64")
    flux, fac = trace_rays(**trace_inputs(d, interleaved), cyl=cyl_inputs(d))
    o_flux, o_fac = oracle_fwd(d, cyl=oracle.cyl_tables(d))
    yard = rel_l2(d["flux"], d64["flux"])                 # what fp32 does to the reference itself
    # Two legs (round-3 review: a 1e-3 defect would have passed the yardstick alone).  (1) HIP against the fp32 restatement, which
    # does the same operations in the same order: tight - measured 1.5e-6 / 7.1e-6 (tools/parity_margins.py) - and the
    # restatement in fp64 is pinned on the reference's fp64 fixtures in the CPU suite (tests/test_oracle_golden.py).  (2) HIP
    # against the reference's fp32 fixture, where torch's other operation order is amplified ~400 times by the cancellation
    # in b^2 - 4ac: within the reference's own fp32-vs-fp64 distance.
    assert rel_l2(n(flux), o_flux) < 3e-5, rel_l2(n(flux), o_flux)
    assert rel_l2(n(flux), d["flux"]) < max(yard, 2e-3), (rel_l2(n(flux), d["flux"]), yard)
    assert rel_l2(n(flux), d64["flux"]) < 2 * max(yard, 2e-3)
    # planar heliostats of the mixed case are untouched by the cylinder launch: same bar as test_trace_forward
    planar = d["target_idx"] < d["target_centers"].shape[0]
    if planar.any():
        assert rel_l2(n(flux)[planar], o_flux[planar]) < 2e-4
        np.testing.assert_array_equal(n(fac)[:, planar], o_fac[:, planar])
    # ray counts: a sin/cos ULP can move a ray across the sector edge; <= 1e-3 of the rays do
    np.testing.assert_allclose(n(fac), o_fac, rtol=0, atol=1e-3)
    np.testing.assert_allclose(n(fac[0]), d["intercept"], rtol=0, atol=1e-3)


@pytest.mark.parametrize("name", CYL_CASES)
def test_trace_backward_cylinders(golden, name):
    from artist_amd import trace_rays
    d, d64 = golden(name), golden(name + "_f64")
    inp = trace_inputs(d)
    inp["origins"].requires_grad_(True)
    inp["normals"].requires_grad_(True)
    flux, _ = trace_rays(**inp, cyl=cyl_inputs(d))
    (flux * t(d["loss_weights"])).sum().backward()
    go, gn = oracle.trace_bwd(d["aligned_points"], d["aligned_normals"], d["incident"], d["distortions_u"],
                              d["distortions_e"], d["target_idx"], d["target_centers"], d["target_normals"],
                              d["target_dims"], d["resolution"], d["loss_weights"], float(d["ray_magnitude"]),
                              float(d["extinction"]), float(d["reflectivity"]), cyl=oracle.cyl_tables(d))
    for got, orc, key in ((inp["origins"].grad, go, "grad_aligned_points"), (inp["normals"].grad, gn, "grad_aligned_normals")):
        yard = rel_l2(d[key], d64[key])
        assert rel_l2(n(got), orc) < 3e-5, (key, rel_l2(n(got), orc))              # (measured <= 3.9e-6: the same two legs as in the forward test)
        assert rel_l2(n(got), d[key]) < max(2 * yard, 1e-3), (key, rel_l2(n(got), d[key]), yard)
        assert rel_l2(n(got), d64[key]) < max(3 * yard, 1e-3), (key, rel_l2(n(got), d64[key]), yard)


def test_published_reflection_directions(golden):
    """``heliostat_group.preferred_reflection_directions`` (heliostat_ray_tracer.py:285-290) comes from ``art_reflect``: the
    oracle's ``reflect`` (= the reference's known answers, tests/test_oracle_golden.py) bit for bit, and the torch
    formula of the mirror's fallback to rounding."""
    from artist_amd import ops
    from artist_amd.raytracing import reflect
    d = golden("small_deg3")
    nrm, inc = t(d["aligned_normals"]), t(d["incident"])
    got = ops.reflect_directions(inc, nrm)
    want = oracle.reflect(d["incident"], d["aligned_normals"])
    np.testing.assert_array_equal(n(got), want)
    with torch.no_grad():
        via_mirror = reflect(inc.unsqueeze(1), nrm)
    np.testing.assert_array_equal(n(via_mirror), n(got))
    with torch.enable_grad():
        formula = reflect(inc.unsqueeze(1), nrm.clone().requires_grad_(True))      # differentiable callers keep torch's ops
    np.testing.assert_allclose(n(formula), n(got), rtol=0, atol=2e-7)


def test_mixed_tower_traces_its_planar_heliostats_like_a_planar_tower(golden, monkeypatch):
    """A tower with planar AND cylindrical receivers is a split call: the lean launch (own geometry) for the heliostats
    that aim at a plane, the cylinder launch for the others.  The planar heliostats' bitmaps, factors and gradients are
    the bits a planar-only call gives for them, and the unsplit call (``ARTIST_HIP_BLOCKING_SPLIT=0``: both launches share
    the generic geometry) gives the same bits for every heliostat."""
    from artist_amd import trace_rays
    d = golden("small_cyl_mixed")
    T = d["target_centers"].shape[0]
    planar = torch.from_numpy(d["target_idx"] < T).to(DEV)
    assert bool(planar.any()) and not bool(planar.all())
    w = t(d["loss_weights"])

    def run(rows=None, **kw):
        inp = trace_inputs(d, interleaved=False)
        if rows is not None:
            for k in ("origins", "normals", "incident", "dist_u", "dist_e", "target_idx"):
                inp[k] = inp[k][rows].contiguous()
        inp["origins"].requires_grad_(True)
        inp["normals"].requires_grad_(True)
        flux, fac = trace_rays(**inp, **kw)
        (flux * (w if rows is None else w[rows])).sum().backward()
        return flux.detach(), fac, inp["origins"].grad, inp["normals"].grad

    # (no sample chunks: a field this small would otherwise be cut into chunks, and the backward call splits only unchunked launches)
    monkeypatch.setenv("ARTIST_HIP_FWD_BLOCKS", "1")
    mixed = run(cyl=cyl_inputs(d))
    alone = run(rows=planar)
    np.testing.assert_array_equal(n(mixed[0][planar]), n(alone[0]))
    np.testing.assert_array_equal(n(mixed[1][:, planar]), n(alone[1]))
    np.testing.assert_array_equal(n(mixed[2][planar]), n(alone[2]))
    np.testing.assert_array_equal(n(mixed[3][planar]), n(alone[3]))
    monkeypatch.setenv("ARTIST_HIP_BLOCKING_SPLIT", "0")
    unsplit = run(cyl=cyl_inputs(d))
    np.testing.assert_array_equal(n(mixed[0]), n(unsplit[0]))
    np.testing.assert_array_equal(n(mixed[1]), n(unsplit[1]))
    for a, b in zip(mixed[2:], unsplit[2:]):      # (the generic and the lean adjoint associate a point's sums differently)
        assert rel_l2(n(a), n(b)) < 1e-6, rel_l2(n(a), n(b))


def test_cylinder_per_target_mode(golden):
    from artist_amd import per_target_sum, trace_rays
    d = golden("small_cyl_mixed")
    T, Tc = d["target_centers"].shape[0], d["cyl_centers"].shape[0]
    inp = trace_inputs(d)
    flux_h, _ = trace_rays(**inp, cyl=cyl_inputs(d))
    flux_t, _ = trace_rays(**inp, cyl=cyl_inputs(d), per_target=True)
    assert flux_t.shape[0] == T + Tc
    summed = per_target_sum(flux_h, inp["target_idx"], T + Tc)
    np.testing.assert_allclose(n(flux_t), n(summed), rtol=0, atol=2e-6 * float(summed.max()))
    assert rel_l2(n(summed), d["per_target"]) < 5e-3


def _wide_cylinder_case(seed=3, H=3, P=640, R=24, res=(192, 64)):
    """Well-conditioned cylinder geometry (radius 25 m, mirrors 60-80 m away: cancellation ~10x, not 400x), where
    fp32 must agree with the fp64 oracle tightly - this is the test that pins the HIP cylinder arithmetic."""
    g = torch.Generator().manual_seed(seed)
    centre = torch.tensor([0.0, 0.0, 40.0])
    pos = torch.tensor([[-30.0, 60.0, 2.0], [5.0, 75.0, 2.0], [40.0, 55.0, 2.0]])[:H]
    sun = torch.tensor([0.2, -0.5, -0.84]); sun = sun / sun.norm()
    hdir = (pos - centre) * torch.tensor([1.0, 1.0, 0.0])
    to_t = centre + 25.0 * hdir / hdir.norm(dim=1, keepdim=True) - pos        # aim at the mantle, mid height
    to_t = to_t / to_t.norm(dim=1, keepdim=True)
    nrm0 = to_t - sun
    nrm0 = nrm0 / nrm0.norm(dim=1, keepdim=True)
    local = (torch.rand((H, P, 3), generator=g) - 0.5) * torch.tensor([3.0, 3.0, 0.0])
    ex = torch.linalg.cross(nrm0, torch.tensor([[0.0, 0.0, 1.0]]).expand(H, 3))
    ex = ex / ex.norm(dim=1, keepdim=True)
    ey = torch.linalg.cross(nrm0, ex)
    pts = pos[:, None] + local[..., :1] * ex[:, None] + local[..., 1:2] * ey[:, None]
    nrm = nrm0[:, None] + 2e-3 * torch.randn((H, P, 3), generator=g)
    nrm = nrm / nrm.norm(dim=-1, keepdim=True)
    origins = torch.cat([pts, torch.ones(H, P, 1)], -1)
    normals = torch.cat([nrm, torch.zeros(H, P, 1)], -1)
    incident = torch.cat([sun, torch.zeros(1)]).expand(H, 4).contiguous()
    both = 2e-3 * torch.randn((H, R, P, 2), generator=g)
    cyl = dict(centers=torch.tensor([[0.0, 0.0, 40.0, 1.0]]), normals=torch.tensor([[0.0, 1.0, 0.0, 0.0]]),
               axes=torch.tensor([[0.0, 0.0, 1.0, 0.0]]), radii=torch.tensor([25.0]), heights=torch.tensor([12.0]),
               opening=torch.tensor([2.6]))
    return origins, normals, incident, both, cyl, res


def test_cylinder_well_conditioned_vs_fp64_oracle():
    from artist_amd import trace_rays
    origins, normals, incident, both, cyl, res = _wide_cylinder_case()
    H = origins.shape[0]
    tix = torch.zeros(H, dtype=torch.int32)                     # T == 0: index 0 is cylinder 0
    dev_both = both.to(DEV)
    empty4, empty2 = torch.zeros((0, 4), device=DEV), torch.zeros((0, 2), device=DEV)
    o, nn = origins.to(DEV).requires_grad_(True), normals.to(DEV).requires_grad_(True)
    cyl_dev = tuple(v.to(DEV) for v in (cyl["centers"], cyl["normals"], cyl["axes"], cyl["radii"], cyl["heights"], cyl["opening"]))
    flux, fac = trace_rays(o, nn, incident.to(DEV), dev_both[..., 0], dev_both[..., 1], tix.to(DEV), empty4, empty4, empty2,
                           ray_magnitude=1.0, extinction=0.1, reflectivity=0.9, resolution=res, cyl=cyl_dev)
    f64 = lambda x: x.double().numpy()
    cyl64 = {k: f64(v) for k, v in cyl.items()}
    z4, z2 = np.zeros((0, 4)), np.zeros((0, 2))
    o_flux, o_fac = oracle.trace_fwd(f64(origins), f64(normals), f64(incident), f64(both[..., 0]), f64(both[..., 1]),
                                     tix.numpy(), z4, z4, z2, res, 1.0, 0.1, 0.9, cyl=cyl64)
    assert o_flux.sum() > 0.5 * both.shape[1] * origins.shape[1] * H * 0.3      # the beam is on the receiver
    assert rel_l2(n(flux), o_flux) < 5e-5, rel_l2(n(flux), o_flux)      # the fp32 oracle sits at 7e-6 here
    np.testing.assert_allclose(n(fac), o_fac, rtol=0, atol=2e-4)
    w = torch.linspace(0.5, 1.5, res[0] * res[1]).reshape(res[1], res[0]).expand(H, -1, -1).contiguous()
    (flux * w.to(DEV)).sum().backward()
    go, gn = oracle.trace_bwd(f64(origins), f64(normals), f64(incident), f64(both[..., 0]), f64(both[..., 1]),
                              tix.numpy(), z4, z4, z2, res, f64(w), 1.0, 0.1, 0.9, cyl=cyl64)
    # smooth loss weights: cell flips cost little and the fp32 oracle sits at 1e-6 here
    assert rel_l2(n(o.grad), go) < 2e-4, rel_l2(n(o.grad), go)
    assert rel_l2(n(nn.grad), gn) < 2e-4, rel_l2(n(nn.grad), gn)


def test_cylinder_through_ray_tracer_mirror():
    """``HeliostatRayTracer`` with a tower that has planar AND cylindrical areas: global target indices, planar
    first (heliostat_ray_tracer.py:337-429), against the oracle."""
    from artist_amd import HeliostatRayTracer
    from artist_amd.scene import SolarTower, TowerTargetAreasCylindrical, build_synthetic_scenario
    H = 6
    scenario, _ = build_synthetic_scenario(H, 20, n_eval=16, device=DEV)
    planar = scenario.solar_tower.target_areas[0]
    cyl = TowerTargetAreasCylindrical(
        names=["cyl"], centers=torch.tensor([[0.0, -12.0, 55.0, 1.0]], device=DEV),
        normals=torch.tensor([[0.0, 1.0, 0.0, 0.0]], device=DEV), axes=torch.tensor([[0.0, 0.0, 1.0, 0.0]], device=DEV),
        radii=torch.tensor([12.0], device=DEV), heights=torch.tensor([30.0], device=DEV),
        opening_angles=torch.tensor([2.0], device=DEV))
    scenario.solar_tower = SolarTower([planar, cyl], device=DEV)
    group = scenario.heliostat_field.heliostat_groups[0]
    mask = torch.ones(H, dtype=torch.int32, device=DEV)
    tix = torch.tensor([0, 1, 1, 0, 1, 0], device=DEV)
    inc = torch.tensor([0.0, 1.0, -1.0, 0.0], device=DEV)
    inc = (inc / inc.norm()).expand(H, 4).contiguous()
    group.activate_heliostats(mask, DEV)
    group.align_surfaces_with_incident_ray_directions(scenario.solar_tower.get_centers_of_target_areas(tix), inc, mask, DEV)
    rt = HeliostatRayTracer(scenario, group, blocking_active=False, bitmap_resolution=torch.tensor([64, 64]))
    flux, intercept, on_target, blocking = rt.trace_rays(inc, mask, tix)
    assert flux.shape == (H, 64, 64) and bool((intercept[tix == 1] > 0.5).all())
    c = scenario.solar_tower.target_areas[1]
    cyl_np = dict(centers=n(c.centers), normals=n(c.normals), axes=n(c.axes), radii=n(c.radii), heights=n(c.heights),
                  opening=n(c.opening_angles))
    o_flux, o_fac = oracle.trace_fwd(n(group.active_surface_points), n(group.active_surface_normals), n(inc),
                                     n(rt.distortions_dataset.distortions_u), n(rt.distortions_dataset.distortions_e),
                                     n(tix), n(planar.centers), n(planar.normals), n(planar.dimensions), (64, 64),
                                     cyl=cyl_np)
    assert rel_l2(n(flux), o_flux) < 1e-3, rel_l2(n(flux), o_flux)
    np.testing.assert_allclose(n(intercept), o_fac[0], rtol=0, atol=1e-3)
    pt, *_ = rt.trace_rays_per_target(inc, mask, tix)
    assert pt.shape[0] == 2
    assert rel_l2(n(pt), n(rt.get_bitmaps_per_target(flux, tix))) < 1e-6


# ---------------------------------------------------------------------------------------------
# Blocking (artist/raytracing/blocking.py): filter (art_blocking_filter) + soft mask inside the trace kernels.
# ---------------------------------------------------------------------------------------------
def blocking_inputs(d, **extra):
    H = d["aligned_points"].shape[0]
    return dict(corners=t(d["prim_corners"]), spans=t(d["prim_spans"]), normals=t(d["prim_normals"]),
                owner=torch.arange(H, dtype=torch.int32, device=DEV), **extra)


@pytest.mark.parametrize("name", BLOCKING_CASES)
@pytest.mark.parametrize("interleaved", [True, False])
def test_blocking_forward(golden, name, interleaved):
    from artist_amd import trace_rays
    d, d64 = golden(name), golden(name + "_f64")
    H = d["aligned_points"].shape[0]
    flux, fac, flags = trace_rays(**trace_inputs(d, interleaved), blocking=blocking_inputs(d))
    np.testing.assert_array_equal(np.nonzero(n(flags))[0], d["filter_indices"])          # the reference's filtered set
    o_flux, o_fac = oracle_fwd(d, blocking=oracle.blocking_tables(d, H))
    yard = rel_l2(d["flux"], d64["flux"])               # sigmoid(1000 x) amplifies fp32 rounding in the edge band
    assert rel_l2(n(flux), o_flux) < 2e-6, rel_l2(n(flux), o_flux)      # HIP against the fp32 restatement: tight (measured 2.5e-7)
    assert rel_l2(n(flux), d["flux"]) < max(yard, 2e-4), (rel_l2(n(flux), d["flux"]), yard)
    rays = d["blocked"][0].size
    np.testing.assert_allclose(n(fac), o_fac, rtol=0, atol=1.5 / rays)                    # <= 1 ray per counter
    for row, key in enumerate(("intercept", "on_target", "blocking")):
        np.testing.assert_allclose(n(fac[row]), d[key], rtol=0, atol=1.5 / rays)
    assert float(fac[2].min()) < 0.9                                                       # the case does block
    # measuring the scatter bound on the device gives the same result as passing it
    bound = float(max(np.abs(d["distortions_u"]).max(), np.abs(d["distortions_e"]).max()))
    flux2, fac2, flags2 = trace_rays(**trace_inputs(d, interleaved), blocking=blocking_inputs(d, max_scatter_angle=bound))
    assert torch.equal(flags, flags2) and torch.equal(fac, fac2)
    np.testing.assert_array_equal(n(flux2), n(flux))


@pytest.mark.parametrize("name", BLOCKING_CASES)
def test_blocking_backward(golden, name):
    """Gradients w.r.t. ray origins / normals AND the rectangles; the aligned points are both ray origins and the
    source of the rectangle corners, so the reference's gradient of them is the sum of both paths."""
    from artist_amd import trace_rays
    from artist_amd.blocking import create_blocking_primitives_rectangles_by_index
    d, d64 = golden(name), golden(name + "_f64")
    H = d["aligned_points"].shape[0]
    inp = trace_inputs(d)
    points = inp["origins"].requires_grad_(True)
    inp["normals"].requires_grad_(True)
    np.testing.assert_array_equal(d["blocking_surfaces"], d["aligned_points"])
    corners, spans, normals = create_blocking_primitives_rectangles_by_index(points)     # one group, all active
    for x in (corners, spans, normals):
        x.retain_grad()
    torch.testing.assert_close(corners.detach().cpu(), torch.from_numpy(d["prim_corners"]), rtol=0, atol=0)
    torch.testing.assert_close(normals.detach().cpu(), torch.from_numpy(d["prim_normals"]), rtol=0, atol=3e-7)
    flux, _, _ = trace_rays(**inp, blocking=dict(corners=corners, spans=spans, normals=normals,
                                                 owner=torch.arange(H, dtype=torch.int32, device=DEV)))
    (flux * t(d["loss_weights"])).sum().backward()
    go, gn, gpc, gps, gpn = oracle.trace_bwd(
        d["aligned_points"], d["aligned_normals"], d["incident"], d["distortions_u"], d["distortions_e"],
        d["target_idx"], d["target_centers"], d["target_normals"], d["target_dims"], d["resolution"], d["loss_weights"],
        float(d["ray_magnitude"]), float(d["extinction"]), float(d["reflectivity"]), blocking=oracle.blocking_tables(d, H))
    for got, orc, key in ((inp["normals"].grad, gn, "grad_aligned_normals"), (normals.grad, gpn, "grad_prim_normals"),
                          (points.grad, None, "grad_aligned_points"), (corners.grad, None, "grad_prim_corners"),
                          (spans.grad, None, "grad_prim_spans")):
        yard = rel_l2(d[key], d64[key])                 # the reference's own fp32-vs-fp64 distance
        if orc is not None:
            assert rel_l2(n(got), orc) < 2e-5, (key, rel_l2(n(got), orc))             # HIP against the fp32 restatement: tight
        assert rel_l2(n(got), d[key]) < max(2 * yard, 1e-3), (key, rel_l2(n(got), d[key]), yard)
        assert rel_l2(n(got), d64[key]) < max(3 * yard, 1e-3), (key, rel_l2(n(got), d64[key]), yard)


def crowd(blk, k, n_extra):
    """The tables ``blk`` (torch or numpy) plus ``n_extra`` small rectangles above rectangle ``k`` (0.3 of its spans, stepped
    along its first span down to its upper edge, each lifted a millimetre more along its normal): whoever had ``k`` in its ray
    cone now has n_extra more candidates, their soft edges cross the beam at different heights, and they block rays that ``k``
    let pass."""
    is_t = torch.is_tensor(blk["corners"])
    c0, su, sv, nn = blk["corners"][k][0], blk["spans"][k][0], blk["spans"][k][1], blk["normals"][k]
    if is_t:
        j = torch.arange(n_extra, dtype=torch.float32, device=c0.device)[:, None]
        w = torch.tensor([1.0, 1.0, 1.0, 0.0], device=c0.device)
        rep = lambda a: a[None].repeat(n_extra, 1)                               # noqa: E731
        stack, cat = torch.stack, lambda a, b: torch.cat([a, b])                 # noqa: E731
    else:
        j = np.arange(n_extra, dtype=np.float32)[:, None]
        w = np.array([1.0, 1.0, 1.0, 0.0], np.float32)
        rep = lambda a: np.repeat(a[None], n_extra, 0)                           # noqa: E731
        stack, cat = np.stack, lambda a, b: np.concatenate([a, b]).astype(np.float32)   # noqa: E731
    c0s = c0[None] + (1.4 - (0.4 / n_extra) * j) * su[None] + 1e-3 * (j + 1) * nn[None] * w
    sus, svs = rep(0.3 * su), rep(0.3 * sv)
    corners = stack([c0s, c0s + sus, c0s + sus + svs, c0s + svs], 1)
    return dict(blk, corners=cat(blk["corners"], corners), spans=cat(blk["spans"], stack([sus, svs], 1)),
                normals=cat(blk["normals"], rep(nn)), lbvh_compat=False)


@pytest.mark.parametrize("body", ["lean", "generic", "unbounded rows"])
@pytest.mark.parametrize("n_extra", [30, 40, 110])
def test_more_candidates_than_the_tables_hold(golden, monkeypatch, n_extra, body):
    """A heliostat with more candidate rectangles than the kernels keep in LDS (32; artist/raytracing/blocking.py:212-354 has no
    such number: every ray meets every filtered rectangle) is traced, not refused: the sigmas of the others are added from the
    caller's tables (40 more: through the wave's 64-bit mask; 110 more: the ones beyond it for every ray; 30 more: a list of
    exactly 32, the last case that is NOT wide - bit 31 of the masks is then a rectangle like any other).  Forward and
    backward - rays and rectangles - against the oracle, which has no limit; through the lean ray bodies (what a planar tower
    takes), the generic ones (``ARTIST_HIP_BLOCK_LEAN=0``: what a tower with cylinders takes), and with candidate rows as wide as
    the table of rectangles (``BLOCKING_CANDIDATES = None``)."""
    from artist_amd import ops, trace_rays
    if body == "generic":
        monkeypatch.setenv("ARTIST_HIP_BLOCK_LEAN", "0")
    if body == "unbounded rows":
        monkeypatch.setattr(ops, "BLOCKING_CANDIDATES", None)
    d = golden("small_blocking")
    H = d["aligned_points"].shape[0]
    k = int(d["filter_indices"][0])
    blk = crowd(blocking_inputs(d), k, n_extra)
    o_blk = crowd(oracle.blocking_tables(d, H), k, n_extra)
    inp = trace_inputs(d)
    inp["origins"].requires_grad_(True); inp["normals"].requires_grad_(True)
    tabs = {key: blk[key].clone().requires_grad_(True) for key in ("corners", "spans", "normals")}
    flux, fac, flags = trace_rays(**inp, blocking=dict(blk, **tabs))
    counts = n(ops._LAST_BLOCKING[1])
    print("candidates per heliostat:", counts, " filtered:", int(n(flags).sum()), "of", flags.shape[0])
    assert counts.max() == 32 if n_extra == 30 else counts.max() > 32 + (64 if n_extra > 100 else 0)     # the case is what it says
    o_flux, o_fac = oracle_fwd(d, blocking=o_blk)
    assert np.isfinite(n(flux)).all() and np.isfinite(n(fac)).all()
    err = rel_l2(n(flux), o_flux)
    print(f"flux {err:.2e}; unblocked fractions {n(fac[2])} (oracle {o_fac[2]})")
    assert err < 2e-4, err
    rays = d["blocked"][0].size
    np.testing.assert_allclose(n(fac), o_fac, rtol=0, atol=1.5 / rays)
    narrow, _, _ = trace_rays(**trace_inputs(d), blocking=blocking_inputs(d))
    assert rel_l2(n(flux), n(narrow)) > 1e-3                                  # the extra rectangles do block
    (flux * t(d["loss_weights"])).sum().backward()
    go, gn, gpc, gps, gpn = oracle.trace_bwd(
        d["aligned_points"], d["aligned_normals"], d["incident"], d["distortions_u"], d["distortions_e"],
        d["target_idx"], d["target_centers"], d["target_normals"], d["target_dims"], d["resolution"], d["loss_weights"],
        float(d["ray_magnitude"]), float(d["extinction"]), float(d["reflectivity"]), blocking=o_blk)
    for got, orc, key in ((inp["origins"].grad, go, "origins"), (inp["normals"].grad, gn, "normals"), (tabs["corners"].grad, gpc, "corners"),
                          (tabs["spans"].grad, gps, "spans"), (tabs["normals"].grad, gpn, "rectangle normals")):
        e = rel_l2(n(got), orc)
        print(f"  gradient w.r.t. {key}: {e:.2e}")
        assert e < 2e-3, (key, e)
    # rectangles beyond the tables (the lists are in ascending order: these sit behind at least 32 others) do receive gradients
    far = H + (96 if n_extra > 100 else (32 if n_extra > 30 else 24))
    assert np.abs(gpc[far:]).sum() > 0 and np.abs(n(tabs["corners"].grad)[far:]).sum() > 0


def test_blocking_filter_matches_reference_tree(golden):
    """art_blocking_filter on the field-like layouts of known_answers.npz: with ``lbvh_compat`` only rectangles that
    the reference's tree can reach are ever flagged (26 of 391, 3 of 2000), without it every hit rectangle is."""
    from artist_amd import trace_rays
    ka = golden("known_answers")
    for i in (1, 2, 3):
        corners_np = ka[f"tree{i}_corners"]
        N = corners_np.shape[0]
        corners = t(corners_np)
        spans = torch.stack((corners[:, 1] - corners[:, 0], corners[:, 3] - corners[:, 0]), dim=1)
        nrm = torch.nn.functional.normalize(torch.linalg.cross(spans[:, 0, :3], spans[:, 1, :3]), dim=-1)
        nrm = torch.cat((nrm, torch.zeros(N, 1, device=DEV)), dim=-1)
        # one ray bundle per rectangle: mirrors = the rectangles themselves (4 corner points + centre), low sun from
        # the south, all aimed at a receiver above the origin: rows behind each other block
        H = N
        pts = torch.cat((corners, corners.mean(1, keepdim=True)), dim=1)                          # [N,5,4]
        pts = 0.9 * pts + 0.1 * corners.mean(1, keepdim=True)
        aim = torch.tensor([0.0, 0.0, 30.0], device=DEV)
        to_t = torch.nn.functional.normalize(aim - pts[:, 4, :3], dim=-1)
        sun = torch.tensor([0.0, 1.0, 0.0], device=DEV)
        mn = torch.nn.functional.normalize(to_t - sun, dim=-1)
        normals_ = torch.cat((mn, torch.zeros(N, 1, device=DEV)), dim=-1)[:, None].expand(-1, 5, -1).contiguous()
        inc = torch.tensor([0.0, 1.0, 0.0, 0.0], device=DEV).expand(H, 4).contiguous()
        both = torch.zeros((H, 2, 5, 2), device=DEV)
        both[:, 1] = 1e-3
        args = (pts.contiguous(), normals_, inc, both[..., 0], both[..., 1], torch.zeros(H, dtype=torch.int32, device=DEV),
                torch.tensor([[0.0, 0.0, 30.0, 1.0]], device=DEV), torch.tensor([[0.0, 1.0, 0.0, 0.0]], device=DEV),
                torch.tensor([[60.0, 60.0]], device=DEV))
        blk = dict(corners=corners, spans=spans, normals=nrm, owner=torch.arange(H, dtype=torch.int32, device=DEV))
        _, _, flags = trace_rays(*args, resolution=(32, 32), blocking=dict(blk, lbvh_compat=True))
        _, _, flags_all = trace_rays(*args, resolution=(32, 32), blocking=dict(blk, lbvh_compat=False))
        got, every = set(np.nonzero(n(flags))[0].tolist()), set(np.nonzero(n(flags_all))[0].tolist())
        reachable = set(ka[f"tree{i}_reachable"].tolist())
        assert got == every & reachable, (i, len(got), len(every), len(reachable))
        assert len(every) > N // 4 and len(got) < len(every)
        # the same rays through the CPU restatement
        o_np, n_np = n(pts), n(normals_)
        _, _, dbg = oracle.trace_fwd(o_np, n_np, n(inc), n(both[..., 0]).copy(), n(both[..., 1]).copy(),
                                     np.zeros(H, np.int32), n(args[6]), n(args[7]), n(args[8]), (32, 32), debug=True,
                                     blocking=dict(corners=corners_np, spans=n(spans), normals=n(nrm),
                                                   owner=np.arange(H, dtype=np.int32)))
        assert got == set(np.nonzero(dbg["filter_flags"])[0].tolist())


def test_blocking_through_ray_tracer_mirror():
    """``HeliostatRayTracer`` with its default ``blocking_active=True`` on a column of heliostats behind each other."""
    from artist_amd import HeliostatRayTracer
    from artist_amd.scene import build_synthetic_scenario
    H = 5
    scenario, _ = build_synthetic_scenario(H, 12, n_eval=12, device=DEV)
    group = scenario.heliostat_field.heliostat_groups[0]
    group.positions = torch.tensor([[0.0, 150.0, 0.0, 1.0], [0.3, 147.2, 0.0, 1.0], [-0.8, 144.0, 0.0, 1.0],
                                    [2.0, 141.0, 0.0, 1.0], [-25.0, 100.0, 0.0, 1.0]], device=DEV)
    mask = torch.ones(H, dtype=torch.int32, device=DEV)
    tix = torch.zeros(H, dtype=torch.long, device=DEV)
    inc = torch.nn.functional.normalize(torch.tensor([0.1, 0.95, -0.1, 0.0], device=DEV), dim=0).expand(H, 4).contiguous()
    group.activate_heliostats(mask, DEV)
    group.align_surfaces_with_incident_ray_directions(scenario.solar_tower.get_centers_of_target_areas(tix), inc, mask, DEV)
    points = group.active_surface_points.detach().requires_grad_(True)
    group.active_surface_points = points
    rt = HeliostatRayTracer(scenario, group, bitmap_resolution=torch.tensor([64, 64]))           # blocking by default
    rt.lbvh_compat = False
    flux, intercept, on_target, blocking = rt.trace_rays(inc, mask, tix)
    assert float(blocking.min()) < 0.8 and float(blocking[4]) == 1.0
    filtered = rt.filtered_blocking_primitive_indices.tolist()
    assert 0 not in filtered and 4 not in filtered and len(filtered) >= 2       # nobody stands behind 0, 4 is alone
    free = HeliostatRayTracer(scenario, group, blocking_active=False, bitmap_resolution=torch.tensor([64, 64]))
    flux_free, *_ = free.trace_rays(inc, mask, tix)
    assert float(flux[0].sum()) < 0.8 * float(flux_free[0].sum())
    assert rel_l2(n(flux[4]), n(flux_free[4])) < 1e-6
    # against the CPU restatement, forward and the gradient of the aligned points (origins + corners paths)
    planar = scenario.solar_tower.target_areas[0]
    prims = oracle.blocking_primitives(n(points))
    blk = dict(corners=prims[0], spans=prims[1], normals=prims[2], owner=np.arange(H, dtype=np.int32), lbvh_compat=False)
    common = (n(points), n(group.active_surface_normals), n(inc), n(rt.distortions_dataset.distortions_u),
              n(rt.distortions_dataset.distortions_e), n(tix), n(planar.centers), n(planar.normals), n(planar.dimensions),
              (64, 64))
    o_flux, o_fac, dbg = oracle.trace_fwd(*common, debug=True, blocking=blk)
    assert sorted(filtered) == np.nonzero(dbg["filter_flags"])[0].tolist()
    assert rel_l2(n(flux), o_flux) < 5e-4, rel_l2(n(flux), o_flux)
    np.testing.assert_allclose(n(blocking), o_fac[2], rtol=0, atol=2.0 / dbg["blocked"][0].size)
    w = torch.rand(flux.shape, generator=torch.Generator().manual_seed(5)).to(DEV)
    (flux * w).sum().backward()
    go, gn, gpc, gps, gpn = oracle.trace_bwd(*common, n(w), blocking=blk)
    import test_oracle_golden as tog
    _, _, _, g_sfc = tog._chain_primitive_grads(n(points), gpc, gps, gpn)
    assert rel_l2(n(points.grad), go + g_sfc) < 5e-3, rel_l2(n(points.grad), go + g_sfc)


def test_dense_rows_under_a_low_sun_through_the_ray_tracer():
    """The drop-in class on a field that NEEDS long candidate lists: 30 rows of four heliostats 2.6 m apart behind a target
    12 m up, the sun 6 degrees above the horizon - the beams of the back rows pass through dozens of mirrors.  Until round 4 such a
    heliostat came back as a NaN bitmap (more than 32 candidate rectangles); now: bitmaps, the three factors and the gradient
    of the aligned points (ray origins + rectangle corners) against the CPU restatement, which has no limit
    (artist/raytracing/blocking.py:212-354, 832-995)."""
    from artist_amd import HeliostatRayTracer, ops
    from artist_amd.scene import build_synthetic_scenario
    H = 120
    scenario, _ = build_synthetic_scenario(H, 8, n_eval=12, device=DEV, target_centers=((0.0, 0.0, 12.0, 1.0),))
    group = scenario.heliostat_field.heliostat_groups[0]
    i = torch.arange(H, device=DEV)
    group.positions = torch.stack([((i % 4) - 1.5) * 3.4, 30.0 + (i // 4) * 2.6, torch.zeros(H, device=DEV), torch.ones(H, device=DEV)], dim=1)
    mask = torch.ones(H, dtype=torch.int32, device=DEV)
    tix = torch.zeros(H, dtype=torch.long, device=DEV)
    inc = torch.nn.functional.normalize(torch.tensor([0.0, 0.9945, -0.1045, 0.0], device=DEV), dim=0).expand(H, 4).contiguous()
    group.activate_heliostats(mask, DEV)
    group.align_surfaces_with_incident_ray_directions(scenario.solar_tower.get_centers_of_target_areas(tix), inc, mask, DEV)
    points = group.active_surface_points.detach().requires_grad_(True)
    group.active_surface_points = points
    rt = HeliostatRayTracer(scenario, group, bitmap_resolution=torch.tensor([64, 64]))           # blocking by default
    rt.lbvh_compat = False
    flux, intercept, on_target, unblocked = rt.trace_rays(inc, mask, tix)
    counts = n(ops._LAST_BLOCKING[1])
    print(f"candidates per heliostat: max {counts.max()}, {int((counts > 32).sum())} of {H} beyond 32; unblocked fraction "
          f"{float(unblocked.min()):.3f} ... {float(unblocked.max()):.3f}")
    assert counts.max() > 40 and (counts > 32).sum() >= 10
    assert np.isfinite(n(flux)).all() and np.isfinite(n(unblocked)).all()
    planar = scenario.solar_tower.target_areas[0]
    prims = oracle.blocking_primitives(n(points))
    blk = dict(corners=prims[0], spans=prims[1], normals=prims[2], owner=np.arange(H, dtype=np.int32), lbvh_compat=False)
    common = (n(points), n(group.active_surface_normals), n(inc), n(rt.distortions_dataset.distortions_u),
              n(rt.distortions_dataset.distortions_e), n(tix), n(planar.centers), n(planar.normals), n(planar.dimensions), (64, 64))
    o_flux, o_fac, dbg = oracle.trace_fwd(*common, debug=True, blocking=blk)
    assert sorted(rt.filtered_blocking_primitive_indices.tolist()) == np.nonzero(dbg["filter_flags"])[0].tolist()
    err = rel_l2(n(flux), o_flux)
    print(f"flux {err:.2e}")
    assert err < 5e-6, err                                  # (measured 3.7e-7; 66 of the 120 heliostats list more than 32, the longest list 96)
    rays = dbg["blocked"][0].size
    np.testing.assert_allclose(n(unblocked), o_fac[2], rtol=0, atol=2.0 / rays)
    np.testing.assert_allclose(n(intercept), o_fac[0], rtol=0, atol=2.0 / rays)
    w = torch.rand(flux.shape, generator=torch.Generator().manual_seed(5)).to(DEV)
    (flux * w).sum().backward()
    go, gn, gpc, gps, gpn = oracle.trace_bwd(*common, n(w), blocking=blk)
    import test_oracle_golden as tog
    _, _, _, g_sfc = tog._chain_primitive_grads(n(points), gpc, gps, gpn)
    gerr = rel_l2(n(points.grad), go + g_sfc)
    print(f"gradient of the aligned points {gerr:.2e}")
    assert gerr < 1e-4, gerr                                # (measured 3.6e-6)


def test_per_point_rectangle_culling_changes_speed_only(monkeypatch):
    """The trace kernels drop, per surface point, the rectangles none of the point's rays can enter (cone against bounding
    sphere, then the three slabs of ``cone_mask``, ray_math.hpp) and the filter does the same with the grown boxes: both
    tests are conservative, so a dense field that shades itself heavily gives the same bits with the slab test off, and
    the same filtered set."""
    from artist_amd import HeliostatRayTracer
    from artist_amd.scene import build_synthetic_scenario
    H, R = 48, 6
    scenario, _ = build_synthetic_scenario(H, R, n_eval=24, device=DEV)
    group = scenario.heliostat_field.heliostat_groups[0]
    i = torch.arange(H, device=DEV)
    group.positions = torch.stack([((i % 8) - 3.5) * 4.2, 160.0 + (i // 8) * 5.0, torch.zeros(H, device=DEV),
                                   torch.ones(H, device=DEV)], dim=1)
    mask = torch.ones(H, dtype=torch.int32, device=DEV)
    tix = torch.zeros(H, dtype=torch.long, device=DEV)
    inc = torch.nn.functional.normalize(torch.tensor([0.0, 0.94, -0.34, 0.0], device=DEV), dim=0).expand(H, 4).contiguous()
    group.activate_heliostats(mask, DEV)
    group.align_surfaces_with_incident_ray_directions(scenario.solar_tower.get_centers_of_target_areas(tix), inc, mask, DEV)

    def run():
        rt = HeliostatRayTracer(scenario, group, bitmap_resolution=torch.tensor([64, 64]))
        rt.lbvh_compat = False
        flux, intercept, on_target, unblocked = rt.trace_rays(inc, mask, tix)
        return flux, unblocked, sorted(rt.filtered_blocking_primitive_indices.tolist())

    flux, unblocked, kept = run()
    assert float(unblocked.mean()) < 0.95 and len(kept) > H // 2              # the field does shade itself
    monkeypatch.setenv("ARTIST_HIP_BLOCK_SLABS", "0")
    flux0, unblocked0, kept0 = run()
    assert kept == kept0
    assert torch.equal(flux, flux0) and torch.equal(unblocked, unblocked0)


@pytest.mark.parametrize("lbvh_compat", [True, False])
def test_sharded_blocking_equals_single_rank(lbvh_compat):
    """Heliostat sharding with blocking on (each rank needs the rectangles of ALL heliostats, which the
    mirror builds on every rank like the reference, heliostat_ray_tracer.py:159-183).  The filtered set F is the union
    over the rays a rank traces, as in the reference it is the union over a batch (:445-461) - a rectangle that only
    another rank's rays hit is missing from this rank's F.  That cannot change this rank's flux beyond the soft mask's
    tails: a ray blocked by rectangle k hits k's box, so k is in the F of the rank that traces the ray; what F may lack
    are rectangles whose sigmoids contribute < 5e-12 to the exponent.  Asserted: per-heliostat rows
    and the summed per-target bitmaps agree to 1e-6, blocking factors to one ray."""
    from artist_amd import HeliostatRayTracer
    from artist_amd.scene import build_synthetic_scenario
    H, R = 7, 12
    scenario, _ = build_synthetic_scenario(H, R, n_eval=12, device=DEV)
    group = scenario.heliostat_field.heliostat_groups[0]
    group.positions = torch.tensor([[0.0, 150.0, 0.0, 1.0], [0.3, 147.2, 0.0, 1.0], [-0.8, 144.0, 0.0, 1.0], [2.0, 141.0, 0.0, 1.0],
                                    [-25.0, 100.0, 0.0, 1.0], [0.9, 138.5, 0.0, 1.0], [-24.5, 97.0, 0.0, 1.0]], device=DEV)
    mask = torch.ones(H, dtype=torch.int32, device=DEV)
    tix = torch.zeros(H, dtype=torch.long, device=DEV)
    inc = torch.nn.functional.normalize(torch.tensor([0.1, 0.95, -0.1, 0.0], device=DEV), dim=0).expand(H, 4).contiguous()
    group.activate_heliostats(mask, DEV)
    group.align_surfaces_with_incident_ray_directions(scenario.solar_tower.get_centers_of_target_areas(tix), inc, mask, DEV)
    res = torch.tensor([64, 64])
    whole = HeliostatRayTracer(scenario, group, bitmap_resolution=res)
    whole.lbvh_compat = lbvh_compat
    flux, intercept, on_target, blocking = whole.trace_rays(inc, mask, tix)
    per_target = whole.get_bitmaps_per_target(flux, tix)
    if not lbvh_compat:
        assert float(blocking.min()) < 0.8                       # the column does shade itself
    rays = R * group.active_surface_points.shape[1]
    acc = torch.zeros_like(per_target)
    for rank in range(3):
        part = HeliostatRayTracer(scenario, group, bitmap_resolution=res, world_size=3, rank=rank)
        part.lbvh_compat = lbvh_compat
        f, ic, ot, bl = part.trace_rays(inc, mask, tix)
        idx = part.get_sampler_indices()
        assert set(part.filtered_blocking_primitive_indices.tolist()) <= set(whole.filtered_blocking_primitive_indices.tolist())
        np.testing.assert_allclose(n(f), n(flux[idx]), rtol=0, atol=1e-6 * float(flux.max()))
        np.testing.assert_allclose(n(bl), n(blocking[idx]), rtol=0, atol=1.5 / rays)
        np.testing.assert_array_equal(n(ot), n(on_target[idx]))
        acc += part.get_bitmaps_per_target(f, tix[idx])
    assert rel_l2(n(acc), n(per_target)) < 1e-6, rel_l2(n(acc), n(per_target))


def test_target_index_out_of_range_is_reported_not_dereferenced(golden):
    """The C ABI trusts no index: a stale target index is found ON THE DEVICE (the heliostat is skipped, nothing is read
    out of bounds), the asynchronous call itself returns ART_OK, art_async_status reports ART_ETARGET, later trace calls
    refuse to start until the status is cleared - and the other heliostats of the launch are untouched."""
    from artist_amd import _lib, ops
    d = golden("small_deg3")
    inp = trace_inputs(d)
    good, _ = ops.trace_rays(**inp)
    bad = dict(inp)
    bad["target_idx"] = inp["target_idx"].clone()
    bad["target_idx"][1] = 7                                     # the tables hold one planar area
    flux, fac = ops.trace_rays(**bad)                            # asynchronous: no error yet
    stream = torch.cuda.current_stream(DEV).cuda_stream
    assert _lib.lib().art_async_status(stream, 0) == -2          # ART_ETARGET, still set
    with pytest.raises(IndexError, match="out of range"):
        ops.trace_rays(**inp)                                    # refused while the status is set
    with pytest.raises(IndexError, match="out of range"):
        ops.check_async_errors(DEV)                              # reports and clears
    assert _lib.lib().art_async_status(stream, 0) == 0
    assert float(flux[1].abs().sum()) == 0 and float(fac[:2, 1].abs().sum()) == 0      # (blocking factor: 1 = nothing blocked)
    for h in (0, 2, 3):
        np.testing.assert_array_equal(n(flux[h]), n(good[h]))
    again, _ = ops.trace_rays(**inp)                             # and the library works as before
    np.testing.assert_array_equal(n(again), n(good))
    # the mirror checks on the host in BOTH entry points before anything is launched
    bad["target_idx"][1] = -3
    flux, fac = ops.trace_rays(**bad)
    with pytest.raises(IndexError):
        ops.check_async_errors(DEV)


@pytest.mark.parametrize("blocking", [False, True])
def test_trace_rays_does_not_wait_for_the_device(blocking):
    """An epoch's ray tracing - ``trace_rays``, its backward pass, ``get_bitmaps_per_target`` - queues its work without a
    single host-device synchronisation once the per-tensor caches are warm (``torch.cuda.set_sync_debug_mode("error")``
    raises on one): the reference's per-call reads (target-area counts, the filtered index list, the active-heliostat
    indices) are cached, made on demand or left on the device, and the candidate-overflow check is the device's."""
    from artist_amd import HeliostatRayTracer
    from artist_amd.scene import build_synthetic_scenario
    H = 12
    scenario, _ = build_synthetic_scenario(H, 8, n_eval=16, device=DEV)
    group = scenario.heliostat_field.heliostat_groups[0]
    mask = torch.ones(H, dtype=torch.int32, device=DEV)
    tix = torch.zeros(H, dtype=torch.long, device=DEV)
    inc = torch.nn.functional.normalize(torch.tensor([0.0, 0.94, -0.34, 0.0], device=DEV), dim=0).expand(H, 4).contiguous()
    group.activate_heliostats(mask, DEV)
    group.align_surfaces_with_incident_ray_directions(scenario.solar_tower.get_centers_of_target_areas(tix), inc, mask, DEV)
    points = group.active_surface_points.detach().requires_grad_(True)
    group.active_surface_points = points
    group.active_surface_normals = group.active_surface_normals.detach()
    rt = HeliostatRayTracer(scenario, group, blocking_active=blocking, bitmap_resolution=torch.tensor([64, 64]))

    def epoch():
        flux, intercept, on_target, unblocked = rt.trace_rays(inc, mask, tix)
        points.grad = None
        (flux * weights).sum().backward(retain_graph=blocking)      # (the rectangles hang off the constructor's graph)
        return rt.get_bitmaps_per_target(flux.detach(), tix)

    weights = torch.rand((H, 64, 64), device=DEV)
    epoch()                                       # warm-up: per-tensor caches, lazy allocations
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        per_target = epoch()
    finally:
        torch.cuda.set_sync_debug_mode("default")
    assert float(per_target.sum()) > 0
    if blocking:
        assert rt.filtered_blocking_primitive_indices is not None       # (made on demand: this read does synchronise)


def test_too_many_candidate_rectangles_are_reported_by_the_device(golden, monkeypatch):
    """More rectangles inside one heliostat's ray cone than its candidate ROW holds (the caller's workspace; 32 entries here -
    the kernels themselves take lists of any length, test_more_candidates_than_the_tables_hold): the filter says so through
    the device status word - no host read of the candidate counts in every call - ``check_async_errors`` raises, later trace
    calls refuse to start until the status is cleared, and the library works as before afterwards."""
    from artist_amd import ArtistHipError, _lib, ops, trace_rays
    monkeypatch.setattr(ops, "BLOCKING_CANDIDATES", 32)
    d = golden("small_blocking")
    inp = trace_inputs(d)
    blk = blocking_inputs(d)
    good, _, flags = trace_rays(**inp, blocking=blk)
    k = int(np.nonzero(n(flags))[0][0])                          # a rectangle that does block somebody
    shifts = torch.arange(1, 41, device=DEV, dtype=torch.float32)[:, None, None] * 1e-3
    extra = blk["corners"][k][None] + shifts * blk["normals"][k][None, None, :] * torch.tensor([1.0, 1.0, 1.0, 0.0], device=DEV)
    crowded = dict(blk, corners=torch.cat([blk["corners"], extra]), spans=torch.cat([blk["spans"], blk["spans"][k][None].expand(40, -1, -1)]),
                   normals=torch.cat([blk["normals"], blk["normals"][k][None].expand(40, -1)]))
    try:
        trace_rays(**inp, blocking=crowded)                       # asynchronous: normally no error yet
    except ArtistHipError:
        pass                                                      # (the status word may already be visible to the host)
    stream = torch.cuda.current_stream(DEV).cuda_stream
    assert _lib.lib().art_async_status(stream, 0) == -5           # ART_ECANDIDATES, still set
    with pytest.raises(ArtistHipError, match="blocking rectangles"):
        trace_rays(**inp, blocking=blk)                           # refused while the status is set
    with pytest.raises(ArtistHipError, match="blocking rectangles"):
        ops.check_async_errors(DEV)                               # reports and clears
    assert _lib.lib().art_async_status(stream, 0) == 0
    again, _, _ = trace_rays(**inp, blocking=blk)
    np.testing.assert_array_equal(n(again), n(good))


# ---------------------------------------------------------------------------------------------
# Flux epilogue (artist/flux/bitmap.py:121-246, artist/optim/loss.py:251-410)
# ---------------------------------------------------------------------------------------------
def _crop_tower():
    from artist_amd.scene import SolarTower, TowerTargetAreasCylindrical, TowerTargetAreasPlanar
    z4 = torch.zeros(1, 4, device=DEV)
    planar = TowerTargetAreasPlanar(["multi_focus_tower"], z4, z4, torch.tensor([[3.0, 3.0]], device=DEV))
    cyl = TowerTargetAreasCylindrical(["receiver"], z4, z4, z4, torch.tensor([1.0], device=DEV),
                                      torch.tensor([3.0], device=DEV), torch.tensor([3.0], device=DEV))
    return SolarTower([planar, cyl], device=DEV)


def test_flux_crop_known_answers_and_autograd(golden):
    from artist_amd.flux import crop_flux_distributions_around_center
    ka = golden("known_answers")
    tower = _crop_tower()
    for i in range(int(ka["crop_count"])):          # tests/flux/test_bitmap.py:66-173, through the drop-in function
        size = float(ka[f"crop{i}_size"])
        got = crop_flux_distributions_around_center(t(ka[f"crop{i}_image"]), tower, t(ka[f"crop{i}_target_idx"]), size, size)
        torch.testing.assert_close(got.cpu(), torch.from_numpy(ka[f"crop{i}_expected"]), rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(n(got), ka[f"crop{i}_reference"], rtol=0, atol=2e-6)
        assert not torch.isnan(got).any()
    img = t(ka["cropgrad_image"]).requires_grad_(True)
    out = crop_flux_distributions_around_center(img, tower, t(ka["cropgrad_target_idx"]))          # default 6 m x 6 m
    (out * t(ka["cropgrad_weights"])).sum().backward()
    # fp32: the reference's own result is 7e-6 / 4e-6 (forward / gradient) from its fp64 run
    assert rel_l2(n(out), ka["cropgrad_f32_out"]) < 2e-5 and rel_l2(n(out), ka["cropgrad_f64_out"]) < 2e-5
    assert rel_l2(n(img.grad), ka["cropgrad_f32_grad"]) < 2e-5 and rel_l2(n(img.grad), ka["cropgrad_f64_grad"]) < 2e-5
    o_out, _ = oracle.flux_crop(ka["cropgrad_image"], ka["cropgrad_dims"])
    assert rel_l2(n(out), o_out) < 2e-5
    # deterministic backward: bit-identical on a second run
    img2 = t(ka["cropgrad_image"]).requires_grad_(True)
    (crop_flux_distributions_around_center(img2, tower, t(ka["cropgrad_target_idx"])) * t(ka["cropgrad_weights"])).sum().backward()
    assert torch.equal(img.grad, img2.grad)


def test_flux_losses(golden):
    from artist_amd.flux import KLDivergenceLoss, PixelLoss
    ka = golden("known_answers")
    truth, w = t(ka["loss_ground_truth"]), t(ka["loss_sample_weights"])
    for name, loss_cls in (("pixel", PixelLoss), ("kl", KLDivergenceLoss)):
        pred = (t(ka["cropgrad_image"]) + 0.05).requires_grad_(True)
        per_sample = loss_cls()(pred, truth, reduction_dimensions=(1, 2))
        (per_sample * w).sum().backward()
        # sums are accumulated in fp64 here: closer to the reference's fp64 run than its own fp32 run is (KL: 3e-5)
        yard = rel_l2(ka[f"loss_{name}_f32"], ka[f"loss_{name}_f64"])
        assert rel_l2(n(per_sample), ka[f"loss_{name}_f64"]) < 1e-5
        assert rel_l2(n(per_sample), ka[f"loss_{name}_f32"]) < max(2 * yard, 1e-5)
        yard = rel_l2(ka[f"loss_{name}_f32_grad"], ka[f"loss_{name}_f64_grad"])
        assert rel_l2(n(pred.grad), ka[f"loss_{name}_f64_grad"]) < 1e-5
        assert rel_l2(n(pred.grad), ka[f"loss_{name}_f32_grad"]) < max(2 * yard, 1e-5)
        with pytest.raises(ValueError, match="reduction_dimensions"):       # artist/optim/loss.py:300-311, 376-383
            loss_cls()(pred, truth)


@pytest.mark.parametrize("parts", ["1", "2", "4", None])
def test_fused_crop_pixel_loss(parts, monkeypatch):
    """art_flux_crop_pixel_loss_fwd/bwd (crop + PixelLoss fused; the backward pass one kernel over the residual the forward pass
    kept) against the two separate ops the reference's epoch calls (bitmap.py:121-246, loss.py:251-318), forward and gradient,
    incl. an empty bitmap, a spot cut by the border and non-square resolutions: the same numbers to 1e-6 (a bitmap's rows are
    summed in four parts added in part order, and the loss adjoint's factor 2 gl / sum(truth) multiplies the gathered residual
    instead of every pixel of dL/dcrop), and the same BITS with one, two or four workgroups per bitmap - i.e. at every batch
    size - and from run to run (round 3: one workgroup per bitmap summed differently from two or four; advisor finding)."""
    from artist_amd import PixelLoss
    from artist_amd.flux import FluxCrop, FluxCropPixelLoss
    monkeypatch.setenv("ARTIST_HIP_DEBUG", "1")
    if parts is not None:
        monkeypatch.setenv("ARTIST_HIP_LOSS_PARTS", parts)
    gen = torch.Generator(device=DEV).manual_seed(4)
    for (B, Hh, W) in [(5, 256, 256), (3, 60, 100), (2, 33, 17)]:
        ys, xs = torch.meshgrid(torch.arange(Hh, device=DEV, dtype=torch.float32), torch.arange(W, device=DEV, dtype=torch.float32),
                                indexing="ij")
        cx = torch.rand(B, generator=gen, device=DEV) * W
        cy = torch.rand(B, generator=gen, device=DEV) * Hh
        sig = 3.0 + 10.0 * torch.rand(B, generator=gen, device=DEV)
        flux = torch.exp(-((xs[None] - cx[:, None, None]) ** 2 + (ys[None] - cy[:, None, None]) ** 2) / (2 * sig[:, None, None] ** 2))
        flux = flux * (0.5 + torch.rand((B, Hh, W), generator=gen, device=DEV))
        flux[0] = 0.0                                            # an empty bitmap
        truth = torch.rand((B, Hh, W), generator=gen, device=DEV) + 0.1
        dims = torch.tensor([[8.0, 8.0], [7.0, 9.0], [3.0, 5.0], [12.0, 4.0], [6.0, 6.0]], device=DEV)[:B].contiguous()
        a = flux.clone().requires_grad_(True)
        loss_a = PixelLoss()(FluxCrop.apply(a, dims, 6.0, 5.0), truth, reduction_dimensions=(1, 2))
        b = flux.clone().requires_grad_(True)
        loss_b = FluxCropPixelLoss.apply(b, dims, truth, 6.0, 5.0)
        w = torch.rand(B, generator=gen, device=DEV)
        (loss_a * w).sum().backward()
        (loss_b * w).sum().backward()
        np.testing.assert_allclose(n(loss_b), n(loss_a), rtol=1e-6, atol=0)
        assert rel_l2(n(b.grad), n(a.grad)) < 1e-6, rel_l2(n(b.grad), n(a.grad))
        with torch.no_grad():                                    # a forward-only call (no residual kept): the same loss bits
            np.testing.assert_array_equal(n(FluxCropPixelLoss.apply(flux, dims, truth, 6.0, 5.0)), n(loss_b))
        c = flux.clone().requires_grad_(True)                    # the same call again: the same bits
        loss_c = FluxCropPixelLoss.apply(c, dims, truth, 6.0, 5.0)
        (loss_c * w).sum().backward()
        np.testing.assert_array_equal(n(loss_c), n(loss_b))
        np.testing.assert_array_equal(n(c.grad), n(b.grad))
        for other in ("1", "2", "4"):                            # ... and with every number of workgroups per bitmap
            monkeypatch.setenv("ARTIST_HIP_LOSS_PARTS", other)
            e = flux.clone().requires_grad_(True)
            loss_e = FluxCropPixelLoss.apply(e, dims, truth, 6.0, 5.0)
            (loss_e * w).sum().backward()
            np.testing.assert_array_equal(n(loss_e), n(loss_b))
            np.testing.assert_array_equal(n(e.grad), n(b.grad))
        if parts is not None:
            monkeypatch.setenv("ARTIST_HIP_LOSS_PARTS", parts)
        else:
            monkeypatch.delenv("ARTIST_HIP_LOSS_PARTS")


def test_flux_epilogue_full_size_properties():
    """256 x 256 bitmaps of a traced field: crop of a centred symmetric spot with crop size = target size is the
    identity, the crop is translation-equivariant, and the whole epilogue is differentiable back to the mirror."""
    from artist_amd import HeliostatRayTracer
    from artist_amd.flux import PixelLoss, crop_flux_distributions_around_center
    from artist_amd.scene import build_synthetic_scenario
    H = 8
    scenario, _ = build_synthetic_scenario(H, 20, n_eval=20, device=DEV)
    group = scenario.heliostat_field.heliostat_groups[0]
    mask = torch.ones(H, dtype=torch.int32, device=DEV)
    tix = torch.zeros(H, dtype=torch.long, device=DEV)
    inc = torch.tensor([0.0, 1.0, 0.0, 0.0], device=DEV).expand(H, 4).contiguous()
    group.activate_heliostats(mask, DEV)
    group.align_surfaces_with_incident_ray_directions(scenario.solar_tower.get_centers_of_target_areas(tix), inc, mask, DEV)
    normals = group.active_surface_normals.detach().requires_grad_(True)
    group.active_surface_normals = normals
    rt = HeliostatRayTracer(scenario, group, blocking_active=False)
    flux, *_ = rt.trace_rays(inc, mask, tix)
    tower = scenario.solar_tower                                              # 8 m x 8 m planar receiver
    same = crop_flux_distributions_around_center(flux, tower, tix, 8.0, 8.0)  # scale 1: only the recentring acts
    com = lambda f: ((f * torch.arange(256, device=DEV)).sum((1, 2)) / f.sum((1, 2)),
                     (f * torch.arange(256, device=DEV)[:, None]).sum((1, 2)) / f.sum((1, 2)))
    cx, cy = com(same)
    assert float((cx - 127.5).abs().max()) < 0.6 and float((cy - 127.5).abs().max()) < 0.6       # spot centred
    # bilinear resampling keeps the integral while nothing leaves the frame (a clipped spot loses its tail)
    inside = flux.sum((1, 2)) > 0
    assert float(((same.sum((1, 2)) - flux.sum((1, 2))).abs() / flux.sum((1, 2)).clamp(min=1))[inside].max()) < 0.05
    zoom = crop_flux_distributions_around_center(flux, tower, tix)            # 6 m of 8 m: magnified 4/3
    assert float((zoom.sum((1, 2)) / same.sum((1, 2)).clamp(min=1))[inside].mean()) > 1.3
    loss = PixelLoss()(zoom, same.detach() + 1.0, reduction_dimensions=(1, 2)).sum()
    loss.backward()
    assert torch.isfinite(normals.grad).all() and float(normals.grad.abs().max()) > 0


# ---------------------------------------------------------------------------------------------
# The reference's own scenario files (fitted surfaces, rigid-body kinematics, planar + cylindrical areas, blocking)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", REAL_CASES)
def test_real_scenarios(golden, name):
    from artist_amd import trace_rays
    from artist_amd.blocking import create_blocking_primitives_rectangles_by_index
    d, d64 = golden(name), golden(name + "_f64")
    inp = trace_inputs(d)
    points = inp["origins"].requires_grad_(True)
    inp["normals"].requires_grad_(True)
    kw = dict(cyl=cyl_inputs(d))
    blocking = "prim_corners" in d
    if blocking:
        np.testing.assert_array_equal(d["blocking_surfaces"], d["aligned_points"])      # every heliostat is active
        corners, spans, normals = create_blocking_primitives_rectangles_by_index(points)
        kw["blocking"] = dict(corners=corners, spans=spans, normals=normals, owner=t(d["owner"]).int())
    out = trace_rays(**inp, **kw)
    flux, fac = out[0], out[1]
    if blocking:
        np.testing.assert_array_equal(np.nonzero(n(out[2]))[0], d["filter_indices"])
    (flux * t(d["loss_weights"])).sum().backward()
    yard = rel_l2(d["flux"], d64["flux"])
    assert rel_l2(n(flux), d["flux"]) < max(yard, 2e-4), (rel_l2(n(flux), d["flux"]), yard)
    assert rel_l2(n(flux), d64["flux"]) < 2 * max(yard, 2e-4)
    rays = d["distortions_u"][0].size
    for row, key in enumerate(("intercept", "on_target", "blocking")):
        np.testing.assert_allclose(n(fac[row]), d[key], rtol=0, atol=2.5 / rays)
    for got, key in ((points.grad, "grad_aligned_points"), (inp["normals"].grad, "grad_aligned_normals")):
        yard = rel_l2(d[key], d64[key])
        assert rel_l2(n(got), d[key]) < max(2 * yard, 1e-3), (key, rel_l2(n(got), d[key]), yard)
        assert rel_l2(n(got), d64[key]) < max(3 * yard, 1e-3), (key, rel_l2(n(got), d64[key]), yard)


# ---------------------------------------------------------------------------------------------
# Randomised scenes: every feature at once (planar + cylindrical areas, blocking rectangles, ragged sizes, sample
# counts above the 4-ray groups and the 128-sample range), HIP vs the CPU oracle on the same inputs.
# ---------------------------------------------------------------------------------------------
def _random_feature_scene(seed, H, P, R):
    g = torch.Generator().manual_seed(seed)
    rnd = lambda *shape: torch.rand(shape, generator=g)
    centre = torch.tensor([0.0, 0.0, 40.0])
    pos = torch.stack([(rnd(H) - 0.5) * 60.0, 50.0 + rnd(H) * 60.0, rnd(H) * 2.0], dim=1)
    sun = torch.nn.functional.normalize(torch.tensor([0.15, 0.9, -0.4]) + 0.1 * (rnd(3) - 0.5), dim=0)
    target_idx = torch.randint(0, 3, (H,), generator=g).int()                    # 0, 1 planar; 2 = the cylinder
    aim = centre.expand(H, 3).clone()
    hdir = (pos - centre) * torch.tensor([1.0, 1.0, 0.0])
    on_cyl = target_idx == 2
    aim[on_cyl] = (centre + 20.0 * hdir / hdir.norm(dim=1, keepdim=True))[on_cyl]   # the mantle of the 20 m cylinder
    to_t = torch.nn.functional.normalize(aim - pos, dim=1)
    n0 = torch.nn.functional.normalize(to_t - sun, dim=1)
    ex = torch.nn.functional.normalize(torch.linalg.cross(n0, torch.tensor([[0.0, 0.0, 1.0]]).expand(H, 3)), dim=1)
    ey = torch.linalg.cross(n0, ex)
    local = (rnd(H, P, 2) - 0.5) * torch.tensor([3.0, 2.5])
    pts = pos[:, None] + local[..., :1] * ex[:, None] + local[..., 1:] * ey[:, None]
    nrm = torch.nn.functional.normalize(n0[:, None] + 1.5e-3 * torch.randn((H, P, 3), generator=g), dim=-1)
    origins = torch.cat([pts, torch.ones(H, P, 1)], -1)
    normals = torch.cat([nrm, torch.zeros(H, P, 1)], -1)
    incident = torch.cat([sun, torch.zeros(1)]).expand(H, 4).contiguous()
    both = 2.5e-3 * torch.randn((H, R, P, 2), generator=g)
    planes = dict(centers=torch.tensor([[0.0, 0.0, 40.0, 1.0], [0.5, -1.0, 41.0, 1.0]]),
                  normals=torch.nn.functional.normalize(torch.tensor([[0.0, 1.0, 0.0, 0.0], [0.1, 0.95, 0.3, 0.0]]), dim=1),
                  dims=torch.tensor([[9.0, 7.0], [6.0, 8.0]]))
    cyl = dict(centers=torch.tensor([[0.0, 0.0, 40.0, 1.0]]), normals=torch.tensor([[0.0, 1.0, 0.0, 0.0]]),
               axes=torch.tensor([[0.0, 0.0, 1.0, 0.0]]), radii=torch.tensor([20.0]), heights=torch.tensor([14.0]),
               opening=torch.tensor([2.8]))
    # blocking rectangles: the mirrors themselves (bounding rectangle of the random points) + two free-standing
    # panels placed into the beams of heliostats 0 and 1, one of them grazing the beam edge
    lo, hi_ = local.amin(1), local.amax(1)
    def rect(c0, su, sv):
        return torch.stack([c0, c0 + su, c0 + su + sv, c0 + sv], dim=0)
    corners = []
    for h in range(H):
        c0 = pos[h] + lo[h, 0] * ex[h] + lo[h, 1] * ey[h]
        corners.append(rect(c0, (hi_[h, 1] - lo[h, 1]) * ey[h], (hi_[h, 0] - lo[h, 0]) * ex[h]))
    for h, shift in ((0, 0.0), (1 % H, 1.6)):
        mid = pos[h] + 12.0 * to_t[h] + shift * ex[h]
        corners.append(rect(mid - 1.0 * ex[h] - 0.8 * ey[h], 1.6 * ey[h], 2.0 * ex[h]))
    corners = torch.stack(corners)
    corners = torch.cat([corners, torch.ones(corners.shape[0], 4, 1)], -1)
    spans = torch.stack((corners[:, 1] - corners[:, 0], corners[:, 3] - corners[:, 0]), dim=1)
    pn = torch.nn.functional.normalize(torch.linalg.cross(spans[:, 0, :3], spans[:, 1, :3]), dim=-1)
    pn = torch.cat([pn, torch.zeros(pn.shape[0], 1)], -1)
    return dict(origins=origins, normals=normals, incident=incident, both=both, target_idx=target_idx, planes=planes,
                cyl=cyl, prims=dict(corners=corners, spans=spans, normals=pn, owner=torch.arange(H, dtype=torch.int32)))


@pytest.mark.parametrize("seed,H,P,R,res,interleaved,lbvh_compat",
                         [(0, 5, 192, 7, (96, 64), True, False), (1, 4, 333, 130, (64, 80), False, False),
                          (2, 7, 64, 5, (48, 48), True, True), (3, 3, 1100, 9, (128, 96), True, False)])
def test_random_scenes_all_features(seed, H, P, R, res, interleaved, lbvh_compat):
    from artist_amd import per_target_sum, trace_rays
    sc = _random_feature_scene(seed, H, P, R)
    dv = lambda x: x.to(DEV)
    both = dv(sc["both"])
    du, de = (both[..., 0], both[..., 1]) if interleaved else (both[..., 0].contiguous(), both[..., 1].contiguous())
    o, nn_ = dv(sc["origins"]).requires_grad_(True), dv(sc["normals"]).requires_grad_(True)
    prims = {k: dv(v) for k, v in sc["prims"].items()}
    prims["corners"].requires_grad_(True)
    cyl_dev = tuple(dv(sc["cyl"][k]) for k in ("centers", "normals", "axes", "radii", "heights", "opening"))
    kw = dict(ray_magnitude=0.7, extinction=0.05, reflectivity=0.9, resolution=res, cyl=cyl_dev,
              blocking=dict(prims, lbvh_compat=lbvh_compat))
    args = (o, nn_, dv(sc["incident"]), du, de, dv(sc["target_idx"]), dv(sc["planes"]["centers"]), dv(sc["planes"]["normals"]),
            dv(sc["planes"]["dims"]))
    flux, fac, flags = trace_rays(*args, **kw)
    f32 = lambda x: np.ascontiguousarray(x.detach().cpu().numpy())
    oracle_args = (f32(sc["origins"]), f32(sc["normals"]), f32(sc["incident"]), f32(sc["both"][..., 0]), f32(sc["both"][..., 1]),
                   f32(sc["target_idx"]), f32(sc["planes"]["centers"]), f32(sc["planes"]["normals"]), f32(sc["planes"]["dims"]), res)
    okw = dict(cyl={k: f32(v) for k, v in sc["cyl"].items()},
               blocking=dict({k: f32(v) for k, v in sc["prims"].items()}, lbvh_compat=lbvh_compat))
    o_flux, o_fac, dbg = oracle.trace_fwd(*oracle_args, 0.7, 0.05, 0.9, debug=True, **okw)
    np.testing.assert_array_equal(np.nonzero(n(flags))[0], np.nonzero(dbg["filter_flags"])[0])
    assert o_flux.sum() > 0 and (dbg["blocked"] > 0.5).any() == bool((n(fac[2]) < 1).any())
    assert rel_l2(n(flux), o_flux) < 1e-5, rel_l2(n(flux), o_flux)           # measured 1e-6 ... 1.8e-6
    np.testing.assert_allclose(n(fac), o_fac, rtol=0, atol=3.0 / (R * P))
    # fused per-target mode on the same scene
    flux_t, _, _ = trace_rays(*args, per_target=True, **kw)
    summed = per_target_sum(flux.detach(), args[5], 3)
    np.testing.assert_allclose(n(flux_t), n(summed), rtol=0, atol=3e-6 * float(summed.max()))
    # gradients (origins, normals, rectangle corners 0 through the three tables)
    w = torch.rand(flux.shape, generator=torch.Generator().manual_seed(seed)).to(DEV)
    (flux * w).sum().backward()
    go, gn, gpc, gps, gpn = oracle.trace_bwd(*oracle_args, f32(w), 0.7, 0.05, 0.9, **okw)
    # measured 3e-7 ... 3.4e-6 (corners 1e-7 ... 1.7e-6); the fp32 oracle itself is 4e-5 ... 1.7e-2 from the fp64 oracle here
    assert rel_l2(n(o.grad), go) < 2e-5, rel_l2(n(o.grad), go)
    assert rel_l2(n(nn_.grad), gn) < 2e-5, rel_l2(n(nn_.grad), gn)
    if np.linalg.norm(gpc) > 0:
        assert rel_l2(n(prims["corners"].grad), gpc) < 2e-5, rel_l2(n(prims["corners"].grad), gpc)


@pytest.mark.parametrize("variant", ["blocking_on_planes", "mixed_tower_no_blocking"])
@pytest.mark.parametrize("seed,H,P,R", [(5, 6, 300, 11), (6, 9, 128, 40), (7, 4, 1500, 6), (8, 12, 77, 17), (9, 5, 640, 24)])
def test_random_scenes_split_calls(seed, H, P, R, variant, monkeypatch):
    """The two kinds of split call on random scenes, against the oracle: blocking on with planar receivers only (lean
    launch for the heliostats without candidate rectangles beside the blocking launch, whose workgroups are mapped to the
    blocked heliostats in order) and a tower with planar and cylindrical receivers without blocking (lean launch for the
    planes beside the cylinder launch).  ``ARTIST_HIP_FWD_BLOCKS=1`` keeps the samples in one chunk, so that the backward
    call splits as well (a field this small would otherwise be chunked, and chunked launches are not split)."""
    from artist_amd import trace_rays
    monkeypatch.setenv("ARTIST_HIP_FWD_BLOCKS", "1")
    sc = _random_feature_scene(seed, H, P, R)
    dv = lambda x: x.to(DEV)
    f32 = lambda x: np.ascontiguousarray(x.detach().cpu().numpy())
    tix = sc["target_idx"].clone()
    kw, okw = {}, {}
    if variant == "blocking_on_planes":
        tix = tix % 2                                                     # the two planes only
        prims = {k: dv(v) for k, v in sc["prims"].items()}
        kw["blocking"] = dict(prims, lbvh_compat=False)
        okw["blocking"] = dict({k: f32(v) for k, v in sc["prims"].items()}, lbvh_compat=False)
    else:
        kw["cyl"] = tuple(dv(sc["cyl"][k]) for k in ("centers", "normals", "axes", "radii", "heights", "opening"))
        okw["cyl"] = {k: f32(v) for k, v in sc["cyl"].items()}
    o, nn_ = dv(sc["origins"]).requires_grad_(True), dv(sc["normals"]).requires_grad_(True)
    both = dv(sc["both"])
    args = (o, nn_, dv(sc["incident"]), both[..., 0], both[..., 1], dv(tix), dv(sc["planes"]["centers"]),
            dv(sc["planes"]["normals"]), dv(sc["planes"]["dims"]))
    res = (96, 64)
    out = trace_rays(*args, ray_magnitude=0.7, extinction=0.05, reflectivity=0.9, resolution=res, **kw)
    flux, fac = out[0], out[1]
    oracle_args = (f32(sc["origins"]), f32(sc["normals"]), f32(sc["incident"]), f32(sc["both"][..., 0]), f32(sc["both"][..., 1]),
                   f32(tix), f32(sc["planes"]["centers"]), f32(sc["planes"]["normals"]), f32(sc["planes"]["dims"]), res)
    o_flux, o_fac = oracle.trace_fwd(*oracle_args, 0.7, 0.05, 0.9, **okw)[:2]
    assert o_flux.sum() > 0
    planar = n(tix) < 2
    assert rel_l2(n(flux)[planar], o_flux[planar]) < 1e-5, rel_l2(n(flux)[planar], o_flux[planar])
    if (~planar).any():                                                   # (cylinders: the fp32 quadratic is ill-conditioned)
        assert rel_l2(n(flux)[~planar], o_flux[~planar]) < 2e-3
    np.testing.assert_allclose(n(fac), o_fac, rtol=0, atol=3.0 / (R * P) if variant == "blocking_on_planes" else 1e-3)
    np.testing.assert_array_equal(n(fac[:2])[:, planar], o_fac[:2][:, planar])       # ray counters of the planes: exact
    w = torch.rand(flux.shape, generator=torch.Generator().manual_seed(seed)).to(DEV)
    (flux * w).sum().backward()
    grads = oracle.trace_bwd(*oracle_args, f32(w), 0.7, 0.05, 0.9, **okw)
    go, gn = grads[0], grads[1]
    tol = 2e-5 if variant == "blocking_on_planes" else 2e-3
    assert np.isfinite(go).all() and np.isfinite(gn).all(), "the oracle's gradients are not finite"
    if not (bool(torch.isfinite(o.grad).all()) and bool(torch.isfinite(nn_.grad).all())):
        bad_o, bad_n = ~np.isfinite(n(o.grad)).all(axis=2), ~np.isfinite(n(nn_.grad)).all(axis=2)
        raise AssertionError(("non-finite HIP gradients (heliostat, point) origins / normals", np.argwhere(bad_o).tolist()[:40],
                              np.argwhere(bad_n).tolist()[:40], "counts", int(bad_o.sum()), int(bad_n.sum()), "unblocked fraction",
                              n(fac[2]).tolist(), "values", n(o.grad)[bad_o][:4].tolist()))
    assert rel_l2(n(o.grad)[planar], go[planar]) < 2e-5, rel_l2(n(o.grad)[planar], go[planar])
    assert rel_l2(n(nn_.grad)[planar], gn[planar]) < 2e-5, rel_l2(n(nn_.grad)[planar], gn[planar])
    assert rel_l2(n(o.grad), go) < tol and rel_l2(n(nn_.grad), gn) < tol


def test_blocking_backward_with_facet_sized_items(monkeypatch):
    """Blocking backward on a planar tower at the metric config's point count (P = 10 000 in four facets, ~100 heliostats): the
    call takes the lean blocking item in FACET-sized blocks, whose rectangle-gradient slabs are more numerous than the generic
    geometry's - round 3 sized the scratch buffer from the latter and this call raised ART_EINVAL (advisor finding; every other
    blocking-backward test runs at P <= 1500, where the two geometries coincide).  Flux and point gradients of a sample of
    heliostats against the oracle (its blocking has no culling: 100 rectangles per ray, so not the whole field); the rectangle
    gradients - sums over every ray of the field - against the same call through the generic item (round 3's path at this
    size), whose slabs are cut differently."""
    from artist_amd import _lib, trace_rays
    H, P, R, facet = 100, 10000, 24, 2500
    lib = _lib.lib()
    # the case is only worth its time while the two geometries differ at this size
    assert lib.art_trace_bwd_scratch_need(H, R, P, facet, 2, 0, 8) != lib.art_trace_bwd_scratch_need(H, R, P, facet, 0, 1, 8)
    sc = _random_feature_scene(21, H, P, R)
    dv = lambda x: x.to(DEV)
    f32 = lambda x: np.ascontiguousarray(x.detach().cpu().numpy())
    tix = sc["target_idx"] % 2
    res = (128, 128)
    both = dv(sc["both"])
    w = torch.rand((H, res[1], res[0]), generator=torch.Generator().manual_seed(21)).to(DEV)

    def run():
        prims = {k: dv(v) for k, v in sc["prims"].items()}
        prims["corners"].requires_grad_(True)
        o, nn_ = dv(sc["origins"]).requires_grad_(True), dv(sc["normals"]).requires_grad_(True)
        args = (o, nn_, dv(sc["incident"]), both[..., 0], both[..., 1], dv(tix), dv(sc["planes"]["centers"]),
                dv(sc["planes"]["normals"]), dv(sc["planes"]["dims"]))
        flux, fac, _ = trace_rays(*args, ray_magnitude=0.7, extinction=0.05, reflectivity=0.9, resolution=res,
                                  blocking=dict(prims, lbvh_compat=False), points_per_facet=facet)
        (flux * w).sum().backward()                       # raised ArtistHipError(ART_EINVAL) before the fix
        torch.cuda.synchronize()
        return n(flux), n(fac), n(o.grad), n(nn_.grad), n(prims["corners"].grad)

    flux, fac, go_hip, gn_hip, gpc_hip = run()
    monkeypatch.setenv("ARTIST_HIP_DEBUG", "1")
    monkeypatch.setenv("ARTIST_HIP_BLOCK_LEAN", "0")
    flux_g, fac_g, go_g, gn_g, gpc_g = run()
    assert rel_l2(flux, flux_g) < 1e-6          # (the lean and the generic ray body differ by the order of two products)
    assert np.linalg.norm(gpc_g) > 0 and rel_l2(gpc_hip, gpc_g) < 1e-5, rel_l2(gpc_hip, gpc_g)
    assert rel_l2(go_hip, go_g) < 1e-5 and rel_l2(gn_hip, gn_g) < 1e-5
    # Oracle on a sample (1 has a free-standing panel in its beam, 50 is shaded by neighbours; heliostat 0, whose panel only
    # grazes the beam, is left out: ONE of its 240 000 rays lands in the neighbouring pixel cell in one implementation and
    # not in the other - measured flux 3.9e-5, gradient 6e-3 from a single point, every other point at 1e-7 - the fp32
    # discontinuity DESIGN.md section 3 describes).  The rectangles are the whole field's; the reference's filter keeps only
    # those that some ray of the BATCH can meet (blocking.py:832-995), so the third factor of a sample is not the field's.
    sel = [1, 3, 17, 50, 64]
    sub = lambda x: f32(x[sel])
    oracle_args = (sub(sc["origins"]), sub(sc["normals"]), sub(sc["incident"]), sub(sc["both"][..., 0]), sub(sc["both"][..., 1]),
                   sub(tix), f32(sc["planes"]["centers"]), f32(sc["planes"]["normals"]), f32(sc["planes"]["dims"]), res)
    okw = dict(blocking=dict({k: (sub(v) if k == "owner" else f32(v)) for k, v in sc["prims"].items()}, lbvh_compat=False))
    o_flux, o_fac = oracle.trace_fwd(*oracle_args, 0.7, 0.05, 0.9, **okw)[:2]
    assert o_flux.sum() > 0 and (fac[2][sel] < 1).any()
    assert rel_l2(flux[sel], o_flux) < 1e-5, rel_l2(flux[sel], o_flux)
    np.testing.assert_array_equal(fac[:2][:, sel], o_fac[:2])
    go, gn = oracle.trace_bwd(*oracle_args, f32(w[sel]), 0.7, 0.05, 0.9, **okw)[:2]
    assert rel_l2(go_hip[sel], go) < 2e-5, rel_l2(go_hip[sel], go)
    assert rel_l2(gn_hip[sel], gn) < 2e-5, rel_l2(gn_hip[sel], gn)


@pytest.mark.parametrize("res", [(700, 40), (40, 700), (2, 2), (3, 1500)])
def test_extreme_bitmap_shapes(golden, res):
    """Very wide / very tall / minimal bitmaps: windows wider than half the LDS capacity, multi-pass row bands, and
    the 2 x 2 bitmap whose only cell quadruple is (0,0)-(1,1)."""
    from artist_amd import trace_rays
    d = golden("mid_256")
    inp = trace_inputs(d)
    inp["resolution"] = res
    inp["origins"].requires_grad_(True)
    flux, fac = trace_rays(**inp)
    dd = dict(d, resolution=np.asarray(res))
    o_flux, o_fac = oracle_fwd(dd)
    assert flux.shape == (d["aligned_points"].shape[0], res[1], res[0])
    assert rel_l2(n(flux), o_flux) < 2e-4 or float(np.abs(o_flux).max()) == 0, rel_l2(n(flux), o_flux)
    np.testing.assert_array_equal(n(fac), o_fac)
    w = torch.linspace(0.5, 1.5, res[0] * res[1], device=DEV).reshape(res[1], res[0])
    (flux * w).sum().backward()
    go, _ = oracle.trace_bwd(d["aligned_points"], d["aligned_normals"], d["incident"], d["distortions_u"], d["distortions_e"],
                             d["target_idx"], d["target_centers"], d["target_normals"], d["target_dims"], res,
                             np.broadcast_to(n(w), o_flux.shape).copy(), float(d["ray_magnitude"]), float(d["extinction"]),
                             float(d["reflectivity"]))
    if np.linalg.norm(go) > 0:
        assert rel_l2(n(inp["origins"].grad), go) < 5e-3, rel_l2(n(inp["origins"].grad), go)


# ---------------------------------------------------------------- rigid-body kinematics
def _rigid_body(c, requires_grad=False):
    from artist_amd.kinematics import RigidBody
    H = c["positions"].shape[0]
    kin = RigidBody(number_of_heliostats=H, heliostat_positions=t(c["positions"], torch.float32),
                    initial_orientations=torch.zeros((H, 4), device=DEV),
                    translation_deviation_parameters=t(c["trans_dev"], torch.float32).requires_grad_(requires_grad),
                    rotation_deviation_parameters=t(c["rot_dev"], torch.float32).requires_grad_(requires_grad),
                    actuator_parameters_non_optimizable=t(c["act_nonopt"], torch.float32),
                    actuator_parameters_optimizable=(t(c["act_opt"], torch.float32).requires_grad_(requires_grad)
                                                     if c["act_opt"].size else torch.tensor([])), device=DEV)
    kin.activate_all()
    return kin


@pytest.mark.parametrize("tag", KINEMATICS_CASES)
def test_rigid_body_orientations(golden, tag):
    """art_rigid_body_fwd on the reference's scenario files (linear actuators with fitted parameters / ideal
    actuators) against the reference's fp32 run, its fp64 run and the CPU restatement."""
    c = kinematics_case(golden("kinematics"), tag, "f32")
    c64 = kinematics_case(golden("kinematics"), tag, "f64")
    kin = _rigid_body(c)
    np.testing.assert_array_equal(n(kin.initial_orientation_offsets[0]), c["offsets"])   # host-side constant, bit-exact
    ori = n(kin.incident_ray_directions_to_orientations(t(c["incident"]), t(c["aim"])))
    motor = n(kin.active_motor_positions)
    o_ori, o_motor, _ = oracle.rigid_body_orientations(c["positions"], c["rot_dev"], c["trans_dev"], c["act_nonopt"],
                                                       c["act_opt"], c["offsets"], incident=c["incident"], aim=c["aim"])
    # fp32 yardstick: how far the reference's own fp32 run is from its fp64 run
    yard = np.abs(c["orientation"] - c64["orientation"]).max()
    assert np.abs(ori - c64["orientation"]).max() <= 4 * yard + 1e-6
    np.testing.assert_allclose(ori, c["orientation"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(ori, o_ori, rtol=0, atol=1e-4)
    np.testing.assert_allclose(motor, c["motor"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(motor, o_motor, rtol=1e-4, atol=1e-4)
    # calibration path: orientations from measured motor positions (kinematics_rigid_body.py:510-538)
    ori_m = n(kin.motor_positions_to_orientations(t(c["motor_given"], torch.float32)))
    np.testing.assert_allclose(ori_m, c["orientation_from_motor"], rtol=0, atol=1e-5)


@pytest.mark.parametrize("tag", KINEMATICS_CASES)
def test_rigid_body_gradients_equal_reference_jacobians(golden, tag):
    """art_rigid_body_bwd (forward-mode replay of the iteration) against torch.autograd.functional.jacobian of the
    reference: d(orientation[h,i,j]) / d(rotation deviations, translation deviations, optimisable actuator
    parameters), fp32 reference and fp64 reference."""
    c = kinematics_case(golden("kinematics"), tag, "f32")
    c64 = kinematics_case(golden("kinematics"), tag, "f64")
    kin = _rigid_body(c, requires_grad=True)
    H = c["positions"].shape[0]
    params = [("jac_rot", kin.active_rotation_deviation_parameters), ("jac_trans", kin.active_translation_deviation_parameters)]
    if c["act_opt"].size:
        params.append(("jac_opt", kin.actuators.active_optimizable_parameters))
    ori = kin.incident_ray_directions_to_orientations(t(c["incident"]), t(c["aim"]))
    jac = {k: np.zeros((H, 4, 4) + tuple(p.shape[1:]), np.float32) for k, p in params}
    for i in range(4):
        for j in range(4):
            seed = torch.zeros_like(ori)
            seed[:, i, j] = 1.0
            grads = torch.autograd.grad(ori, [p for _, p in params], grad_outputs=seed, retain_graph=True)
            for (k, _), g in zip(params, grads):
                jac[k][:, i, j] = n(g)
    for k, _ in params:
        ref32, ref64 = c[k], c64[k]
        scale = max(1.0, float(np.abs(ref64).max()))
        yard = np.abs(ref32 - ref64).max()
        assert np.abs(jac[k] - ref64).max() <= 4 * yard + 1e-5 * scale, k
        np.testing.assert_allclose(jac[k], ref32, rtol=0, atol=2e-3 * scale, err_msg=k)


def test_rigid_body_motor_path_gradients_vs_finite_differences(golden):
    """Calibration path: gradients of a scalar of the orientations from GIVEN motor positions against central
    differences of the fp64 restatement."""
    c = kinematics_case(golden("kinematics"), "blocking", "f32")
    c64 = kinematics_case(golden("kinematics"), "blocking", "f64")
    kin = _rigid_body(c, requires_grad=True)
    rng = np.random.default_rng(3)
    w = rng.standard_normal(c["orientation"].shape)
    motor = t(c["motor_given"], torch.float32).requires_grad_(True)      # what AimPointOptimizer learns
    ori = kin.motor_positions_to_orientations(motor)
    (ori * t(w, torch.float32)).sum().backward()

    def scalar(**over):
        a = dict(c64, **over)
        return float((oracle.rigid_body_orientations(a["positions"], a["rot_dev"], a["trans_dev"], a["act_nonopt"], a["act_opt"],
                                                     a["offsets"], motor_positions=a["motor_given"])[0] * w).sum())

    for key, param in (("rot_dev", kin.rotation_deviation_parameters), ("trans_dev", kin.translation_deviation_parameters),
                       ("act_opt", kin.actuators.optimizable_parameters), ("motor_given", motor)):
        g = n(param.grad).reshape(-1)
        base = c64[key]
        fd = np.zeros(base.size)
        for q in range(base.size):
            step = 1e-6 * max(1.0, abs(float(base.reshape(-1)[q])))
            plus, minus = base.copy().reshape(-1), base.copy().reshape(-1)
            plus[q] += step
            minus[q] -= step
            fd[q] = (scalar(**{key: plus.reshape(base.shape)}) - scalar(**{key: minus.reshape(base.shape)})) / (2 * step)
        assert rel_l2(g, fd) < 1e-3, (key, rel_l2(g, fd))


def test_rigid_body_field_scale_and_stopping_rule():
    """2000 heliostats (more than one pass of the single workgroup), ideal actuators: every heliostat's result equals
    the restatement's, including the FIELD-WIDE stopping rule - one slow heliostat keeps everybody iterating."""
    from artist_amd.kinematics import initial_orientation_offsets, rigid_body_orientations
    rng = np.random.default_rng(11)
    H = 2000
    pos = np.concatenate([rng.uniform(-100, 100, (H, 2)), rng.uniform(0, 3, (H, 1)), np.ones((H, 1))], 1).astype(np.float32)
    rot = rng.normal(0, 0.01, (H, 4)).astype(np.float32)
    trans = rng.normal(0, 0.05, (H, 9)).astype(np.float32)
    nonopt = np.zeros((H, 4, 2), np.float32)
    nonopt[:, 2], nonopt[:, 3] = -10.0, 10.0
    inc = np.tile(np.array([0.2, 0.9, -0.39, 0.0], np.float32) / np.linalg.norm([0.2, 0.9, -0.39]), (H, 1)).astype(np.float32)
    aim = np.tile(np.array([0.0, -5.0, 50.0, 1.0], np.float32), (H, 1))
    off = n(initial_orientation_offsets(DEV))[0]
    for max_iter, eps in ((4, 1e-4), (10, 1e-3), (10, 1e-30), (1, 1e-4)):
        ori, motor = rigid_body_orientations(1, t(pos), t(rot), t(trans), t(nonopt), None, t(off), t(inc), t(aim), None, max_iter, eps)
        o_ori, o_motor, evals = oracle.rigid_body_orientations(pos, rot, trans, nonopt, None, off, incident=inc, aim=aim,
                                                               max_iter=max_iter, min_eps=eps)
        np.testing.assert_allclose(n(ori), o_ori, rtol=0, atol=5e-5, err_msg=f"{max_iter} {eps}")
        np.testing.assert_allclose(n(motor), o_motor, rtol=0, atol=5e-5)
    assert rigid_body_orientations(1, t(pos[:0]), t(rot[:0]), t(trans[:0]), t(nonopt[:0]), None, t(off), t(inc[:0]), t(aim[:0]),
                                   None, 4, 1e-4)[0].shape == (0, 4, 4)


def _real_group(d):
    """The reference scenario's heliostat group as artist_amd objects: surfaces as sampled by ARTIST's loader,
    rigid-body kinematics with the scenario's deviation and actuator parameters as leaves."""
    from artist_amd import RigidBody, scene
    H = d["surface_points"].shape[0]
    leaf = lambda key: t(d[key], torch.float32).requires_grad_(True)
    kin = RigidBody(number_of_heliostats=H, heliostat_positions=t(d["kin_positions"], torch.float32),
                    initial_orientations=torch.zeros((H, 4), device=DEV), translation_deviation_parameters=leaf("kin_trans_dev"),
                    rotation_deviation_parameters=leaf("kin_rot_dev"),
                    actuator_parameters_non_optimizable=t(d["kin_act_nonopt"], torch.float32),
                    actuator_parameters_optimizable=leaf("kin_act_opt") if d["kin_act_opt"].size else torch.tensor([]), device=DEV)
    group = scene.HeliostatGroup(names=[f"h{i}" for i in range(H)], positions=t(d["kin_positions"], torch.float32),
                                 surface_points=t(d["surface_points"], torch.float32),
                                 surface_normals=t(d["surface_normals"], torch.float32), canting=torch.zeros((H, 4, 2, 4), device=DEV),
                                 facet_translations=torch.zeros((H, 4, 4), device=DEV),
                                 nurbs_control_points=t(d["control_points"])[:1].expand(H, -1, -1, -1, -1) if d["control_points"].shape[0] != H
                                 else t(d["control_points"]),
                                 nurbs_degrees=torch.tensor([3, 3]), device=DEV, kinematics=kin)
    mask = torch.ones(H, dtype=torch.int32, device=DEV)
    group.activate_heliostats(active_heliostats_mask=mask)
    group.align_surfaces_with_incident_ray_directions(aim_points=t(d["aim_points"], torch.float32),
                                                      incident_ray_directions=t(d["incident"], torch.float32),
                                                      active_heliostats_mask=mask)
    return group, kin


@pytest.mark.parametrize("name", REAL_CASES)
def test_real_scenarios_kinematics_chain(golden, name):
    """Reference scenario: sampled surfaces -> RigidBody (HIP) -> alignment (HIP) equals the reference's aligned
    surfaces, and with the REFERENCE's dL/d(aligned surfaces) as upstream gradient the kinematic-parameter gradients
    equal the reference's end-to-end autograd (flux -> surfaces -> orientations -> deviations / actuator parameters)."""
    d, d64 = golden(name), golden(name + "_f64")
    group, kin = _real_group(d)           # the fixture's ACTIVE rows, each as a heliostat of its own
    ap, an = group.active_surface_points, group.active_surface_normals
    # 2e-5 on the orientation entries (fp32 actuator geometry, see test_rigid_body_orientations) x 2 m lever + position rounding
    np.testing.assert_allclose(n(ap), d["aligned_points"], rtol=0, atol=3e-4)
    np.testing.assert_allclose(n(an), d["aligned_normals"], rtol=0, atol=1e-4)
    torch.autograd.backward([ap, an], [t(d["grad_aligned_points"]), t(d["grad_aligned_normals"])])
    checks = [(kin.rotation_deviation_parameters, "grad_kin_rot_dev"), (kin.translation_deviation_parameters, "grad_kin_trans_dev")]
    if d["kin_act_opt"].size:
        checks.append((kin.actuators.optimizable_parameters, "grad_kin_act_opt"))
    for param, key in checks:
        # the actuator-parameter gradients go through acos near its clamp: the reference's own fp32 run is 3e-3 from its
        # fp64 run there (the yardstick); the deviation gradients are well conditioned
        tol = max(2e-3, 3 * rel_l2(d[key], d64[key])) if key == "grad_kin_act_opt" else 2e-3
        assert rel_l2(n(param.grad), d[key]) < tol, (key, rel_l2(n(param.grad), d[key]), tol)


def test_real_scenario_end_to_end_kinematic_gradients(golden):
    """The whole calibration-style step on the GPU - kinematics -> alignment -> blocking rectangles -> trace -> weighted
    flux sum -> backward down to the kinematic parameters - against the reference's end-to-end run.  The bitmaps are
    only as close as fp32 kinematics allow: a 2e-5 rad orientation difference moves the image by ~4 mm at 100 m."""
    from artist_amd import trace_rays
    from artist_amd.blocking import create_blocking_primitives_rectangles_by_index
    d = golden("real_blocking")
    group, kin = _real_group(d)
    inp = trace_inputs(d)
    inp["origins"], inp["normals"] = group.active_surface_points, group.active_surface_normals
    corners, spans, normals = create_blocking_primitives_rectangles_by_index(inp["origins"])
    out = trace_rays(**inp, cyl=cyl_inputs(d), blocking=dict(corners=corners, spans=spans, normals=normals,
                                                             owner=t(d["owner"]).int()))
    (out[0] * t(d["loss_weights"])).sum().backward()
    assert rel_l2(n(out[0]), d["flux"]) < 3e-2, rel_l2(n(out[0]), d["flux"])
    np.testing.assert_allclose(n(out[1][0]), d["intercept"], rtol=0, atol=5e-3)
    for param, key in ((kin.rotation_deviation_parameters, "grad_kin_rot_dev"), (kin.translation_deviation_parameters, "grad_kin_trans_dev"),
                       (kin.actuators.optimizable_parameters, "grad_kin_act_opt")):
        assert rel_l2(n(param.grad), d[key]) < 5e-2, (key, rel_l2(n(param.grad), d[key]))


# ---------------------------------------------------------------------------------------------
# From the scenario FILE to the flux: the reference's own HDF5 scenarios (tests/golden/scenarios, copies of the data
# files of ARTIST's test suite) through artist_amd's loader, kinematics, alignment and ray tracer - the sequence of
# tests/field/test_integration_alignment.py and tutorials/01 - against the reference's run of the same sequence.
# ---------------------------------------------------------------------------------------------
SCENARIO_RUNS = {
    "real_blocking": dict(file="test_blocking.h5", points=[10, 10], rays=6, blocking=True, resolution=[64, 64],
                          mapping=[(f"heliostat_{i}", "target_0", [0.0, 1.0, 0.0, 0.0]) for i in range(6)]),
    "real_paint_mixed": dict(file="test_scenario_paint_four_heliostats.h5", points=[8, 8], rays=5, blocking=False, resolution=[96, 64],
                             mapping=[("AA28", "receiver", [0.3, 0.8, -0.52, 0.0]), ("AA31", "multi_focus_tower", [0.3, 0.8, -0.52, 0.0]),
                                      ("AA39", "receiver", [0.3, 0.8, -0.52, 0.0]), ("AC43", "solar_tower_juelich_upper", [0.3, 0.8, -0.52, 0.0])]),
    "real_stral_single": dict(file="test_scenario_stral_single_heliostat.h5", points=[12, 12], rays=6, blocking=False, resolution=[64, 64],
                              mapping=[("heliostat_1", "receiver", [0.0, 1.0, 0.0, 0.0]), ("heliostat_1", "receiver", [-1.0, 0.0, 0.0, 0.0]),
                                       ("heliostat_1", "receiver", [1.0, 0.0, 0.0, 0.0]), ("heliostat_1", "receiver", [0.0, 0.0, -1.0, 0.0])]),
}


@pytest.mark.parametrize("name", SCENARIO_RUNS)
def test_scenario_file_to_flux(golden, name):
    import pathlib

    from artist_amd import HeliostatRayTracer
    from artist_amd.scenario import Scenario, open_scenario_file
    run, d = SCENARIO_RUNS[name], golden(name)
    path = pathlib.Path(__file__).resolve().parent / "golden" / "scenarios" / run["file"]
    with open_scenario_file(path) as scenario_file:
        scenario = Scenario.load_scenario_from_hdf5(scenario_file=scenario_file,
                                                    number_of_surface_points_per_facet=torch.tensor(run["points"]), device=DEV)
    assert scenario.get_number_of_heliostat_groups_from_hdf5(path) == len(scenario.heliostat_field.heliostat_groups)
    group = scenario.heliostat_field.heliostat_groups[0]
    mapping = [(h, tgt, torch.nn.functional.normalize(torch.tensor(s), dim=-1)) for h, tgt, s in run["mapping"]]
    mask, target_idx, incident = scenario.index_mapping(heliostat_group=group, string_mapping=mapping, device=DEV)
    np.testing.assert_array_equal(n(mask), d["active_mask"])
    np.testing.assert_array_equal(n(target_idx), d["target_idx"])
    group.activate_heliostats(active_heliostats_mask=mask, device=DEV)
    # the loader's surfaces: one batched NURBS launch per group against the reference's per-heliostat evaluation
    np.testing.assert_allclose(n(group.active_surface_points), d["surface_points"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(n(group.active_surface_normals), d["surface_normals"], rtol=0, atol=2e-6)
    aim = scenario.solar_tower.get_centers_of_target_areas(target_area_indices=target_idx, device=DEV)
    np.testing.assert_array_equal(n(aim), d["aim_points"])
    group.align_surfaces_with_incident_ray_directions(aim_points=aim, incident_ray_directions=incident,
                                                      active_heliostats_mask=mask, device=DEV)
    # yardstick: the reference's own fp32 run against its fp64 run of the same file (fp32 `acos` in the actuator law moves the
    # mirror by up to 3e-5 m); the HIP chain has to be as close to the fp64 truth as the reference's fp32 chain is (x2)
    d64 = golden(name + "_f64")
    for got, key, floor in ((group.active_surface_points, "aligned_points", 1e-6), (group.active_surface_normals, "aligned_normals", 3e-7)):
        yard = float(np.abs(d[key] - d64[key]).max())
        assert float(np.abs(n(got) - d64[key]).max()) <= 2.0 * yard + floor, (key, float(np.abs(n(got) - d64[key]).max()), yard)
    scenario.set_number_of_rays(number_of_rays=run["rays"])
    tracer = HeliostatRayTracer(scenario=scenario, heliostat_group=group, blocking_active=run["blocking"], batch_size=100,
                                bitmap_resolution=torch.tensor(run["resolution"]))
    assert tuple(tracer.distortions_dataset.distortions_u.shape) == d["distortions_u"].shape
    # the sun sample is a seeded CPU draw in the reference: take the fixture's so that the bitmaps are comparable
    tracer.distortions_dataset.distortions_u, tracer.distortions_dataset.distortions_e = interleave(d["distortions_u"], d["distortions_e"])
    flux, intercept, on_target, unblocked = tracer.trace_rays(incident_ray_directions=incident, active_heliostats_mask=mask,
                                                              target_area_indices=target_idx, device=DEV)
    # bitmaps: within 2x the reference's fp32-vs-fp64 distance (4.5e-5 ... 5.8e-4 on these files) of the fp64 run; energy to 1e-5;
    # the ray-count factors within one ray of the reference's fp32 run
    yard = rel_l2(d["flux"].astype(np.float64), d64["flux"])
    assert rel_l2(n(flux), d64["flux"]) < 2.0 * yard, (rel_l2(n(flux), d64["flux"]), yard)
    np.testing.assert_allclose(n(flux).sum((1, 2)), d64["flux"].sum((1, 2)), rtol=1e-5, atol=1e-5 * float(d64["flux"].sum((1, 2)).max()))
    rays = d["distortions_u"][0].size
    for got, key in ((intercept, "intercept"), (on_target, "on_target"), (unblocked, "blocking")):
        np.testing.assert_allclose(n(got), d[key], rtol=0, atol=1.0 / rays + 1e-7)
    if run["blocking"]:
        np.testing.assert_array_equal(n(tracer.filtered_blocking_primitive_indices), d["filter_indices"])


@pytest.mark.parametrize("knobs", [dict(ARTIST_HIP_PERSISTENT="0"), dict(ARTIST_HIP_LEAN="0"), dict(ARTIST_HIP_FWD_PBLOCK="512", ARTIST_HIP_BWD_PBLOCK="640"),
                                   dict(ARTIST_HIP_FWD_BLOCKS="4096"), dict(ARTIST_HIP_BWD_PACK="0"), dict(ARTIST_HIP_BWD_PACK="200"),
                                   dict(ARTIST_HIP_FWD_TILE_KB="24", ARTIST_HIP_BWD_PACK="64"), dict(ARTIST_HIP_TAIL="2"),
                                   dict(ARTIST_HIP_TAIL="2", ARTIST_HIP_FWD_PBLOCK="256", ARTIST_HIP_BWD_PBLOCK="320"),
                                   dict(ARTIST_HIP_WINDOW_SAMPLE="0"), dict(ARTIST_HIP_BWD_THREADS="1024"), dict(ARTIST_HIP_BWD_THREADS="768"),
                                   dict(ARTIST_HIP_BWD_THREADS="1024", ARTIST_HIP_BWD_PBLOCK="640"),
                                   dict(ARTIST_HIP_BWD_THREADS="1024", ARTIST_HIP_BWD_PACK="200"),
                                   dict(ARTIST_HIP_BWD_THREADS="1024", ARTIST_HIP_TAIL="2", ARTIST_HIP_BWD_PBLOCK="320"),
                                   dict(ARTIST_HIP_WINDOW_TABLE="0"), dict(ARTIST_HIP_WINDOW_TABLE="2", ARTIST_HIP_TAIL="2"),
                                   dict(ARTIST_HIP_WINDOW_TABLE="0", ARTIST_HIP_BWD_REVERSE="1")])
def test_work_queue_variants_give_the_same_results(golden, monkeypatch, knobs):
    """The windowed kernels hand out (heliostat, point block, sample chunk) items through a work queue: persistent
    workgroups (default) or one workgroup per item, the lean or the generic ray body, other point-block sizes, samples cut
    into more chunks, the backward kernel's edge points packed or not (and with a margin that makes every point an edge point,
    and with a small window), the lean backward kernel launched with 768 or 1024 threads per workgroup, the queue's end cut into finer point blocks (ARTIST_HIP_TAIL=2: in both kernels, any field
    size), the window phase on every point instead of a sample, the items' windows from a table made ahead of the launch (the default
    for a field this small) or by the items themselves.  However the items are dealt, the bitmaps are the same BITS
    (integer pixel accumulators); the gradients are the same bits as long as a point's samples are summed in the same order
    (everything but the chunking)."""
    from artist_amd import trace_rays
    d = golden("mid_256")

    def run():
        inp = trace_inputs(d)
        inp["origins"].requires_grad_(True)
        inp["normals"].requires_grad_(True)
        flux, fac = trace_rays(**inp)[:2]
        (flux * t(d["loss_weights"])).sum().backward()
        return n(flux), n(fac), n(inp["origins"].grad), n(inp["normals"].grad)

    base = run()
    again = run()
    for x, y in zip(base, again):                     # same launch twice: the same bits, bitmaps and gradients alike
        np.testing.assert_array_equal(x, y)
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    other = run()
    if "ARTIST_HIP_LEAN" in knobs:                    # the generic body rounds the four weights in another order before it quantises them
        assert rel_l2(other[0], base[0]) < 1e-6, rel_l2(other[0], base[0])
    else:
        np.testing.assert_array_equal(other[0], base[0])
    np.testing.assert_array_equal(other[1], base[1])
    if "ARTIST_HIP_FWD_BLOCKS" in knobs or "ARTIST_HIP_LEAN" in knobs or "ARTIST_HIP_FWD_TILE_KB" in knobs:
        # another summation order (sample chunks; the row bands a small window is swept in) / another gradient arithmetic
        assert rel_l2(other[2], base[2]) < 2e-6 and rel_l2(other[3], base[3]) < 2e-6, (rel_l2(other[2], base[2]), rel_l2(other[3], base[3]))
    else:
        np.testing.assert_array_equal(other[2], base[2])
        np.testing.assert_array_equal(other[3], base[3])
    again = run()                                     # and every variant repeats itself
    for x, y in zip(other, again):
        np.testing.assert_array_equal(x, y)


def test_scale_free_square_root_is_the_ieee_square_root():
    """The cylinder hit takes its two square roots (geometry.py:336, :384) with ray_math.hpp's sqrt_noscale - the compiler's
    correctly rounded sequence without the scaling of arguments below 2^-96 and the fix-up of 0 / inf / NaN (15 -> 9 instructions).
    EVERY float from 2^-96 up to the largest finite one - 1.9e9 values - and zero give the bits of sqrtf (tests/sqrt_check.hip)."""
    import ctypes
    so = ROOT / "tests" / "bin" / "libsqrt_check.so"
    assert so.exists(), "tests/bin/libsqrt_check.so is built by __graft_entry__.build()"
    lib = ctypes.CDLL(str(so))
    lib.sqrt_check.argtypes = [ctypes.c_uint, ctypes.c_ulonglong, ctypes.c_void_p, ctypes.c_void_p]
    bad = torch.zeros(1, dtype=torch.int64, device=DEV)
    lo = 0x0F800000                      # 2^-96
    hi = 0x7F800000                      # +inf (excluded)
    stream = torch.cuda.current_stream().cuda_stream
    assert lib.sqrt_check(lo, hi - lo, bad.data_ptr(), stream) == 0
    assert lib.sqrt_check(0, 1, bad.data_ptr(), stream) == 0          # 0.0
    assert int(bad.item()) == 0, f"{int(bad.item())} arguments differ"


@pytest.mark.parametrize("name", CYL_CASES)
@pytest.mark.parametrize("knobs", [dict(ARTIST_HIP_CYL_LEAN="0"), dict(ARTIST_HIP_CYL_LEAN="0", ARTIST_HIP_FWD_BLOCKS="1"), dict(ARTIST_HIP_FWD_BLOCKS="1"),
                                   dict(ARTIST_HIP_FWD_PBLOCK="200"), dict(ARTIST_HIP_PERSISTENT="0")])
def test_cylinder_bodies_agree(golden, monkeypatch, name, knobs):
    """Cylindrical receivers take the lean forward AND backward items with the cylinder hit in place of the plane's (round 3;
    ARTIST_HIP_CYL_LEAN=0: the generic items of round 2), on cylinder-only and mixed towers, with the samples in chunks or in one
    item, other point blocks, one workgroup per item.  Same rays and the same hit: ray counters equal, bitmaps within the two
    bodies' rounding of the four weights (1e-6) and the same bits where only the items change, gradients within 2e-5, every
    variant bit-reproducible."""
    from artist_amd import trace_rays
    d = golden(name)

    def run():
        inp = trace_inputs(d)
        inp["cyl"] = cyl_inputs(d)
        inp["origins"].requires_grad_(True)
        inp["normals"].requires_grad_(True)
        flux, fac = trace_rays(**inp)[:2]
        (flux * t(d["loss_weights"])).sum().backward()
        return n(flux), n(fac), n(inp["origins"].grad), n(inp["normals"].grad)

    base = run()
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    other = run()
    if "ARTIST_HIP_CYL_LEAN" in knobs:
        assert rel_l2(other[0], base[0]) < 1e-6, rel_l2(other[0], base[0])
    else:
        np.testing.assert_array_equal(other[0], base[0])
    np.testing.assert_array_equal(other[1], base[1])
    # the backward kernels: the lean item (round 3) against the generic one, and other geometries of either - the same adjoint
    # arithmetic, sums in other orders
    assert np.isfinite(other[2]).all() and np.isfinite(other[3]).all()
    assert rel_l2(other[2], base[2]) < 2e-5 and rel_l2(other[3], base[3]) < 2e-5, (rel_l2(other[2], base[2]), rel_l2(other[3], base[3]))
    again = run()
    for x, y in zip(again, other):
        np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("knobs", [dict(ARTIST_HIP_BLOCK_LEAN="0"), dict(ARTIST_HIP_BLOCK_FACETS="0"), dict(ARTIST_HIP_BLOCKING_SPLIT="0"),
                                   dict(ARTIST_HIP_BLOCK_LEAN="0", ARTIST_HIP_BLOCKING_SPLIT="0"), dict(ARTIST_HIP_FWD_BLOCKS="1"),
                                   dict(ARTIST_HIP_FWD_BLOCKS="1", ARTIST_HIP_BLOCK_LEAN="0")])
def test_blocking_bodies_agree(golden, monkeypatch, knobs):
    """Blocking on planes runs the heliostats WITH candidate rectangles through the lean ray body with the soft mask (round 3;
    ARTIST_HIP_BLOCK_LEAN=0: the generic item of round 2), in facet-sized items or not, next to a lean launch for the
    heliostats without candidates or in one launch, with the samples in chunks or in one item (ARTIST_HIP_FWD_BLOCKS=1:
    the launches of a split call).  Same rays, same mask: bitmaps within the two bodies' rounding of the four weights
    (1e-6; the same bits where only the items change), ray counters equal, gradients - the rectangles' included - within 2e-6."""
    from artist_amd import trace_rays
    d = golden("mid_blocking")

    def run():
        inp = trace_inputs(d)
        blk = blocking_inputs(d)
        leaves = [inp["origins"].requires_grad_(True), inp["normals"].requires_grad_(True)]
        blk = {k: (v.clone().requires_grad_(True) if k in ("corners", "spans", "normals") else v) for k, v in blk.items()}
        leaves += [blk["corners"], blk["spans"], blk["normals"]]
        flux, fac = trace_rays(**inp, blocking=blk)[:2]
        grads = torch.autograd.grad(flux, leaves, t(d["loss_weights"]))
        return [n(flux), n(fac)] + [n(g) for g in grads]

    base = run()
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    other = run()
    same_body = "ARTIST_HIP_BLOCK_LEAN" not in knobs
    if same_body:
        np.testing.assert_array_equal(other[0], base[0])
    else:
        assert rel_l2(other[0], base[0]) < 1e-6, rel_l2(other[0], base[0])
    np.testing.assert_array_equal(other[1], base[1])
    for got, ref in zip(other[2:], base[2:]):
        assert np.isfinite(got).all()
        assert rel_l2(got, ref) < 2e-6, rel_l2(got, ref)
    again = run()                                     # every variant repeats itself bit for bit
    for x, y in zip(other, again):
        np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("name", ["small_deg3", "mid_256", "mid_cyl", "mid_blocking"])
def test_flux_is_bit_reproducible(golden, name):
    """SURVEY.md section 5 (race surface): the reference needs torch.use_deterministic_algorithms(True) for a stable
    scatter_add_ (tests/conftest.py:109); here every launch of the forward gives the same bits - window flushes, cell
    carries and stray rays all go through 64-bit integer pixel accumulators."""
    from artist_amd import trace_rays
    d = golden(name)
    inp = trace_inputs(d)
    if name in CYL_CASES:
        inp["cyl"] = cyl_inputs(d)
    if name in BLOCKING_CASES:
        inp["blocking"] = blocking_inputs(d)
    runs = [n(trace_rays(**inp)[0]) for _ in range(3)]
    np.testing.assert_array_equal(runs[0], runs[1])
    np.testing.assert_array_equal(runs[0], runs[2])
    # ... and so do the gradients - with blocking on also those of the rectangle tables, which round 2 still added up with
    # float atomics (wave reductions on the DPP network, wave order, item slabs added in item order: DESIGN.md 4.2b)
    def grads():
        leaves = [inp["origins"].clone().requires_grad_(True), inp["normals"].clone().requires_grad_(True)]
        kw = dict(inp, origins=leaves[0], normals=leaves[1])
        if name in BLOCKING_CASES:
            blk = {k: (v.clone().requires_grad_(True) if k in ("corners", "spans", "normals") else v) for k, v in inp["blocking"].items()}
            kw["blocking"] = blk
            leaves += [blk["corners"], blk["spans"], blk["normals"]]
        flux = trace_rays(**kw)[0]
        return [n(g) for g in torch.autograd.grad(flux, leaves, t(d["loss_weights"]))]
    g0, g1, g2 = grads(), grads(), grads()
    for a_, b_, c_ in zip(g0, g1, g2):
        np.testing.assert_array_equal(a_, b_)
        np.testing.assert_array_equal(a_, c_)
    if name in BLOCKING_CASES:
        assert np.abs(g0[2]).sum() > 0 and np.abs(g0[4]).sum() > 0           # the rectangles do receive gradients


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("name", STAGE_CASES)
def test_control_point_gradient_is_bit_reproducible(golden, name, fused):
    """... and on through nurbs_bwd_kernel to the tensor a surface reconstruction trains: control points -> NURBS (tensor-product
    scheme: every gradient element has one owner that adds in index order) -> alignment (separate kernel or fused into the
    evaluation) -> trace -> loss gives the same control-point gradient bits on every run (round 3 summed the control-point
    gradients with LDS atomics, in whatever order the lanes arrived)."""
    from artist_amd import NURBSSurfaces, align_surfaces, trace_rays
    d = golden(name)
    ori = t(d["orientation"])
    H = ori.shape[0]
    args = (t(d["eval_points"]), t(d["canting"]), t(d["facet_translations"]))

    def grad():
        cp = t(d["control_points"]).requires_grad_(True)
        surf = NURBSSurfaces(torch.from_numpy(d["degrees"]), cp, device=DEV)
        if fused:
            ap, an = surf.calculate_surface_points_and_normals(*args, orientations=ori)
            ap, an = ap.reshape(H, -1, 4), an.reshape(H, -1, 4)
        else:
            pts, nrm = surf(*args)
            ap, an = align_surfaces(pts.reshape(H, -1, 4), nrm.reshape(H, -1, 4), ori)
        inp = trace_inputs(d)
        inp.update(origins=ap, normals=an)
        flux, _ = trace_rays(**inp)
        (flux * t(d["loss_weights"])).sum().backward()
        return n(cp.grad)

    g0, g1, g2 = grad(), grad(), grad()
    np.testing.assert_array_equal(g0, g1)
    np.testing.assert_array_equal(g0, g2)
    assert np.isfinite(g0).all() and (name == "small_offtarget" or np.abs(g0).sum() > 0)


def test_kinematics_reconstruction_loop_converges():
    """Acceptance run in the shape of tutorials/04 (kinematics reconstruction through ray tracing), entirely on
    artist_amd: scenario file -> measured flux with the true parameters -> perturbed deviation and actuator parameters
    -> with the measured motor positions (the calibration path) Adam on the kinematics' learnable tensors through
    kinematics, alignment, trace and pixel loss.  The loss must
    fall five-fold and the focal spots move back towards the measured ones, and a second run must retrace the first.  (The parameters themselves
    are not identifiable from one sun position - a joint tilt and an actuator's initial angle move the spot alike.)"""
    import pathlib

    from artist_amd import HeliostatRayTracer, PixelLoss
    from artist_amd.scenario import Scenario, open_scenario_file
    path = pathlib.Path(__file__).resolve().parent / "golden" / "scenarios" / "test_blocking.h5"
    with open_scenario_file(path) as scenario_file:
        scenario = Scenario.load_scenario_from_hdf5(scenario_file=scenario_file,
                                                    number_of_surface_points_per_facet=torch.tensor([12, 12]), device=DEV)
    group = scenario.heliostat_field.heliostat_groups[0]
    kin = group.kinematics
    sun = torch.nn.functional.normalize(torch.tensor([0.1, 1.0, -0.2, 0.0]), dim=0)
    mapping = [(name, "target_3", sun) for name in group.names if name != "heliostat_3"]     # heliostat_3 stands behind the target
    mask, targets, incident = scenario.index_mapping(heliostat_group=group, string_mapping=mapping, device=DEV)
    scenario.set_number_of_rays(number_of_rays=20)

    # calibration data: the motor positions the TRUE kinematics drives to for this sun and aim point
    group.activate_heliostats(active_heliostats_mask=mask, device=DEV)
    group.align_surfaces_with_incident_ray_directions(
        aim_points=scenario.solar_tower.get_centers_of_target_areas(target_area_indices=targets, device=DEV),
        incident_ray_directions=incident, active_heliostats_mask=mask, device=DEV)
    motors = kin.active_motor_positions.detach().clone()

    def flux_now():
        group.activate_heliostats(active_heliostats_mask=mask, device=DEV)
        group.align_surfaces_with_motor_positions(motor_positions=motors, active_heliostats_mask=mask, device=DEV)
        tracer = HeliostatRayTracer(scenario=scenario, heliostat_group=group, blocking_active=False,
                                    bitmap_resolution=torch.tensor([128, 128]))
        return tracer.trace_rays(incident_ray_directions=incident, active_heliostats_mask=mask,
                                 target_area_indices=targets, device=DEV)[0]

    def spots(flux):                      # centre of mass of every bitmap, pixels
        ys, xs = torch.meshgrid(torch.arange(128.0, device=DEV), torch.arange(128.0, device=DEV), indexing="ij")
        total = flux.sum((1, 2))
        return torch.stack(((flux * xs).sum((1, 2)) / total, (flux * ys).sum((1, 2)) / total), dim=1)

    with torch.no_grad():
        measured = flux_now().clone()
    true_rotation = kin.rotation_deviation_parameters.detach().clone()
    true_actuators = kin.actuators.optimizable_parameters.detach().clone()

    def optimise():
        g = torch.Generator(device="cpu").manual_seed(5)
        kin.rotation_deviation_parameters = (true_rotation + 3e-3 * torch.randn((6, 4), generator=g).to(DEV)).requires_grad_(True)
        kin.actuators.optimizable_parameters = true_actuators.clone().requires_grad_(True)
        optimizer = torch.optim.Adam([kin.rotation_deviation_parameters, kin.actuators.optimizable_parameters], lr=5e-4)
        loss_fn = PixelLoss()
        history, spot_error = [], []
        for _ in range(250):
            optimizer.zero_grad()
            flux = flux_now()
            loss = loss_fn(flux, measured, reduction_dimensions=(1, 2)).sum()
            loss.backward()
            optimizer.step()
            history.append(float(loss))
            spot_error.append(float((spots(flux.detach()) - spots(measured)).norm(dim=1).mean()))
        return history, spot_error

    history, spot_error = optimise()
    # Every kernel on this path sums in a fixed order (integer pixel accumulators, chunk slabs, ordered reductions): a
    # second run of the 250 steps retraces the first one exactly.  (Round 1's float atomics let the final loss wander
    # between 0.26 and 0.61 of the initial 5.36 from run to run.)
    history2, spot_error2 = optimise()
    assert history == history2 and spot_error == spot_error2
    assert min(history[-5:]) < 0.2 * history[0], (history[0], history[-5:])
    assert spot_error[0] > 3.0 and min(spot_error[-5:]) < 0.3 * spot_error[0], (spot_error[0], spot_error[-5:])


def test_surface_reconstruction_loop_converges():
    """Acceptance run in the shape of tutorials/03 / SurfaceReconstructor's epoch (surface_reconstructor.py:452-779): the
    measured flux comes from surfaces with deflections, the model starts from flat control nets, and Adam on the
    control points through NURBS, alignment, trace, crop and pixel loss brings the flux back."""
    from artist_amd import NURBSSurfaces, PixelLoss, ops, scene
    from artist_amd.flux import FluxCrop
    H, R, n_eval = 4, 20, 24
    P = 4 * n_eval * n_eval
    scenario, uv = scene.build_synthetic_scenario(H, n_rays=R, n_cp=(6, 6), n_eval=n_eval, z_noise=4e-4, device=DEV)
    group = scenario.heliostat_field.heliostat_groups[0]
    group.activate_heliostats(torch.ones(H, dtype=torch.int32, device=DEV))
    targets = torch.zeros(H, dtype=torch.long, device=DEV)
    incident = torch.tensor([[0.0, 1.0, 0.0, 0.0]], device=DEV).repeat(H, 1)
    planar = scenario.solar_tower.target_areas[0]
    orientation = scene.ideal_orientations(group.active_positions, scenario.solar_tower.get_centers_of_target_areas(targets), incident)
    dist_u, dist_e = scenario.light_sources.light_source_list[0].get_distortions(P, H)
    dims = planar.dimensions.index_select(0, targets).contiguous()

    def flux_of(cp):
        pts, nrm = NURBSSurfaces(group.nurbs_degrees, cp, device=DEV).calculate_surface_points_and_normals(
            uv, group.active_canting, group.active_facet_translations)
        ap, an = ops.align_surfaces(pts.reshape(H, P, 4), nrm.reshape(H, P, 4), orientation)
        flux = ops.trace_rays(ap, an, incident, dist_u, dist_e, targets, planar.centers, planar.normals, planar.dimensions,
                              1.0, 0.0, 0.935, (128, 128))[0]
        return FluxCrop.apply(flux, dims, 6.0, 6.0)

    true_cp = group.active_nurbs_control_points
    with torch.no_grad():
        measured = flux_of(true_cp).clone()
    loss_fn = PixelLoss()

    def reconstruct(make_optimizer, epochs=150):
        cp = true_cp.clone()
        cp[..., 2] = 0.0                                    # the model starts from ideal (flat) facets
        cp.requires_grad_(True)
        optimizer = make_optimizer(cp)
        history = []
        for _ in range(epochs):
            optimizer.zero_grad()
            loss = loss_fn(flux_of(cp), measured, reduction_dimensions=(1, 2)).sum()
            loss.backward()
            optimizer.step()
            history.append(float(loss))
        return history, n(cp)

    history, cp_end = reconstruct(lambda cp: torch.optim.Adam([cp], lr=2e-5))
    assert min(history[-5:]) < 0.3 * history[0], (history[0], history[-5:])
    # the whole loop retraces itself: same loss history, same control points, bit for bit (flux through integer accumulators,
    # trace gradients through plain stores, control-point gradients through ordered sums, crop + loss through fixed trees)
    history2, cp_end2 = reconstruct(lambda cp: torch.optim.Adam([cp], lr=2e-5))
    assert history == history2
    np.testing.assert_array_equal(cp_end, cp_end2)
    # ... and with the optimiser step on the HIP kernel (artist_amd.optim.Adam): the same trajectory up to the rounding of the
    # update (tests/test_gpu_optim.py), and as reproducible
    from artist_amd.optim import Adam
    history3, cp_end3 = reconstruct(lambda cp: Adam([cp], lr=2e-5), epochs=60)
    history4, cp_end4 = reconstruct(lambda cp: Adam([cp], lr=2e-5), epochs=60)
    assert history3 == history4
    np.testing.assert_array_equal(cp_end3, cp_end4)
    np.testing.assert_allclose(history3[:20], history[:20], rtol=1e-3)


def test_aim_point_optimisation_loop_converges():
    """Acceptance run in the shape of AimPointOptimizer's epoch (aim_point_optimizer.py:352-470): the learnable tensor is
    the motor positions, the path is align_surfaces_with_motor_positions -> trace with blocking -> per-target flux.
    Wanted: the field's flux when every heliostat aims 0.8 m above the target centre; start: aimed at the centre."""
    import pathlib

    from artist_amd import HeliostatRayTracer, PixelLoss
    from artist_amd.scenario import Scenario, open_scenario_file
    path = pathlib.Path(__file__).resolve().parent / "golden" / "scenarios" / "test_blocking.h5"
    with open_scenario_file(path) as scenario_file:
        scenario = Scenario.load_scenario_from_hdf5(scenario_file=scenario_file,
                                                    number_of_surface_points_per_facet=torch.tensor([12, 12]), device=DEV)
    group = scenario.heliostat_field.heliostat_groups[0]
    kin = group.kinematics
    sun = torch.tensor([0.0, 1.0, 0.0, 0.0])
    mapping = [(name, "target_3", sun) for name in group.names if name != "heliostat_3"]
    mask, targets, incident = scenario.index_mapping(heliostat_group=group, string_mapping=mapping, device=DEV)
    scenario.set_number_of_rays(number_of_rays=20)
    centre = scenario.solar_tower.get_centers_of_target_areas(target_area_indices=targets, device=DEV)

    def motors_for(aim):
        group.activate_heliostats(active_heliostats_mask=mask, device=DEV)
        group.align_surfaces_with_incident_ray_directions(aim_points=aim, incident_ray_directions=incident,
                                                          active_heliostats_mask=mask, device=DEV)
        return kin.active_motor_positions.detach().clone()

    def field_flux(motors):
        group.activate_heliostats(active_heliostats_mask=mask, device=DEV)
        group.align_surfaces_with_motor_positions(motor_positions=motors, active_heliostats_mask=mask, device=DEV)
        tracer = HeliostatRayTracer(scenario=scenario, heliostat_group=group, blocking_active=True,
                                    bitmap_resolution=torch.tensor([128, 128]))
        flux = tracer.trace_rays(incident_ray_directions=incident, active_heliostats_mask=mask,
                                 target_area_indices=targets, device=DEV)[0]
        return tracer.get_bitmaps_per_target(flux, targets, device=DEV)[3:4]          # the field's flux on target_3

    with torch.no_grad():
        wanted = field_flux(motors_for(centre + torch.tensor([0.0, 0.0, 0.8, 0.0], device=DEV))).clone()
    start = motors_for(centre)
    offset = torch.zeros_like(start, requires_grad=True)          # motor steps (1e4-scale values): learn an offset
    optimizer = torch.optim.Adam([offset], lr=40.0)
    loss_fn = PixelLoss()
    history = []
    for _ in range(120):
        optimizer.zero_grad()
        loss = loss_fn(field_flux(start + offset), wanted, reduction_dimensions=(1, 2)).sum()
        loss.backward()
        optimizer.step()
        history.append(float(loss))
    assert min(history[-5:]) < 0.2 * history[0], (history[0], history[-5:])


def test_from_hdf5_classmethods_equal_the_scenario_loader():
    """SolarTower / target areas / LightSourceArray / Sun / HeliostatField.from_hdf5 (the reference's per-class loaders)
    give the objects Scenario.load_scenario_from_hdf5 assembles."""
    import pathlib

    from artist_amd import scene
    from artist_amd.scenario import Scenario, open_scenario_file
    path = pathlib.Path(__file__).resolve().parent / "golden" / "scenarios" / "test_scenario_paint_four_heliostats.h5"
    points = torch.tensor([6, 6])
    with open_scenario_file(path) as f:
        whole = Scenario.load_scenario_from_hdf5(scenario_file=f, number_of_surface_points_per_facet=points, device=DEV)
        tower = scene.SolarTower.from_hdf5(config_file=f, device=DEV)
        lights = scene.LightSourceArray.from_hdf5(config_file=f, device=DEV)
        sun = scene.Sun.from_hdf5(config_file=f["lightsources"]["sun_1"], light_source_name="sun_1", device=DEV)
        field = scene.HeliostatField.from_hdf5(config_file=f, number_of_surface_points_per_facet=points, device=DEV)
    assert tower.target_name_to_index == whole.solar_tower.target_name_to_index
    assert torch.equal(tower.target_areas[0].centers, whole.solar_tower.target_areas[0].centers)
    assert torch.equal(tower.target_areas[1].radii, whole.solar_tower.target_areas[1].radii)
    assert len(lights.light_source_list) == 1 and lights.light_source_list[0].number_of_rays == sun.number_of_rays == 10
    assert sun.distribution_parameters == whole.light_sources.light_source_list[0].distribution_parameters
    assert [g.names for g in field.heliostat_groups] == [["AA28", "AC43"], ["AA31", "AA39"]]      # ideal, then linear actuators
    for ours, ref in zip(field.heliostat_groups, whole.heliostat_field.heliostat_groups):
        assert torch.equal(ours.surface_points, ref.surface_points) and torch.equal(ours.surface_normals, ref.surface_normals)
        assert torch.equal(ours.kinematics.actuators.non_optimizable_parameters, ref.kinematics.actuators.non_optimizable_parameters)


def _sharded_worker(rank, world, port, out_dir):
    """One rank of the two-process run below: owns heliostats i = rank (mod world), traces them with the HIP kernels,
    reduces the per-target flux and gathers the per-point gradients' norms."""
    import os

    import torch.distributed as dist

    from artist_amd import ops
    from artist_amd.distributed import all_reduce_sum_async, gather_owned_rows, owned_heliostats
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d = dict(np.load(__import__("conftest").GOLDEN / "mid_256.npz"))
        H = 8
        rep = lambda x: np.concatenate([x] * (H // x.shape[0]))        # an 8-heliostat field from the 2-heliostat fixture
        own = owned_heliostats(H, world, rank)
        sel = lambda x: t(rep(x)[own])
        inp = dict(origins=sel(d["aligned_points"]).requires_grad_(True), normals=sel(d["aligned_normals"]), incident=sel(d["incident"]),
                   target_idx=sel(d["target_idx"]), centers=t(d["target_centers"]), plane_normals=t(d["target_normals"]),
                   dims=t(d["target_dims"]), ray_magnitude=float(d["ray_magnitude"]), extinction=float(d["extinction"]),
                   reflectivity=float(d["reflectivity"]), resolution=tuple(int(v) for v in d["resolution"]))
        inp["dist_u"], inp["dist_e"] = interleave(rep(d["distortions_u"])[own], rep(d["distortions_e"])[own])
        flux = ops.trace_rays(**inp)[0]
        T = d["target_centers"].shape[0]
        per_target = ops.per_target_sum(flux.detach(), inp["target_idx"], T)
        pending = all_reduce_sum_async(per_target)
        flux.sum().backward()
        norms = gather_owned_rows(inp["origins"].grad.flatten(1).norm(dim=1, keepdim=True), H)
        pending.wait()
        if rank == 0:
            np.savez(out_dir / "sharded.npz", per_target=n(per_target), norms=n(norms))
    finally:
        dist.destroy_process_group()


def test_two_processes_share_the_field(tmp_path):
    """Two real processes (gloo rendezvous, both on this GPU): sharded trace + flux all-reduce + gradient gather equal
    the single-process result."""
    import socket

    import torch.multiprocessing as mp

    from artist_amd import ops
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    mp.spawn(_sharded_worker, args=(2, port, tmp_path), nprocs=2, join=True)
    got = np.load(tmp_path / "sharded.npz")
    d = dict(np.load(__import__("conftest").GOLDEN / "mid_256.npz"))
    rep = lambda x: np.concatenate([x] * 4)
    inp = dict(origins=t(rep(d["aligned_points"])).requires_grad_(True), normals=t(rep(d["aligned_normals"])), incident=t(rep(d["incident"])),
               target_idx=t(rep(d["target_idx"])), centers=t(d["target_centers"]), plane_normals=t(d["target_normals"]),
               dims=t(d["target_dims"]), ray_magnitude=float(d["ray_magnitude"]), extinction=float(d["extinction"]),
               reflectivity=float(d["reflectivity"]), resolution=tuple(int(v) for v in d["resolution"]))
    inp["dist_u"], inp["dist_e"] = interleave(rep(d["distortions_u"]), rep(d["distortions_e"]))
    flux = ops.trace_rays(**inp)[0]
    flux.sum().backward()
    want = n(ops.per_target_sum(flux.detach(), inp["target_idx"], d["target_centers"].shape[0]))
    assert rel_l2(got["per_target"], want) < 1e-6
    np.testing.assert_allclose(got["norms"], n(inp["origins"].grad.flatten(1).norm(dim=1, keepdim=True)), rtol=1e-5)
