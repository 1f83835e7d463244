"""BASELINE.json's workloads at FULL size on the GPU (``-m gpu``): configs 3, 4, 5 and the metric config
(config 4 = examples/field_optimizations/config.yaml:25-28).  Each test
  (a) compares a seeded sample of heliostats - near, middle and far end of the field - with the oracle on the same inputs
      (flux relative L2 < 1e-5 = the north-star bound, ray counters exact; config 4 and the metric config also the
      gradients, against the oracle within the fp32-vs-fp64 yardstick the oracle itself supplies), and
  (b) checks the size-independent invariants on the WHOLE field: fused per-target mode == segment sum of the
      per-heliostat bitmaps, and sum over three ranks' shards == the single-rank result.
Everything goes through the C ABI (artist_amd.ops -> ctypes -> libartist_hip.so).  Distortions are generated on the
device (4 GB for config 3, 8 GB for the metric config).
"""
import numpy as np
import pytest
import torch

import oracle
from conftest import rel_l2

pytestmark = pytest.mark.gpu

DEV = torch.device("cuda:0")


def n(x):
    return x.detach().cpu().numpy()


def spread(H, k=4):
    """k heliostats from the near end to the far end of the list (the synthetic fan runs near to far)."""
    return sorted({int(round(i * (H - 1) / (k - 1))) for i in range(k)}) if H >= k else list(range(H))


def build_field(H, R, n_cp=10, n_eval=50):
    from artist_amd import scene
    scenario, uv = scene.build_synthetic_scenario(H, n_rays=R, n_cp=(n_cp, n_cp), n_eval=n_eval, device=DEV)
    group = scenario.heliostat_field.heliostat_groups[0]
    mask = torch.ones(H, dtype=torch.int32, device=DEV)
    group.activate_heliostats(mask)
    tix = torch.zeros(H, dtype=torch.long, device=DEV)
    inc = torch.tensor([[0.0, 1.0, 0.0, 0.0]], device=DEV).repeat(H, 1)
    aim = scenario.solar_tower.get_centers_of_target_areas(tix)
    group.align_surfaces_with_incident_ray_directions(aim, inc, mask)
    return scenario, group, mask, tix, inc, uv


def oracle_rows(group, rt, inc, tix, planar, rows, dtype=np.float32, grad_weights=None):
    """Oracle forward (and backward) of the heliostat samples ``rows`` on the tracer's own inputs."""
    sel = torch.tensor(rows, device=DEV)
    cast = lambda a: np.ascontiguousarray(n(a)).astype(dtype)  # noqa: E731
    args = (cast(group.active_surface_points[sel]), cast(group.active_surface_normals[sel]), cast(inc[sel]),
            cast(rt.distortions_dataset.distortions_u[sel]), cast(rt.distortions_dataset.distortions_e[sel]),
            n(tix[sel]).astype(np.int32), cast(planar.centers), cast(planar.normals), cast(planar.dimensions), (256, 256))
    flux, fac = oracle.trace_fwd(*args)
    if grad_weights is None:
        return flux, fac
    go, gn = oracle.trace_bwd(*args, cast(grad_weights))
    return flux, fac, go, gn


def check_sample_against_oracle(flux, intercept, on_target, group, rt, inc, tix, planar, rows, label):
    o_flux, o_fac = oracle_rows(group, rt, inc, tix, planar, rows)
    for k, h in enumerate(rows):
        err = rel_l2(n(flux[h]), o_flux[k])
        print(f"{label}: heliostat {h}: flux rel L2 vs oracle {err:.2e}, intercept {float(intercept[h]):.4f}")
        assert err < 1e-5, (label, h, err)                       # north_star: flux-bitmap L2 error < 1e-5
    sel = torch.tensor(rows, device=DEV)
    np.testing.assert_array_equal(n(intercept[sel]), o_fac[0])   # ray counters are integers: exact
    np.testing.assert_array_equal(n(on_target[sel]), o_fac[1])


def check_invariants(scenario, group, mask, tix, inc, flux, per_target_fused, T, label):
    """Fused per-target == segment sum; three ranks' shards add up to the single-rank bitmaps."""
    from artist_amd import HeliostatRayTracer
    H = flux.shape[0]
    rt = HeliostatRayTracer(scenario, group, blocking_active=False)
    segment = rt.get_bitmaps_per_target(flux, tix)
    assert segment.shape == (T, 256, 256)
    # fp32 sums of up to 1e4 bitmaps in a different order: a few 1e-7 relative
    assert rel_l2(n(per_target_fused), n(segment)) < 2e-6, (label, rel_l2(n(per_target_fused), n(segment)))
    acc = torch.zeros_like(segment)
    seen = []
    for rank in range(3):
        rtr = HeliostatRayTracer(scenario, group, blocking_active=False, world_size=3, rank=rank)
        assert rtr.distortions_dataset.distortions_u.shape[0] == len(rtr.distortions_sampler.rank_indices)   # owned rows only
        local, *_ = rtr.trace_rays_per_target(inc, mask, tix)
        idx = rtr.get_sampler_indices()
        seen.append(idx)
        acc += local
        del rtr, local
    assert sorted(torch.cat(seen).tolist()) == list(range(H))
    assert rel_l2(n(acc), n(segment)) < 2e-6, (label, rel_l2(n(acc), n(segment)))


def test_config3_field_forward_per_target():
    """Config 3: 500 heliostats x 100 rays x 10^4 points = 5e8 rays, forward, accumulated to [1,256,256]."""
    from artist_amd import HeliostatRayTracer
    H, R = 500, 100
    scenario, group, mask, tix, inc, _ = build_field(H, R)
    planar = scenario.solar_tower.target_areas[0]
    rt = HeliostatRayTracer(scenario, group, blocking_active=False)
    assert rt.distortions_dataset.distortions_u.shape == (H, R, 10000)
    fused, ic, ot, bl = rt.trace_rays_per_target(inc, mask, tix)
    flux, intercept, on_target, blocking = rt.trace_rays(inc, mask, tix)
    assert fused.shape == (1, 256, 256) and flux.shape == (H, 256, 256)
    assert bool(torch.isfinite(fused).all()) and bool((blocking == 1).all())
    np.testing.assert_array_equal(n(ic), n(intercept))          # same rays in both modes
    np.testing.assert_array_equal(n(ot), n(on_target))
    check_sample_against_oracle(flux, intercept, on_target, group, rt, inc, tix, planar, spread(H), "config 3")
    del rt
    check_invariants(scenario, group, mask, tix, inc, flux, fused, 1, "config 3")


def test_config5_ten_thousand_heliostats_one_ray_per_point():
    """Config 5: 10 000 heliostats x 1 ray x 10^4 points = 1e8 rays, forward (ten points per thread, one workgroup per
    heliostat: the pipelined window loop of the forward kernel), per-target accumulation."""
    from artist_amd import HeliostatRayTracer
    H, R = 10000, 1
    scenario, group, mask, tix, inc, _ = build_field(H, R)
    planar = scenario.solar_tower.target_areas[0]
    rt = HeliostatRayTracer(scenario, group, blocking_active=False)
    fused, ic, ot, _ = rt.trace_rays_per_target(inc, mask, tix)
    flux, intercept, on_target, _ = rt.trace_rays(inc, mask, tix)       # [10000,256,256] = 2.6 GB
    np.testing.assert_array_equal(n(ic), n(intercept))
    check_sample_against_oracle(flux, intercept, on_target, group, rt, inc, tix, planar, spread(H, 6), "config 5")
    del rt
    check_invariants(scenario, group, mask, tix, inc, flux, fused, 1, "config 5")


def _epoch(group, uv, inc, tix, planar, scenario, cp, orientation, du, de, weights, n_cp):
    """control points -> NURBS -> alignment -> trace -> weighted flux; returns flux, factors and d/d(control points)."""
    from artist_amd import NURBSSurfaces, ops
    H = cp.shape[0]
    pts, nrm = NURBSSurfaces(group.nurbs_degrees, cp, device=DEV).calculate_surface_points_and_normals(
        uv, group.active_canting, group.active_facet_translations)
    P = pts.shape[1] * pts.shape[2]
    ap, an = ops.align_surfaces(pts.reshape(H, P, 4), nrm.reshape(H, P, 4), orientation)
    flux, factors = ops.trace_rays(ap, an, inc, du, de, tix, planar.centers, planar.normals, planar.dimensions, 1.0, 0.0, 0.935,
                                   (256, 256))
    (g_cp,) = torch.autograd.grad((flux * weights).sum(), cp)
    return flux, factors, g_cp, ap.detach(), an.detach()


def _oracle_epoch(dtype, cp, uv, canting, transl, orientation, inc, du, de, tix, planar, weights, ap_hip, an_hip):
    """The oracle's epoch.  Surfaces and alignment are evaluated by the oracle too and must agree with the HIP stages to
    rounding; the trace itself is then given the SAME aligned surfaces the HIP trace saw - one ULP of a normal moves
    every ray of that point by ~4e-4 px, i.e. the flux by ~1e-4, which would drown the 1e-5 bound of the trace stage."""
    c = lambda a: np.ascontiguousarray(n(a)).astype(dtype)  # noqa: E731
    H = cp.shape[0]
    pts, nrm = oracle.nurbs_fwd(c(cp), c(uv), [3, 3], c(canting), c(transl))
    P = pts.shape[1] * pts.shape[2]
    ori = c(orientation)
    ap = (pts.reshape(H, P, 4) @ ori.transpose(0, 2, 1)).astype(dtype)
    an = (nrm.reshape(H, P, 4) @ ori.transpose(0, 2, 1)).astype(dtype)
    assert rel_l2(n(ap_hip), ap) < 1e-6 and rel_l2(n(an_hip), an) < 1e-6          # stages upstream of the trace
    args = (c(ap_hip), c(an_hip), c(inc), c(du), c(de), n(tix).astype(np.int32), c(planar.centers), c(planar.normals),
            c(planar.dimensions), (256, 256))
    flux, fac = oracle.trace_fwd(*args)
    go, gn = oracle.trace_bwd(*args, c(weights))
    g_cp = oracle.nurbs_bwd(c(cp), c(uv), [3, 3], (go @ ori).reshape(pts.shape), (gn @ ori).reshape(nrm.shape), c(canting))
    # ... and END TO END: the oracle's trace on the oracle's OWN aligned surfaces (nothing taken from the HIP run)
    flux_e2e, _ = oracle.trace_fwd(ap, an, *args[2:])
    return flux, fac, g_cp, flux_e2e


@pytest.mark.parametrize("H,R,n_cp,label", [(100, 180, 6, "config 4"), (1000, 100, 10, "metric config")])
def test_reconstruction_epoch_forward_and_control_point_gradients(H, R, n_cp, label):
    """Config 4 (100 heliostats x 180 rays, 6x6 degree-3 control nets) and the metric config (1000 x 100, 10x10): one
    surface-reconstruction epoch forward + backward on the whole field; a sample of heliostats against the oracle's
    epoch (flux < 1e-5 with the trace stage fed the same aligned surfaces, counters exact, control-point gradients within the
    oracle's own fp32-vs-fp64 distance) AND end to end - control points to flux with every stage the oracle's own - within the
    distance between the oracle's fp32 and fp64 chains (printed)."""
    from artist_amd import scene
    scenario, group, mask, tix, inc, uv = build_field(H, R, n_cp=n_cp)
    planar = scenario.solar_tower.target_areas[0]
    aim = scenario.solar_tower.get_centers_of_target_areas(tix)
    orientation = scene.ideal_orientations(group.active_positions, aim, inc)
    cp = group.active_nurbs_control_points.clone().requires_grad_(True)
    du, de = scenario.light_sources.light_source_list[0].get_distortions(10000, H, random_seed=7)
    gen = torch.Generator(device=DEV).manual_seed(3)
    weights = torch.rand((H, 256, 256), generator=gen, device=DEV)
    flux, factors, g_cp, ap, an = _epoch(group, uv, inc, tix, planar, scenario, cp, orientation, du, de, weights, n_cp)
    assert g_cp.shape == (H, 4, n_cp, n_cp, 3) and bool(torch.isfinite(g_cp).all()) and float(g_cp.abs().sum()) > 0
    rows = spread(H)
    sel = torch.tensor(rows, device=DEV)
    take = lambda a: a[sel]  # noqa: E731
    both = torch.stack((du[sel], de[sel]), dim=-1).contiguous()
    o32 = _oracle_epoch(np.float32, take(cp), take(uv), take(group.active_canting), take(group.active_facet_translations),
                        take(orientation), take(inc), both[..., 0], both[..., 1], take(tix), planar, take(weights), take(ap), take(an))
    o64 = _oracle_epoch(np.float64, take(cp), take(uv), take(group.active_canting), take(group.active_facet_translations),
                        take(orientation), take(inc), both[..., 0], both[..., 1], take(tix), planar, take(weights), take(ap), take(an))
    for k, h in enumerate(rows):
        err = rel_l2(n(flux[h]), o32[0][k])
        assert err < 1e-5, (label, h, err)
        yard = rel_l2(o32[2][k], o64[2][k])                       # how far fp32 arithmetic alone moves this gradient
        got = rel_l2(n(g_cp[h]), o32[2][k])
        print(f"{label}: heliostat {h}: flux {err:.2e}; d/d(control points) vs oracle fp32 {got:.2e}, "
              f"vs fp64 {rel_l2(n(g_cp[h]), o64[2][k]):.2e}, oracle fp32-vs-fp64 {yard:.2e}")
        assert got < max(2.0 * yard, 2e-4), (label, h, got, yard)
        assert rel_l2(n(g_cp[h]), o64[2][k]) < max(3.0 * yard, 2e-4), (label, h)
        # end to end (control points -> flux, every stage the oracle's own): one ULP of a normal moves every ray of its point by
        # ~4e-4 px, so the yardstick is how far the oracle's fp32 chain is from its fp64 chain
        e2e, e2e_yard = rel_l2(n(flux[h]), o32[3][k]), rel_l2(o32[3][k], o64[3][k])
        print(f"{label}: heliostat {h}: END TO END flux vs the oracle's own chain {e2e:.2e} (oracle fp32-vs-fp64 chain: {e2e_yard:.2e}; "
              f"vs the fp64 chain {rel_l2(n(flux[h]), o64[3][k]):.2e})")
        assert e2e < max(3.0 * e2e_yard, 1e-4), (label, h, e2e, e2e_yard)
        assert rel_l2(n(flux[h]), o64[3][k]) < max(3.0 * e2e_yard, 1e-4), (label, h)
    np.testing.assert_array_equal(n(factors[0][sel]), o32[1][0])
    np.testing.assert_array_equal(n(factors[1][sel]), o32[1][1])
    # invariants on the whole field (the tracer mirror reads the aligned surfaces from the group)
    group.active_surface_points, group.active_surface_normals = ap, an
    from artist_amd import HeliostatRayTracer
    rt = HeliostatRayTracer(scenario, group, blocking_active=False)
    fused, *_ = rt.trace_rays_per_target(inc, mask, tix)
    flux_rt, *_ = rt.trace_rays(inc, mask, tix)
    del rt
    check_invariants(scenario, group, mask, tix, inc, flux_rt, fused, 1, label)


def test_field_groups_change_speed_only(monkeypatch):
    """Field-scale prediction (per-target bitmaps, a few samples per point) groups consecutive heliostats into one work
    item that shares a window and its flush (trace_fwd_item_field).  The grouping is a launch geometry like any other: the
    bitmaps and the ray counters are the same BITS with groups of any size, without groups, and as the segment sum's
    inputs - on a field with two targets in runs of irregular length, a run that crosses group borders, two heliostats
    that miss everything and one with a stale target index (skipped, reported)."""
    from artist_amd import ops
    from artist_amd.scene import build_synthetic_scenario
    H, R = 700, 2
    scenario, _ = build_synthetic_scenario(H, n_rays=R, n_cp=(6, 6), n_eval=16, device=DEV,
                                           target_centers=((0.0, 0.0, 55.0, 1.0), (3.0, 0.0, 48.0, 1.0)),
                                           target_normals=((0.0, 1.0, 0.0, 0.0), (0.0, 1.0, 0.0, 0.0)), target_dims=((8.0, 8.0), (6.0, 7.0)))
    group = scenario.heliostat_field.heliostat_groups[0]
    mask = torch.ones(H, dtype=torch.int32, device=DEV)
    group.activate_heliostats(mask)
    gen = torch.Generator().manual_seed(5)
    tix = torch.zeros(H, dtype=torch.long)
    k = 0
    while k < H:                                                   # runs of 1 .. 40 heliostats per target
        run = int(torch.randint(1, 41, (1,), generator=gen))
        tix[k:k + run] = int(torch.randint(0, 2, (1,), generator=gen))
        k += run
    tix = tix.to(DEV)
    inc = torch.tensor([[0.0, 1.0, 0.0, 0.0]], device=DEV).repeat(H, 1)
    group.align_surfaces_with_incident_ray_directions(scenario.solar_tower.get_centers_of_target_areas(tix), inc, mask)
    points, normals = group.active_surface_points.clone(), group.active_surface_normals.clone()
    up = torch.tensor([0.0, 0.0, 1.0, 0.0], device=DEV)            # two mirrors lying flat: the sun's rays go on northwards,
    normals[17] = up                                               # away from the tower - no ray reaches a target
    normals[400] = up
    P = points.shape[1]
    both = torch.randn((H, R, P, 2), device=DEV, generator=torch.Generator(device=DEV).manual_seed(3)) * 2.09e-3
    planar = scenario.solar_tower.target_areas[0]
    args = (points, normals, inc, both[..., 0], both[..., 1], tix, planar.centers, planar.normals, planar.dimensions, 1.0, 0.0, 0.935,
            (256, 256))

    def run(env, target_idx=tix):
        with monkeypatch.context() as m:
            for key, value in env.items():
                m.setenv(key, value)
            flux, fac = ops.trace_rays(*args[:5], target_idx, *args[6:], per_target=True)
            return n(flux), n(fac)

    plain_flux, plain_fac = run({"ARTIST_HIP_FIELD_GROUP": "0"})
    assert plain_flux.sum() > 0 and plain_fac[0, 17] == 0 and plain_fac[0, 400] == 0 and (plain_fac[0] > 0.5).sum() > H - 10
    for env in ({}, {"ARTIST_HIP_FIELD_GROUP": "2"}, {"ARTIST_HIP_FIELD_GROUP": "7"}, {"ARTIST_HIP_FIELD_GROUP": "32"},
                {"ARTIST_HIP_FIELD_GROUP": "13", "ARTIST_HIP_FWD_TILE_KB": "16"}):
        flux, fac = run(env)
        np.testing.assert_array_equal(flux, plain_flux, err_msg=str(env))
        np.testing.assert_array_equal(fac, plain_fac, err_msg=str(env))
    # ... and the per-heliostat bitmaps' segment sum agrees up to its fp32 additions
    per_h, _ = ops.trace_rays(*args)
    summed = n(ops.per_target_sum(per_h, tix, 2))
    assert rel_l2(plain_flux, summed) < 1e-6
    # a stale target index inside a group: that heliostat is skipped and reported, the others are untouched
    bad = tix.clone()
    bad[333] = 9
    flux_bad, fac_bad = run({"ARTIST_HIP_FIELD_GROUP": "8"}, bad)
    with pytest.raises(IndexError):
        ops.check_async_errors(DEV)
    one, fac_one = ops.trace_rays(*(a_[333:334] if isinstance(a_, torch.Tensor) and a_.shape[:1] == (H,) else a_ for a_ in args),
                                  per_target=True)
    assert fac_bad[0, 333] == 0 and np.array_equal(np.delete(fac_bad, 333, axis=1), np.delete(plain_fac, 333, axis=1))
    assert rel_l2(flux_bad + n(one), plain_flux) < 1e-6
