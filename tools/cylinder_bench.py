#!/usr/bin/env python3
"""Cost of a cylindrical receiver at the metric size: 1000 heliostats x 100 rays x 10000 points, every heliostat aimed at
the mantle of one cylinder (radius 6 m, height 12 m, opening 2 rad) - forward and forward + backward, against the same
field aimed at the planar receiver, and with every tenth heliostat on the cylinder."""
import json, sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
from artist_amd import HeliostatRayTracer
from artist_amd.scene import SolarTower, TowerTargetAreasCylindrical, build_synthetic_scenario

dev = torch.device("cuda:0")


def timed(fn, steps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main(H=1000, R=100, steps=5):
    scenario, _ = build_synthetic_scenario(H, n_rays=R, device=dev)
    planar = scenario.solar_tower.target_areas[0]
    cyl = TowerTargetAreasCylindrical(
        names=["cyl"], centers=torch.tensor([[0.0, -6.0, 55.0, 1.0]], device=dev),
        normals=torch.tensor([[0.0, 1.0, 0.0, 0.0]], device=dev), axes=torch.tensor([[0.0, 0.0, 1.0, 0.0]], device=dev),
        radii=torch.tensor([6.0], device=dev), heights=torch.tensor([12.0], device=dev),
        opening_angles=torch.tensor([2.0], device=dev))
    scenario.solar_tower = SolarTower([planar, cyl], device=dev)
    g = scenario.heliostat_field.heliostat_groups[0]
    mask = torch.ones(H, dtype=torch.int32, device=dev)
    inc = torch.nn.functional.normalize(torch.tensor([[0.0, 0.94, -0.34, 0.0]], device=dev), dim=1).repeat(H, 1)
    out = {"H": H, "R": R}
    for label, t in (("planar", 0), ("cylinder", 1), ("every_tenth_on_the_cylinder", -1)):
        tix = torch.full((H,), max(t, 0), dtype=torch.long, device=dev)
        if t < 0:
            tix[::10] = 1
        g.activate_heliostats(mask)              # (alignment starts from the unaligned surfaces)
        g.align_surfaces_with_incident_ray_directions(scenario.solar_tower.get_centers_of_target_areas(tix), inc, mask)
        pts = g.active_surface_points.detach().requires_grad_(True)
        g.active_surface_points = pts
        g.active_surface_normals = g.active_surface_normals.detach()
        out["rays"] = H * R * pts.shape[1]
        rt = HeliostatRayTracer(scenario, g, blocking_active=False)
        flux, intercept, on_target, unblocked = rt.trace_rays(inc, mask, tix)
        w = torch.rand_like(flux)

        def fwd():
            return rt.trace_rays(inc, mask, tix)[0]

        def fwd_bwd():
            pts.grad = None
            (rt.trace_rays(inc, mask, tix)[0] * w).sum().backward()

        out[label] = {"fwd_ms": timed(fwd, steps), "fwd_bwd_ms": timed(fwd_bwd, steps),
                      "mean_intercept": float(intercept.mean()), "flux_sum": float(flux.sum())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
