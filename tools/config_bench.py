#!/usr/bin/env python3
"""Forward flux prediction timings for BASELINE.json configs 2, 3 and 5 (single GPU share), per-heliostat
bitmaps + segment sum vs the fused per-target mode."""
import json, sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
from artist_amd import HeliostatRayTracer, ops
from artist_amd.scene import build_synthetic_scenario

dev = torch.device("cuda:0")


def run(name, H, R, steps=5):
    scenario, uv = build_synthetic_scenario(H, n_rays=R, device=dev)
    g = scenario.heliostat_field.heliostat_groups[0]
    mask = torch.ones(H, dtype=torch.int32, device=dev)
    g.activate_heliostats(mask)
    tix = torch.zeros(H, dtype=torch.long, device=dev)
    inc = torch.tensor([[0.0, 1.0, 0.0, 0.0]], device=dev).repeat(H, 1)
    g.align_surfaces_with_incident_ray_directions(scenario.solar_tower.get_centers_of_target_areas(tix), inc, mask)
    gen = torch.Generator(device=dev).manual_seed(7)       # (the same sample in every run: the printed flux sums are comparable)
    both = torch.randn((H, R, g.active_surface_points.shape[1], 2), device=dev, generator=gen).mul_(4.3681e-06 ** 0.5)
    planar = scenario.solar_tower.target_areas[0]
    args = (g.active_surface_points, g.active_surface_normals, inc, both[..., 0], both[..., 1], tix, planar.centers,
            planar.normals, planar.dimensions, 1.0, 0.0, 0.935, (256, 256))
    out = {}
    for label, fn in (("per_heliostat+segment_sum", lambda: ops.per_target_sum(ops.trace_rays(*args)[0], tix, 1)),
                      ("fused_per_target", lambda: ops.trace_rays(*args, per_target=True)[0])):
        for _ in range(2):
            r = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            r = fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        out[label] = {"ms": dt * 1e3, "rays_per_s": H * R * 10000 / dt}
        out[label + "_sum"] = float(r.sum())
    print(json.dumps({"config": name, "H": H, "R": R, **out}))


if __name__ == "__main__":
    run("config2 (1 heliostat, 1e6 rays)", 1, 100)
    run("config3 (500 heliostats, 5e8 rays, per-target)", 500, 100)
    run("config5 share (1250 heliostats x 1 ray/point, 1.25e7 rays: one of 8 ranks)", 1250, 1)
    run("config5 whole (10000 heliostats x 1 ray/point, 1e8 rays)", 10000, 1)
