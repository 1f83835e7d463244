#!/usr/bin/env python3
"""Why is the backward trace kernel ~4 % slower inside the epoch than in a row of its own launches?
One process, the metric field, the backward launch timed with HIP events in four settings, interleaved rounds:
  row       ten launches back to back
  copy      each launch preceded by an HBM-bound pass of the epoch's size (a 786 MB read + write: the crop / loss kernels' traffic)
  forward   each launch preceded by a forward trace (the epoch's other long kernel)
  idle      each launch preceded by ~1 ms of idle GPU (the host sleeps)
usage (GPU box): python tools/bwd_context.py [--heliostats 1000]"""
import argparse, pathlib, statistics, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
from artist_amd import ops, scene


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--heliostats", type=int, default=1000)
    ap.add_argument("--rounds", type=int, default=5)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    H, R = args.heliostats, 100
    scenario, uv = scene.build_synthetic_scenario(H, n_rays=R, device=dev)
    group = scenario.heliostat_field.heliostat_groups[0]
    mask = torch.ones(H, dtype=torch.int32, device=dev)
    group.activate_heliostats(mask)
    tix = torch.zeros(H, dtype=torch.long, device=dev)
    inc = torch.tensor([[0.0, 1.0, 0.0, 0.0]], device=dev).repeat(H, 1)
    group.align_surfaces_with_incident_ray_directions(scenario.solar_tower.get_centers_of_target_areas(tix), inc, mask)
    ap_, an_ = group.active_surface_points.contiguous(), group.active_surface_normals.contiguous()
    P = ap_.shape[1]
    ppf = P // group.number_of_facets_per_heliostat
    gen = torch.Generator(device=dev).manual_seed(7)
    both = torch.randn((H, R, P, 2), generator=gen, device=dev).mul_(4.3681e-06 ** 0.5)
    du, de = both[..., 0], both[..., 1]
    planar = scenario.solar_tower.target_areas[0]
    gflux = torch.rand((H, 256, 256), device=dev)
    scratch_a = torch.rand((H, 256, 256), device=dev)
    scratch_b = torch.empty_like(scratch_a)
    apg, ang = ap_.clone().requires_grad_(True), an_.clone().requires_grad_(True)
    flux, _ = ops.trace_rays(apg, ang, inc, du, de, tix, planar.centers, planar.normals, planar.dimensions, 1.0, 0.0, 0.935, (256, 256),
                             points_per_facet=ppf)

    def fwd():
        with torch.no_grad():
            ops.trace_rays(ap_, an_, inc, du, de, tix, planar.centers, planar.normals, planar.dimensions, 1.0, 0.0, 0.935, (256, 256),
                           points_per_facet=ppf)

    def before(kind):
        if kind == "copy":
            for _ in range(3):
                scratch_b.copy_(scratch_a)
        elif kind == "forward":
            fwd()
        elif kind == "idle":
            torch.cuda.synchronize()
            time.sleep(1e-3)

    def measure(kind, n=10):
        ev = []
        for _ in range(n):
            before(kind)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            torch.autograd.grad(flux, (apg, ang), gflux, retain_graph=True)
            b.record()
            ev.append((a, b))
        torch.cuda.synchronize()
        return [a.elapsed_time(b) for a, b in ev]

    kinds = ["row", "copy", "forward", "idle"]
    out = {k: [] for k in kinds}
    for rnd in range(args.rounds + 1):
        for k in kinds:
            t = measure(k)
            if rnd > 0:
                out[k].extend(t[2:])                 # (the first two launches of a setting still see the previous one)
    for k in kinds:
        print(f"{k:8s} backward launch {statistics.median(out[k]):.3f} ms (min {min(out[k]):.3f}, max {max(out[k]):.3f}, n {len(out[k])})")


if __name__ == "__main__":
    main()
