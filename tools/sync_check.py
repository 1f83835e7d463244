#!/usr/bin/env python3
"""Which calls of a ray-tracing epoch wait for the device?  Runs HeliostatRayTracer.trace_rays (blocking off / on) and a
backward pass under torch.cuda.set_sync_debug_mode("warn") after a warm-up call and prints the warnings."""
import sys, pathlib, warnings
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
from artist_amd import HeliostatRayTracer
from artist_amd.scene import build_synthetic_scenario

dev = torch.device("cuda:0")
H = 64
scenario, _ = build_synthetic_scenario(H, n_rays=10, n_eval=20, device=dev)
uv_grid = _
g = scenario.heliostat_field.heliostat_groups[0]
mask = torch.ones(H, dtype=torch.int32, device=dev)
g.activate_heliostats(mask)
tix = torch.zeros(H, dtype=torch.long, device=dev)
inc = torch.nn.functional.normalize(torch.tensor([[0.0, 0.94, -0.34, 0.0]], device=dev), dim=1).repeat(H, 1)
g.align_surfaces_with_incident_ray_directions(scenario.solar_tower.get_centers_of_target_areas(tix), inc, mask)
pts = g.active_surface_points.detach().requires_grad_(True)
g.active_surface_points = pts
g.active_surface_normals = g.active_surface_normals.detach()
for blocking in (False, True):
    rt = HeliostatRayTracer(scenario, g, blocking_active=blocking)
    w = None
    for it in range(3):
        if it == 2:
            torch.cuda.synchronize()
            torch.cuda.set_sync_debug_mode("warn")
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            flux = rt.trace_rays(inc, mask, tix)[0]
            if w is None:
                w = torch.rand_like(flux)
            pts.grad = None
            (flux * w).sum().backward(retain_graph=True)
            per_target = rt.get_bitmaps_per_target(flux.detach(), tix)
        if it == 2:
            torch.cuda.set_sync_debug_mode("default")
            print(f"blocking={blocking}: {len(caught)} synchronising call(s) in the third epoch")
            for c in caught:
                print("   ", str(c.message)[:160], f"({pathlib.Path(c.filename).name}:{c.lineno})")

# ---- the rest of a surface-reconstruction epoch through the mirror classes: NURBS evaluation, alignment, crop + pixel loss ----
from artist_amd import NURBSSurfaces
from artist_amd.flux import crop_and_pixel_loss
cp = g.nurbs_control_points.detach().clone().requires_grad_(True)
uv = uv_grid
rt = HeliostatRayTracer(scenario, g, blocking_active=False)
target = None
for it in range(3):
    if it == 2:
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("warn")
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        surf = NURBSSurfaces(g.nurbs_degrees, cp, device=dev)
        pts_, nrm_ = surf.calculate_surface_points_and_normals(uv, g.canting, g.facet_translations)
        g.surface_points, g.surface_normals = pts_.reshape(H, -1, 4), nrm_.reshape(H, -1, 4)
        g.activate_heliostats(mask)
        g.align_surfaces_with_incident_ray_directions(scenario.solar_tower.get_centers_of_target_areas(tix), inc, mask)
        flux = rt.trace_rays(inc, mask, tix)[0]
        if target is None:
            target = torch.rand_like(flux) + 0.1
        loss = crop_and_pixel_loss(flux, scenario.solar_tower, tix, target).sum()
        cp.grad = None
        loss.backward()
    if it == 2:
        torch.cuda.set_sync_debug_mode("default")
        ours = [c for c in caught if pathlib.Path(c.filename).name != "scene.py"]
        print(f"NURBS -> activate/align -> trace -> crop + pixel loss -> backward: {len(ours)} synchronising call(s) in the third epoch "
              f"outside scene.py (+ {len(caught) - len(ours)} in scene.py's stand-ins of ARTIST's HeliostatGroup: activate_heliostats repeats "
              "its tensors with repeat_interleave(mask) like the original, heliostat_group.py:225-315, which reads the mask back)")
        for c in ours:
            print("   ", str(c.message)[:160], f"({pathlib.Path(c.filename).name}:{c.lineno})")
