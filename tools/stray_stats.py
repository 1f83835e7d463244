#!/usr/bin/env python3
"""Share of stray rays (valid, on the bitmap, outside the workgroup's LDS window) per heliostat of the bench field.
Needs the diagnostic build artist_amd/libablate_COUNT_STRAYS.so (-DART_DEBUG_COUNT_STRAYS) via ARTIST_HIP_LIB."""
import sys, pathlib, json
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
from artist_amd import ops
from artist_amd.scene import build_synthetic_scenario
dev = torch.device("cuda:0")
H, R = int(sys.argv[1]) if len(sys.argv) > 1 else 200, 100
scenario, uv = build_synthetic_scenario(H, n_rays=R, device=dev)
g = scenario.heliostat_field.heliostat_groups[0]
mask = torch.ones(H, dtype=torch.int32, device=dev)
g.activate_heliostats(mask)
tix = torch.zeros(H, dtype=torch.long, device=dev)
inc = torch.tensor([[0.0, 1.0, 0.0, 0.0]], device=dev).repeat(H, 1)
g.align_surfaces_with_incident_ray_directions(scenario.solar_tower.get_centers_of_target_areas(tix), inc, mask)
both = torch.randn((H, R, g.active_surface_points.shape[1], 2), device=dev).mul_(4.3681e-06 ** 0.5)
planar = scenario.solar_tower.target_areas[0]
flux, fac = ops.trace_rays(g.active_surface_points, g.active_surface_normals, inc, both[..., 0], both[..., 1], tix,
                           planar.centers, planar.normals, planar.dimensions, 1.0, 0.0, 0.935, (256, 256))
stray = fac[2].cpu()
pos = g.positions.cpu()
order = torch.argsort(stray, descending=True)
print(json.dumps({"mean_stray_share": float(stray.mean()), "max": float(stray.max()),
                  "heliostats_with_strays": int((stray > 0).sum()), "H": H,
                  "share_above_1e-3": int((stray > 1e-3).sum()),
                  "worst": [(round(float(pos[i, 0]), 1), round(float(pos[i, 1]), 1), round(float(stray[i]), 5)) for i in order[:8]],
                  "quantiles": [round(float(q), 6) for q in torch.quantile(stray, torch.tensor([0.5, 0.75, 0.9, 0.99]))]}))
