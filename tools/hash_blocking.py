import sys, hashlib, pathlib, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from artist_amd import HeliostatRayTracer
from artist_amd.scene import build_synthetic_scenario
dev = torch.device("cuda:0")
H, R = 1000, 10
scenario, _ = build_synthetic_scenario(H, n_rays=R, device=dev)
g = scenario.heliostat_field.heliostat_groups[0]
i = torch.arange(H, device=dev)
g.positions = torch.stack([((i % 40) - 19.5) * 4.2, 60.0 + (i // 40) * 5.0, torch.zeros(H, device=dev), torch.ones(H, device=dev)], dim=1)
mask = torch.ones(H, dtype=torch.int32, device=dev)
g.activate_heliostats(mask)
tix = torch.zeros(H, dtype=torch.long, device=dev)
inc = torch.nn.functional.normalize(torch.tensor([[0.0, 0.94, -0.34, 0.0]], device=dev), dim=1).repeat(H, 1)
g.align_surfaces_with_incident_ray_directions(scenario.solar_tower.get_centers_of_target_areas(tix), inc, mask)
pts = g.active_surface_points.detach().requires_grad_(True)
g.active_surface_points = pts
rt = HeliostatRayTracer(scenario, g, blocking_active=True)
rt.lbvh_compat = False
torch.manual_seed(0)
flux, a, b, c = rt.trace_rays(inc, mask, tix)
w = torch.rand_like(flux)
(flux * w).sum().backward()
hh = lambda t: hashlib.sha1(t.detach().cpu().numpy().tobytes()).hexdigest()[:12]
print("flux", hh(flux), "unblocked", hh(c), float(c.mean()), "grad", hh(pts.grad), float(pts.grad.abs().sum()))
