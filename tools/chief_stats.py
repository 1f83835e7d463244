"""Where do the chief rays of the metric field land?  Per heliostat: spread of the chief-ray hits (pixels), sun-shape pad,
share of points outside / near the edge of a centred window of the LDS capacity."""
import sys

import torch

sys.path.insert(0, ".")
from artist_amd import NURBSSurfaces, ops, scene  # noqa: E402
from artist_amd.scene import build_synthetic_scenario  # noqa: E402

dev = torch.device("cuda:0")
H = 1000
scenario, uv = build_synthetic_scenario(H, n_rays=100, n_cp=(10, 10), n_eval=50, device=dev)
group = scenario.heliostat_field.heliostat_groups[0]
group.activate_heliostats(torch.ones(H, dtype=torch.int32, device=dev))
tix = torch.zeros(H, dtype=torch.long, device=dev)
inc = torch.tensor([[0.0, 1.0, 0.0, 0.0]], device=dev).repeat(H, 1)
aim = scenario.solar_tower.get_centers_of_target_areas(tix)
ori = scene.ideal_orientations(group.active_positions, aim, inc)
P = 10000
pts, nrm = NURBSSurfaces(group.nurbs_degrees, group.active_nurbs_control_points, device=dev).calculate_surface_points_and_normals(
    uv[:1].expand(H, -1, -1, -1), group.active_canting, group.active_facet_translations)
ap, an = ops.align_surfaces(pts.reshape(H, P, 4), nrm.reshape(H, P, 4), ori)
planar = scenario.solar_tower.target_areas[0]
c, nn, dims = planar.centers[0], planar.normals[0], planar.dimensions[0]
d = inc[:, None, :3] - 2 * (inc[:, None, :3] * an[..., :3]).sum(-1, keepdim=True) * an[..., :3]
t = ((c[:3] - ap[..., :3]) * nn[:3]).sum(-1) / (d * nn[:3]).sum(-1)
hit = ap[..., :3] + d * t[..., None]
be = (hit[..., 0] - c[0] + dims[0] / 2) / dims[0] * 255
bu = (hit[..., 2] - c[2] + dims[1] / 2) / dims[1] * 255
dist = t.abs().mean(1)
for h in (0, 100, 250, 500, 750, 900, 999):
    e, u = be[h], bu[h]
    pad = 1.15 * 4.5 * 0.00209 * float(dist[h]) * 255 / float(dims[0]) + 2
    for blk in (0, 5):
        eb, ub = e[blk * 1024:(blk + 1) * 1024], u[blk * 1024:(blk + 1) * 1024]
        half = 98.0
        ce, cu = eb.mean(), ub.mean()
        me = half - (eb - ce).abs()
        mu = half - (ub - cu).abs()
        m = torch.minimum(me, mu)
        print(f"h={h} blk={blk} dist={float(dist[h]):.0f} m  e [{float(eb.min()):.0f},{float(eb.max()):.0f}] std {float(eb.std()):.1f}  "
              f"u [{float(ub.min()):.0f},{float(ub.max()):.0f}] std {float(ub.std()):.1f}  pad {pad:.0f} px  "
              f"outside {float((m < 0).float().mean()):.3f}  <10px {float((m < 10).float().mean()):.3f}  "
              f"<20px {float((m < 20).float().mean()):.3f} <40px {float((m < 40).float().mean()):.3f}")
