#!/usr/bin/env python3
"""Does running the epoch of one rank's heliostats as K independent sub-batches on K streams hide the kernels' tails?
(The heliostats of an epoch are independent up to the optimiser step: NURBS -> trace -> crop + loss -> backward per heliostat.)
usage: python tools/pipeline_probe.py [H] [K ...]    prints ms per epoch for each K (K = 1: the bench's single-stream step)"""
import json, pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
from artist_amd import NURBSSurfaces, ops, scene
from artist_amd.flux import FluxCrop, FluxCropPixelLoss
from artist_amd.scene import build_synthetic_scenario

dev = torch.device("cuda:0")
H = int(sys.argv[1]) if len(sys.argv) > 1 else 125
Ks = [int(x) for x in sys.argv[2:]] or [1, 2, 3, 4]
R, n_eval, n_cp = 100, 50, 10
P = 4 * n_eval * n_eval
scenario, uv = build_synthetic_scenario(H, n_rays=R, n_cp=(n_cp, n_cp), n_eval=n_eval, device=dev)
group = scenario.heliostat_field.heliostat_groups[0]
group.activate_heliostats(torch.ones(H, dtype=torch.int32, device=dev))
tix = torch.zeros(H, dtype=torch.long, device=dev)
inc = torch.tensor([[0.0, 1.0, 0.0, 0.0]], device=dev).repeat(H, 1)
aim = scenario.solar_tower.get_centers_of_target_areas(tix)
orientation = scene.ideal_orientations(group.active_positions, aim, inc)
planar = scenario.solar_tower.target_areas[0]
sun = scenario.light_sources.light_source_list[0]
dist_u, dist_e = sun.get_distortions_rows(list(range(H)), number_of_points=P, number_of_active_heliostats=H, random_seed=7)
cp_all = group.active_nurbs_control_points
canting, transl = group.active_canting, group.active_facet_translations
degrees = group.nurbs_degrees
crop_dims = planar.dimensions.index_select(0, tix).contiguous()
res = {}
for K in Ks:
    # sub-batch k takes heliostats k, k + K, ... (near and far heliostats in every sub-batch)
    idx = [torch.arange(k, H, K, device=dev) for k in range(K)]
    part = []
    for ix in idx:
        h = len(ix)
        part.append(dict(h=h, cp=cp_all[ix].clone().requires_grad_(True), cant=canting[ix].contiguous(), tr=transl[ix].contiguous(),
                         ori=orientation[ix].contiguous(), inc=inc[ix].contiguous(), tix=tix[ix].contiguous(),
                         du=dist_u[ix].contiguous() if K > 1 else dist_u, de=None, dims=crop_dims[ix].contiguous(),
                         uv=uv[:1].expand(h, -1, -1, -1), stream=torch.cuda.Stream(dev) if K > 1 else torch.cuda.current_stream(dev)))
    if K > 1:
        for p_, ix in zip(part, idx):
            both = torch.stack((dist_u[ix], dist_e[ix]), dim=-1).contiguous()
            p_["du"], p_["de"] = both[..., 0], both[..., 1]
    else:
        part[0]["de"] = dist_e
    from artist_amd.optim import Adam
    opt = Adam([p_["cp"] for p_ in part], lr=1e-6)

    def fwd(p_):
        ap, an = NURBSSurfaces(degrees, p_["cp"], device=dev).calculate_surface_points_and_normals(p_["uv"], p_["cant"], p_["tr"], orientations=p_["ori"])
        flux, _ = ops.trace_rays(ap.reshape(p_["h"], P, 4), an.reshape(p_["h"], P, 4), p_["inc"], p_["du"], p_["de"], p_["tix"], planar.centers,
                                 planar.normals, planar.dimensions, 1.0, 0.0, 0.935, (256, 256), points_per_facet=n_eval * n_eval)
        return flux

    for p_ in part:
        with torch.no_grad(), torch.cuda.stream(p_["stream"]):
            p_["target"] = (FluxCrop.apply(fwd(p_), p_["dims"], 6.0, 6.0) * 1.05 + 1e-3).detach()
    torch.cuda.synchronize()
    main = torch.cuda.current_stream(dev)

    def step():
        opt.zero_grad(set_to_none=True)
        if K > 1:
            ev0 = torch.cuda.Event(); ev0.record(main)
        sums = []
        for p_ in part:
            with torch.cuda.stream(p_["stream"]):
                if K > 1:
                    p_["stream"].wait_event(ev0)
                flux = fwd(p_)
                sums.append(ops.per_target_sum(flux.detach(), p_["tix"], 1))
                loss = FluxCropPixelLoss.apply(flux, p_["dims"], p_["target"], 6.0, 6.0).sum()
                loss.backward()
                if K > 1:
                    ev = torch.cuda.Event(); ev.record(p_["stream"]); main.wait_event(ev)
        opt.step()
        return sums

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    res[K] = round((time.perf_counter() - t0) / n * 1e3, 4)
    print(f"H {H} K {K}: {res[K]} ms per epoch", flush=True)
print(json.dumps({"heliostats": H, "ms_per_epoch_by_streams": res}))
