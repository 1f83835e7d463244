#!/usr/bin/env python3
"""NURBS evaluation at the metric size, the two schemes side by side on the same box (A/B):
tensor-product (cartesian evaluation grid, the default) vs scattered (ARTIST_HIP_DEBUG=1 ARTIST_HIP_NURBS_GRID=0), forward and
backward, with and without the fused alignment; points must be the same bits, control-point gradients agree to rounding.
usage: python tools/nurbs_bench.py [H ...]   -> one JSON line (profiles/r04_nurbs_bench.json)"""
import json, os, pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
from artist_amd import NURBSSurfaces
from artist_amd.scene import build_synthetic_scenario

dev = torch.device("cuda:0")
os.environ["ARTIST_HIP_DEBUG"] = "1"


def timed(fn, steps=20):
    for _ in range(3):
        fn()
    ms = []
    for _ in range(steps):
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record(); fn(); en.record()
        torch.cuda.synchronize()
        ms.append(st.elapsed_time(en))
    ms.sort()
    return ms[len(ms) // 2]


out = {"what": "median of 20 launches, HIP events; 4 facets x 50 x 50 evaluation points, 10 x 10 degree-3 control nets", "cases": []}
for H in [int(x) for x in sys.argv[1:]] or [1000, 125]:
    scenario, uv = build_synthetic_scenario(H, n_rays=1, n_cp=(10, 10), n_eval=50, device=dev)
    group = scenario.heliostat_field.heliostat_groups[0]
    group.activate_heliostats(torch.ones(H, dtype=torch.int32, device=dev))
    cp = group.active_nurbs_control_points.clone().requires_grad_(True)
    cant, tr = group.active_canting, group.active_facet_translations
    uvx = uv[:1].expand(H, -1, -1, -1)
    gen = torch.Generator(device=dev).manual_seed(1)
    ori = torch.linalg.qr(torch.randn(H, 4, 4, device=dev, generator=gen))[0].contiguous()
    res = {}
    keep = {}
    for scheme, flag in (("tensor_product", "1"), ("scattered", "0")):
        os.environ["ARTIST_HIP_NURBS_GRID"] = flag
        for fused in (False, True):
            surf = NURBSSurfaces(group.nurbs_degrees, cp, device=dev)
            kw = dict(orientations=ori) if fused else {}
            pts, nrm = surf.calculate_surface_points_and_normals(uvx, cant, tr, **kw)
            gp, gn = torch.rand(pts.shape, device=dev, generator=gen), torch.rand(nrm.shape, device=dev, generator=gen)
            key = "fused_alignment" if fused else "plain"
            if key not in keep:
                keep[key] = (gp, gn)
            gp, gn = keep[key]
            (g_cp,) = torch.autograd.grad([pts, nrm], [cp], [gp, gn], retain_graph=True)
            with torch.no_grad():
                f_ms = timed(lambda: surf.calculate_surface_points_and_normals(uvx, cant, tr, **kw))
            b_ms = timed(lambda: torch.autograd.grad([pts, nrm], [cp], [gp, gn], retain_graph=True))
            (g2,) = torch.autograd.grad([pts, nrm], [cp], [gp, gn], retain_graph=True)
            res[(scheme, key)] = dict(fwd_ms=round(f_ms, 4), bwd_ms=round(b_ms, 4), pts=pts.detach(), nrm=nrm.detach(), g=g_cp,
                                      bwd_bit_reproducible=bool(torch.equal(g_cp, g2)))
    for key in ("plain", "fused_alignment"):
        a, b = res[("tensor_product", key)], res[("scattered", key)]
        out["cases"].append({
            "heliostats": H, "alignment": key,
            "tensor_product": {"fwd_ms": a["fwd_ms"], "bwd_ms": a["bwd_ms"], "bwd_bit_reproducible": a["bwd_bit_reproducible"]},
            "scattered": {"fwd_ms": b["fwd_ms"], "bwd_ms": b["bwd_ms"], "bwd_bit_reproducible": b["bwd_bit_reproducible"]},
            "points_bit_equal": bool(torch.equal(a["pts"], b["pts"])), "normals_bit_equal": bool(torch.equal(a["nrm"], b["nrm"])),
            "grad_rel_l2": float((a["g"].double() - b["g"].double()).norm() / b["g"].double().norm()),
            "fwd_GBps_tensor_product": round(H * 4 * 2500 * 32 / a["fwd_ms"] / 1e6, 1),
            "bwd_GBps_tensor_product": round(H * 4 * 2500 * 32 / a["bwd_ms"] / 1e6, 1)})
print(json.dumps(out))
