#!/usr/bin/env python3
"""NURBS evaluation at the metric size, the two schemes side by side on the same box (A/B):
tensor-product (cartesian evaluation grid, the default) vs scattered (ARTIST_HIP_DEBUG=1 ARTIST_HIP_NURBS_GRID=0), forward and
backward, with and without the fused alignment; points must be the same bits, control-point gradients agree to rounding.
Kernel times: 20 launches back to back through the C ABI between one pair of HIP events (no host gap between the launches).
usage: python tools/nurbs_bench.py [H ...]   -> one JSON line (profiles/r04_nurbs_bench.json)"""
import json, os, pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
from artist_amd import NURBSSurfaces, _lib
from artist_amd.scene import build_synthetic_scenario

dev = torch.device("cuda:0")
os.environ["ARTIST_HIP_DEBUG"] = "1"
lib = _lib.lib()


def timed(fn, steps=20, reps=5):
    for _ in range(3):
        fn()
    best = []
    for _ in range(reps):
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        st.record()
        for _ in range(steps):
            fn()
        en.record()
        torch.cuda.synchronize()
        best.append(st.elapsed_time(en) / steps)
    best.sort()
    return best[len(best) // 2]


out = {"what": "median of 5 x (20 launches back to back through the C ABI, one HIP-event pair); 4 facets x 50 x 50 evaluation points, "
               "10 x 10 degree-3 control nets; bytes = 32 B per point written (forward) / read (backward)", "cases": []}
for H in [int(x) for x in sys.argv[1:]] or [1000, 125]:
    scenario, uv = build_synthetic_scenario(H, n_rays=1, n_cp=(10, 10), n_eval=50, device=dev)
    group = scenario.heliostat_field.heliostat_groups[0]
    group.activate_heliostats(torch.ones(H, dtype=torch.int32, device=dev))
    cp = group.active_nurbs_control_points.clone().requires_grad_(True)
    cant, tr = group.active_canting.contiguous(), group.active_facet_translations.reshape(H, 4, 4).contiguous()
    uvx = uv[:1].expand(H, -1, -1, -1)
    F, M = 4, uv.shape[2]
    gen = torch.Generator(device=dev).manual_seed(1)
    ori = torch.linalg.qr(torch.randn(H, 4, 4, device=dev, generator=gen))[0].contiguous()
    surf = NURBSSurfaces(group.nurbs_degrees, cp, device=dev)
    ku = surf.knot_vectors_u.contiguous()
    kv = surf.knot_vectors_v.contiguous()
    pts, nrm = torch.empty(H, F, M, 4, device=dev), torch.empty(H, F, M, 4, device=dev)
    gp, gn = torch.rand(pts.shape, device=dev, generator=gen), torch.rand(nrm.shape, device=dev, generator=gen)
    g_cp = torch.empty_like(cp)
    s = torch.cuda.current_stream().cuda_stream
    p = lambda t: t.data_ptr()
    res = {}
    for scheme, flag in (("tensor_product", "1"), ("scattered", "0")):
        os.environ["ARTIST_HIP_NURBS_GRID"] = flag
        for key, o in (("plain", None), ("fused_alignment", ori)):
            fwd = lambda: lib.art_nurbs_fwd(p(cp), p(uvx), uvx.stride(0), uvx.stride(1), p(ku), p(kv), p(cant), p(tr), 3, 3, 1, 8, 8, H, F, M, 10, 10,
                                            None if o is None else p(o), p(pts), p(nrm), s)
            bwd = lambda: lib.art_nurbs_bwd(p(cp), p(uvx), uvx.stride(0), uvx.stride(1), p(ku), p(kv), p(cant), 3, 3, 1, 8, 8, H, F, M, 10, 10,
                                            None if o is None else p(o), p(gp), p(gn), p(g_cp), s)
            assert fwd() == 0 and bwd() == 0
            torch.cuda.synchronize()
            first = g_cp.clone()
            f_ms, b_ms = timed(fwd), timed(bwd)
            torch.cuda.synchronize()
            res[(scheme, key)] = dict(fwd_ms=round(f_ms, 4), bwd_ms=round(b_ms, 4), pts=pts.clone(), nrm=nrm.clone(), g=g_cp.clone(),
                                      bwd_bit_reproducible=bool(torch.equal(first, g_cp)))
    for key in ("plain", "fused_alignment"):
        a, b = res[("tensor_product", key)], res[("scattered", key)]
        out["cases"].append({
            "heliostats": H, "alignment": key,
            "tensor_product": {"fwd_ms": a["fwd_ms"], "bwd_ms": a["bwd_ms"], "bwd_bit_reproducible": a["bwd_bit_reproducible"]},
            "scattered": {"fwd_ms": b["fwd_ms"], "bwd_ms": b["bwd_ms"], "bwd_bit_reproducible": b["bwd_bit_reproducible"]},
            "points_bit_equal": bool(torch.equal(a["pts"], b["pts"])), "normals_bit_equal": bool(torch.equal(a["nrm"], b["nrm"])),
            "grad_rel_l2": float((a["g"].double() - b["g"].double()).norm() / b["g"].double().norm()),
            "fwd_GBps_tensor_product": round(H * F * M * 32 / a["fwd_ms"] / 1e6, 1),
            "bwd_GBps_tensor_product": round(H * F * M * 32 / a["bwd_ms"] / 1e6, 1)})
print(json.dumps(out))
