#!/usr/bin/env python3
"""Launch-geometry sweep of the tensor-product NURBS kernels (ARTIST_HIP_DEBUG knobs): waves per facet in the forward, waves per
workgroup in the backward.  usage: python tools/nurbs_sweep.py [H ...]"""
import os, pathlib, sys
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
        torch.cuda.synchronize(); st.record()
        for _ in range(steps):
            fn()
        en.record(); torch.cuda.synchronize()
        best.append(st.elapsed_time(en) / steps)
    return sorted(best)[len(best) // 2]

for H in [int(x) for x in sys.argv[1:]] or [1000, 125]:
    scenario, uv = build_synthetic_scenario(H, n_rays=1, n_cp=(10, 10), n_eval=50, device=dev)
    group = scenario.heliostat_field.heliostat_groups[0]
    group.activate_heliostats(torch.ones(H, dtype=torch.int32, device=dev))
    cp = group.active_nurbs_control_points.clone()
    cant, tr = group.active_canting.contiguous(), group.active_facet_translations.reshape(H, 4, 4).contiguous()
    uvx = uv[:1].expand(H, -1, -1, -1)
    F, M = 4, uv.shape[2]
    surf = NURBSSurfaces(group.nurbs_degrees, cp, device=dev)
    ku, kv = surf.knot_vectors_u.contiguous(), surf.knot_vectors_v.contiguous()
    pts, nrm = torch.empty(H, F, M, 4, device=dev), torch.empty(H, F, M, 4, device=dev)
    gp, gn = torch.rand_like(pts), torch.rand_like(nrm)
    g_cp = torch.empty_like(cp)
    s = torch.cuda.current_stream().cuda_stream
    p = lambda t: t.data_ptr()
    fwd = lambda: lib.art_nurbs_fwd(p(cp), p(uvx), uvx.stride(0), uvx.stride(1), p(ku), p(kv), p(cant), p(tr), 3, 3, 1, 8, 8, H, F, M, 10, 10, None, p(pts), p(nrm), s)
    bwd = lambda: lib.art_nurbs_bwd(p(cp), p(uvx), uvx.stride(0), uvx.stride(1), p(ku), p(kv), p(cant), 3, 3, 1, 8, 8, H, F, M, 10, 10, None, p(gp), p(gn), p(g_cp), s)
    for g in (1, 2, 3, 4, 6, 8, 12, 16):
        os.environ["ARTIST_HIP_NURBS_GROUPS"] = str(g)
        print(f"H {H} fwd groups {g}: {timed(fwd) * 1e3:.1f} us", flush=True)
    os.environ.pop("ARTIST_HIP_NURBS_GROUPS")
    for b in (64, 128, 256, 512):
        os.environ["ARTIST_HIP_NURBS_BWD_BLOCK"] = str(b)
        print(f"H {H} bwd block {b}: {timed(bwd) * 1e3:.1f} us", flush=True)
    os.environ.pop("ARTIST_HIP_NURBS_BWD_BLOCK")
