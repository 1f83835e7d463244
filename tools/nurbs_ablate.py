import os, pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
from artist_amd import NURBSSurfaces, _lib
from artist_amd.scene import build_synthetic_scenario
dev = torch.device("cuda:0")
os.environ["ARTIST_HIP_DEBUG"] = "1"
lib = _lib.lib()
def timed(fn, steps=20, reps=5):
    for _ in range(3): fn()
    best = []
    for _ in range(reps):
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); st.record()
        for _ in range(steps): fn()
        en.record(); torch.cuda.synchronize()
        best.append(st.elapsed_time(en) / steps)
    return sorted(best)[len(best) // 2]
H = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
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
bwd = lambda: lib.art_nurbs_bwd(p(cp), p(uvx), uvx.stride(0), uvx.stride(1), p(ku), p(kv), p(cant), 3, 3, 1, 8, 8, H, F, M, 10, 10, None, p(gp), p(gn), p(g_cp), s)
for ab, what in ((0, "full"), (1, "no stage A"), (2, "no points"), (3, "no points, no stage A"), (7, "no stage1/points/A"), (8, "no stage B"), (16, "no strips"), (24, "prologue only")):
    os.environ["ARTIST_HIP_NURBS_ABLATE"] = str(ab)
    print(f"H {H} bwd ablate {ab:2d} ({what}): {timed(bwd) * 1e3:.1f} us", flush=True)
