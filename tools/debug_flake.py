"""Debug: the blocking split backward of the random scene [8-12-77-17]: who owns which heliostat, and a stress loop."""
import os, sys, pathlib
os.environ["ARTIST_HIP_FWD_BLOCKS"] = "1"
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent / "tests"))
import numpy as np, torch
from test_gpu_parity import _random_feature_scene, DEV
from artist_amd import trace_rays, ops, _lib
seed, H, P, R = 8, 12, 77, 17
sc = _random_feature_scene(seed, H, P, R)
dv = lambda x: x.to(DEV).contiguous()
tix = sc["target_idx"].clone() % 2
prims = {k: dv(v) for k, v in sc["prims"].items()}
both = dv(sc["both"])
res = (96, 64)
saved = {}
orig_fwd = ops.TraceRays.forward
# peek at the candidate counts the forward saves for the backward
o, nn_ = dv(sc["origins"]).requires_grad_(True), dv(sc["normals"]).requires_grad_(True)
flux, fac, flags = trace_rays(o, nn_, dv(sc["incident"]), both[..., 0], both[..., 1], dv(tix), dv(sc["planes"]["centers"]),
                              dv(sc["planes"]["normals"]), dv(sc["planes"]["dims"]), ray_magnitude=0.7, extinction=0.05,
                              reflectivity=0.9, resolution=res, blocking=dict(prims, lbvh_compat=False))
ctx = flux.grad_fn
cand_count = [t_ for t_ in ctx.saved_tensors if t_.dtype == torch.int32 and t_.shape == (H,)]
print("int32 [H] saved tensors (target_idx, cand_count):", [t_.cpu().tolist() for t_ in cand_count])
print("fac[2]", fac[2].cpu().numpy().round(4).tolist())
w = torch.rand(flux.shape, generator=torch.Generator().manual_seed(seed)).to(DEV)
big = torch.randn(64 << 20, device=DEV)
ref = None
bad = 0
for rep in range(600):
    o, nn_ = dv(sc["origins"]).requires_grad_(True), dv(sc["normals"]).requires_grad_(True)
    flux, fac, flags = trace_rays(o, nn_, dv(sc["incident"]), both[..., 0], both[..., 1], dv(tix), dv(sc["planes"]["centers"]),
                                  dv(sc["planes"]["normals"]), dv(sc["planes"]["dims"]), ray_magnitude=0.7, extinction=0.05,
                                  reflectivity=0.9, resolution=res, blocking=dict(prims, lbvh_compat=False))
    if rep % 3 == 0:
        big.mul_(1.0001)                       # unrelated work queued in front
    (flux * w).sum().backward()
    g = o.grad
    if ref is None:
        ref = g.clone()
    elif not torch.equal(g, ref):
        bad += 1
        diff = (g != ref) | torch.isnan(g)
        hs = diff.any(dim=2).nonzero().cpu().tolist()
        print("rep", rep, "differs at (h, p):", hs[:12], "n", len(hs), "nan", int(torch.isnan(g).sum()))
print("differing reps", bad, "of 600")
