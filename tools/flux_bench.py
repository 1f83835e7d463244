#!/usr/bin/env python3
"""Flux epilogue at the metric size: 1000 bitmaps of 256 x 256 (crop forward/backward, both losses), with the
algorithmic HBM bytes of each kernel (one read + one write of the bitmaps, two reads for kernels with two inputs)."""
import json, sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
from artist_amd import _lib

dev = torch.device("cuda:0")
B, Hh, W = int(sys.argv[1]) if len(sys.argv) > 1 else 1000, 256, 256
nbytes = B * Hh * W * 4


def timed(fn, steps=20):
    for _ in range(3):
        fn()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(steps):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) / steps


g = torch.Generator(device=dev).manual_seed(1)
yy, xx = torch.meshgrid(torch.arange(Hh, device=dev), torch.arange(W, device=dev), indexing="ij")
cx = 128 + 40 * torch.rand(B, 1, 1, device=dev, generator=g) - 20
cy = 128 + 40 * torch.rand(B, 1, 1, device=dev, generator=g) - 20
flux = torch.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * 18.0 ** 2)).contiguous()
dims = torch.full((B, 2), 8.0, device=dev)
out, com = torch.empty_like(flux), torch.empty(B, 3, device=dev)
gout, gflux, ws = torch.rand_like(flux), torch.empty_like(flux), torch.empty(B, 3, device=dev)
truth, loss, gl, gp = torch.rand_like(flux) + 0.1, torch.empty(B, device=dev), torch.ones(B, device=dev), torch.empty_like(flux)
lib, s = _lib.lib(), torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr()
cases = {
    "crop_fwd": (lambda: lib.art_flux_crop_fwd(p(flux), p(dims), B, Hh, W, 6.0, 6.0, p(out), p(com), s), 2 * nbytes),
    "crop_bwd": (lambda: lib.art_flux_crop_bwd(p(flux), p(dims), p(com), B, Hh, W, 6.0, 6.0, p(gout), p(gflux), p(ws), s), 3 * nbytes),
    "pixel_loss_fwd": (lambda: lib.art_flux_loss(p(out), p(truth), B, Hh * W, 0, p(loss), None, None, s), 2 * nbytes),
    "pixel_loss_bwd": (lambda: lib.art_flux_loss(p(out), p(truth), B, Hh * W, 0, None, p(gl), p(gp), s), 3 * nbytes),
    "kl_loss_fwd": (lambda: lib.art_flux_loss(p(out), p(truth), B, Hh * W, 1, p(loss), None, None, s), 2 * nbytes),
    "kl_loss_bwd": (lambda: lib.art_flux_loss(p(out), p(truth), B, Hh * W, 1, None, p(gl), p(gp), s), 3 * nbytes),
}
c4 = torch.empty(B, 4, device=dev)
resid, unit = torch.empty_like(flux), torch.empty(B, 2, device=dev)
cases["crop_pixel_loss_fwd"] = (lambda: lib.art_flux_crop_pixel_loss_fwd(p(flux), p(dims), p(truth), B, Hh, W, 6.0, 6.0, p(loss), p(c4), None, None, None, s),
                                2 * nbytes)
cases["crop_pixel_loss_fwd_keep"] = (lambda: lib.art_flux_crop_pixel_loss_fwd(p(flux), p(dims), p(truth), B, Hh, W, 6.0, 6.0, p(loss), p(c4),
                                                                              p(resid), p(unit), None, s), 3 * nbytes)
cases["crop_pixel_loss_bwd"] = (lambda: lib.art_flux_crop_pixel_loss_bwd(p(dims), p(c4), p(gl), 1, p(resid), p(unit), B, Hh, W, 6.0, 6.0,
                                                                         p(gflux), s), 2 * nbytes)
# the same pass with the centre-of-mass sums handed over (as the trace's conversion pass leaves them: include/artist_hip.h `moments`)
lin = lambda k: torch.linspace(-1, 1, k, device=dev, dtype=torch.float64)
f64 = flux.double()
parts = [f64[:, (Hh * v) // 4:(Hh * (v + 1)) // 4] for v in range(4)]
ysl = [lin(Hh)[(Hh * v) // 4:(Hh * (v + 1)) // 4] for v in range(4)]
mom = torch.stack([torch.stack([q.sum((1, 2)), (q * lin(W)[None, None, :]).sum((1, 2)), (q * y[None, :, None]).sum((1, 2))], 1) for q, y in zip(parts, ysl)], 1).contiguous()
cases["crop_pixel_loss_fwd_keep_moments"] = (lambda: lib.art_flux_crop_pixel_loss_fwd(p(flux), p(dims), p(truth), B, Hh, W, 6.0, 6.0, p(loss), p(c4),
                                                                                      p(resid), p(unit), p(mom), s), 3 * nbytes)
res = {}
for name, (fn, alg) in cases.items():
    ms = timed(fn)
    res[name] = {"ms": round(ms, 4), "algorithmic_GBps": round(alg / ms / 1e6, 1), "frac_of_8TBps": round(alg / ms / 1e6 / 8000, 3)}
print(json.dumps({"bitmaps": B, "resolution": [Hh, W], **res}))
