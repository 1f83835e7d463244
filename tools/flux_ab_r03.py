import ctypes, json, sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
from artist_amd import _lib
dev = torch.device("cuda:0")
B, Hh, W = int(sys.argv[1]) if len(sys.argv) > 1 else 1000, 256, 256
new = _lib.lib()
old = ctypes.CDLL(str(pathlib.Path(__file__).resolve().parent / "bin" / "libflux_r03.so"))
vp, i64, dbl = ctypes.c_void_p, ctypes.c_int64, ctypes.c_double
old.art_flux_crop_pixel_loss_fwd.argtypes = [vp, vp, vp, i64, i64, i64, dbl, dbl, vp, vp, vp]
old.art_flux_crop_pixel_loss_bwd.argtypes = [vp, vp, vp, vp, vp, i64, i64, i64, dbl, dbl, vp, vp, vp]
def timed(fn, steps=20):
    for _ in range(3): fn()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); st.record()
    for _ in range(steps): fn()
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / steps
g = torch.Generator(device=dev).manual_seed(1)
yy, xx = torch.meshgrid(torch.arange(Hh, device=dev), torch.arange(W, device=dev), indexing="ij")
cx = 128 + 40 * torch.rand(B, 1, 1, device=dev, generator=g) - 20
cy = 128 + 40 * torch.rand(B, 1, 1, device=dev, generator=g) - 20
flux = torch.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * 18.0 ** 2)).contiguous()
dims = torch.full((B, 2), 8.0, device=dev)
truth, loss, gl = torch.rand_like(flux) + 0.1, torch.empty(B, device=dev), torch.ones(B, device=dev)
c4, gflux = torch.empty(B, 4, device=dev), torch.empty_like(flux)
ws2 = torch.empty(B * Hh * W + 5 * B, device=dev)
resid, unit = torch.empty_like(flux), torch.empty(B, 2, device=dev)
s = torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr()
print("r03 fwd", timed(lambda: old.art_flux_crop_pixel_loss_fwd(p(flux), p(dims), p(truth), B, Hh, W, 6.0, 6.0, p(loss), p(c4), s)))
print("r03 bwd", timed(lambda: old.art_flux_crop_pixel_loss_bwd(p(flux), p(dims), p(truth), p(c4), p(gl), B, Hh, W, 6.0, 6.0, p(gflux), p(ws2), s)))
print("new fwd", timed(lambda: new.art_flux_crop_pixel_loss_fwd(p(flux), p(dims), p(truth), B, Hh, W, 6.0, 6.0, p(loss), p(c4), None, None, None, s)))
print("new fwd keep", timed(lambda: new.art_flux_crop_pixel_loss_fwd(p(flux), p(dims), p(truth), B, Hh, W, 6.0, 6.0, p(loss), p(c4), p(resid), p(unit), None, s)))
print("new bwd", timed(lambda: new.art_flux_crop_pixel_loss_bwd(p(dims), p(c4), p(gl), 1, p(resid), p(unit), B, Hh, W, 6.0, 6.0, p(gflux), s)))
