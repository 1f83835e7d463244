import ctypes, sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
dev = torch.device("cuda:0")
B, Hh, W = 1000, 256, 256
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
truth, loss = torch.rand_like(flux) + 0.1, torch.empty(B, device=dev)
c4 = torch.empty(B, 4, device=dev)
s = torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr()
vp, i64, dbl = ctypes.c_void_p, ctypes.c_int64, ctypes.c_double
for name in ("", "_nocom", "_nocrop"):
    old = ctypes.CDLL(str(pathlib.Path(__file__).resolve().parent / "bin" / f"libflux_r03{name}.so"))
    old.art_flux_crop_pixel_loss_fwd.argtypes = [vp, vp, vp, i64, i64, i64, dbl, dbl, vp, vp, vp]
    print("r03" + name, timed(lambda: old.art_flux_crop_pixel_loss_fwd(p(flux), p(dims), p(truth), B, Hh, W, 6.0, 6.0, p(loss), p(c4), s)))
