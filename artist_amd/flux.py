"""The flux epilogue of an optimisation epoch, on MI355X.

Drop-ins for ``artist.flux.bitmap.crop_flux_distributions_around_center`` (artist/flux/bitmap.py:121-246) and
for ``artist.optim.loss.PixelLoss`` / ``KLDivergenceLoss`` (artist/optim/loss.py:251-410): same names,
arguments and return values; the arithmetic runs in ``artist_amd/csrc/flux_kernels.hip`` through
``art_flux_crop_fwd/bwd`` and ``art_flux_loss``.  They consume the ray tracer's ``[H,res_u,res_e]`` bitmaps where
they are, in HBM, and their backward passes are deterministic (the crop's gradient is a gather, not the
atomic scatter of ``grid_sample``'s backward).
"""
from __future__ import annotations

from typing import Any

import torch

from . import _lib
from .ops import _f32c, _require_cuda, _stream


def target_dimensions(solar_tower, target_area_indices: torch.Tensor) -> torch.Tensor:
    """``[B,2]`` (width, height) in metres of each bitmap's target area by GLOBAL index, planar first, cylindrical
    second: planar ``dimensions``, or ``radius * opening_angle`` and ``height`` (bitmap.py:183-216)."""
    tables = []
    from .raytracing import target_area_counts
    n_per_type = target_area_counts(solar_tower)          # (host side: no read of the tower's device tensor)
    if n_per_type[0] > 0:
        tables.append(solar_tower.target_areas[0].dimensions.to(torch.float32))
    if n_per_type[1] > 0:
        cyl = solar_tower.target_areas[1]
        tables.append(torch.stack((cyl.radii.reshape(-1) * cyl.opening_angles.reshape(-1), cyl.heights.reshape(-1)), dim=1)
                      .to(torch.float32))
    table = torch.cat([t.to(target_area_indices.device) for t in tables])
    return table.index_select(0, target_area_indices.long())


class FluxCrop(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flux, dims, crop_width, crop_height):
        dev = _require_cuda(flux, dims)
        flux, dims = _f32c(flux), _f32c(dims)
        if flux.dim() != 3 or dims.shape != (flux.shape[0], 2):
            raise ValueError("flux must be [B,Hh,W] and the target dimensions [B,2]")
        B, Hh, W = flux.shape
        out = torch.empty_like(flux)
        centers = torch.empty((B, 3), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.lib().art_flux_crop_fwd(flux.data_ptr(), dims.data_ptr(), B, Hh, W, float(crop_width),
                                              float(crop_height), out.data_ptr(), centers.data_ptr(), _stream(dev))
        _lib.check(rc, "art_flux_crop_fwd")
        ctx.save_for_backward(flux, dims, centers)
        ctx.crop = (float(crop_width), float(crop_height))
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        flux, dims, centers = ctx.saved_tensors
        dev = flux.device
        B, Hh, W = flux.shape
        grad_out = _f32c(grad_out)
        grad_flux = torch.empty_like(flux)
        workspace = torch.empty((B, 3), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.lib().art_flux_crop_bwd(flux.data_ptr(), dims.data_ptr(), centers.data_ptr(), B, Hh, W, *ctx.crop,
                                              grad_out.data_ptr(), grad_flux.data_ptr(), workspace.data_ptr(),
                                              _stream(dev))
        _lib.check(rc, "art_flux_crop_bwd")
        return grad_flux, None, None, None


def crop_flux_distributions_around_center(flux_distributions: torch.Tensor, solar_tower, target_area_indices: torch.Tensor,
                                          crop_width: float = 6, crop_height: float = 6,
                                          device: torch.device | None = None) -> torch.Tensor:
    """Crop a ``crop_width`` x ``crop_height`` metre region centred on each bitmap's centre of mass, resampled to the
    bitmap's own resolution (bitmap.py:121-246; defaults = ``constants.utis_crop_width/height``)."""
    if device is not None:
        flux_distributions = flux_distributions.to(device)
    dims = target_dimensions(solar_tower, target_area_indices.to(flux_distributions.device))
    return FluxCrop.apply(flux_distributions, dims, crop_width, crop_height)


class _FluxLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, prediction, ground_truth, kind):
        dev = _require_cuda(prediction, ground_truth)
        prediction, ground_truth = _f32c(prediction), _f32c(ground_truth)
        if prediction.shape != ground_truth.shape or prediction.dim() != 3:
            raise ValueError("prediction and ground truth must both be [number_of_samples, res_e, res_u]")
        B = prediction.shape[0]
        npix = prediction.shape[1] * prediction.shape[2]
        loss = torch.empty((B,), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.lib().art_flux_loss(prediction.data_ptr(), ground_truth.data_ptr(), B, npix, kind, loss.data_ptr(),
                                          None, None, _stream(dev))
        _lib.check(rc, "art_flux_loss")
        ctx.save_for_backward(prediction, ground_truth)
        ctx.kind = kind
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_loss):
        prediction, ground_truth = ctx.saved_tensors
        dev = prediction.device
        B = prediction.shape[0]
        npix = prediction.shape[1] * prediction.shape[2]
        grad_loss = _f32c(grad_loss)
        grad_prediction = torch.empty_like(prediction)
        with torch.cuda.device(dev):
            rc = _lib.lib().art_flux_loss(prediction.data_ptr(), ground_truth.data_ptr(), B, npix, ctx.kind, None,
                                          grad_loss.data_ptr(), grad_prediction.data_ptr(), _stream(dev))
        _lib.check(rc, "art_flux_loss")
        return grad_prediction, None, None


class FluxCropPixelLoss(torch.autograd.Function):
    """``PixelLoss()(crop_flux_distributions_around_center(flux, ...), ground_truth, reduction_dimensions=(1, 2))`` fused
    (``art_flux_crop_pixel_loss_fwd/bwd``): the same numbers up to the rounding of the sums (< 1e-6), the same bits for every
    batch size, without the cropped bitmaps' round trip through HBM; when a gradient is wanted the forward pass keeps the
    residual ``crop - ground_truth`` and the backward pass is one kernel over it.  Differentiable w.r.t. ``flux``."""

    calls_with_moments = 0          # (tests: how often the bitmaps came with their centre-of-mass sums)

    @staticmethod
    def forward(ctx, flux, dims, ground_truth, crop_width, crop_height):
        dev = _require_cuda(flux, dims, ground_truth)
        flux, dims, ground_truth = _f32c(flux), _f32c(dims), _f32c(ground_truth)
        if flux.dim() != 3 or dims.shape != (flux.shape[0], 2) or ground_truth.shape != flux.shape:
            raise ValueError("flux and ground truth must be [B,Hh,W] and the target dimensions [B,2]")
        B, Hh, W = flux.shape
        loss = torch.empty((B,), dtype=torch.float32, device=dev)
        centers = torch.empty((B, 4), dtype=torch.float32, device=dev)
        keep = bool(ctx.needs_input_grad[0])
        residual = torch.empty_like(flux) if keep else None
        unit = torch.empty((B, 2), dtype=torch.float32, device=dev) if keep else None
        from .ops import bitmap_moments
        moments = bitmap_moments(flux)              # bitmaps straight from the tracer bring their centre-of-mass sums along
        if moments is not None and (moments.shape[0] != B or moments.device != dev):
            moments = None
        FluxCropPixelLoss.calls_with_moments += moments is not None
        with torch.cuda.device(dev):
            rc = _lib.lib().art_flux_crop_pixel_loss_fwd(flux.data_ptr(), dims.data_ptr(), ground_truth.data_ptr(), B, Hh, W,
                                                         float(crop_width), float(crop_height), loss.data_ptr(), centers.data_ptr(),
                                                         residual.data_ptr() if keep else None, unit.data_ptr() if keep else None,
                                                         None if moments is None else moments.data_ptr(), _stream(dev))
        _lib.check(rc, "art_flux_crop_pixel_loss_fwd")
        if keep:
            ctx.save_for_backward(dims, centers, residual, unit)
        ctx.crop = (float(crop_width), float(crop_height))
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_loss):
        dims, centers, residual, unit = ctx.saved_tensors
        dev = residual.device
        B, Hh, W = residual.shape
        # the gradient of `loss.sum()` arrives as an expanded scalar (stride 0): the kernel reads the one value - no copy to [B]
        stride = 1
        if grad_loss.dtype == torch.float32 and grad_loss.dim() == 1 and grad_loss.stride(0) == 0 and grad_loss.is_cuda:
            stride = 0
        else:
            grad_loss = _f32c(grad_loss)
        grad_flux = torch.empty_like(residual)
        with torch.cuda.device(dev):
            rc = _lib.lib().art_flux_crop_pixel_loss_bwd(dims.data_ptr(), centers.data_ptr(), grad_loss.data_ptr(), stride,
                                                         residual.data_ptr(), unit.data_ptr(), B, Hh, W, *ctx.crop, grad_flux.data_ptr(),
                                                         _stream(dev))
        _lib.check(rc, "art_flux_crop_pixel_loss_bwd")
        return grad_flux, None, None, None, None


def crop_and_pixel_loss(flux_distributions: torch.Tensor, solar_tower, target_area_indices: torch.Tensor,
                        ground_truth: torch.Tensor, crop_width: float = 6, crop_height: float = 6) -> torch.Tensor:
    """Per-sample pixel loss of the flux cropped around its centre of mass against the measured (cropped) flux: the
    epilogue of ``SurfaceReconstructor``'s epoch (surface_reconstructor.py:575-590, 664-676) in one fused pass."""
    dims = target_dimensions(solar_tower, target_area_indices.to(flux_distributions.device))
    return FluxCropPixelLoss.apply(flux_distributions, dims, ground_truth, crop_width, crop_height)


class FluxCropKLLoss(torch.autograd.Function):
    """``KLDivergenceLoss()(crop_flux_distributions_around_center(flux, ...), ground_truth, reduction_dimensions=(1, 2))``
    as one pass per direction (``art_flux_crop_kl_loss_fwd/bwd``): the cropped bitmaps never reach HBM.  Differentiable
    w.r.t. ``flux``."""

    @staticmethod
    def forward(ctx, flux, dims, ground_truth, crop_width, crop_height):
        dev = _require_cuda(flux, dims, ground_truth)
        flux, dims, ground_truth = _f32c(flux), _f32c(dims), _f32c(ground_truth)
        if flux.dim() != 3 or dims.shape != (flux.shape[0], 2) or ground_truth.shape != flux.shape:
            raise ValueError("flux and ground truth must be [B,Hh,W] and the target dimensions [B,2]")
        B, Hh, W = flux.shape
        loss = torch.empty((B,), dtype=torch.float32, device=dev)
        record = torch.empty((B, 8), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.lib().art_flux_crop_kl_loss_fwd(flux.data_ptr(), dims.data_ptr(), ground_truth.data_ptr(), B, Hh, W,
                                                      float(crop_width), float(crop_height), loss.data_ptr(),
                                                      record.data_ptr(), _stream(dev))
        _lib.check(rc, "art_flux_crop_kl_loss_fwd")
        ctx.save_for_backward(flux, dims, ground_truth, record)
        ctx.crop = (float(crop_width), float(crop_height))
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_loss):
        flux, dims, ground_truth, record = ctx.saved_tensors
        dev = flux.device
        B, Hh, W = flux.shape
        grad_loss = _f32c(grad_loss)
        grad_flux = torch.empty_like(flux)
        workspace = torch.empty((B * Hh * W + 5 * B,), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.lib().art_flux_crop_kl_loss_bwd(flux.data_ptr(), dims.data_ptr(), ground_truth.data_ptr(),
                                                      record.data_ptr(), grad_loss.data_ptr(), B, Hh, W, *ctx.crop,
                                                      grad_flux.data_ptr(), workspace.data_ptr(), _stream(dev))
        _lib.check(rc, "art_flux_crop_kl_loss_bwd")
        return grad_flux, None, None, None, None


def crop_and_kl_loss(flux_distributions: torch.Tensor, solar_tower, target_area_indices: torch.Tensor,
                     ground_truth: torch.Tensor, crop_width: float = 6, crop_height: float = 6) -> torch.Tensor:
    """Per-sample KL divergence of the flux cropped around its centre of mass against the measured (cropped) flux
    (artist/flux/bitmap.py:121-246 + artist/optim/loss.py:321-410) in one fused pass."""
    dims = target_dimensions(solar_tower, target_area_indices.to(flux_distributions.device))
    return FluxCropKLLoss.apply(flux_distributions, dims, ground_truth, crop_width, crop_height)


class _CenterOfMass(torch.autograd.Function):
    @staticmethod
    def forward(ctx, bitmaps):
        dev = _require_cuda(bitmaps)
        bitmaps = _f32c(bitmaps)
        if bitmaps.dim() != 3:
            raise ValueError("bitmaps must be [number_of_active_heliostats, bitmap_resolution_u, bitmap_resolution_e]")
        B, Hh, W = bitmaps.shape
        com = torch.empty((B, 3), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.lib().art_flux_center_of_mass(bitmaps.data_ptr(), B, Hh, W, com.data_ptr(), _stream(dev))
        _lib.check(rc, "art_flux_center_of_mass")
        ctx.save_for_backward(com)
        ctx.shape = (B, Hh, W)
        return com[:, :2]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_com):
        (com,) = ctx.saved_tensors
        B, Hh, W = ctx.shape
        dev = com.device
        grad_com = _f32c(grad_com)
        grad = torch.empty((B, Hh, W), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.lib().art_flux_center_of_mass_bwd(com.data_ptr(), grad_com.data_ptr(), B, Hh, W, grad.data_ptr(),
                                                        _stream(dev))
        _lib.check(rc, "art_flux_center_of_mass_bwd")
        return grad


def get_center_of_mass(bitmaps: torch.Tensor, device: torch.device | None = None) -> torch.Tensor:
    """Bitmap coordinates ``[B,2]`` = (e pixel, u pixel) of each bitmap's centre of mass; (0, 0) for an empty bitmap
    (artist/flux/bitmap.py:12-71).  One streaming pass (``art_flux_center_of_mass``); differentiable."""
    if device is not None:
        bitmaps = bitmaps.to(device)
    return _CenterOfMass.apply(bitmaps)


def bitmap_coordinates_to_target_coordinates(bitmap_coordinates: torch.Tensor, bitmap_resolution, solar_tower,
                                             target_area_indices: torch.Tensor, device: torch.device | None = None) -> torch.Tensor:
    """Pixel coordinates (e, u) -> homogeneous world coordinates ``[B,4]`` on the target surface
    (artist/geometry/coordinates.py:119-249): planar areas linearly, cylindrical ones over angle and height; pixel centres,
    e axis flipped.  ``[B]``-sized arithmetic on the host side of the op (torch, differentiable); unlike the reference no
    branch reads a device tensor, so nothing here waits for the device."""
    from .raytracing import target_area_counts
    bc = bitmap_coordinates if device is None else bitmap_coordinates.to(device)
    dev, dt = bc.device, bc.dtype
    width, height = float(bitmap_resolution[0]), float(bitmap_resolution[1])
    e_norm = (bc[:, 0] + 0.5) / width
    u_norm = (bc[:, 1] + 0.5) / height
    n_planar, n_cyl = target_area_counts(solar_tower)
    tix = target_area_indices.to(dev).long()
    out3 = torch.zeros((bc.shape[0], 3), dtype=dt, device=dev)
    if n_planar > 0:
        planar = solar_tower.target_areas[0]
        pi_ = tix.clamp(0, n_planar - 1)
        centers, dims = planar.centers.to(dev, dt)[pi_][:, :3], planar.dimensions.to(dev, dt)[pi_]
        e_local, u_local = (0.5 - e_norm) * dims[:, 0], (0.5 - u_norm) * dims[:, 1]
        on_plane = centers + e_local[:, None] * torch.tensor([1.0, 0.0, 0.0], dtype=dt, device=dev) \
            + u_local[:, None] * torch.tensor([0.0, 0.0, 1.0], dtype=dt, device=dev)
        out3 = torch.where((tix < n_planar)[:, None], on_plane, out3)
    if n_cyl > 0:
        cyl = solar_tower.target_areas[1]
        ci = (tix - n_planar).clamp(0, n_cyl - 1)
        centers, axes, normals = (t.to(dev, dt)[ci][:, :3] for t in (cyl.centers, cyl.axes, cyl.normals))
        radii, heights, opening = (t.to(dev, dt).reshape(-1)[ci] for t in (cyl.radii, cyl.heights, cyl.opening_angles))
        v = torch.cross(axes, normals, dim=-1)
        theta, z = (e_norm - 0.5) * opening, (0.5 - u_norm) * heights
        on_cyl = centers + radii[:, None] * torch.cos(theta)[:, None] * normals + radii[:, None] * torch.sin(theta)[:, None] * v \
            + z[:, None] * axes
        out3 = torch.where((tix >= n_planar)[:, None], on_cyl, out3)
    return torch.cat((out3, torch.ones((bc.shape[0], 1), dtype=dt, device=dev)), dim=1)


class FocalSpotLoss:
    """Distance between the predicted and the ground-truth focal spot on the target area, per sample
    (artist/optim/loss.py:124-250): centres of mass of both bitmaps (``get_center_of_mass``) mapped to world coordinates."""

    def __init__(self, scenario) -> None:
        self.loss_function = None
        self.scenario = scenario

    def __call__(self, prediction: torch.Tensor, ground_truth: torch.Tensor, **kwargs: Any) -> torch.Tensor:
        expected_kwargs = ["device", "target_area_indices"]
        errors = [f"Please add '{key}' as keyword argument." for key in expected_kwargs if key not in kwargs]
        if errors:                                   # same message as artist/optim/loss.py:187-196
            raise ValueError(f"The focal spot loss expects {expected_kwargs} as keyword arguments. " + " ".join(errors))
        tower = self.scenario.solar_tower
        resolution = (prediction.shape[2], prediction.shape[1])             # (width, height), loss.py:212-216
        spots = [bitmap_coordinates_to_target_coordinates(get_center_of_mass(b), resolution, tower, kwargs["target_area_indices"])
                 for b in (prediction, ground_truth)]
        return torch.norm(spots[0][:, :3] - spots[1][:, :3], dim=1)


def _check_reduction(kwargs: dict, what: str) -> None:
    if "reduction_dimensions" not in kwargs:          # same messages as artist/optim/loss.py:300-311, 376-383
        if what == "pixel":
            raise ValueError("The vector loss expects ['reduction_dimensions'] as keyword arguments. "
                             "Please add 'reduction_dimensions' as keyword argument.")
        raise ValueError("The KL-divergence loss expects 'reduction_dimensions' as keyword argument. "
                         "Please add this argument.")
    if tuple(int(d) for d in kwargs["reduction_dimensions"]) not in ((1, 2), (-2, -1)):
        raise NotImplementedError("the fused flux losses reduce over the two bitmap dimensions (1, 2) - the only "
                                  "reduction ARTIST's optimisers use")


class PixelLoss:
    """``sum (prediction - ground_truth)^2 / sum ground_truth`` per sample (artist/optim/loss.py:251-318)."""

    def __call__(self, prediction: torch.Tensor, ground_truth: torch.Tensor, **kwargs: Any) -> torch.Tensor:
        _check_reduction(kwargs, "pixel")
        return _FluxLoss.apply(prediction, ground_truth, 0)


class KLDivergenceLoss:
    """``D_KL(ground truth || prediction)`` of the L1-normalised bitmaps per sample (artist/optim/loss.py:321-410)."""

    def __call__(self, prediction: torch.Tensor, ground_truth: torch.Tensor, **kwargs: Any) -> torch.Tensor:
        _check_reduction(kwargs, "kl")
        return _FluxLoss.apply(prediction, ground_truth, 1)
