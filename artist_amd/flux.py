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
    """``PixelLoss()(crop_flux_distributions_around_center(flux, ...), ground_truth, reduction_dimensions=(1, 2))`` as one
    pass per direction (``art_flux_crop_pixel_loss_fwd/bwd``): the same numbers, bit for bit, without the cropped
    bitmaps' round trip through HBM.  Differentiable w.r.t. ``flux``."""

    @staticmethod
    def forward(ctx, flux, dims, ground_truth, crop_width, crop_height):
        dev = _require_cuda(flux, dims, ground_truth)
        flux, dims, ground_truth = _f32c(flux), _f32c(dims), _f32c(ground_truth)
        if flux.dim() != 3 or dims.shape != (flux.shape[0], 2) or ground_truth.shape != flux.shape:
            raise ValueError("flux and ground truth must be [B,Hh,W] and the target dimensions [B,2]")
        B, Hh, W = flux.shape
        loss = torch.empty((B,), dtype=torch.float32, device=dev)
        centers = torch.empty((B, 4), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.lib().art_flux_crop_pixel_loss_fwd(flux.data_ptr(), dims.data_ptr(), ground_truth.data_ptr(), B, Hh, W,
                                                         float(crop_width), float(crop_height), loss.data_ptr(),
                                                         centers.data_ptr(), _stream(dev))
        _lib.check(rc, "art_flux_crop_pixel_loss_fwd")
        ctx.save_for_backward(flux, dims, ground_truth, centers)
        ctx.crop = (float(crop_width), float(crop_height))
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_loss):
        flux, dims, ground_truth, centers = ctx.saved_tensors
        dev = flux.device
        B, Hh, W = flux.shape
        grad_loss = _f32c(grad_loss)
        grad_flux = torch.empty_like(flux)
        workspace = torch.empty((B * Hh * W + 5 * B,), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.lib().art_flux_crop_pixel_loss_bwd(flux.data_ptr(), dims.data_ptr(), ground_truth.data_ptr(),
                                                         centers.data_ptr(), grad_loss.data_ptr(), B, Hh, W, *ctx.crop,
                                                         grad_flux.data_ptr(), workspace.data_ptr(), _stream(dev))
        _lib.check(rc, "art_flux_crop_pixel_loss_bwd")
        return grad_flux, None, None, None, None


def crop_and_pixel_loss(flux_distributions: torch.Tensor, solar_tower, target_area_indices: torch.Tensor,
                        ground_truth: torch.Tensor, crop_width: float = 6, crop_height: float = 6) -> torch.Tensor:
    """Per-sample pixel loss of the flux cropped around its centre of mass against the measured (cropped) flux: the
    epilogue of ``SurfaceReconstructor``'s epoch (surface_reconstructor.py:575-590, 664-676) in one fused pass."""
    dims = target_dimensions(solar_tower, target_area_indices.to(flux_distributions.device))
    return FluxCropPixelLoss.apply(flux_distributions, dims, ground_truth, crop_width, crop_height)


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
