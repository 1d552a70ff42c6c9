"""dn-splatter's per-pixel training loss as two HIP launches.

``dn_loss_fused`` computes what ``torch_losses.dn_loss`` (the PyTorch restatement of
``DNSplatterModel.get_loss_dict``, dn_splatter/dn_model.py:614-729) computes — same value, same gradients w.r.t. the
rendered rgb / depth / normal images — but with the cotangents produced directly by ``dnsplat_dn_loss`` instead of by
autograd over ~120 torch kernels; the per-Gaussian min-scale term is one more launch (``dnsplat_scale_reg``).
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional

import torch
from torch import Tensor

from . import _lib
from ._lib import DnLossArgs
from ._ops import _f32c, _ptr, _stream


def depth_counts(gt_depth: Tensor, depth_tolerance: float = 0.1) -> Tensor:
    """Normalisers of the two EdgeAwareLogL1 means (losses.py:216-222): valid pixels in columns < W-1 and rows < H-1.
    Depends on the batch only — compute once per image, no host sync."""
    valid = gt_depth.reshape(gt_depth.shape[0], gt_depth.shape[1]) > depth_tolerance
    return torch.stack([valid[:, :-1].sum(), valid[:-1, :].sum()]).to(torch.float32)


class _DnLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb, depth, normal, image, gt_depth, gt_normal, counts, ssim_lambda, depth_lambda, depth_tolerance):
        rgb = _f32c(rgb, "rgb"); depth = _f32c(depth, "depth"); normal = _f32c(normal, "normal")
        image = _f32c(image, "image")
        H, W = rgb.shape[0], rgb.shape[1]
        dev = rgb.device
        f32 = dict(dtype=torch.float32, device=dev)
        maps = torch.empty(9 * H * W + 512, **f32)      # dnsplat_dn_loss_args.maps: scratch
        v_rgb = torch.empty(H, W, 3, **f32)
        v_depth = torch.empty(depth.shape, **f32)
        v_normal = torch.empty(H, W, 3, **f32)
        sums = torch.empty(8, **f32)
        a = DnLossArgs()
        a.width, a.height = W, H
        a.rgb, a.depth, a.normal, a.gt_rgb = _ptr(rgb), _ptr(depth), _ptr(normal), _ptr(image)
        if gt_depth is not None:
            gt_depth = _f32c(gt_depth, "mono_depth")
            counts = _f32c(counts, "depth_counts")
        if gt_normal is not None:
            gt_normal = _f32c(gt_normal, "normal gt")
        a.gt_depth, a.gt_normal, a.depth_counts = _ptr(gt_depth), _ptr(gt_normal), _ptr(counts)
        a.ssim_lambda, a.depth_weight, a.depth_tolerance = ssim_lambda, 1.0 + depth_lambda, depth_tolerance
        a.maps, a.v_rgb, a.v_depth, a.v_normal, a.sums = _ptr(maps), _ptr(v_rgb), _ptr(v_depth), _ptr(v_normal), _ptr(sums)
        _lib.run("dnsplat_dn_loss", _lib.lib().dnsplat_dn_loss, ctypes.byref(a), _stream())
        P = float(W * H)
        M = 3.0 * (W - 10) * (H - 10)
        loss = (1 - ssim_lambda) * sums[1] / (3 * P) + ssim_lambda * (1 - sums[0] / M)
        if gt_depth is not None:
            loss = loss + (1.0 + depth_lambda) * (sums[2] / counts[0] + sums[3] / counts[1])
        if gt_normal is not None:
            loss = loss + sums[4] / (3 * P) + sums[5] / (3.0 * H * (W - 1)) + sums[6] / (3.0 * (H - 1) * W)
        ctx.save_for_backward(v_rgb, v_depth, v_normal)
        return loss

    @staticmethod
    def backward(ctx, g):
        v_rgb, v_depth, v_normal = ctx.saved_tensors
        return (v_rgb * g, v_depth * g, v_normal * g) + (None,) * 7


class _SsimFn(torch.autograd.Function):
    """Mean SSIM of two [H,W,3] images and its gradient w.r.t. the first in two launches (``dnsplat_ssim``)."""

    @staticmethod
    def forward(ctx, x, y):
        x = _f32c(x, "ssim x"); y = _f32c(y, "ssim y")
        if x.dim() != 3 or x.shape[2] != 3 or x.shape != y.shape:
            raise ValueError(f"dnsplat_ssim takes two [H,W,3] images, got {tuple(x.shape)} and {tuple(y.shape)}")
        H, W = x.shape[0], x.shape[1]
        f32 = dict(dtype=torch.float32, device=x.device)
        maps = torch.empty(9 * H * W + 512, **f32)
        need_grad = ctx.needs_input_grad[0]
        v_x = torch.empty(H, W, 3, **f32) if need_grad else None
        sums = torch.empty(8, **f32)
        _lib.run("dnsplat_ssim", _lib.lib().dnsplat_ssim, W, H, _ptr(x), _ptr(y), _ptr(maps), _ptr(v_x), _ptr(sums), _stream())
        if need_grad:
            ctx.save_for_backward(v_x)
        return sums[0] / (3.0 * (W - 10) * (H - 10))

    @staticmethod
    def backward(ctx, g):
        (v_x,) = ctx.saved_tensors
        return v_x * g, None


def ssim_hip(pred: Tensor, gt: Tensor) -> Tensor:
    """Mean SSIM of two [H,W,3] images in [0,1] (pytorch_msssim's definition: 11-tap Gaussian, sigma 1.5, valid padding), with the
    gradient w.r.t. ``pred`` — ``torch_losses.ssim`` as ONE autograd node on two HIP launches instead of ~45 torch kernels."""
    return _SsimFn.apply(pred, gt)


class SSIM(torch.nn.Module):
    """Stand-in for the module ``DNSplatterModel`` holds as ``self.ssim`` — ``torchmetrics.StructuralSimilarityIndexMeasure(
    data_range=1.0, kernel_size=11)`` (dn_model.py:180; nerfstudio's own splatfacto holds ``pytorch_msssim.SSIM(data_range=1.0,
    size_average=True, channel=3)``: the same 11-tap sigma-1.5 Gaussian statistics averaged over the (W-10)(H-10) windows that do
    not touch the border — torchmetrics reflect-pads and crops the padded rim away again) — as the inherited RGB term calls it:
    ``1 - self.ssim(gt.permute(2,0,1)[None], pred.permute(2,0,1)[None])`` (two [1,3,H,W] views of [H,W,3] images: read in place;
    any other layout is copied), and as the evaluation calls it (dn_model.py:855).  SSIM is symmetric in its arguments; the gradient
    goes to whichever of the two requires it (the rendered image).  Not restated: recent torchmetrics clamp the two variances at 0
    before the ratio (a decision on fp32 rounding noise of a quantity that is >= 0 in exact arithmetic).  ``install_ssim(model)``
    puts it in place."""

    def __init__(self, data_range: float = 1.0, size_average: bool = True, channel: int = 3, kernel_size: int = 11, sigma: float = 1.5):
        super().__init__()
        if data_range != 1.0 or not size_average or channel != 3 or kernel_size != 11 or sigma != 1.5:
            raise NotImplementedError("dnsplat SSIM: data_range=1.0, kernel_size=11, sigma=1.5, mean over 3 channels (the module "
                                      "dn-splatter / splatfacto construct)")

    def forward(self, X: Tensor, Y: Tensor) -> Tensor:
        if X.dim() != 4 or X.shape[0] != 1 or X.shape[1] != 3 or X.shape != Y.shape:
            raise NotImplementedError(f"dnsplat SSIM takes two [1,3,H,W] images, got {tuple(X.shape)} and {tuple(Y.shape)}")
        x, y = X[0].permute(1, 2, 0), Y[0].permute(1, 2, 0)
        if y.requires_grad and not x.requires_grad:
            x, y = y, x
        elif y.requires_grad:
            raise NotImplementedError("dnsplat SSIM differentiates one argument (the rendered image)")
        return _SsimFn.apply(x, y)


class _EdgeAwareLogL1Fn(torch.autograd.Function):
    """EdgeAwareLogL1 ("scalar", losses.py:187-224): value and gradient w.r.t. the prediction in one pass (``dnsplat_edge_aware_logl1``)."""

    @staticmethod
    def forward(ctx, pred, gt, rgb, mask):
        shape = pred.shape
        H, W = shape[0], shape[1]
        p2 = _f32c(pred.reshape(H, W), "pred"); g2 = _f32c(gt.reshape(H, W).float(), "gt"); rgb = _f32c(rgb, "rgb")
        if rgb.shape != (H, W, 3):
            raise ValueError(f"EdgeAwareLogL1: rgb must be [H,W,3] for a [{H},{W}] depth, got {tuple(rgb.shape)}")
        m = None
        if mask is not None:
            if mask.dtype != torch.bool or mask.numel() != H * W:
                raise ValueError("EdgeAwareLogL1: mask must be a bool tensor of the depth's shape")
            m = mask.reshape(H, W).contiguous()
        f32 = dict(dtype=torch.float32, device=p2.device)
        need = ctx.needs_input_grad[0]
        v_x = torch.empty(H, W, **f32) if need else None
        v_y = torch.empty(H, W, **f32) if need else None
        scratch = torch.empty(512, **f32)
        sums = torch.empty(8, **f32)
        _lib.run("dnsplat_edge_aware_logl1", _lib.lib().dnsplat_edge_aware_logl1, W, H, _ptr(p2), _ptr(g2), _ptr(rgb), _ptr(m),
                 _ptr(v_x), _ptr(v_y), _ptr(scratch), _ptr(sums), _stream())
        if need:
            ctx.save_for_backward(v_x, v_y, sums)
            ctx.shape = shape
        return sums[0] / sums[2] + sums[1] / sums[3]        # mean over the counted pixels of each term (an empty mask: nan, as the reference)

    @staticmethod
    def backward(ctx, g):
        v_x, v_y, sums = ctx.saved_tensors
        return (v_x * (g / sums[2]) + v_y * (g / sums[3])).reshape(ctx.shape), None, None, None


class EdgeAwareLogL1(torch.nn.Module):
    """Drop-in for ``dn_splatter.losses.EdgeAwareLogL1(implementation="scalar")`` (losses.py:187-224) as
    ``DNRegularization.get_depth_loss`` calls it (regularization_strategy.py:162-170): ``loss(pred_depth, gt_depth, gt_img, valid_mask)``
    with [H,W,1] depths, an [H,W,3] image and an [H,W,1] bool mask (or None).  One launch instead of ~20 torch kernels and — what costs
    more in an eager step — instead of the two boolean-mask gathers ``loss_x[mask]``, whose data-dependent size is a host
    synchronisation each.  Gradient w.r.t. ``pred`` only (the reference differentiates nothing else here)."""

    def __init__(self, implementation: str = "scalar", **kwargs):
        super().__init__()
        if implementation != "scalar":
            raise NotImplementedError('dnsplat EdgeAwareLogL1: implementation="scalar" (what DNRegularization constructs)')
        self.implementation = implementation

    def forward(self, pred: Tensor, gt: Tensor, rgb: Tensor, mask: Optional[Tensor]) -> Tensor:
        if pred.dim() not in (2, 3) or (pred.dim() == 3 and pred.shape[2] != 1):
            raise NotImplementedError(f"dnsplat EdgeAwareLogL1 takes one [H,W,1] depth image, got {tuple(pred.shape)}")
        if gt.requires_grad or rgb.requires_grad:
            raise NotImplementedError("dnsplat EdgeAwareLogL1 differentiates the prediction only")
        return _EdgeAwareLogL1Fn.apply(pred, gt, rgb, mask)


class _TVLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred):
        pred = _f32c(pred, "pred")
        H, W, C = pred.shape
        f32 = dict(dtype=torch.float32, device=pred.device)
        v = torch.empty_like(pred) if ctx.needs_input_grad[0] else None
        scratch = torch.empty(512, **f32)
        sums = torch.empty(8, **f32)
        _lib.run("dnsplat_tv_loss", _lib.lib().dnsplat_tv_loss, W, H, C, _ptr(pred), _ptr(v), _ptr(scratch), _ptr(sums), _stream())
        if v is not None:
            ctx.save_for_backward(v)
        return sums[0] / float(C * H * (W - 1)) + sums[1] / float(C * (H - 1) * W)

    @staticmethod
    def backward(ctx, g):
        (v,) = ctx.saved_tensors
        return v * g


class TVLoss(torch.nn.Module):
    """Drop-in for ``dn_splatter.losses.TVLoss`` (losses.py:279-295) on one [H,W,C] image, as ``NormalLoss(Smooth)`` applies it to
    the rendered normals (regularization_strategy.py:188-193): one launch for value and gradient instead of ~14 torch kernels."""

    def forward(self, pred: Tensor) -> Tensor:
        if pred.dim() != 3:
            raise NotImplementedError(f"dnsplat TVLoss takes one [H,W,C] image, got {tuple(pred.shape)}")
        return _TVLossFn.apply(pred)


class _ScaleRegFn(torch.autograd.Function):
    """mean_g min_k exp(scales[g, k]) (regularization_strategy.py:195-199) and its gradient in one launch (dnsplat_scale_reg)."""

    @staticmethod
    def forward(ctx, scales):
        scales = _f32c(scales, "scales")
        N = scales.shape[0]
        v = torch.empty_like(scales)
        total = torch.zeros((), dtype=torch.float32, device=scales.device)
        _lib.run("dnsplat_scale_reg", _lib.lib().dnsplat_scale_reg, N, _ptr(scales), 1.0 / max(N, 1), _ptr(v), _ptr(total), _stream())
        ctx.save_for_backward(v)
        return total

    @staticmethod
    def backward(ctx, g):
        (v,) = ctx.saved_tensors
        return v * g


def scale_reg(scales: Tensor) -> Tensor:
    """``torch.min(torch.exp(scales), dim=1, keepdim=True)[0].mean()`` (``DNRegularization.get_scale_loss``,
    regularization_strategy.py:195-199) and its gradient in one launch (``dnsplat_scale_reg``) instead of ten kernels over [N,3]."""
    return _ScaleRegFn.apply(scales)


def dn_loss_fused(outputs: Dict[str, Tensor], batch: Dict[str, Tensor], scales: Tensor, ssim_lambda: float = 0.2,
                  depth_lambda: float = 0.2, depth_tolerance: float = 0.1, counts: Optional[Tensor] = None) -> Tensor:
    """Drop-in for ``torch_losses.dn_loss`` (mono depth + mono normal supervision)."""
    gt_depth = batch.get("mono_depth")
    if gt_depth is not None and counts is None:
        counts = depth_counts(gt_depth, depth_tolerance)
    per_pixel = _DnLossFn.apply(outputs["rgb"], outputs["depth"], outputs["normal"], batch["image"], gt_depth,
                                batch.get("normal"), counts, ssim_lambda, depth_lambda, depth_tolerance)
    return per_pixel + _ScaleRegFn.apply(scales)                                      # regularization_strategy.py:195-199
