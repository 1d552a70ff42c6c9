"""The per-pixel losses of dn-splatter, kept in PyTorch-ROCm (BASELINE.json north star: "the per-pixel
depth/normal losses ... in dn_splatter/losses.py stay in PyTorch-ROCm").

This is the host-side restatement that BASELINE config C5 ("depth + mono-normal loss enabled") times together
with the renderer; nothing here is a kernel of ours.  It follows

* ``DNSplatterModel.get_loss_dict``                      dn_splatter/dn_model.py:614-729
* ``DNRegularization.get_loss`` / depth / normal / scale dn_splatter/regularization_strategy.py:146-199
* ``EdgeAwareLogL1``, ``LogL1``, ``L1``, ``TVLoss``      dn_splatter/losses.py:154-224, 279-295
* the inherited RGB term of nerfstudio's ``SplatfactoModel.get_loss_dict``: ``(1 - l) * L1 + l * (1 - SSIM)`` with
  ``l = ssim_lambda = 0.2``; ``self.ssim`` is, in dn-splatter, torchmetrics' ``StructuralSimilarityIndexMeasure(data_range=1.0,
  kernel_size=11)`` (dn_model.py:180; nerfstudio itself holds pytorch_msssim's SSIM): an 11 x 11 Gaussian window, sigma 1.5, averaged
  over the windows that do not touch the border (pytorch_msssim: valid convolution; torchmetrics: reflect-pad, then crop the rim) —
  neither package is vendored in the reference, so that term is restated from their published definitions.
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch
import torch.nn.functional as F
from torch import Tensor


def _gaussian_window(size: int = 11, sigma: float = 1.5, device=None) -> Tensor:
    x = torch.arange(size, dtype=torch.float32, device=device) - size // 2
    g = torch.exp(-(x ** 2) / (2 * sigma ** 2))
    return g / g.sum()


def ssim(pred: Tensor, gt: Tensor, data_range: float = 1.0) -> Tensor:
    """Mean SSIM of two [H,W,3] images (separable 11-tap Gaussian, valid padding)."""
    x = pred.permute(2, 0, 1)[None]
    y = gt.permute(2, 0, 1)[None]
    C = x.shape[1]
    w = _gaussian_window(device=x.device).to(x.dtype)
    wh = w.view(1, 1, -1, 1).repeat(C, 1, 1, 1)
    ww = w.view(1, 1, 1, -1).repeat(C, 1, 1, 1)

    def blur(t):
        return F.conv2d(F.conv2d(t, wh, groups=C), ww, groups=C)

    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    mu_x, mu_y = blur(x), blur(y)
    sxx = blur(x * x) - mu_x * mu_x
    syy = blur(y * y) - mu_y * mu_y
    sxy = blur(x * y) - mu_x * mu_y
    cs = (2 * sxy + c2) / (sxx + syy + c2)
    s = ((2 * mu_x * mu_y + c1) / (mu_x * mu_x + mu_y * mu_y + c1)) * cs
    return s.mean()


_TOEPLITZ: Dict = {}
SSIM_GEMM_BLOCK = int(os.environ.get("DNSPLAT_SSIM_GEMM_BLOCK", "64"))
SSIM_FAST_IMPL = os.environ.get("DNSPLAT_SSIM_IMPL", "conv")      # what dn_loss(capturable=True) uses: "conv" (ten grouped conv2d calls, the fastest measured) | "gemm" | "stacked"


def _toeplitz(block: int, device, size: int = 11, sigma: float = 1.5) -> Tensor:
    """[block + size - 1, block] banded matrix: column j holds the Gaussian window at rows j .. j + size - 1, so that a block of
    ``block + size - 1`` consecutive samples times it is the VALID 11-tap blur of that block (``block`` outputs)."""
    key = (block, str(device), size, sigma)
    T = _TOEPLITZ.get(key)
    if T is None:
        w = _gaussian_window(size, sigma)
        T = torch.zeros(block + size - 1, block)
        for j in range(block):
            T[j:j + size, j] = w
        T = T.to(device)
        _TOEPLITZ[key] = T
    return T


def _blur_valid_gemm(t: Tensor, block: int = 64, size: int = 11) -> Tensor:
    """Separable 11-tap Gaussian blur (valid padding) of the last two dims of ``t`` [..., H, W] as two block-Toeplitz GEMMs: the
    windows of ``block + 10`` samples at a stride of ``block`` are a view (``unfold``), the taps a [block + 10, block] banded matrix.
    Same sums as conv2d with the separable window, in hipBLASLt's order.  Why not conv2d: MIOpen runs the depthwise 11 x 1 / 1 x 11
    convolutions of a 1600 x 1200 image through general-purpose convolution kernels at ~250-500 us each, thirty of them per step
    (profiles/r06_c5_torch_loss_kernel_stats.txt): 5.5 ms, more than the whole render step."""
    T = _toeplitz(block, t.device, size)

    def along_last(x):
        n = x.shape[-1]
        out = n - size + 1
        nb = -(-out // block)
        need = nb * block + size - 1
        if need > n:
            x = F.pad(x, (0, need - n))
        y = x.unfold(-1, block + size - 1, block) @ T                  # [..., nb, block]
        return y.reshape(*x.shape[:-1], nb * block)[..., :out]

    t = along_last(t)                                                   # along W
    return along_last(t.transpose(-1, -2)).transpose(-1, -2)            # along H


def ssim_stacked(pred: Tensor, gt: Tensor, data_range: float = 1.0) -> Tensor:
    """``ssim`` with the five maps (x, y, x^2, y^2, xy) stacked into ONE [1, 15, H, W] tensor: two grouped conv2d calls (and two in the
    backward) instead of ten (and six)."""
    x = pred.permute(2, 0, 1)
    y = gt.permute(2, 0, 1)
    C = x.shape[0]
    t = torch.cat([x, y, x * x, y * y, x * y], 0)[None]
    w = _gaussian_window(device=x.device)
    wh = w.view(1, 1, -1, 1).repeat(5 * C, 1, 1, 1)
    ww = w.view(1, 1, 1, -1).repeat(5 * C, 1, 1, 1)
    b = F.conv2d(F.conv2d(t, wh, groups=5 * C), ww, groups=5 * C)[0]
    mu_x, mu_y, bxx, byy, bxy = b[:C], b[C:2 * C], b[2 * C:3 * C], b[3 * C:4 * C], b[4 * C:]
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    cs = (2 * (bxy - mu_x * mu_y) + c2) / ((bxx - mu_x * mu_x) + (byy - mu_y * mu_y) + c2)
    s = ((2 * mu_x * mu_y + c1) / (mu_x * mu_x + mu_y * mu_y + c1)) * cs
    return s.mean()


def ssim_gemm(pred: Tensor, gt: Tensor, data_range: float = 1.0) -> Tensor:
    """``ssim`` with the five blurs (x, y, x^2, y^2, xy) stacked into one [15, H, W] tensor and run as two GEMMs (see
    ``_blur_valid_gemm``) instead of ten grouped conv2d calls: same value and gradient up to fp32 summation order, 5x fewer kernels."""
    x = pred.permute(2, 0, 1)
    y = gt.permute(2, 0, 1)
    C = x.shape[0]
    b = _blur_valid_gemm(torch.cat([x, y, x * x, y * y, x * y], 0), block=SSIM_GEMM_BLOCK)
    mu_x, mu_y, bxx, byy, bxy = b[:C], b[C:2 * C], b[2 * C:3 * C], b[3 * C:4 * C], b[4 * C:]
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    sxx = bxx - mu_x * mu_x
    syy = byy - mu_y * mu_y
    sxy = bxy - mu_x * mu_y
    cs = (2 * sxy + c2) / (sxx + syy + c2)
    s = ((2 * mu_x * mu_y + c1) / (mu_x * mu_x + mu_y * mu_y + c1)) * cs
    return s.mean()


def _dcol(t: Tensor) -> Tensor:
    """t[i, j] - t[i, j+1] for an [H,W,C] image."""
    return t[:, :-1] - t[:, 1:]


def _drow(t: Tensor) -> Tensor:
    """t[i, j] - t[i+1, j]."""
    return t[:-1] - t[1:]


def edge_aware_log_l1(pred: Tensor, gt: Tensor, rgb: Tensor, mask: Optional[Tensor], capturable: bool = False) -> Tensor:
    """Scalar EdgeAwareLogL1 (losses.py:187-224): log(1 + |pred - gt|), down-weighted across image edges by
    exp(-mean_c |colour difference to the right / lower neighbour|), averaged over the masked pixels separately for the
    horizontal and the vertical term.
    ``capturable``: the masked means as sum(term x mask) / sum(mask) instead of the reference's boolean-mask gather
    ``term[mask].mean()`` — the gather's output size is data dependent (a host synchronisation: the step cannot be captured into
    a HIP graph); same value and gradients up to the order of the fp32 sums."""
    err = torch.log(1 + (pred - gt).abs())
    edge_x = torch.exp(-_dcol(rgb).abs().mean(dim=-1, keepdim=True))
    edge_y = torch.exp(-_drow(rgb).abs().mean(dim=-1, keepdim=True))
    term_x = edge_x * err[:, :-1]
    term_y = edge_y * err[:-1]
    if mask is not None and capturable:
        mx, my = mask[:, :-1].to(term_x.dtype), mask[:-1].to(term_y.dtype)
        return (term_x * mx).sum() / mx.sum() + (term_y * my).sum() / my.sum()
    if mask is not None:
        term_x = term_x[mask[:, :-1]]
        term_y = term_y[mask[:-1]]
    return term_x.mean() + term_y.mean()


def tv_loss(pred: Tensor) -> Tensor:
    """TVLoss (losses.py:279-295): mean absolute difference to the right neighbour plus to the lower neighbour."""
    return _dcol(pred).abs().mean() + _drow(pred).abs().mean()


def _ssim_hip(pred: Tensor, gt: Tensor) -> Tensor:
    from .fused_loss import ssim_hip
    return ssim_hip(pred, gt)


def rgb_term(outputs: Dict[str, Tensor], batch: Dict[str, Tensor], ssim_lambda: float = 0.2, fast: bool = False,
             ssim_impl: Optional[str] = None) -> Tensor:
    """nerfstudio splatfacto's main_loss, which DNSplatterModel.get_loss_dict takes over unchanged (dn_model.py:624-627, :663):
    (1 - l) * L1 + l * (1 - SSIM).  Restated from nerfstudio / pytorch_msssim's published definitions (not vendored: unpinned).
    ``ssim_impl``: "conv" (pytorch_msssim's ten grouped conv2d calls), "gemm", "stacked" (the same sums phrased for other library
    kernels; both measured slower) or "hip" — the SSIM module alone as one autograd node on ``dnsplat_ssim`` (what
    ``fused_loss.SSIM`` installs as splatfacto's ``self.ssim``), everything else of the stack in PyTorch."""
    pred_img = outputs["rgb"]
    ll1 = torch.abs(batch["image"] - pred_img).mean()
    if ssim_impl is None:
        ssim_impl = SSIM_FAST_IMPL if fast else "conv"
    impl = {"gemm": ssim_gemm, "stacked": ssim_stacked, "hip": _ssim_hip}.get(ssim_impl, ssim)
    simloss = 1 - impl(pred_img, batch["image"])
    return (1 - ssim_lambda) * ll1 + ssim_lambda * simloss


def regularization_term(outputs: Dict[str, Tensor], batch: Dict[str, Tensor], scales: Tensor, depth_lambda: float = 0.2,
                        depth_tolerance: float = 0.1, use_depth_loss: bool = True, use_normal_loss: bool = True,
                        capturable: bool = False, hip_modules: bool = False) -> Tensor:
    """What dn-splatter itself adds in get_loss_dict (dn_model.py:629-727) for regularization_strategy == "dn-splatter" with mono
    depth / mono normal supervision: ``DNRegularization.get_loss`` (regularization_strategy.py:146-199) on the ground truths the
    method picks — the image clamped at 10/255 for the edge weights (:633), depth, both normals and their ground truths multiplied
    by ``batch["mask"]`` if there is one (:646-659).  Pinned to the reference's own text: tests/golden/reference_regularization.npz.
    ``hip_modules``: the two stencil modules as ``install_losses(model)`` swaps them (``fused_loss.EdgeAwareLogL1`` / ``TVLoss``)."""
    gt_img = batch["image"].clamp(min=10 / 255.0)                                   # dn_model.py:633
    depth_out, pred_normal = outputs["depth"], outputs["normal"]
    gt_depth, gt_normal = batch.get("mono_depth"), batch.get("normal")
    if "mask" in batch:                                                             # dn_model.py:646-659
        mask = batch["mask"]
        depth_out = depth_out * mask
        gt_depth = gt_depth * mask if gt_depth is not None else None
        gt_normal = gt_normal * mask if gt_normal is not None else None
        pred_normal = pred_normal * mask
    if hip_modules:
        from .fused_loss import scale_reg
        loss = scale_reg(scales)
    else:
        loss = torch.min(torch.exp(scales), dim=1, keepdim=True)[0].mean()          # regularization_strategy.py:195-199
    if use_depth_loss and gt_depth is not None:
        valid = gt_depth > depth_tolerance                                          # :162
        if hip_modules:
            from .fused_loss import EdgeAwareLogL1
            d = EdgeAwareLogL1()(depth_out, gt_depth.float(), gt_img, valid)
        else:
            d = edge_aware_log_l1(depth_out, gt_depth.float(), gt_img, valid, capturable)
        loss = loss + (d + depth_lambda * d)                                        # :184
    if use_normal_loss and gt_normal is not None:
        if hip_modules:
            from .fused_loss import TVLoss
            tv = TVLoss()(pred_normal)
        else:
            tv = tv_loss(pred_normal)
        loss = loss + torch.abs(pred_normal - gt_normal).mean() + tv                     # :188-193
    return loss


def dn_loss(outputs: Dict[str, Tensor], batch: Dict[str, Tensor], scales: Tensor, ssim_lambda: float = 0.2,
            depth_lambda: float = 0.2, depth_tolerance: float = 0.1, use_depth_loss: bool = True,
            use_normal_loss: bool = True, capturable: bool = False, ssim_impl: Optional[str] = None, hip_modules: bool = False) -> Tensor:
    """main_loss of ``DNSplatterModel.get_loss_dict`` for regularization_strategy == "dn-splatter" with mono depth
    and mono normal supervision (dn_model.py:614-729): rgb_loss + regularization_strategy_loss (:727).
    ``capturable``: the same terms in PyTorch ops that need no host synchronisation and no convolution library — masked means as
    sum / count (``edge_aware_log_l1``), SSIM blurs as GEMMs (``ssim_gemm``): the whole step can be captured into a HIP graph."""
    return rgb_term(outputs, batch, ssim_lambda, fast=capturable, ssim_impl=ssim_impl) + regularization_term(outputs, batch, scales, depth_lambda, depth_tolerance,
                                                                       use_depth_loss, use_normal_loss, capturable, hip_modules)


def synthetic_batch(width: int, height: int, device, seed: int = 0) -> Dict[str, Tensor]:
    """Random ground truth with the shapes the datamanager hands over (dn_datamanager.py:90-150)."""
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(height, width, 3, generator=g)
    depth = torch.rand(height, width, 1, generator=g) * 9.0 + 0.5
    nrm = F.normalize(torch.randn(height, width, 3, generator=g), dim=-1)
    return {"image": img.to(device), "mono_depth": depth.to(device), "normal": ((nrm + 1) / 2).to(device)}
