"""The per-pixel losses of dn-splatter, kept in PyTorch-ROCm (BASELINE.json north star: "the per-pixel
depth/normal losses ... in dn_splatter/losses.py stay in PyTorch-ROCm").

This is the host-side restatement that BASELINE config C5 ("depth + mono-normal loss enabled") times together
with the renderer; nothing here is a kernel of ours.  It follows

* ``DNSplatterModel.get_loss_dict``                      dn_splatter/dn_model.py:614-729
* ``DNRegularization.get_loss`` / depth / normal / scale dn_splatter/regularization_strategy.py:146-199
* ``EdgeAwareLogL1``, ``LogL1``, ``L1``, ``TVLoss``      dn_splatter/losses.py:154-224, 279-295
* the inherited RGB term of nerfstudio's ``SplatfactoModel.get_loss_dict``: ``(1 - l) * L1 + l * (1 - SSIM)`` with
  ``l = ssim_lambda = 0.2`` and pytorch_msssim's SSIM (11x11 Gaussian window, sigma 1.5, valid padding,
  data_range 1) — neither nerfstudio nor pytorch_msssim is vendored in the reference, so that term is restated
  from their published definitions.
"""
from __future__ import annotations

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
    w = _gaussian_window(device=x.device)
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


def rgb_term(outputs: Dict[str, Tensor], batch: Dict[str, Tensor], ssim_lambda: float = 0.2) -> Tensor:
    """nerfstudio splatfacto's main_loss, which DNSplatterModel.get_loss_dict takes over unchanged (dn_model.py:624-627, :663):
    (1 - l) * L1 + l * (1 - SSIM).  Restated from nerfstudio / pytorch_msssim's published definitions (not vendored: unpinned)."""
    pred_img = outputs["rgb"]
    ll1 = torch.abs(batch["image"] - pred_img).mean()
    simloss = 1 - ssim(pred_img, batch["image"])
    return (1 - ssim_lambda) * ll1 + ssim_lambda * simloss


def regularization_term(outputs: Dict[str, Tensor], batch: Dict[str, Tensor], scales: Tensor, depth_lambda: float = 0.2,
                        depth_tolerance: float = 0.1, use_depth_loss: bool = True, use_normal_loss: bool = True,
                        capturable: bool = False) -> Tensor:
    """What dn-splatter itself adds in get_loss_dict (dn_model.py:629-727) for regularization_strategy == "dn-splatter" with mono
    depth / mono normal supervision: ``DNRegularization.get_loss`` (regularization_strategy.py:146-199) on the ground truths the
    method picks — the image clamped at 10/255 for the edge weights (:633), depth, both normals and their ground truths multiplied
    by ``batch["mask"]`` if there is one (:646-659).  Pinned to the reference's own text: tests/golden/reference_regularization.npz."""
    gt_img = batch["image"].clamp(min=10 / 255.0)                                   # dn_model.py:633
    depth_out, pred_normal = outputs["depth"], outputs["normal"]
    gt_depth, gt_normal = batch.get("mono_depth"), batch.get("normal")
    if "mask" in batch:                                                             # dn_model.py:646-659
        mask = batch["mask"]
        depth_out = depth_out * mask
        gt_depth = gt_depth * mask if gt_depth is not None else None
        gt_normal = gt_normal * mask if gt_normal is not None else None
        pred_normal = pred_normal * mask
    loss = torch.min(torch.exp(scales), dim=1, keepdim=True)[0].mean()              # regularization_strategy.py:195-199
    if use_depth_loss and gt_depth is not None:
        valid = gt_depth > depth_tolerance                                          # :162
        d = edge_aware_log_l1(depth_out, gt_depth.float(), gt_img, valid, capturable)
        loss = loss + (d + depth_lambda * d)                                        # :184
    if use_normal_loss and gt_normal is not None:
        loss = loss + torch.abs(pred_normal - gt_normal).mean() + tv_loss(pred_normal)   # :188-193
    return loss


def dn_loss(outputs: Dict[str, Tensor], batch: Dict[str, Tensor], scales: Tensor, ssim_lambda: float = 0.2,
            depth_lambda: float = 0.2, depth_tolerance: float = 0.1, use_depth_loss: bool = True,
            use_normal_loss: bool = True, capturable: bool = False) -> Tensor:
    """main_loss of ``DNSplatterModel.get_loss_dict`` for regularization_strategy == "dn-splatter" with mono depth
    and mono normal supervision (dn_model.py:614-729): rgb_loss + regularization_strategy_loss (:727)."""
    return rgb_term(outputs, batch, ssim_lambda) + regularization_term(outputs, batch, scales, depth_lambda, depth_tolerance,
                                                                       use_depth_loss, use_normal_loss, capturable)


def synthetic_batch(width: int, height: int, device, seed: int = 0) -> Dict[str, Tensor]:
    """Random ground truth with the shapes the datamanager hands over (dn_datamanager.py:90-150)."""
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(height, width, 3, generator=g)
    depth = torch.rand(height, width, 1, generator=g) * 9.0 + 0.5
    nrm = F.normalize(torch.randn(height, width, 3, generator=g), dim=-1)
    return {"image": img.to(device), "mono_depth": depth.to(device), "normal": ((nrm + 1) / 2).to(device)}
