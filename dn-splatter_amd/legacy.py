"""Drop-ins for the three legacy gsplat symbols dn-splatter imports.

- ``rasterize_gaussians``  <- ``from gsplat import rasterize_gaussians``            (dn_model.py:33, used :564-575)
- ``quat_to_rotmat``       <- ``gsplat.cuda_legacy._torch_impl.quat_to_rotmat``     (dn_model.py:34, used :222,:547,...)
- ``num_sh_bases``         <- ``gsplat.cuda_legacy._wrapper.num_sh_bases``          (dn_model.py:35, used :139)
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F
from torch import Tensor

from . import _ops


def num_sh_bases(degree: int) -> int:
    if degree == 0:
        return 1
    if degree == 1:
        return 4
    if degree == 2:
        return 9
    if degree == 3:
        return 16
    if degree == 4:
        return 25
    assert False, "Invalid SH degree (must be 0..4)"


def quat_to_rotmat(quat: Tensor) -> Tensor:
    """wxyz quaternion(s) -> rotation matrices; the input is normalised first."""
    assert quat.shape[-1] == 4, quat.shape
    w, x, y, z = torch.unbind(F.normalize(quat, dim=-1), dim=-1)
    mat = torch.stack(
        [
            1 - 2 * (y**2 + z**2), 2 * (x * y - w * z), 2 * (x * z + w * y),
            2 * (x * y + w * z), 1 - 2 * (x**2 + z**2), 2 * (y * z - w * x),
            2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x**2 + y**2),
        ],
        dim=-1,
    )
    return mat.reshape(quat.shape[:-1] + (3, 3))


def rasterize_gaussians(
    xys: Tensor,            # [N, 2]
    depths: Tensor,         # [N]
    radii: Tensor,          # [N] int32
    conics: Tensor,         # [N, 3]
    num_tiles_hit: Tensor,  # [N] int32
    colors: Tensor,         # [N, Ch]
    opacity: Tensor,        # [N, 1]
    img_height: int,
    img_width: int,
    block_width: int,
    background: Optional[Tensor] = None,
    return_alpha: Optional[bool] = False,
):
    """N-channel compositing of already-projected Gaussians (the second pass of
    ``DNSplatterModel.get_outputs``, dn_model.py:564-575).  Differentiable w.r.t. ``xys``, ``conics``,
    ``colors`` and ``opacity``; ``background`` defaults to ones (legacy gsplat behaviour).

    Tile membership follows the v1.0 bounding-box rule that produced ``num_tiles_hit``; the legacy
    kernel's ``(int)(c + r + 1)`` rule differs from it only where ``(x + r)/16`` is an exact integer, a
    case in which the reference over-runs its own ``cum_tiles_hit`` slots.
    """
    assert 1 < block_width <= 16, "block_width must be between 2 and 16"
    if block_width != 16:
        raise NotImplementedError("libdnsplat composites 16x16 tiles (dn_model.py:470-472 uses 16)")
    if colors.dtype == torch.uint8:
        colors = colors.float() / 255
    if xys.ndimension() != 2 or xys.size(1) != 2:
        raise ValueError("xys must have dimensions (N, 2)")
    if colors.ndimension() != 2:
        raise ValueError("colors must have dimensions (N, D)")
    C = colors.shape[-1]
    if background is not None:
        assert background.shape[0] == C, f"incorrect shape of background color tensor, expected shape {C}"
    else:
        background = torch.ones(C, dtype=torch.float32, device=colors.device)
    if C > 8:
        raise NotImplementedError(f"{C} channels requested; libdnsplat records carry at most 8")

    splats = _ops._PackFn.apply(xys, conics, opacity, colors)
    render, alphas = _ops.rasterize(xys.detach(), splats, depths, radii.to(torch.int32).contiguous(),
                                    num_tiles_hit.to(torch.int32).contiguous(), background=background,
                                    width=img_width, height=img_height, tile_size=block_width, D=C)
    render, alphas = render.squeeze(0), alphas.squeeze(0)
    if return_alpha:
        return render, alphas
    return render
