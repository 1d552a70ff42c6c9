"""Drop-in for ``gsplat.rendering.rasterization`` as dn-splatter uses it.

Reference call site: ``dn_splatter/dn_model.py:495-516`` (import at ``dn_model.py:29-32``).  Same
argument names, same return triple ``(render_colors, render_alphas, info)``, same ``info`` keys
(``dn_model.py:517-524`` reads ``means2d`` — with ``.retain_grad()`` / ``.absgrad`` — ``radii``,
``depths``, ``conics``, ``tiles_per_gauss``).  Everything runs in hand-written HIP kernels through
``libdnsplat.so``; there is no torch/CPU fallback.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from . import _ops
from ._ops import ProjCfg

_RENDER_MODES = ("RGB", "D", "ED", "RGB+D", "RGB+ED")


def rasterization(
    means: Tensor,  # [N, 3]
    quats: Tensor,  # [N, 4]
    scales: Tensor,  # [N, 3]
    opacities: Tensor,  # [N]
    colors: Tensor,  # [N, D] or [N, K, 3]
    viewmats: Tensor,  # [C, 4, 4]
    Ks: Tensor,  # [C, 3, 3]
    width: int,
    height: int,
    near_plane: float = 0.01,
    far_plane: float = 1e10,
    radius_clip: float = 0.0,
    eps2d: float = 0.3,
    sh_degree: Optional[int] = None,
    packed: bool = True,
    tile_size: int = 16,
    backgrounds: Optional[Tensor] = None,
    render_mode: str = "RGB",
    sparse_grad: bool = False,
    absgrad: bool = False,
    rasterize_mode: str = "classic",
    channel_chunk: int = 32,
) -> Tuple[Tensor, Tensor, Dict]:
    """Rasterize 3D Gaussians to one image (see module docstring for the contract).

    ``viewmats`` [C,4,4] / ``Ks`` [C,3,3]: C = 1 is the training call (``dn_model.py:421`` asserts a single camera); C > 1
    renders the batch in ONE launch sequence — one binning pass over (camera, tile, depth) keys and one compositing launch
    over all C tile grids (the render loops of export_mesh.py:960-1017, dn_pipeline.py:199-214,
    scripts/render_model.py:47-69), with the same results, bit for bit, as C separate calls.

    Differences from gsplat 1.0.0, all outside what dn-splatter exercises: ``packed`` is accepted and ignored (the info
    dict is always the dense ``[C, N, ...]`` layout dn-splatter asks for with ``packed=False``), at most 8 feature
    channels, one background row shared by the batch, ``sparse_grad`` unsupported.
    """
    N = means.shape[0]
    C = viewmats.shape[0]
    assert viewmats.shape == (C, 4, 4) and Ks.shape == (C, 3, 3), (viewmats.shape, Ks.shape)
    assert means.shape == (N, 3), means.shape
    assert quats.shape == (N, 4), quats.shape
    assert scales.shape == (N, 3), scales.shape
    assert opacities.shape == (N,), opacities.shape
    assert render_mode in _RENDER_MODES, render_mode
    if rasterize_mode not in ("classic", "antialiased"):
        raise ValueError(f"Unknown rasterize_mode: {rasterize_mode}")
    if sparse_grad:
        raise NotImplementedError("sparse_grad=True is not used by dn-splatter (dn_model.py:511) and not provided")

    coeffs = direct = None
    if sh_degree is None:
        # treat colors as post-activation values [N, D] (dn_model.py:492 path: sigmoid(colors) with dim_sh == 1)
        direct = colors.reshape(N, -1)
        n_feat = direct.shape[-1]
        if render_mode in ("D", "ED"):
            direct, n_feat = None, 0
    else:
        assert colors.dim() == 3 and colors.shape[0] == N and colors.shape[2] == 3, colors.shape
        assert (sh_degree + 1) ** 2 <= colors.shape[1], colors.shape
        coeffs = colors
        n_feat = 3
        if render_mode in ("D", "ED"):
            coeffs, n_feat, sh_degree = None, 0, None

    with_depth = render_mode in ("D", "ED", "RGB+D", "RGB+ED")
    D = n_feat + (1 if with_depth else 0)
    if D > 8:
        raise NotImplementedError(f"{D} feature channels requested; libdnsplat records carry at most 8")
    ed_channel = D - 1 if render_mode in ("ED", "RGB+ED") else -1

    cfg = ProjCfg(width=width, height=height, tile_size=tile_size, eps2d=eps2d, near_plane=near_plane,
                  far_plane=far_plane, radius_clip=radius_clip, antialiased=(rasterize_mode == "antialiased"),
                  sh_degree=-1 if sh_degree is None else int(sh_degree), with_depth=with_depth)
    pr = _ops.project(means, quats, scales, opacities, coeffs=coeffs, colors=direct, viewmat=viewmats, K=Ks, cfg=cfg)

    holder: Dict = {}
    bg = _background_row(backgrounds, n_feat, with_depth, D)
    render, alphas = _ops.rasterize(pr["means2d"], pr["splats"], pr["depths"], pr["radii"], pr["tiles_per_gauss"],
                                    background=bg, width=width, height=height, tile_size=tile_size, D=D,
                                    ed_channel=ed_channel, absgrad=absgrad, holder=holder, tile_boxes=pr["tile_boxes"])
    b: _ops.Binning = holder["binning"]
    tw, th = b.tile_width, b.tile_height
    meta = {
        "camera_ids": None,
        "gaussian_ids": None,
        "radii": pr["radii"],
        "means2d": pr["means2d"],
        "depths": pr["depths"],
        "conics": pr["conics"],
        "opacities": pr["splats"][:, 5].reshape(C, N),
        "tile_width": tw,
        "tile_height": th,
        "tiles_per_gauss": pr["tiles_per_gauss"],
        "flatten_ids": b.flatten_ids[: b.n_isects],
        "isect_offsets": b.tile_offsets[:-1].reshape(C, th, tw),
        "width": width,
        "height": height,
        "tile_size": tile_size,
        "n_cameras": C,
        "n_isects": b.n_isects,
        "_binning": b,             # not a gsplat key: handle for _ops.binning_status / lazily built entries
    }
    if pr["compensations"] is not None:
        meta["compensations"] = pr["compensations"]
    # gsplat's sorted 64-bit keys: nothing in dn-splatter reads them and the binning never forms them, so the entry is built (one
    # small kernel) when it is first read — and is then a plain int64 Tensor like every other entry
    depths = pr["depths"]
    meta = _ops.LazyInfo(meta, lazy={"isect_ids": lambda: _ops.isect_ids(b, depths.detach())})
    return render, alphas[..., None], meta


def _background_row(backgrounds: Optional[Tensor], n_feat: int, with_depth: bool, D: int) -> Optional[Tensor]:
    """gsplat takes ``backgrounds`` [C, channels of ``colors``] and gives the depth channel of the RGB+D / RGB+ED / D / ED
    modes a zero background itself (the call at dn_model.py:495-516 passes none).  The kernel reads ``background[k]`` for
    every composited channel k < D, so the row handed to it must be exactly D wide."""
    if backgrounds is None:
        return None
    if backgrounds.dim() != 2:
        raise ValueError(f"backgrounds must be [C, channels], got {tuple(backgrounds.shape)}")
    if backgrounds.shape[0] > 1 and not bool((backgrounds == backgrounds[:1]).all()):
        raise NotImplementedError("per-camera backgrounds: the batch shares one background row")
    row = backgrounds[0]
    if with_depth and n_feat == 0:
        row = row.new_zeros(1)                      # "D" / "ED": gsplat replaces the background by zeros
    elif with_depth and row.shape[0] == n_feat:
        row = torch.cat([row, row.new_zeros(1)])
    if row.shape[0] != D:
        raise ValueError(f"backgrounds has {backgrounds.shape[-1]} channels; expected {n_feat} (the colour channels)"
                         + (f" or {D} (colour channels + depth)" if with_depth else ""))
    return row
