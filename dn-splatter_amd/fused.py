"""One-pass colour + expected-depth + normal rendering — the MI355X-native form of the two gsplat
calls in ``DNSplatterModel.get_outputs`` (``dn_splatter/dn_model.py:495-516`` and ``:564-575``).

The reference renders RGB+ED with ``rasterization`` and then bins, sorts and composites a second
time (legacy ``rasterize_gaussians``) just to splat the per-Gaussian normals — "about 20% slower"
(reference README.md:60).  Both passes use the same xys / conics / opacities / tile lists, so their
transmittance is identical; here the 3 normal channels ride along as channels 4..6 of a single
7-channel compositing pass, with the two quirks of the second pass kept:

* its background defaults to ones  -> background vector (0,0,0,0,1,1,1);
* it is fed ``xys.detach()`` (dn_model.py:562) -> ``xy_split=4``: the normal channels contribute to
  v_conics / v_opacities / v_normals but not to ``means2d.grad`` / ``means2d.absgrad``.

Activations (exp / sigmoid / quaternion normalisation, dn_model.py:497-499), SH evaluation and the
normal derivation of dn_model.py:543-560 are fused into the projection kernel, and features_dc /
features_rest are read in place (no ``torch.cat`` of dn_model.py:466-468).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from . import _ops
from ._ops import ProjCfg


def normal_frame_from_c2w(camera_to_world: Tensor) -> Tensor:
    """[12] = (M row-major, camera centre) with n_cam = M n_world; dn_model.py:560 computes
    ``normals @ c2w[:3,:3]`` i.e. M = c2w[:3,:3]^T, and dn_model.py:550-552 flips against c2w[:3,3]."""
    c2w = camera_to_world.reshape(-1, 4)[:3]  # [3,4]
    return torch.cat([c2w[:, :3].t().reshape(9), c2w[:, 3].reshape(3)]).to(torch.float32).contiguous()


def render_dn(
    means: Tensor,            # [N,3]
    quats: Tensor,            # [N,4] raw (un-normalised) parameters
    scales: Tensor,           # [N,3] log-scales
    opacities: Tensor,        # [N] or [N,1] logits
    features_dc: Tensor,      # [N,3]
    features_rest: Tensor,    # [N,K-1,3]
    viewmat: Tensor,          # [4,4] world->camera (OpenCV), device tensor
    K: Tensor,                # [3,3]
    camera_to_world: Tensor,  # [3,4] nerfstudio (OpenGL) c2w used for the normal frame
    width: int,
    height: int,
    sh_degree: int,
    predict_normals: bool = True,
    rasterize_mode: str = "classic",
    near_plane: float = 0.01,
    far_plane: float = 1e10,
    eps2d: float = 0.3,
    absgrad: bool = True,
    activated: bool = False,
) -> Tuple[Tensor, Tensor, Optional[Tensor], Dict]:
    """Returns ``(render[H,W,4] (RGB + expected depth), alpha[H,W,1], normals[H,W,3] | None, info)``.

    ``activated=False`` means scales/opacities are the raw log / logit parameters of
    ``gauss_params`` (dn_model.py:227-237) and the kernel applies exp / sigmoid itself.
    """
    if rasterize_mode == "antialiased" and predict_normals:
        raise NotImplementedError(
            "fused normals need the normal pass to share the colour pass' opacities; with "
            "rasterize_mode='antialiased' the reference's second pass uses un-compensated opacities "
            "(dn_model.py:571) — use the two-call path (model.DNSplatterRenderer(fused=False))")
    N = means.shape[0]
    D = 4 + (3 if predict_normals else 0)
    cfg = ProjCfg(width=width, height=height, tile_size=16, eps2d=eps2d, near_plane=near_plane, far_plane=far_plane,
                  antialiased=(rasterize_mode == "antialiased"), scales_are_log=not activated,
                  opacities_are_logit=not activated, sh_degree=int(sh_degree), with_depth=True,
                  with_normals=predict_normals, want_normals_world=predict_normals, tight_tiles=_ops.TIGHT_TILES)
    nf = normal_frame_from_c2w(camera_to_world).to(means.device) if predict_normals else None
    pr = _ops.project(means, quats, scales, opacities.reshape(N), sh0=features_dc, shN=features_rest,
                      viewmat=viewmat, K=K, normal_frame=nf, cfg=cfg)
    bg = None
    if predict_normals:
        bg = torch.tensor([0.0, 0.0, 0.0, 0.0, 1.0, 1.0, 1.0], device=means.device)
    holder: Dict = {}
    # the tile lists of this path are internal: tight tile boxes (counts and flag both from the projection result)
    out, alphas = _ops.rasterize(pr["means2d"], pr["splats"], pr["depths"], pr["radii"], pr["tiles_bin"],
                                 background=bg, width=width, height=height, tile_size=16, D=D, ed_channel=3,
                                 xy_split=4, absgrad=absgrad, holder=holder, tight=pr["tight_tiles"], tile_boxes=pr["tile_boxes"])
    b = holder["binning"]
    info = {
        "means2d": pr["means2d"], "radii": pr["radii"], "depths": pr["depths"], "conics": pr["conics"],
        "tiles_per_gauss": pr["tiles_per_gauss"], "tiles_bin": pr["tiles_bin"],
        "normals_world": None if pr["normals_world"] is None else pr["normals_world"][-1],
        "flatten_ids": b.flatten_ids[: b.n_isects], "isect_offsets": b.tile_offsets[:-1].reshape(1, b.tile_height, b.tile_width),
        "n_isects": b.n_isects, "tile_width": b.tile_width, "tile_height": b.tile_height,
        "width": width, "height": height, "tile_size": 16, "n_cameras": 1, "tight_tiles": cfg.tight_tiles,
    }
    out, alphas = out.squeeze(0), alphas.squeeze(0)
    normals = out[..., 4:7] if predict_normals else None
    return out[..., :4], alphas[..., None], normals, info


def render_dn_outputs(
    means: Tensor, quats: Tensor, scales: Tensor, opacities: Tensor, features_dc: Tensor, features_rest: Tensor,
    camera_to_world: Tensor,  # [3,4] nerfstudio (OpenGL) c2w on the GPU
    fx: float, fy: float, cx: float, cy: float, width: int, height: int, sh_degree: int, background_rgb: Tensor,
    near_plane: float = 0.01, far_plane: float = 1e10, eps2d: float = 0.3, absgrad: bool = True,
    pair_counters: Optional[Tensor] = None, sigmoid_colors: bool = False,
) -> Tuple[Dict[str, Tensor], Dict]:
    """The whole replaced part of ``get_outputs`` (classic mode, predict_normals=True) in six launches:
    camera prepare, fused projection, binning, compositing with the dn-splatter epilogue, depth fill +
    depth->normal stencil; backward = compositing backward (taking the image cotangents directly) + fused
    projection backward.  Returns ``(outputs, info)`` with the output keys of dn_model.py:605-612 minus
    ``background``."""
    viewmat, K, nf, flag = _ops.camera_prepare(camera_to_world, fx, fy, cx, cy, with_flag=True, n_depth_max=1)
    outs, info = _render_dn_batch(means, quats, scales, opacities, features_dc, features_rest, viewmat[None], K[None], nf[None],
                                  [(fx, fy, cx, cy)], width, height, sh_degree, background_rgb, near_plane, far_plane, eps2d,
                                  absgrad, pair_counters, flag, sigmoid_colors=sigmoid_colors)
    # squeeze, not [0]: the backward of a select would zero-fill a full-size gradient and copy the slice into it (a fill + a copy
    # kernel per output image, ~60 us per frame at 1080p); the backward of a squeeze is a view
    return {k: v.squeeze(0) for k, v in outs.items()}, info


def render_dn_outputs_batch(
    means: Tensor, quats: Tensor, scales: Tensor, opacities: Tensor, features_dc: Tensor, features_rest: Tensor,
    cameras, width: int, height: int, sh_degree: int, background_rgb: Tensor,
    near_plane: float = 0.01, far_plane: float = 1e10, eps2d: float = 0.3, absgrad: bool = True, sigmoid_colors: bool = False,
) -> Tuple[Dict[str, Tensor], Dict]:
    """``render_dn_outputs`` for C cameras (records with camera_to_worlds [1,3,4], fx, fy, cx, cy; one image size) in ONE
    binning pass and ONE compositing launch: outputs are [C,H,W,.] stacks whose slices equal the
    per-camera results bit for bit."""
    viewmats, Ks, nfs, flag = _ops.camera_prepare_batch(cameras, with_flag=True)
    intr = [(float(c.fx), float(c.fy), float(c.cx), float(c.cy)) for c in cameras]
    return _render_dn_batch(means, quats, scales, opacities, features_dc, features_rest, viewmats, Ks, nfs, intr, width, height,
                            sh_degree, background_rgb, near_plane, far_plane, eps2d, absgrad, None, flag, sigmoid_colors=sigmoid_colors)


def _render_dn_batch(means, quats, scales, opacities, features_dc, features_rest, viewmats, Ks, nfs, intr, width, height,
                     sh_degree, background_rgb, near_plane, far_plane, eps2d, absgrad, pair_counters, saturation_flag=None,
                     sigmoid_colors=False):
    N = means.shape[0]
    C = viewmats.shape[0]
    # sigmoid_colors: the config.sh_degree == 0 branch of get_outputs (dn_model.py:491-493) — colours = sigmoid(features_dc), no SH
    # evaluation (sh_degree = None for gsplat); the sigmoid and its backward run inside the projection kernels
    cfg = ProjCfg(width=width, height=height, tile_size=16, eps2d=eps2d, near_plane=near_plane, far_plane=far_plane,
                  antialiased=False, scales_are_log=True, opacities_are_logit=True, sh_degree=-1 if sigmoid_colors else int(sh_degree),
                  colors_are_logit=bool(sigmoid_colors), with_depth=True, with_normals=True, want_normals_world=True,
                  tight_tiles=_ops.TIGHT_TILES, split_colours=_ops.SPLIT_COLOURS, skip_culled_records=_ops.SKIP_CULLED_RECORDS)
    side: Dict = {}
    if sigmoid_colors:
        pr = _ops.project(means, quats, scales, opacities.reshape(N), colors=features_dc.reshape(N, 3), viewmat=viewmats,
                          K=Ks, normal_frame=nfs, cfg=cfg, saturation_flag=saturation_flag, side=side)
    else:
        pr = _ops.project(means, quats, scales, opacities.reshape(N), sh0=features_dc, shN=features_rest, viewmat=viewmats,
                          K=Ks, normal_frame=nfs, cfg=cfg, saturation_flag=saturation_flag, side=side)
    # the tile lists of this path are internal: tight tile boxes; the projection's "some visible opacity > 0.999" word
    holder: Dict = {"saturation_flag": saturation_flag, "colours_ready": side.get("colours_ready")}
    if pair_counters is not None:
        holder["pair_counters"] = pair_counters
    rgb, depth, normal, acc, surface_normal = _ops.rasterize_dn(
        pr["means2d"], pr["splats"], pr["depths"], pr["radii"], pr["tiles_bin"], background_rgb=background_rgb,
        width=width, height=height, intrinsics=intr, absgrad=absgrad, holder=holder, tight=pr["tight_tiles"], tile_boxes=pr["tile_boxes"])
    b = holder["binning"]
    info = _ops.LazyInfo({
        "means2d": pr["means2d"], "radii": pr["radii"], "depths": pr["depths"], "conics": pr["conics"],
        # tiles_per_gauss: gsplat's count (what dn_model.py:524 stores as num_tiles_hit); tiles_bin: the count this path's binning
        # walked (tight tile boxes) — sum(tiles_bin) == n_isects
        "tiles_per_gauss": pr["tiles_per_gauss"], "tiles_bin": pr["tiles_bin"],
        "normals_world": pr["normals_world"][-1],      # dn_model.py:558 keeps the last camera's
        "tile_width": b.tile_width, "tile_height": b.tile_height,
        "width": width, "height": height, "tile_size": 16, "n_cameras": C, "_binning": b, "tight_tiles": cfg.tight_tiles,
        "_saturation_flag": saturation_flag,
    }, lazy={     # need the intersection count on the host: under the "deferred" bin policy reading them is what waits for it
        "flatten_ids": lambda: b.flatten_ids[: b.n_isects], "n_isects": lambda: b.n_isects,
        # this path keeps [start, end) per tile; gsplat's offsets of the empty tiles are filled in when somebody asks
        "isect_offsets": lambda: b.filled_offsets()[:-1].reshape(C, b.tile_height, b.tile_width),
    })
    return {"rgb": rgb, "depth": depth, "normal": normal, "surface_normal": surface_normal, "accumulation": acc}, info
