"""Host-side mirror of ``DNSplatterModel.get_outputs`` (``dn_splatter/dn_model.py:404-612``).

nerfstudio is not a dependency here, so this module restates only what the hot path touches:
the 7-entry ``gauss_params`` dict (dn_model.py:227-237), a minimal pinhole ``Camera`` record standing
in for ``nerfstudio.cameras.Cameras`` (the fields read at dn_model.py:474-479, :585-597),
nerfstudio's ``get_viewmat`` and the per-pixel post-ops.  Same output keys and shapes
(dn_model.py:605-612): ``rgb[H,W,3] depth[H,W,1] normal[H,W,3] surface_normal[H,W,3]
accumulation[H,W,1] background[3]``.

``DNSplatterRenderer(fused=True)`` (default) renders colour, depth and normals in one compositing
pass (``fused.render_dn``); ``fused=False`` makes the reference's two calls through the drop-in
``rasterization`` / ``rasterize_gaussians`` — bit-for-bit the reference's own sequence of torch ops
around them.  ``rasterization_fn`` / ``rasterize_gaussians_fn`` exist so that parity tests can run
this very code against another implementation of the two calls; the package itself only ever
passes its HIP implementations.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, Optional

import torch
import torch.nn.functional as F
from torch import Tensor

from . import fused as _fused
from .legacy import quat_to_rotmat, rasterize_gaussians
from .rendering import rasterization


@dataclass
class Camera:
    """The slice of ``nerfstudio.cameras.Cameras`` that get_outputs reads (one camera)."""
    camera_to_worlds: Tensor  # [1,3,4], nerfstudio/OpenGL convention (x right, y up, z back)
    fx: float
    fy: float
    cx: float
    cy: float
    width: int
    height: int

    def get_intrinsics_matrices(self) -> Tensor:
        K = torch.zeros(1, 3, 3, dtype=torch.float32, device=self.camera_to_worlds.device)
        K[0, 0, 0], K[0, 1, 1], K[0, 0, 2], K[0, 1, 2], K[0, 2, 2] = self.fx, self.fy, self.cx, self.cy, 1.0
        return K

    def to(self, device) -> "Camera":
        return Camera(self.camera_to_worlds.to(device), self.fx, self.fy, self.cx, self.cy, self.width, self.height)


def get_viewmat(optimized_camera_to_world: Tensor) -> Tensor:
    """nerfstudio.models.splatfacto.get_viewmat (imported at dn_model.py:48): OpenGL c2w -> OpenCV w2c,
    analytic inverse (SURVEY.md Appendix A conventions)."""
    R = optimized_camera_to_world[:, :3, :3]
    T = optimized_camera_to_world[:, :3, 3:4]
    R = R * torch.tensor([[[1.0, -1.0, -1.0]]], device=R.device, dtype=R.dtype)
    R_inv = R.transpose(1, 2)
    T_inv = -torch.bmm(R_inv, T)
    viewmat = torch.zeros(R.shape[0], 4, 4, device=R.device, dtype=R.dtype)
    viewmat[:, 3, 3] = 1.0
    viewmat[:, :3, :3] = R_inv
    viewmat[:, :3, 3:4] = T_inv
    return viewmat


# ---- dn_splatter/utils/normal_utils.py:9-48 and utils/camera_utils.py:70-144, restated (stay in torch)


def pcd_to_normal(xyz: Tensor) -> Tensor:
    """Normals of an organised point cloud [H,W,3] from its 4-neighbourhood (utils/normal_utils.py:9-22):
    normalize(cross(right - left, top - bottom)) on the interior, zeros on the one-pixel border."""
    horizontal = xyz[1:-1, 2:] - xyz[1:-1, :-2]      # right minus left neighbour
    vertical = xyz[:-2, 1:-1] - xyz[2:, 1:-1]        # upper minus lower neighbour
    inner = F.normalize(torch.cross(horizontal, vertical, dim=-1), p=2, dim=-1)
    out = torch.zeros_like(xyz)
    out[1:-1, 1:-1] = inner
    return out


def normal_from_depth_image(depths: Tensor, fx: float, fy: float, cx: float, cy: float, img_size: tuple,
                            c2w: Tensor) -> Tensor:
    """Back-project with +0.5 pixel centres, 4-neighbour cross product (normal_utils.py:25-48)."""
    W, H = int(img_size[0]), int(img_size[1])
    dev = depths.device
    d = depths.reshape(-1).float()
    xs = torch.arange(W, device=dev, dtype=torch.float32) + 0.5
    ys = torch.arange(H, device=dev, dtype=torch.float32) + 0.5
    gx = xs[None, :].expand(H, W).reshape(-1)
    gy = ys[:, None].expand(H, W).reshape(-1)
    means3d = torch.stack([(gx - cx) * d / fx, (gy - cy) * d / fy, d], dim=-1)
    means3d = means3d @ torch.linalg.inv(c2w[:3, :3]) + c2w[:3, 3]
    return pcd_to_normal(means3d.view(H, W, 3))


@dataclass
class RendererConfig:
    """The renderer-relevant subset of DNSplatterModelConfig (dn_model.py:55-123)."""
    sh_degree: int = 3
    sh_degree_interval: int = 1000
    predict_normals: bool = True
    rasterize_mode: str = "classic"
    background_color: tuple = (0.1490, 0.1647, 0.2157)  # dn_model.py:161-163


class DNSplatterRenderer:
    """``get_outputs`` over a ``gauss_params`` dict (means, scales, quats, features_dc, features_rest,
    opacities[, normals]) — dn_model.py:227-237."""

    def __init__(self, gauss_params: Dict[str, Tensor], config: Optional[RendererConfig] = None, fused: bool = True,
                 rasterization_fn: Optional[Callable] = None, rasterize_gaussians_fn: Optional[Callable] = None,
                 fused_postops: bool = True):
        self.gauss_params = gauss_params
        self.config = config or RendererConfig()
        self.fused = fused
        self.fused_postops = fused_postops   # also run dn_model.py:526-537,577-603 inside the HIP kernels
        self._bg_cache: Dict = {}
        self.step = 10 ** 9  # all SH bands active unless the trainer says otherwise (dn_model.py:488-490)
        self._rasterization = rasterization_fn or rasterization
        self._rasterize_gaussians = rasterize_gaussians_fn or rasterize_gaussians
        self.training = True
        self.xys = self.radii = self.depths = self.conics = self.num_tiles_hit = None
        self.last_info: Dict = {}

    def forget(self) -> None:
        """Drops what the last get_outputs left on the renderer (xys, depths, last_info ...), and with it the autograd graph of
        that frame — e.g. before the step is captured into a HIP graph on another stream (graph.GraphedStep)."""
        self.xys = self.radii = self.depths = self.conics = self.num_tiles_hit = None
        self.last_info = {}

    @torch.no_grad()
    def get_outputs_batch(self, cameras, max_batch: int = 8):
        """Forward-only rendering of several cameras of one image size (the render loops of the offline
        consumers — export_mesh.py:960-1017, dn_pipeline.py:199-214, scripts/render_model.py:47-69 — call get_outputs once per
        camera, one after the other).  Up to ``max_batch`` cameras go through ONE launch sequence: their projections are
        stacked, binned with (camera, tile, depth) keys in one pass and composited by one launch over all their tile grids,
        so the bandwidth-bound binning of the batch and the tails of the compositing kernels are paid once.  Returns one
        output dict per camera, bit-identical to what get_outputs returns for it.  Renderer state (xys, radii, last_info,
        gauss_params["normals"]) is left as after the LAST camera, as a sequential loop would leave it."""
        gp = self.gauss_params
        cfg = self.config
        fused_ok = (self.fused and self.fused_postops and cfg.sh_degree >= 0 and cfg.predict_normals and cfg.rasterize_mode == "classic")
        same = all((int(c.width), int(c.height)) == (int(cameras[0].width), int(cameras[0].height)) for c in cameras)
        if not (fused_ok and same) or len(cameras) == 1:
            was_training, self.training = self.training, False
            try:
                return [self.get_outputs(c) for c in cameras]
            finally:
                self.training = was_training
        dev = cameras[0].camera_to_worlds.device
        background = self._background(dev)
        W, H = int(cameras[0].width), int(cameras[0].height)
        outs = []
        for i in range(0, len(cameras), max_batch):
            chunk = cameras[i:i + max_batch]
            out, info = _fused.render_dn_outputs_batch(
                gp["means"], gp["quats"], gp["scales"], gp["opacities"], gp["features_dc"], gp["features_rest"], chunk, W, H,
                sh_degree=self._sh_degree_to_use(), background_rgb=background, absgrad=False, sigmoid_colors=(cfg.sh_degree == 0))
            for c in range(len(chunk)):
                d = {k: v[c] for k, v in out.items()}
                d["background"] = background
                outs.append(d)
        gp["normals"] = info["normals_world"]
        self.xys = info["means2d"][-1:]
        self.radii = info["radii"][-1]
        self.depths = info["depths"][-1:]
        self.conics = info["conics"][-1:]
        self.num_tiles_hit = info["tiles_per_gauss"][-1:]
        self.last_info = info
        return outs

    def _background(self, device):
        background = self._bg_cache.get(device)
        if background is None:
            background = torch.tensor(self.config.background_color, dtype=torch.float32, device=device)
            self._bg_cache[device] = background
        return background

    def _sh_degree_to_use(self):
        c = self.config
        return min(self.step // c.sh_degree_interval, c.sh_degree)

    def get_outputs(self, camera: Camera) -> Dict[str, Tensor]:
        gp = self.gauss_params
        cfg = self.config
        if cfg.rasterize_mode not in ["antialiased", "classic"]:
            raise ValueError("Unknown rasterize_mode: %s", cfg.rasterize_mode)
        c2w = camera.camera_to_worlds
        W, H = int(camera.width), int(camera.height)
        background = self._background(c2w.device)

        means, scales, quats = gp["means"], gp["scales"], gp["quats"]
        features_dc, features_rest, opacities = gp["features_dc"], gp["features_rest"], gp["opacities"]

        if (self.fused and self.fused_postops and cfg.sh_degree >= 0 and cfg.predict_normals
                and cfg.rasterize_mode == "classic"):
            # everything between the parameters and the output dict in HIP; config.sh_degree == 0
            # (dn_model.py:486-493: sigmoid(colours), no SH) is the same pass with the sigmoid inside the projection kernels
            out, info = _fused.render_dn_outputs(
                means, quats, scales, opacities, features_dc, features_rest, c2w[0], float(camera.fx), float(camera.fy),
                float(camera.cx), float(camera.cy), W, H, sh_degree=self._sh_degree_to_use(), background_rgb=background,
                absgrad=True, sigmoid_colors=(cfg.sh_degree == 0))
            gp["normals"] = info["normals_world"]          # dn_model.py:558
            if self.training and info["means2d"].requires_grad:
                info["means2d"].retain_grad()              # dn_model.py:517-518
            self.xys = info["means2d"]
            self.radii = info["radii"][0]
            self.depths = info["depths"]
            self.conics = info["conics"]
            self.num_tiles_hit = info["tiles_per_gauss"]
            self.last_info = info
            out["background"] = background
            return out

        viewmat = get_viewmat(c2w)                       # dn_model.py:475
        K = camera.get_intrinsics_matrices().to(c2w.device)   # dn_model.py:476
        if self.fused and cfg.sh_degree > 0 and not (cfg.rasterize_mode == "antialiased" and cfg.predict_normals):
            render, alpha, normals_im, info = _fused.render_dn(
                means, quats, scales, opacities, features_dc, features_rest, viewmat[0], K[0], c2w[0], W, H,
                sh_degree=self._sh_degree_to_use(), predict_normals=cfg.predict_normals,
                rasterize_mode=cfg.rasterize_mode, absgrad=True)
            render, alpha = render[None], alpha[None]
            if cfg.predict_normals:
                gp["normals"] = info["normals_world"]      # dn_model.py:558
        else:
            colors_crop = torch.cat((features_dc[:, None, :], features_rest), dim=1)   # dn_model.py:466-468
            if cfg.sh_degree > 0:
                sh_degree_to_use = self._sh_degree_to_use()
            else:
                colors_crop = torch.sigmoid(colors_crop)
                sh_degree_to_use = None
            render, alpha, info = self._rasterization(                                 # dn_model.py:495-516
                means=means,
                quats=quats / quats.norm(dim=-1, keepdim=True),
                scales=torch.exp(scales),
                opacities=torch.sigmoid(opacities).squeeze(-1),
                colors=colors_crop,
                viewmats=viewmat,
                Ks=K,
                width=W,
                height=H,
                tile_size=16,
                packed=False,
                near_plane=0.01,
                far_plane=1e10,
                render_mode="RGB+ED",
                sh_degree=sh_degree_to_use,
                sparse_grad=False,
                absgrad=True,
                rasterize_mode=cfg.rasterize_mode,
            )
            normals_im = None
        if self.training and info["means2d"].requires_grad:
            info["means2d"].retain_grad()                                              # dn_model.py:517-518
        self.xys = info["means2d"]
        self.radii = info["radii"][0]
        self.depths = info["depths"]
        self.conics = info["conics"]
        self.num_tiles_hit = info["tiles_per_gauss"]
        self.last_info = info

        rgb = render[:, ..., :3] + (1 - alpha) * background                            # dn_model.py:526-528
        rgb = torch.clamp(rgb, 0.0, 1.0)
        depth_im = render[:, ..., 3:4]
        depth_im = torch.where(alpha > 0, depth_im, depth_im.detach().max()).squeeze(0)   # dn_model.py:533-537

        if cfg.predict_normals and normals_im is None:
            # the reference's second pass, verbatim (dn_model.py:543-575)
            quats_n = quats / quats.norm(dim=-1, keepdim=True)
            normals = F.one_hot(torch.argmin(scales, dim=-1), num_classes=3).to(scales.dtype)   # .float() in the reference
            rots = quat_to_rotmat(quats_n)
            # the reference's torch.bmm(rots, normals[:, :, None]).squeeze(-1): hipBLASLt runs a batch of 1 M 3x3 matrices as one
            # 13 ms launch (+ 9 ms in the backward); `normals` is one-hot, so the same products summed elementwise give the
            # same bits in 0.1 ms
            normals = (rots * normals[:, None, :]).sum(-1)
            normals = F.normalize(normals, dim=1)
            viewdirs = -means.detach() + c2w.detach()[..., :3, 3]
            viewdirs = viewdirs / viewdirs.norm(dim=-1, keepdim=True)
            dots = (normals * viewdirs).sum(-1)
            negative_dot_indices = dots < 0
            normals = torch.where(negative_dot_indices[:, None], -normals, normals)
            gp["normals"] = normals
            normals = normals @ c2w.squeeze(0)[:3, :3]
            xys = self.xys[0, ...].detach()
            normals_im = self._rasterize_gaussians(
                xys, self.depths[0, ...], self.radii, self.conics[0, ...], self.num_tiles_hit[0, ...], normals,
                torch.sigmoid(opacities).reshape(-1, 1), H, W, 16)
        if cfg.predict_normals:
            normals_im = normals_im / normals_im.norm(dim=-1, keepdim=True)            # dn_model.py:577-578
            normals_im = (normals_im + 1) / 2
        else:
            normals_im = torch.full(rgb.shape[1:], 0.0, device=rgb.device)

        surface_normal = normal_from_depth_image(                                       # dn_model.py:589-603
            depths=depth_im.detach(), fx=camera.fx, fy=camera.fy, cx=camera.cx, cy=camera.cy, img_size=(W, H),
            c2w=torch.eye(4, dtype=torch.float, device=depth_im.device))
        surface_normal = surface_normal @ torch.diag(torch.tensor([1.0, -1.0, -1.0], device=depth_im.device))
        surface_normal = (1 + surface_normal) / 2

        return {
            "rgb": rgb.squeeze(0),
            "depth": depth_im,
            "normal": normals_im,
            "surface_normal": surface_normal,
            "accumulation": alpha.squeeze(0),
            "background": background,
        }
