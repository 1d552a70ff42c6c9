"""``install(model_cls)``: put the fused MI355X pass behind ``DNSplatterModel.get_outputs`` at run time.

INTEGRATION.md section A (the import swap of ``dn_splatter/dn_model.py:29-35``) keeps the reference's two passes and ~40 torch
kernels around them; section B replaces the BODY of ``get_outputs`` (``dn_model.py:404-612``) by one call — which so far meant
editing the reference file by hand.  This module is section B as a function::

    import dn_splatter_amd
    from dn_splatter.dn_model import DNSplatterModel
    dn_splatter_amd.install(DNSplatterModel)          # once, before training / rendering

``get_outputs`` of the class is then the method below; the original is kept as ``DNSplatterModel._dnsplat_original_get_outputs``
(``uninstall`` restores it) and is what runs for the one configuration the fused pass does not cover
(``rasterize_mode="antialiased"`` with ``predict_normals``: the reference's second pass uses un-compensated opacities).

The method mirrors the reference line by line outside the two gsplat calls: the camera optimiser (``:420-424``), binary opacities
(``:426-438``), the evaluation crop box (``:440-464``), the down-scaled resolution (``:473-479``), the attributes nerfstudio's
``after_train`` / ``refinement_after`` read back (``:517-524``, ``:531``, ``:558``), camera bookkeeping (``:580-583``) and the output
dict (``:605-612``).  No nerfstudio import: the camera is whatever object the trainer hands over (``camera_to_worlds``, ``fx`` ...,
``rescale_output_resolution``).
"""
from __future__ import annotations

from typing import Dict

import torch

from . import fused as _fused

_ORIGINAL = "_dnsplat_original_get_outputs"


def _scalar(v) -> float:
    return float(v.item()) if torch.is_tensor(v) else float(v)


def fused_get_outputs(self, camera) -> Dict[str, torch.Tensor]:
    """Drop-in body of ``DNSplatterModel.get_outputs`` (``dn_model.py:404-612``) on the fused HIP pass."""
    if not hasattr(camera, "camera_to_worlds"):                       # dn_model.py:416-418 `not isinstance(camera, Cameras)`
        print("Called get_outputs with not a camera")
        return {}
    cfg = self.config
    if cfg.rasterize_mode not in ["antialiased", "classic"]:          # dn_model.py:482-483
        raise ValueError("Unknown rasterize_mode: %s", cfg.rasterize_mode)
    if cfg.rasterize_mode == "antialiased" and cfg.predict_normals:
        original = getattr(type(self), _ORIGINAL, None)
        if original is None:
            raise NotImplementedError("antialiased + predict_normals needs the reference's two-call get_outputs (INTEGRATION.md A)")
        return original(self, camera)

    if self.training:                                                  # dn_model.py:420-424
        assert camera.shape[0] == 1, "Only one camera at a time"
        optimized_camera_to_world = self.camera_optimizer.apply_to_camera(camera)
    else:
        optimized_camera_to_world = camera.camera_to_worlds

    if getattr(cfg, "use_binary_opacities", False) and self.step > cfg.warmup_length:      # dn_model.py:426-438
        skip_steps = cfg.reset_alpha_every * cfg.refine_every
        margin = 200
        if not self.step % skip_steps == 0 and self.step % skip_steps not in range(1, margin + 1):
            self.opacities = torch.where(self.opacities >= cfg.binary_opacities_threshold, torch.ones_like(self.opacities),
                                         torch.zeros_like(self.opacities))

    crop_ids = None                                                    # dn_model.py:440-464
    if getattr(self, "crop_box", None) is not None and not self.training:
        crop_ids = self.crop_box.within(self.means).squeeze()
        if crop_ids.sum() == 0:
            return self.get_empty_outputs(int(_scalar(camera.width)), int(_scalar(camera.height)), self.background_color)
    names = ("means", "quats", "scales", "opacities", "features_dc", "features_rest")
    p = {k: (getattr(self, k)[crop_ids] if crop_ids is not None else getattr(self, k)) for k in names}

    camera_scale_fac = self._get_downscale_factor()                    # dn_model.py:473-479
    camera.rescale_output_resolution(1 / camera_scale_fac)
    fx, fy, cx, cy = _scalar(camera.fx), _scalar(camera.fy), _scalar(camera.cx), _scalar(camera.cy)
    W, H = int(_scalar(camera.width)), int(_scalar(camera.height))
    self.last_size = (H, W)
    camera.rescale_output_resolution(camera_scale_fac)

    background = self._get_background_color()                          # dn_model.py:526
    sigmoid_colors = not (cfg.sh_degree > 0)                           # dn_model.py:486-493
    sh_degree = min(self.step // cfg.sh_degree_interval, cfg.sh_degree) if cfg.sh_degree > 0 else 0
    c2w = optimized_camera_to_world.reshape(-1, 3, 4)[0]
    if cfg.predict_normals:
        out, info = _fused.render_dn_outputs(
            p["means"], p["quats"], p["scales"], p["opacities"], p["features_dc"], p["features_rest"], c2w, fx, fy, cx, cy, W, H,
            sh_degree=sh_degree, background_rgb=background.to(c2w.device), absgrad=True, sigmoid_colors=sigmoid_colors)
        if crop_ids is None:
            self.gauss_params["normals"] = info["normals_world"]       # dn_model.py:558
    else:
        # no normal pass in the reference either (dn_model.py:542): colour + depth through the fused projection / compositing, the
        # post-ops of dn_model.py:526-537 in torch, a zero normal image
        from .model import DNSplatterRenderer, RendererConfig, Camera

        r = DNSplatterRenderer(p, RendererConfig(sh_degree=cfg.sh_degree, sh_degree_interval=cfg.sh_degree_interval,
                                                 predict_normals=False, rasterize_mode=cfg.rasterize_mode,
                                                 background_color=tuple(float(x) for x in background)), fused=True)
        r.step, r.training = self.step, self.training
        out = r.get_outputs(Camera(optimized_camera_to_world.reshape(1, 3, 4), fx, fy, cx, cy, W, H))
        info = {"means2d": r.xys, "radii": r.radii[None], "depths": r.depths, "conics": r.conics, "tiles_per_gauss": r.num_tiles_hit}
        out.pop("background", None)
    if self.training and info["means2d"].requires_grad and not info["means2d"].retains_grad:
        info["means2d"].retain_grad()                                  # dn_model.py:517-518
    self.xys = info["means2d"]                                         # dn_model.py:519-524
    self.radii = info["radii"][0]
    self.depths = info["depths"]
    self.conics = info["conics"]
    self.num_tiles_hit = info["tiles_per_gauss"]
    self.vis_indices = torch.where(self.radii > 0)[0]                  # dn_model.py:531

    if hasattr(camera, "metadata"):                                    # dn_model.py:580-583
        if camera.metadata is not None and "cam_idx" in camera.metadata:
            self.camera_idx = camera.metadata["cam_idx"]
    self.camera = camera
    return {"rgb": out["rgb"], "depth": out["depth"], "normal": out["normal"], "surface_normal": out["surface_normal"],
            "accumulation": out["accumulation"], "background": background}           # dn_model.py:605-612


def install(model_cls):
    """Replaces ``model_cls.get_outputs`` by the fused body (idempotent).  Returns ``model_cls``."""
    if getattr(model_cls, "get_outputs", None) is fused_get_outputs:
        return model_cls
    if hasattr(model_cls, "get_outputs"):
        setattr(model_cls, _ORIGINAL, model_cls.get_outputs)
    model_cls.get_outputs = fused_get_outputs
    return model_cls


def install_ssim(model):
    """Replaces ``model.ssim`` (dn_model.py:180: torchmetrics' ``StructuralSimilarityIndexMeasure(data_range=1.0, kernel_size=11)``,
    used by the inherited RGB loss term and by the evaluation, :855) by ``fused_loss.SSIM`` — the one term of the PyTorch loss stack
    whose library kernels (sixteen depthwise conv2d calls through MIOpen, ~3 ms at 1600 x 1200) cost as much as the render step;
    everything else of ``get_loss_dict`` stays as it is.  The old module is kept as ``model._dnsplat_original_ssim``."""
    from .fused_loss import SSIM

    if not isinstance(getattr(model, "ssim", None), SSIM):
        object.__setattr__(model, "_dnsplat_original_ssim", getattr(model, "ssim", None))
        model.ssim = SSIM()
    return model


def install_losses(model):
    """Swaps the two stencil modules of the model's regularisation strategy for their one-launch HIP drop-ins, when they are the ones
    ``DNRegularization`` / ``AGSMeshRegularization`` construct (regularization_strategy.py:131-144): ``strategy.depth_loss.loss``
    (``EdgeAwareLogL1(implementation="scalar")`` -> ``fused_loss.EdgeAwareLogL1``) and ``strategy.normal_smooth_loss.loss``
    (``TVLoss`` -> ``fused_loss.TVLoss``), the strategy's ``get_scale_loss`` method (-> ``fused_loss.scale_reg``); also
    ``install_ssim(model)``.  Everything else of ``get_loss_dict`` stays the reference's
    PyTorch code.  Returns the list of what was swapped."""
    from . import fused_loss

    swapped = []
    strategy = getattr(model, "regularization_strategy", None)
    for holder_name, cls_name, make in (("depth_loss", "EdgeAwareLogL1", fused_loss.EdgeAwareLogL1),
                                        ("normal_smooth_loss", "TVLoss", fused_loss.TVLoss)):
        holder = getattr(strategy, holder_name, None)
        inner = getattr(holder, "loss", None)
        if inner is None or type(inner).__name__ != cls_name or isinstance(inner, make):
            continue
        if cls_name == "EdgeAwareLogL1" and getattr(inner, "implementation", "scalar") != "scalar":
            continue
        holder.loss = make()
        swapped.append(f"regularization_strategy.{holder_name}.loss")
    if strategy is not None and type(strategy).__name__ in ("DNRegularization", "AGSMeshRegularization") \
            and getattr(strategy.get_scale_loss, "__name__", "") != "_hip_scale_loss":
        # a method, not a module (regularization_strategy.py:195-199): mean_g min_k exp(scales[g, k]), ten torch kernels over [N,3]
        def _hip_scale_loss(scales):
            return fused_loss.scale_reg(scales)

        strategy.get_scale_loss = _hip_scale_loss
        swapped.append("regularization_strategy.get_scale_loss")
    if hasattr(model, "ssim"):
        install_ssim(model)
        swapped.append("ssim")
    return swapped


def uninstall(model_cls):
    """Puts the original ``get_outputs`` back."""
    original = model_cls.__dict__.get(_ORIGINAL)
    if original is not None:
        model_cls.get_outputs = original
        delattr(model_cls, _ORIGINAL)
    return model_cls
