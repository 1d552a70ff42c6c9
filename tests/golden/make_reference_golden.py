"""Golden vectors produced by THE REFERENCE's own Python code (not by the oracle).

Only the parts of the path that are plain PyTorch in the reference can be run in this container (gsplat 1.0.0 needs CUDA,
nerfstudio is not installed):

  * depth -> normal image      dn_splatter/utils/normal_utils.py:9-48 (pcd_to_normal, normal_from_depth_image) with
                               dn_splatter/utils/camera_utils.py:92-144 (get_means3d_backproj), as called at
                               dn_model.py:589-603 (c2w = identity, then @ diag(1,-1,-1) and (1 + n) / 2)
  * per-pixel loss terms       dn_splatter/losses.py:154-224 (L1, LogL1, EdgeAwareLogL1) and :279-295 (TVLoss), the terms
                               DNRegularization combines at regularization_strategy.py:146-199
  * DNSplatterModel.get_outputs   dn_splatter/dn_model.py:404-612 ITSELF: the method's source text is cut out of the class
                               with ``ast`` and executed as it stands.  The names it pulls from absent packages are
                               supplied: the two gsplat raster calls by recording stand-ins that return seeded leaf tensors
                               (so everything AROUND them — the activations handed to gsplat dn_model.py:496-499, the
                               background blend / clamp / depth fill :526-537, the per-Gaussian normal derivation :543-560,
                               the normalisation :577-578, the depth -> normal call :589-603 — is the reference's own
                               arithmetic and autograd), gsplat's quat_to_rotmat and nerfstudio's get_viewmat by their
                               published formulas (SURVEY.md A.1 / Appendix A conventions), normal_from_depth_image by
                               the reference's real function.

The module files are loaded by path from /root/reference (read-only); the package's __init__ (which pulls in nerfstudio
data parsers) is bypassed by registering empty parent packages, and the imports losses.py makes but these classes never
use (torchmetrics, nerfstudio.field_components, dn_splatter.metrics) are satisfied with empty stand-ins.  Nothing of the
reference is copied: only its OUTPUTS on seeded inputs are stored.

    python tests/golden/make_reference_golden.py     # needs /root/reference; rewrites reference_*.npz
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("DNSPLAT_REFERENCE", "/root/reference")


def _load(name, relpath):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    for pkg in ("dn_splatter", "dn_splatter.utils"):
        m = types.ModuleType(pkg)
        m.__path__ = []
        sys.modules[pkg] = m
    # stand-ins for imports the functions used here never touch
    metrics = types.ModuleType("dn_splatter.metrics")
    metrics.mean_angular_error = None
    sys.modules["dn_splatter.metrics"] = metrics
    tm = types.ModuleType("torchmetrics"); tmi = types.ModuleType("torchmetrics.image")
    tmi.MultiScaleStructuralSimilarityIndexMeasure = object
    tmi.StructuralSimilarityIndexMeasure = object
    sys.modules.setdefault("torchmetrics", tm); sys.modules.setdefault("torchmetrics.image", tmi)
    for name in ("nerfstudio", "nerfstudio.field_components", "nerfstudio.field_components.field_heads"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["nerfstudio.field_components.field_heads"].FieldHeadNames = object
    cam = _load("dn_splatter.utils.camera_utils", "dn_splatter/utils/camera_utils.py")
    nrm = _load("dn_splatter.utils.normal_utils", "dn_splatter/utils/normal_utils.py")
    los = _load("dn_splatter.losses", "dn_splatter/losses.py")
    return cam, nrm, los


def depth_normal_case(nrm, path, W=80, H=56, seed=3):
    g = torch.Generator().manual_seed(seed)
    # a smooth surface plus noise, one flat region and a depth step, fx != fy, off-centre principal point
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    depth = 3.0 + 0.02 * xx + 0.5 * torch.sin(yy / 7.0) + 0.05 * torch.rand(H, W, generator=g)
    depth[10:20, 30:50] = 2.5
    depth[40:, 60:] += 1.5
    fx, fy, cx, cy = 61.5, 58.25, 41.0, 26.5
    depth_in = depth[..., None]                                  # [H,W,1] as dn_model.py:590 passes it
    normals = nrm.normal_from_depth_image(depths=depth_in, fx=fx, fy=fy, cx=cx, cy=cy, img_size=(W, H),
                                          c2w=torch.eye(4, dtype=torch.float), device=torch.device("cpu"), smooth=False)
    # dn_model.py:599-603
    surface_normal = normals @ torch.diag(torch.tensor([1, -1, -1], dtype=depth.dtype))
    surface_normal = (1 + surface_normal) / 2
    np.savez_compressed(path, W=W, H=H, fx=fx, fy=fy, cx=cx, cy=cy, depth=depth.numpy(), normals_raw=normals.numpy(),
                        surface_normal=surface_normal.numpy())
    print(path, os.path.getsize(path) // 1024, "KiB")


def loss_case(los, path, W=72, H=48, seed=5):
    g = torch.Generator().manual_seed(seed)
    pred = (torch.rand(H, W, 1, generator=g) * 4 + 0.5).requires_grad_(True)
    gt = torch.rand(H, W, 1, generator=g) * 4 + 0.5
    rgb = torch.rand(H, W, 3, generator=g)
    mask = torch.rand(H, W, 1, generator=g) > 0.2
    pn = torch.nn.functional.normalize(torch.randn(H, W, 3, generator=g), dim=-1)
    pn = ((pn + 1) / 2).requires_grad_(True)
    out = dict(W=W, H=H, pred=pred.detach().numpy(), gt=gt.numpy(), rgb=rgb.numpy(), mask=mask.numpy(), pred_normal=pn.detach().numpy())

    def grad_of(value, wrt):
        (gr,) = torch.autograd.grad(value, wrt, retain_graph=True)
        return gr.numpy()

    for name, fn in (("edge_aware_logl1_masked", lambda: los.EdgeAwareLogL1(implementation="scalar")(pred, gt, rgb, mask)),
                     ("edge_aware_logl1_nomask", lambda: los.EdgeAwareLogL1(implementation="scalar")(pred, gt, rgb, None)),
                     ("logl1_scalar", lambda: los.LogL1(implementation="scalar")(pred, gt)),
                     ("l1_scalar", lambda: los.L1(implementation="scalar")(pred, gt))):
        v = fn()
        out[name] = np.float64(v.item())
        out[name + "_grad"] = grad_of(v, pred)
    tv = los.TVLoss()(pn)
    out["tv_normal"] = np.float64(tv.item())
    out["tv_normal_grad"] = grad_of(tv, pn)
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KiB", {k: float(v) for k, v in out.items() if np.ndim(v) == 0 and k not in ("W", "H")})


# ------------------------------------------------------------------------------------------------------------------
# DNSplatterModel.get_outputs, executed from the reference's own text


def extract_method(path, cls, name):
    """Source text of ``cls.name`` in the file at ``path``, dedented so that it compiles as a free function."""
    import ast
    import textwrap

    src = open(path).read()
    for node in ast.parse(src).body:
        if isinstance(node, ast.ClassDef) and node.name == cls:
            for item in node.body:
                if isinstance(item, ast.FunctionDef) and item.name == name:
                    return textwrap.dedent(ast.get_source_segment(src, item, padded=True))
    raise KeyError(f"{cls}.{name} not found in {path}")


def extract_function(path, name):
    """Source text of the module-level function ``name`` in the file at ``path``."""
    import ast

    src = open(path).read()
    for node in ast.parse(src).body:
        if isinstance(node, ast.FunctionDef) and node.name == name:
            return ast.get_source_segment(src, node, padded=True)
    raise KeyError(f"{name} not found in {path}")


def helpers_case(path, N=48, seed=29):
    """The module-level plain-torch helpers of dn_model.py, executed from their own text: they state the conventions the gsplat
    symbols are USED with, in the reference's own words —
      * ``random_quat_tensor`` (:1497-1509) with a seeded generator: the initial rotations;
      * ``SH2RGB`` (:1512-1517): band 0 of the SH colour is ``0.28209479177387814 * coefficient + 0.5``;
      * ``rotate_vector_to_vector`` + ``matrix_to_quaternion`` (:1520-1600, the normal initialisation of :200-218): quaternions
        are (w, x, y, z) and ``quat_to_rotmat(matrix_to_quaternion(R)) == R`` — the file feeds these quaternions to gsplat's
        ``quat_to_rotmat`` (:34, :552, :1199) and reads the columns of the result as the Gaussian's axes;
      * ``invert_quaternion`` (:1615-1626): the inverse rotation is the conjugate in that convention;
      * ``scale_rot_to_inv_cov3d`` (:1603-1612) with quat_to_rotmat supplied by the caller: Sigma^-1 = R diag(1/s^2) R^T."""
    import math

    src = os.path.join(REF, "dn_splatter/dn_model.py")
    ns = {"torch": torch, "math": math, "Tensor": torch.Tensor, "quat_to_rotmat": quat_to_rotmat_published}
    for fn in ("random_quat_tensor", "SH2RGB", "rotate_vector_to_vector", "matrix_to_quaternion", "invert_quaternion",
               "scale_rot_to_inv_cov3d"):
        exec(compile(extract_function(src, fn), src + ":" + fn, "exec"), ns)
    g = torch.Generator().manual_seed(seed)
    quats0 = ns["random_quat_tensor"](N, generator=torch.Generator().manual_seed(seed + 1))
    sh = torch.randn(N, 3, generator=g) * 2
    v1 = torch.randn(N, 3, generator=g)
    v2 = torch.randn(N, 3, generator=g)
    R = ns["rotate_vector_to_vector"](v1, v2)
    q = ns["matrix_to_quaternion"](R)
    scale = torch.rand(N, 3, generator=g) * 0.5 + 0.01
    np.savez_compressed(path, N=N, seed=seed, random_quats=quats0.numpy(), sh=sh.numpy(), sh2rgb=ns["SH2RGB"](sh).numpy(),
                        v1=v1.numpy(), v2=v2.numpy(), R=R.numpy(), quat_of_R=q.numpy(),
                        quat_inverse=ns["invert_quaternion"](q).numpy(), scale=scale.numpy(),
                        inv_cov3d=ns["scale_rot_to_inv_cov3d"](scale, q).numpy())
    print("wrote", path)


class StubCameras:
    """What get_outputs reads from nerfstudio.cameras.Cameras (dn_model.py:417-421, 474-479, 585-597)."""

    def __init__(self, c2w, fx, fy, cx, cy, W, H):
        self.camera_to_worlds = c2w                          # [1,3,4]
        t = lambda v: torch.tensor([[float(v)]])             # noqa: E731
        self.fx, self.fy, self.cx, self.cy = t(fx), t(fy), t(cx), t(cy)
        self.width, self.height = torch.tensor([[W]]), torch.tensor([[H]])
        self.shape = (1,)
        self.metadata = None

    def rescale_output_resolution(self, _f):
        pass

    def get_intrinsics_matrices(self):
        K = torch.zeros(1, 3, 3)
        K[0, 0, 0], K[0, 1, 1], K[0, 0, 2], K[0, 1, 2], K[0, 2, 2] = self.fx.item(), self.fy.item(), self.cx.item(), self.cy.item(), 1.0
        return K


def quat_to_rotmat_published(quat):
    """gsplat.cuda_legacy._torch_impl.quat_to_rotmat (not vendored): wxyz, normalised first."""
    w, x, y, z = torch.unbind(torch.nn.functional.normalize(quat, dim=-1), dim=-1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=-1).reshape(quat.shape[:-1] + (3, 3))


def get_viewmat_published(c2w):
    """nerfstudio.models.splatfacto.get_viewmat (not vendored): flip y/z columns, analytic inverse."""
    R = c2w[:, :3, :3] * torch.tensor([[[1.0, -1.0, -1.0]]])
    T = c2w[:, :3, 3:4]
    Rt = R.transpose(1, 2)
    vm = torch.zeros(c2w.shape[0], 4, 4)
    vm[:, 3, 3] = 1.0
    vm[:, :3, :3] = Rt
    vm[:, :3, 3:4] = -torch.bmm(Rt, T)
    return vm


def get_outputs_case(nrm, path, N=60, W=40, H=24, seed=11):
    import types as _t

    g = torch.Generator().manual_seed(seed)
    rnd = lambda *shape: torch.rand(*shape, generator=g)        # noqa: E731
    gauss = {
        "means": (rnd(N, 3) - 0.5) * 4,
        "scales": torch.log(rnd(N, 3) * 0.3 + 0.02),
        "quats": torch.randn(N, 4, generator=g) * 1.7,           # deliberately un-normalised
        "features_dc": rnd(N, 3),
        "features_rest": torch.randn(N, 15, 3, generator=g) * 0.1,
        "opacities": torch.randn(N, 1, generator=g),
    }
    gauss["scales"][3] = gauss["scales"][3, 0]                   # a tie for argmin(scales)
    params = {k: v.clone().requires_grad_(True) for k, v in gauss.items()}
    # camera: a rotation that is not axis aligned, looking roughly at the origin
    ang = torch.tensor(0.7)
    Ry = torch.tensor([[torch.cos(ang), 0, torch.sin(ang)], [0, 1, 0], [-torch.sin(ang), 0, torch.cos(ang)]])
    ang2 = torch.tensor(-0.3)
    Rx = torch.tensor([[1, 0, 0], [0, torch.cos(ang2), -torch.sin(ang2)], [0, torch.sin(ang2), torch.cos(ang2)]])
    c2w = torch.cat([Ry @ Rx, torch.tensor([[1.5], [0.4], [5.0]])], dim=1)[None].contiguous()
    fx, fy, cx, cy = 31.5, 29.0, 19.25, 12.5
    camera = StubCameras(c2w, fx, fy, cx, cy, W, H)

    # what the two stand-ins hand back: seeded "rendered" images (leaf tensors, so the reference's post-op autograd runs)
    render = torch.cat([rnd(1, H, W, 3) * 1.6 - 0.3, rnd(1, H, W, 1) * 5 + 0.5], dim=-1)      # rgb beyond both clamp corners
    alpha = rnd(1, H, W, 1)
    alpha[0, :3, :5] = 0.0                                                                   # where(alpha > 0, depth, max)
    render = render.requires_grad_(True)
    alpha = alpha.requires_grad_(True)
    mix = torch.randn(H * W, N, generator=g) * 0.3                                           # normals image = mix @ normals
    calls = {}

    def rasterization(**kw):
        calls["rasterization"] = kw
        info = {"means2d": (rnd(1, N, 2) * 30).requires_grad_(True), "radii": torch.ones(1, N, dtype=torch.int32),
                "depths": rnd(1, N) + 1, "conics": rnd(1, N, 3), "tiles_per_gauss": torch.ones(1, N, dtype=torch.int32)}
        return render, alpha, info

    def rasterize_gaussians(xys, depths, radii, conics, num_tiles_hit, colors, opacity, img_height, img_width, block_width,
                            background=None, return_alpha=False):
        colors.retain_grad()
        calls["rasterize_gaussians"] = dict(xys=xys, colors=colors, opacity=opacity, img_height=img_height, img_width=img_width,
                                            block_width=block_width, background=background, xys_requires_grad=xys.requires_grad)
        return (mix @ colors).reshape(img_height, img_width, 3) + 0.2

    cfg = _t.SimpleNamespace(use_binary_opacities=False, rasterize_mode="classic", sh_degree=3, sh_degree_interval=1000,
                             predict_normals=True)
    me = _t.SimpleNamespace(training=True, config=cfg, step=2500, crop_box=None, device=torch.device("cpu"),
                            camera_optimizer=_t.SimpleNamespace(apply_to_camera=lambda cam: cam.camera_to_worlds),
                            _get_downscale_factor=lambda: 1, gauss_params=dict(params),
                            _get_background_color=lambda: torch.tensor([0.1490, 0.1647, 0.2157]),
                            means=params["means"], scales=params["scales"], quats=params["quats"],
                            features_dc=params["features_dc"], features_rest=params["features_rest"],
                            opacities=params["opacities"])
    from typing import Dict, List, Union
    ns = dict(torch=torch, F=torch.nn.functional, Tensor=torch.Tensor, Dict=Dict, List=List, Union=Union, Cameras=StubCameras,
              rasterization=rasterization, rasterize_gaussians=rasterize_gaussians, quat_to_rotmat=quat_to_rotmat_published,
              get_viewmat=get_viewmat_published, normal_from_depth_image=nrm.normal_from_depth_image)
    text = extract_method(os.path.join(REF, "dn_splatter/dn_model.py"), "DNSplatterModel", "get_outputs")
    exec(compile(text, "dn_model.py::DNSplatterModel.get_outputs", "exec"), ns)
    cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self           # dn_model.py:476 `.cuda()`: this container has no GPU
    try:
        out = ns["get_outputs"](me, camera)
    finally:
        torch.Tensor.cuda = cuda
    cot = {k: rnd(*out[k].shape) * 2 - 1 for k in ("rgb", "depth", "normal", "accumulation")}
    torch.autograd.backward([out[k] for k in cot], [cot[k] for k in cot])
    kw = calls["rasterization"]
    rg = calls["rasterize_gaussians"]
    save = dict(N=N, W=W, H=H, fx=fx, fy=fy, cx=cx, cy=cy, c2w=c2w.numpy(), step=me.step,
                render=render.detach().numpy(), alpha=alpha.detach().numpy(), mix=mix.numpy(),
                # A0: what the reference hands to gsplat.rasterization (dn_model.py:495-513)
                call_quats=kw["quats"].detach().numpy(), call_scales=kw["scales"].detach().numpy(),
                call_opacities=kw["opacities"].detach().numpy(), call_colors=kw["colors"].detach().numpy(),
                call_viewmats=kw["viewmats"].numpy(), call_Ks=kw["Ks"].numpy(), call_sh_degree=kw["sh_degree"],
                call_scalars=np.array([kw["width"], kw["height"], kw["tile_size"], kw["near_plane"], kw["far_plane"]], dtype=np.float64),
                call_flags=np.array([kw["packed"], kw["sparse_grad"], kw["absgrad"], kw["render_mode"] == "RGB+ED",
                                     kw["rasterize_mode"] == "classic"]),
                # A7: what it hands to the legacy rasterize_gaussians (dn_model.py:564-575) and stores at :558
                normals_cam=rg["colors"].detach().numpy(), normals_world=me.gauss_params["normals"].detach().numpy(),
                legacy_opacity=rg["opacity"].detach().numpy(), legacy_xys_requires_grad=rg["xys_requires_grad"],
                legacy_background_is_none=rg["background"] is None, legacy_block_width=rg["block_width"],
                v_normals_cam=rg["colors"].grad.numpy(),
                # gradients of the seeded loss
                v_render=render.grad.numpy(), v_alpha=alpha.grad.numpy(), v_quats=params["quats"].grad.numpy(),
                grad_is_none=np.array([params[k].grad is None for k in ("means", "scales", "features_dc", "features_rest", "opacities")]))
    for k, v in gauss.items():
        save["param_" + k] = v.numpy()
    for k, v in out.items():
        save["out_" + k] = v.detach().numpy()
    for k, v in cot.items():
        save["cot_" + k] = v.numpy()
    np.savez_compressed(path, **save)
    print(path, os.path.getsize(path) // 1024, "KiB", "outputs", {k: tuple(v.shape) for k, v in out.items()})


# ------------------------------------------------------------------------------------------------------------------
# DNRegularization.get_loss and DNSplatterModel.get_loss_dict (N2: how the per-pixel terms are COMBINED)


def regularization_case(los, path, W=56, H=40, N=300, seed=17):
    """regularization_strategy.py is loaded as it stands (its only import beside torch is dn_splatter.losses, the reference's
    own file loaded above): DNRegularization() with its defaults, get_loss on seeded images -> value and gradients.  Then
    DNSplatterModel.get_loss_dict (dn_model.py:614-729) executed from its source text inside a class whose parent's
    get_loss_dict (nerfstudio's, absent) is a stand-in returning a recorded rgb term: what is pinned is everything dn-splatter
    itself does — gt image clamp at 10/255, which depth / normal ground truth is chosen, the mask products, main_loss =
    rgb_loss + regularization.  The SSIM + L1 rgb term stays nerfstudio's (unpinned)."""
    import types as _t

    reg_mod = _load("dn_splatter.regularization_strategy", "dn_splatter/regularization_strategy.py")
    g = torch.Generator().manual_seed(seed)
    rnd = lambda *shape: torch.rand(*shape, generator=g)       # noqa: E731
    image = rnd(H, W, 3)
    image[:6, :9] *= 0.02                                       # below the 10/255 clamp of dn_model.py:633
    gt_depth = rnd(H, W, 1) * 6 + 0.2
    gt_depth[10:16, 20:31] = 0.05                               # below depth_tolerance: masked out of the depth term
    gt_normal = (torch.nn.functional.normalize(torch.randn(H, W, 3, generator=g), dim=-1) + 1) / 2
    pred_depth = (rnd(H, W, 1) * 6 + 0.2).requires_grad_(True)
    pred_normal = ((torch.nn.functional.normalize(torch.randn(H, W, 3, generator=g), dim=-1) + 1) / 2).requires_grad_(True)
    pred_rgb = rnd(H, W, 3).requires_grad_(True)
    scales = (torch.randn(N, 3, generator=g) * 0.7 - 3.0).requires_grad_(True)
    save = dict(W=W, H=H, N=N, image=image.numpy(), gt_depth=gt_depth.numpy(), gt_normal=gt_normal.numpy(),
                pred_depth=pred_depth.detach().numpy(), pred_normal=pred_normal.detach().numpy(),
                pred_rgb=pred_rgb.detach().numpy(), scales=scales.detach().numpy())

    # (1) the strategy object on its own, defaults of regularization_strategy.py:126-144
    strat = reg_mod.DNRegularization()
    save["defaults"] = np.array([strat.depth_tolerance, strat.depth_lambda, strat.normal_lambda], dtype=np.float64)
    gt_img = image.clamp(min=10 / 255.0)
    val = strat(pred_depth=pred_depth, gt_depth=gt_depth, pred_normal=pred_normal, gt_normal=gt_normal, scales=scales, gt_img=gt_img)
    gd, gn, gs = torch.autograd.grad(val, [pred_depth, pred_normal, scales])
    save.update(reg_value=np.float64(val.item()), reg_v_depth=gd.numpy(), reg_v_normal=gn.numpy(), reg_v_scales=gs.numpy())
    # its three parts, for the record
    save["reg_depth_term"] = np.float64(strat.get_depth_loss(pred_depth, gt_depth, gt_img=gt_img).item())
    save["reg_normal_term"] = np.float64(strat.get_normal_loss(pred_normal, gt_normal).item())
    save["reg_scale_term"] = np.float64(strat.get_scale_loss(scales=scales).item())

    # (2) get_loss_dict from the reference's text
    text = extract_method(os.path.join(REF, "dn_splatter/dn_model.py"), "DNSplatterModel", "get_loss_dict")
    rgb_term = (pred_rgb - image).abs().mean() * 0.8 + 0.05                # stand-in for nerfstudio's main_loss (recorded)
    src = ("class _Base:\n"
           "    def get_loss_dict(self, outputs, batch, metrics_dict=None):\n"
           "        return {'main_loss': _RGB_TERM, 'scale_reg': _SCALE_REG}\n"
           "class _M(_Base):\n" + "\n".join("    " + ln for ln in text.splitlines()) + "\n")
    from typing import Dict, List, Union
    ns = dict(torch=torch, Dict=Dict, List=List, Union=Union, _RGB_TERM=rgb_term, _SCALE_REG=torch.tensor(0.0),
              normal_from_depth_image=None, CONSOLE=_t.SimpleNamespace(log=lambda *a, **k: None))
    exec(compile(src, "dn_model.py::DNSplatterModel.get_loss_dict", "exec"), ns)
    me = ns["_M"]()
    me.config = _t.SimpleNamespace(normal_supervision="mono", use_depth_loss=True, regularization_strategy="dn-splatter")
    me.regularization_strategy = strat
    me.get_gt_img = lambda im: im                                # num_downscales == 0: the image as it is
    me.scales = scales
    me.device = torch.device("cpu")
    outputs = {"rgb": pred_rgb, "depth": pred_depth, "normal": pred_normal, "surface_normal": rnd(H, W, 3)}
    batch = {"image": image, "mono_depth": gt_depth, "normal": gt_normal.clone()}
    ld = me.get_loss_dict(outputs, batch)
    main = ld["main_loss"]
    gd2, gn2, gs2, gr2 = torch.autograd.grad(main, [pred_depth, pred_normal, scales, pred_rgb])
    save.update(loss_dict_main=np.float64(main.item()), loss_dict_rgb_term=np.float64(rgb_term.item()),
                loss_dict_v_depth=gd2.numpy(), loss_dict_v_normal=gn2.numpy(), loss_dict_v_scales=gs2.numpy())
    # with a mask in the batch (dn_model.py:646-659): depth, both normals and the ground truths are multiplied by it
    mask = (rnd(H, W, 1) > 0.3).float()
    pd3 = pred_depth.detach().clone().requires_grad_(True)
    pn3 = pred_normal.detach().clone().requires_grad_(True)
    outputs3 = {"rgb": pred_rgb, "depth": pd3, "normal": pn3, "surface_normal": outputs["surface_normal"]}
    batch3 = {"image": image, "mono_depth": gt_depth, "normal": gt_normal.clone(), "mask": mask}
    main3 = me.get_loss_dict(outputs3, batch3)["main_loss"]
    gd3, gn3 = torch.autograd.grad(main3, [pd3, pn3])
    save.update(mask=mask.numpy(), masked_main=np.float64(main3.item()), masked_v_depth=gd3.numpy(), masked_v_normal=gn3.numpy())
    np.savez_compressed(path, **save)
    print(path, os.path.getsize(path) // 1024, "KiB", {k: float(save[k]) for k in ("reg_value", "loss_dict_main", "masked_main")})


# ------------------------------------------------------------------------------------------------------------------
# DNSplatterModel.refinement_after (N3), executed from the reference's own text


def refinement_case(path, N=200, seed=23):
    """dn_model.py:271-386 is cut out and executed as a method.  The five helpers it inherits from nerfstudio's SplatfactoModel
    (split_gaussians, dup_gaussians, cull_gaussians, dup_in_all_optim, remove_from_all_optim — not vendored) are RECORDING
    stand-ins that delegate to oracle/densify_ref.py's restatement of them; what this pins is the reference's own text: which
    masks it forms from which thresholds and statistics, in which order it calls the helpers with which arguments, how it
    concatenates parameters / max_2Dsize, the splits_mask it hands to the cull, the opacity-reset branch, the statistics reset.
    One scenario per branch."""
    import types as _t

    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import densify_ref as ref

    text = extract_method(os.path.join(REF, "dn_splatter/dn_model.py"), "DNSplatterModel", "refinement_after")
    ns = dict(torch=torch, Optimizers=object)
    exec(compile(text, "dn_model.py::DNSplatterModel.refinement_after", "exec"), ns)
    reference_refinement_after = ns["refinement_after"]

    class RefModel(ref.Model):
        def __init__(self, *a, noise_seed=0):
            super().__init__(*a)
            self.__dict__["log"] = []
            self.__dict__["device"] = torch.device("cpu")
            self.__dict__["_gen"] = torch.Generator().manual_seed(noise_seed)
            self.__dict__["noise"] = None

        def split_gaussians(self, split_mask, samps):
            self.log.append(("split_gaussians", split_mask.clone(), int(samps)))
            self.__dict__["noise"] = torch.randn(samps * int(split_mask.sum()), 3, generator=self._gen)
            return super().split_gaussians(split_mask, samps, self.noise)

        def dup_gaussians(self, dup_mask):
            self.log.append(("dup_gaussians", dup_mask.clone()))
            return super().dup_gaussians(dup_mask)

        def cull_gaussians(self, extra_cull_mask=None):
            self.log.append(("cull_gaussians", None if extra_cull_mask is None else extra_cull_mask.clone()))
            deleted = super().cull_gaussians(extra_cull_mask)
            self.log.append(("cull_result", deleted.clone()))
            return deleted

        def dup_in_all_optim(self, optimizers, idcs, n):
            self.log.append(("dup_in_all_optim", idcs.clone(), int(n)))
            return super().dup_in_all_optim(idcs, n)

        def remove_from_all_optim(self, optimizers, deleted_mask):
            self.log.append(("remove_from_all_optim", deleted_mask.clone()))
            return super().remove_from_all_optim(deleted_mask)

    g = torch.Generator().manual_seed(seed)
    gp = {"means": (torch.rand(N, 3, generator=g) - 0.5) * 6, "scales": torch.randn(N, 3, generator=g) * 1.5 - 4.0,
          "quats": torch.randn(N, 4, generator=g), "features_dc": torch.rand(N, 3, generator=g),
          "features_rest": torch.randn(N, 15, 3, generator=g) * 0.1, "opacities": torch.randn(N, 1, generator=g) * 2 - 1,
          "normals": torch.randn(N, 3, generator=g)}
    xys_grad_norm = torch.rand(N, generator=g) * 0.01
    vis_counts = torch.randint(1, 5, (N,), generator=g).float()
    max_2Dsize = torch.rand(N, generator=g) * 0.1
    adam = {k: {"exp_avg": torch.randn(v.shape, generator=g), "exp_avg_sq": torch.rand(v.shape, generator=g)} for k, v in gp.items() if k != "normals"}
    base = dict(warmup_length=500, refine_every=100, reset_alpha_every=30, stop_split_at=15000, stop_screen_size_at=4000,
                densify_grad_thresh=0.0008, densify_size_thresh=0.01, split_screen_size=0.05, n_split_samples=2,
                cull_alpha_thresh=0.1, cull_scale_thresh=0.5, cull_screen_size=0.15, continue_cull_post_densification=True)
    scenarios = [("densify_screen", 3500, {}), ("densify_early", 2500, {}), ("opacity_reset", 3100, {}), ("densify_no_screen", 6500, {}),
                 ("cull_only", 16000, {}), ("no_cull", 16000, dict(continue_cull_post_densification=False)),
                 ("big_thresholds", 3500, dict(cull_alpha_thresh=0.005)), ("warmup", 400, {})]
    save = dict(N=N, num_train_data=100, last_size=np.array([480, 640]), scenario_names=np.array([s[0] for s in scenarios]),
                xys_grad_norm=xys_grad_norm.numpy(), vis_counts=vis_counts.numpy(), max_2Dsize=max_2Dsize.numpy())
    for k, v in gp.items():
        save["param_" + k] = v.numpy()
    for k, st in adam.items():
        save["adam_avg_" + k] = st["exp_avg"].numpy(); save["adam_sq_" + k] = st["exp_avg_sq"].numpy()
    save["config_keys"] = np.array(sorted(base))
    for name, step, kw in scenarios:
        cfg = _t.SimpleNamespace(**{**base, **kw})
        m = RefModel(gp, cfg, step, 100, (480, 640), xys_grad_norm.clone(), vis_counts.clone(), max_2Dsize.clone(), adam, noise_seed=step)
        opt_state = {} if m.adam is None else m.adam["opacities"]
        p_key = torch.zeros(1)
        optimizers = _t.SimpleNamespace(optimizers={"opacities": _t.SimpleNamespace(param_groups=[{"params": [p_key]}], state={p_key: opt_state})})
        reference_refinement_after(m, optimizers, step)
        save[name + "__step"] = step
        save[name + "__config"] = np.array([float(getattr(cfg, k)) for k in sorted(base)])
        save[name + "__calls"] = np.array([e[0] for e in m.log])
        for i, e in enumerate(m.log):
            for j, a in enumerate(e[1:]):
                if a is None:
                    save[f"{name}__call{i}_arg{j}_none"] = True
                elif torch.is_tensor(a):
                    save[f"{name}__call{i}_arg{j}"] = a.numpy()
                else:
                    save[f"{name}__call{i}_arg{j}"] = a
        if m.noise is not None:
            save[name + "__noise"] = m.noise.numpy()
        for k, v in m.gauss_params.items():
            save[f"{name}__out_{k}"] = v.detach().numpy()
        for k, st in m.adam.items():
            save[f"{name}__adam_avg_{k}"] = st["exp_avg"].numpy(); save[f"{name}__adam_sq_{k}"] = st["exp_avg_sq"].numpy()
        save[name + "__stats_reset"] = np.array([m.xys_grad_norm is None, m.vis_counts is None, m.max_2Dsize is None])
        print(f"  refinement {name:18s} step {step:6d}: calls {[e[0] for e in m.log]}, {N} -> {m.gauss_params['means'].shape[0]} Gaussians")
    np.savez_compressed(path, **save)
    print(path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    cam_mod, nrm_mod, los_mod = load_reference()
    helpers_case(os.path.join(HERE, "reference_helpers.npz"))
    depth_normal_case(nrm_mod, os.path.join(HERE, "reference_depth_normal.npz"))
    loss_case(los_mod, os.path.join(HERE, "reference_losses.npz"))
    get_outputs_case(nrm_mod, os.path.join(HERE, "reference_get_outputs.npz"))
    regularization_case(los_mod, os.path.join(HERE, "reference_regularization.npz"))
    refinement_case(os.path.join(HERE, "reference_refinement.npz"))
