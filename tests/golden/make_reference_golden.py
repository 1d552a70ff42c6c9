"""Golden vectors produced by THE REFERENCE's own Python code (not by the oracle).

Only the parts of the path that are plain PyTorch in the reference can be run in this container (gsplat 1.0.0 needs CUDA,
nerfstudio is not installed):

  * depth -> normal image      dn_splatter/utils/normal_utils.py:9-48 (pcd_to_normal, normal_from_depth_image) with
                               dn_splatter/utils/camera_utils.py:92-144 (get_means3d_backproj), as called at
                               dn_model.py:589-603 (c2w = identity, then @ diag(1,-1,-1) and (1 + n) / 2)
  * per-pixel loss terms       dn_splatter/losses.py:154-224 (L1, LogL1, EdgeAwareLogL1) and :279-295 (TVLoss), the terms
                               DNRegularization combines at regularization_strategy.py:146-199

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


if __name__ == "__main__":
    cam_mod, nrm_mod, los_mod = load_reference()
    depth_normal_case(nrm_mod, os.path.join(HERE, "reference_depth_normal.npz"))
    loss_case(los_mod, os.path.join(HERE, "reference_losses.npz"))
