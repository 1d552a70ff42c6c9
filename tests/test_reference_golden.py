"""Parity against vectors produced by THE REFERENCE's own Python code (tests/golden/make_reference_golden.py):
the depth -> normal step of get_outputs (dn_model.py:589-603) and the per-pixel loss terms of losses.py.  These are the
rows of the path whose reference implementation is plain PyTorch and therefore runs here; the compositing rows stay
"parity unpinned" (gsplat 1.0.0 needs CUDA)."""
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name):
    return np.load(os.path.join(HERE, "golden", name))


def test_depth_to_normal_restatement_equals_the_reference():
    """model.normal_from_depth_image (the host mirror's A12) == utils/normal_utils.py on the reference's output."""
    import dn_splatter_amd as dns
    from dn_splatter_amd import model

    g = _load("reference_depth_normal.npz")
    W, H = int(g["W"]), int(g["H"])
    depth = torch.from_numpy(g["depth"])
    n = model.normal_from_depth_image(depth[..., None], float(g["fx"]), float(g["fy"]), float(g["cx"]), float(g["cy"]),
                                      (W, H), torch.eye(4))
    ref = torch.from_numpy(g["normals_raw"])
    assert n.shape == ref.shape == (H, W, 3)
    assert float((n - ref).abs().max()) < 2e-6
    sn = (1 + n @ torch.diag(torch.tensor([1.0, -1.0, -1.0]))) / 2
    assert float((sn - torch.from_numpy(g["surface_normal"])).abs().max()) < 2e-6
    assert dns is not None


def test_loss_terms_equal_the_reference():
    """torch_losses (what bench.py --losses torch times and what dnsplat_dn_loss is checked against) == losses.py."""
    from dn_splatter_amd import torch_losses as tl

    g = _load("reference_losses.npz")
    pred = torch.from_numpy(g["pred"]).requires_grad_(True)
    gt, rgb, mask = torch.from_numpy(g["gt"]), torch.from_numpy(g["rgb"]), torch.from_numpy(g["mask"])
    pn = torch.from_numpy(g["pred_normal"]).requires_grad_(True)

    def check(value, wrt, key, tol=2e-6):
        assert abs(float(value) - float(g[key])) < tol * max(1.0, abs(float(g[key]))), key
        (gr,) = torch.autograd.grad(value, wrt)
        ref = torch.from_numpy(g[key + "_grad"])
        assert float((gr - ref).abs().max()) < tol * max(1.0, float(ref.abs().max())), key + " gradient"

    check(tl.edge_aware_log_l1(pred, gt, rgb, mask), pred, "edge_aware_logl1_masked")
    check(tl.edge_aware_log_l1(pred, gt, rgb, None), pred, "edge_aware_logl1_nomask")
    check(tl.tv_loss(pn), pn, "tv_normal")
    # the plain terms the strategy falls back to (losses.py:154-185): trivial, pinned for completeness
    check(torch.log(1 + (pred - gt).abs()).mean(), pred, "logl1_scalar")
    check((pred - gt).abs().mean(), pred, "l1_scalar")


@pytest.mark.gpu
def test_hip_depth_normals_kernel_equals_the_reference():
    """dnsplat_dn_depth_normals (postops.hip) on the reference's own depth -> surface-normal vector."""
    import dn_splatter_amd as dns
    from dn_splatter_amd import _lib, _ops

    g = _load("reference_depth_normal.npz")
    W, H = int(g["W"]), int(g["H"])
    dev = "cuda:0"
    depth = torch.from_numpy(g["depth"]).to(dev).contiguous()
    alphas = torch.ones(H, W, device=dev)
    dmax = depth.max().reshape(1).contiguous()
    depth_out = torch.empty_like(depth)
    sn = torch.empty(H, W, 3, device=dev)
    _lib.check(_lib.lib().dnsplat_dn_depth_normals(W, H, float(g["fx"]), float(g["fy"]), float(g["cx"]), float(g["cy"]),
                                                   _ops._ptr(depth), _ops._ptr(alphas), _ops._ptr(dmax), _ops._ptr(depth_out),
                                                   _ops._ptr(sn), _ops._stream()), "dnsplat_dn_depth_normals")
    torch.cuda.synchronize()
    assert torch.equal(depth_out, depth)
    ref = torch.from_numpy(g["surface_normal"]).to(dev)
    assert float((sn - ref).abs().max()) < 5e-6
    assert dns is not None


@pytest.mark.gpu
def test_hip_loss_kernels_equal_the_reference_terms():
    """dnsplat_dn_loss is validated against torch_losses.dn_loss elsewhere; here its depth and normal parts are isolated
    (zero rgb error, constant images) so that what remains are exactly EdgeAwareLogL1 x (1 + depth_lambda) and
    L1 + TV of the normals — compared with the reference's values for the same inputs."""
    from dn_splatter_amd import fused_loss, torch_losses as tl

    g = _load("reference_losses.npz")
    dev = "cuda:0"
    H, W = int(g["H"]), int(g["W"])
    pred = torch.from_numpy(g["pred"]).to(dev)
    gt = torch.from_numpy(g["gt"]).to(dev)
    rgb = torch.from_numpy(g["rgb"]).to(dev).clamp(min=10 / 255.0)     # the strategy clamps the image it takes edges from
    pn = torch.from_numpy(g["pred_normal"]).to(dev)
    scales = torch.zeros(4, 3, device=dev)
    out = {"rgb": rgb.clone().requires_grad_(True), "depth": pred.clone().requires_grad_(True),
           "normal": pn.clone().requires_grad_(True), "accumulation": torch.ones(H, W, 1, device=dev)}
    batch = {"image": rgb, "mono_depth": gt, "normal": pn.detach().clone()}
    loss = fused_loss.dn_loss_fused(out, batch, scales, counts=fused_loss.depth_counts(gt))
    ref_torch = tl.dn_loss({k: v.detach() for k, v in out.items()}, batch, scales)
    torch.cuda.synchronize()
    assert abs(float(loss) - float(ref_torch)) < 1e-5 * max(1.0, abs(float(ref_torch)))
    # reference terms: rgb error 0 (pred == gt image: L1 = 0, SSIM = 1), normal L1 = 0, so
    # loss = 1.2 x EdgeAwareLogL1(pred, gt, clamp(rgb), gt > 0.1) + TV(normal) + mean(min exp(scales)) = ... + 1
    valid = (gt > 0.1)
    ea = tl.edge_aware_log_l1(pred.cpu(), gt.cpu(), rgb.cpu(), valid.cpu())
    want = 1.2 * float(ea) + float(g["tv_normal"]) + 1.0
    assert abs(float(loss) - want) < 2e-5 * want
