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


# ------------------------------------------------------------------------------------------------------------------
# DNSplatterModel.get_outputs executed from the reference's own text (reference_get_outputs.npz): pins A0 (what is
# handed to gsplat.rasterization), A7 (per-Gaussian normals, dn_model.py:543-560) and A9 (the per-pixel post-ops,
# dn_model.py:526-537, 577-578) — values and autograd gradients — to the reference instead of to our restatement.


def _reference_frame():
    from dn_splatter_amd.model import Camera

    g = _load("reference_get_outputs.npz")
    params = {k[6:]: torch.from_numpy(g[k]).clone() for k in g.files if k.startswith("param_")}
    cam = Camera(torch.from_numpy(g["c2w"]), float(g["fx"]), float(g["fy"]), float(g["cx"]), float(g["cy"]), int(g["W"]), int(g["H"]))
    return g, params, cam


def test_host_mirror_equals_get_outputs_of_the_reference():
    """model.DNSplatterRenderer.get_outputs(fused=False) around the same two stand-ins for the gsplat calls: every
    argument it hands to them, every output image and every gradient equals what the reference's own method produced."""
    import dn_splatter_amd as dns

    g, params, cam = _reference_frame()
    N, W, H = int(g["N"]), int(g["W"]), int(g["H"])
    params = {k: v.requires_grad_(True) for k, v in params.items()}
    render = torch.from_numpy(g["render"]).requires_grad_(True)
    alpha = torch.from_numpy(g["alpha"]).requires_grad_(True)
    mix = torch.from_numpy(g["mix"])
    calls = {}

    def rasterization(**kw):
        calls["rasterization"] = kw
        info = {"means2d": torch.zeros(1, N, 2, requires_grad=True), "radii": torch.ones(1, N, dtype=torch.int32),
                "depths": torch.ones(1, N), "conics": torch.ones(1, N, 3), "tiles_per_gauss": torch.ones(1, N, dtype=torch.int32)}
        return render, alpha, info

    def rasterize_gaussians(xys, depths, radii, conics, num_tiles_hit, colors, opacity, img_height, img_width, block_width,
                            background=None, return_alpha=False):
        colors.retain_grad()
        calls["legacy"] = dict(xys=xys, colors=colors, opacity=opacity, background=background, block_width=block_width)
        return (mix @ colors).reshape(img_height, img_width, 3) + 0.2

    m = dns.DNSplatterRenderer(params, fused=False, rasterization_fn=rasterization, rasterize_gaussians_fn=rasterize_gaussians)
    m.step = int(g["step"])
    out = m.get_outputs(cam)
    kw = calls["rasterization"]
    close = lambda a, b, tol=2e-6: float((torch.as_tensor(a).detach() - torch.from_numpy(np.asarray(b))).abs().max()) <= tol * max(1.0, float(np.abs(b).max()))  # noqa: E731
    # A0: the activations and the call arguments (dn_model.py:495-513)
    for key in ("quats", "scales", "opacities", "colors", "viewmats", "Ks"):
        assert kw[key].shape == g["call_" + key].shape, key
        assert close(kw[key], g["call_" + key]), "rasterization argument " + key
    assert kw["sh_degree"] == int(g["call_sh_degree"])
    assert [kw["width"], kw["height"], kw["tile_size"], kw["near_plane"], kw["far_plane"]] == list(g["call_scalars"])
    assert [kw["packed"], kw["sparse_grad"], kw["absgrad"], kw["render_mode"] == "RGB+ED", kw["rasterize_mode"] == "classic"] == list(g["call_flags"])
    # A7: the per-Gaussian normals handed to the legacy pass and stored in gauss_params (dn_model.py:543-575)
    lg = calls["legacy"]
    assert close(lg["colors"], g["normals_cam"]) and close(params["normals"], g["normals_world"])
    assert close(lg["opacity"], g["legacy_opacity"])
    assert lg["xys"].requires_grad == bool(g["legacy_xys_requires_grad"]) and (lg["background"] is None) == bool(g["legacy_background_is_none"])
    assert lg["block_width"] == int(g["legacy_block_width"])
    # A9 / A12: the output dict (dn_model.py:605-612)
    for k in ("rgb", "depth", "normal", "surface_normal", "accumulation", "background"):
        assert out[k].shape == g["out_" + k].shape, k
        assert close(out[k], g["out_" + k]), k
    keys = ("rgb", "depth", "normal", "accumulation")
    torch.autograd.backward([out[k] for k in keys], [torch.from_numpy(g["cot_" + k]) for k in keys])
    assert close(render.grad, g["v_render"]) and close(alpha.grad, g["v_alpha"])
    assert close(lg["colors"].grad, g["v_normals_cam"]) and close(params["quats"].grad, g["v_quats"], 1e-5)
    for k, none in zip(("means", "scales", "features_dc", "features_rest", "opacities"), g["grad_is_none"]):
        assert (params[k].grad is None) == bool(none), k


@pytest.mark.gpu
def test_hip_gaussian_normals_equal_the_reference():
    """A7 in the projection kernels: normals_world (dn_model.py:558) and the camera-frame normal channels of the splat
    records == what the reference's text derived, and dnsplat_project_bwd turns the reference's cotangent of those normals
    into the reference's quaternion gradient (visible Gaussians: the real second pass gives culled ones no gradient)."""
    from dn_splatter_amd import _ops, fused
    from dn_splatter_amd._ops import ProjCfg

    g, params, cam = _reference_frame()
    dev = "cuda:0"
    N, W, H = int(g["N"]), int(g["W"]), int(g["H"])
    p = {k: v.to(dev).requires_grad_(True) for k, v in params.items()}
    c2w = cam.camera_to_worlds.to(dev)
    viewmat, K, nf = _ops.camera_prepare(c2w[0], cam.fx, cam.fy, cam.cx, cam.cy)
    assert float((viewmat.cpu() - torch.from_numpy(g["call_viewmats"][0])).abs().max()) < 2e-6
    assert float((K.cpu() - torch.from_numpy(g["call_Ks"][0])).abs().max()) < 2e-6
    assert float((nf.cpu() - fused.normal_frame_from_c2w(c2w[0]).cpu()).abs().max()) < 1e-6
    cfg = ProjCfg(width=W, height=H, scales_are_log=True, opacities_are_logit=True, sh_degree=int(g["call_sh_degree"]),
                  with_depth=True, with_normals=True, want_normals_world=True)
    pr = _ops.project(p["means"], p["quats"], p["scales"], p["opacities"].reshape(N), sh0=p["features_dc"], shN=p["features_rest"],
                      viewmat=viewmat, K=K, normal_frame=nf, cfg=cfg)
    vis = (pr["radii"][0] > 0).cpu()
    assert 40 <= int(vis.sum()) < N
    assert float((pr["normals_world"][0].cpu() - torch.from_numpy(g["normals_world"])).abs().max()) < 2e-6
    rec = pr["splats"].detach().cpu()
    assert float((rec[vis][:, 10:13] - torch.from_numpy(g["normals_cam"])[vis]).abs().max()) < 2e-6
    # A0 inside the kernel: opacity column = sigmoid(logit) as the reference hands it to gsplat
    assert float((rec[vis][:, 5] - torch.from_numpy(g["call_opacities"])[vis]).abs().max()) < 2e-6
    # backward: only the normal channels carry a cotangent
    v = torch.zeros(N, 16, device=dev)
    v[:, 10:13] = torch.from_numpy(g["v_normals_cam"]).to(dev)
    pr["splats"].backward(v)
    torch.cuda.synchronize()
    ref = torch.from_numpy(g["v_quats"])
    got = p["quats"].grad.cpu()
    assert float((got[vis] - ref[vis]).abs().max()) < 1e-5 * max(1.0, float(ref.abs().max()))
    assert float(got[~vis].abs().max()) == 0.0
    for k in ("means", "scales", "opacities"):
        assert float(p[k].grad.abs().max()) == 0.0, k      # argmin / flip are not differentiable, means are detached (dn_model.py:551)


@pytest.mark.gpu
def test_hip_postops_equal_the_reference_pinned_torch_postops():
    """A9 in the compositing kernels: the HIP epilogue / backward prologue (fused_postops=True) against the torch post-ops
    of the host mirror — the code test_host_mirror_equals_get_outputs_of_the_reference pins to the reference's text — on
    the SAME raw composite (same fused pass, same GPU), values and gradients."""
    import dn_splatter_amd as dns
    from dn_splatter_amd import synthetic

    dev = "cuda:0"
    N, W, H = 6000, 320, 240
    gp = synthetic.make_gauss_params(N, sh_rest_std=0.3, seed=4)
    gp["features_dc"] = gp["features_dc"].detach() * 3 - 1          # colours beyond the upper corner of clamp(rgb, 0, 1)
    # small, fairly opaque splats: ~5 % of the pixels stay empty (alpha == 0: they take the depth fill of dn_model.py:533-537)
    gp["scales"] = gp["scales"].detach() + torch.randn(N, 3, generator=torch.Generator().manual_seed(5)) * 0.5 - 2.5
    gp["opacities"] = gp["opacities"].detach() + 3.0
    cam = synthetic.orbit_camera(2, width=W, height=H, focal=200.0).to(dev)
    keys = ("rgb", "depth", "normal", "accumulation")
    gen = torch.Generator().manual_seed(6)
    cot = None
    res = {}
    for mode in (False, True):
        p = {k: v.detach().to(dev).clone().requires_grad_(k != "normals") for k, v in gp.items()}
        out = dns.DNSplatterRenderer(p, fused=True, fused_postops=mode).get_outputs(cam)
        if cot is None:
            cot = {k: (torch.rand(out[k].shape, generator=gen) * 2 - 1).to(dev) for k in keys}
        torch.autograd.backward([out[k] for k in keys], [cot[k] for k in keys])
        res[mode] = (out, p)
    torch.cuda.synchronize()
    acc = res[False][0]["accumulation"]
    assert float((acc == 0).float().mean()) > 0.01 and float((res[False][0]["rgb"] == 1).float().mean()) > 0.001
    for k in keys + ("surface_normal",):
        a, b = res[True][0][k], res[False][0][k]
        assert float((a - b).abs().max()) <= 1e-5 * max(1.0, float(b.abs().max())), k
    for k in ("means", "scales", "quats", "features_dc", "features_rest", "opacities"):
        a, b = res[True][1][k].grad, res[False][1][k].grad
        assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max()), k
