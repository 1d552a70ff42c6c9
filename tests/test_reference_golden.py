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
        assert abs(float(value.detach()) - float(g[key])) < tol * max(1.0, abs(float(g[key]))), key
        (gr,) = torch.autograd.grad(value, wrt)
        ref = torch.from_numpy(g[key + "_grad"])
        assert float((gr - ref).abs().max()) < tol * max(1.0, float(ref.abs().max())), key + " gradient"

    check(tl.edge_aware_log_l1(pred, gt, rgb, mask), pred, "edge_aware_logl1_masked")
    check(tl.edge_aware_log_l1(pred, gt, rgb, None), pred, "edge_aware_logl1_nomask")
    check(tl.tv_loss(pn), pn, "tv_normal")
    # the plain terms the strategy falls back to (losses.py:154-185): trivial, pinned for completeness
    check(torch.log(1 + (pred - gt).abs()).mean(), pred, "logl1_scalar")
    check((pred - gt).abs().mean(), pred, "l1_scalar")


def test_conventions_stated_by_the_reference_helpers():
    """dn_model.py's own module-level torch helpers (executed from their text: golden/reference_helpers.npz) state the conventions
    the un-vendored gsplat symbols are used with.  Pinned to them: the quaternion -> rotation map of the package, of the oracle and
    of the stand-in the get_outputs fixture was generated with (w, x, y, z; R, not its transpose: ``quat_to_rotmat(
    matrix_to_quaternion(R)) == R`` for the reference's own matrix_to_quaternion, :1554-1600), the conjugate as inverse (:1615-1626),
    Sigma^-1 = R diag(1 / s^2) R^T (:1603-1612), band 0 of the SH colour (SH2RGB, :1512-1517: C0 x coefficient + 0.5) and the seeded
    initial rotations (random_quat_tensor, :1497-1509)."""
    import dn_splatter_amd as dns
    from dn_splatter_amd import synthetic
    from oracle import oracle as orc

    g = _load("reference_helpers.npz")
    N, seed = int(g["N"]), int(g["seed"])
    R, q = torch.from_numpy(g["R"]), torch.from_numpy(g["quat_of_R"])
    assert torch.allclose(q.norm(dim=-1), torch.ones(N), atol=1e-5)
    for name, fn in (("package", dns.quat_to_rotmat), ("oracle", orc.quat_to_rotmat)):
        got = fn(q)
        assert float((got - R).abs().max()) < 5e-6, name
        assert float((fn(torch.from_numpy(g["quat_inverse"])) - R.transpose(1, 2)).abs().max()) < 5e-6, name
        M = got * (1.0 / torch.from_numpy(g["scale"]).clamp(min=1e-3))[:, None, :]
        ref = torch.from_numpy(g["inv_cov3d"])
        assert float((M @ M.transpose(1, 2) - ref).abs().max()) < 2e-5 * float(ref.abs().max()), name
    # the rotation really moves v1 onto v2 (rotate_vector_to_vector, the normal initialisation dn_model.py:200-218)
    v1, v2 = torch.from_numpy(g["v1"]), torch.from_numpy(g["v2"])
    moved = torch.bmm(dns.quat_to_rotmat(q), torch.nn.functional.normalize(v1, dim=-1)[:, :, None])[:, :, 0]
    assert float((moved - torch.nn.functional.normalize(v2, dim=-1)).abs().max()) < 1e-5
    # SH band 0: what gsplat's rasterization() makes of degree-0 coefficients, clamp_min(SH + 0.5, 0), is clamp_min(SH2RGB, 0)
    sh = torch.from_numpy(g["sh"])
    dirs = torch.nn.functional.normalize(torch.randn(N, 3, generator=torch.Generator().manual_seed(1)), dim=-1)
    col = orc.sh_fwd(0, dirs, sh[:, None, :].contiguous())
    assert float((col + 0.5 - torch.from_numpy(g["sh2rgb"])).abs().max()) < 1e-6
    # seeded initial rotations: same variates, same order, same formula
    mine = synthetic.random_quat_tensor(N, generator=torch.Generator().manual_seed(seed + 1))
    assert torch.equal(mine, torch.from_numpy(g["random_quats"]))


@pytest.mark.gpu
def test_hip_projection_follows_the_reference_helpers_conventions():
    """The same fixture through the HIP projection: Gaussians whose rotations are the reference's matrix_to_quaternion(R) report the
    column of R that belongs to their smallest scale as their normal (A7, dn_model.py:543-556) and degree-0 SH2RGB colours."""
    from dn_splatter_amd import _ops, fused, synthetic
    from dn_splatter_amd._ops import ProjCfg

    g = _load("reference_helpers.npz")
    dev = "cuda:0"
    N = int(g["N"])
    R, q = torch.from_numpy(g["R"]), torch.from_numpy(g["quat_of_R"])
    gen = torch.Generator().manual_seed(4)
    means = (torch.rand(N, 3, generator=gen) - 0.5) * 2.0
    scales = torch.from_numpy(g["scale"])
    sh0 = torch.from_numpy(g["sh"]) * 0.5
    cam = synthetic.orbit_camera(0, width=64, height=48, focal=40.0).to(dev)
    viewmat, K, nf = _ops.camera_prepare(cam.camera_to_worlds[0], cam.fx, cam.fy, cam.cx, cam.cy)
    cfg = ProjCfg(width=64, height=48, scales_are_log=True, opacities_are_logit=True, sh_degree=0, with_depth=True, with_normals=True,
                  want_normals_world=True)
    pr = _ops.project(means.to(dev), (q * 3.0).to(dev), torch.log(scales).to(dev), torch.zeros(N, device=dev), sh0=sh0.to(dev),
                      shN=torch.zeros(N, 15, 3, device=dev), viewmat=viewmat, K=K, normal_frame=nf, cfg=cfg)
    vis = (pr["radii"][0] > 0).cpu()
    assert int(vis.sum()) > N // 2
    axis = R[torch.arange(N), :, torch.argmin(scales, dim=-1)]                  # column argmin(scale) of R
    nw = pr["normals_world"][0].cpu()
    dots = (nw * axis).sum(-1)
    assert float((dots.abs() - 1.0).abs().max()) < 1e-5                          # +- that column (flipped towards the camera)
    to_cam = cam.camera_to_worlds[0, :3, 3].cpu() - means
    assert bool(((nw * to_cam).sum(-1) >= 0).all())
    rec = pr["splats"].detach().cpu()
    want = torch.from_numpy(g["sh2rgb"]) * 0.5 + 0.25                            # SH2RGB(0.5 sh) = 0.5 SH2RGB(sh) + 0.25
    assert float((rec[vis][:, 6:9] - want.clamp(min=0)[vis]).abs().max()) < 1e-6


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
    assert abs(float(loss.detach()) - float(ref_torch.detach())) < 1e-5 * max(1.0, abs(float(ref_torch)))
    # reference terms: rgb error 0 (pred == gt image: L1 = 0, SSIM = 1), normal L1 = 0, so
    # loss = 1.2 x EdgeAwareLogL1(pred, gt, clamp(rgb), gt > 0.1) + TV(normal) + mean(min exp(scales)) = ... + 1
    valid = (gt > 0.1)
    ea = tl.edge_aware_log_l1(pred.cpu(), gt.cpu(), rgb.cpu(), valid.cpu())
    want = 1.2 * float(ea) + float(g["tv_normal"]) + 1.0
    assert abs(float(loss.detach()) - want) < 2e-5 * want


@pytest.mark.gpu
def test_hip_loss_modules_equal_the_reference_modules():
    """fused_loss.EdgeAwareLogL1 / TVLoss (dnsplat_edge_aware_logl1 / dnsplat_tv_loss, what install_losses(model) puts in place of
    the reference's modules) against the values and gradients the reference's OWN classes produced (losses.py:187-224 with and
    without a mask, :279-295; tests/golden/reference_losses.npz), and the swap itself on a stand-in model."""
    import dn_splatter_amd as dns
    from dn_splatter_amd import fused_loss

    g = _load("reference_losses.npz")
    dev = "cuda:0"
    pred = torch.from_numpy(g["pred"]).to(dev)
    gt, rgb, mask = torch.from_numpy(g["gt"]).to(dev), torch.from_numpy(g["rgb"]).to(dev), torch.from_numpy(g["mask"]).to(dev)
    pn = torch.from_numpy(g["pred_normal"]).to(dev)

    def check(fn, x, key, tol=2e-6):
        x = x.clone().requires_grad_(True)
        value = fn(x)
        (3.0 * value).backward()                                            # an upstream factor reaches the gradient
        ref, ref_g = float(g[key]), torch.from_numpy(g[key + "_grad"]).to(dev)
        assert abs(float(value.detach()) - ref) < tol * max(1.0, abs(ref)), (key, float(value.detach()), ref)
        assert x.grad.shape == ref_g.shape
        assert float((x.grad / 3.0 - ref_g).abs().max()) < tol * max(1.0, float(ref_g.abs().max())), key + " gradient"

    ea, tv = fused_loss.EdgeAwareLogL1(), fused_loss.TVLoss()
    check(lambda x: ea(x, gt, rgb, mask), pred, "edge_aware_logl1_masked")
    check(lambda x: ea(x, gt, rgb, None), pred, "edge_aware_logl1_nomask")
    check(lambda x: tv(x), pn, "tv_normal")
    with torch.no_grad():                                                   # value only: no gradient planes are written
        assert abs(float(ea(pred, gt, rgb, mask)) - float(g["edge_aware_logl1_masked"])) < 2e-6
    with pytest.raises(NotImplementedError):
        fused_loss.EdgeAwareLogL1(implementation="per-pixel")
    # the swap: a model whose strategy holds the reference's kind of modules (class NAMES are what install_losses goes by)
    EdgeAwareLogL1 = type("EdgeAwareLogL1", (torch.nn.Module,), {"implementation": "scalar"})
    TVLoss = type("TVLoss", (torch.nn.Module,), {})
    Holder = type("Holder", (torch.nn.Module,), {})
    strategy = torch.nn.Module()
    strategy.depth_loss, strategy.normal_smooth_loss, strategy.normal_loss = Holder(), Holder(), Holder()
    strategy.depth_loss.loss, strategy.normal_smooth_loss.loss, strategy.normal_loss.loss = EdgeAwareLogL1(), TVLoss(), torch.nn.L1Loss()
    model = torch.nn.Module()
    model.regularization_strategy, model.ssim = strategy, torch.nn.Identity()
    swapped = dns.install_losses(model)
    assert swapped == ["regularization_strategy.depth_loss.loss", "regularization_strategy.normal_smooth_loss.loss", "ssim"]
    sc = torch.randn(1000, 3, device=dev, requires_grad=True)
    fused_loss.scale_reg(sc).backward()
    sc2 = sc.detach().clone().requires_grad_(True)
    want = torch.min(torch.exp(sc2), dim=1, keepdim=True)[0].mean()            # regularization_strategy.py:195-199
    want.backward()
    assert abs(float(fused_loss.scale_reg(sc.detach())) - float(want.detach())) < 1e-6 and float((sc.grad - sc2.grad).abs().max()) < 1e-9
    assert isinstance(strategy.depth_loss.loss, fused_loss.EdgeAwareLogL1) and isinstance(strategy.normal_smooth_loss.loss, fused_loss.TVLoss)
    assert isinstance(strategy.normal_loss.loss, torch.nn.L1Loss) and isinstance(model.ssim, fused_loss.SSIM)
    assert dns.install_losses(model) == ["ssim"]                            # idempotent (ssim reports itself, nothing is replaced twice)


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
        assert float((a - b).detach().abs().max()) <= 1e-5 * max(1.0, float(b.detach().abs().max())), k
    for k in ("means", "scales", "quats", "features_dc", "features_rest", "opacities"):
        a, b = res[True][1][k].grad, res[False][1][k].grad
        assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max()), k


# ------------------------------------------------------------------------------------------------------------------
# N2: how the loss terms are combined — DNRegularization.get_loss and DNSplatterModel.get_loss_dict, executed from the
# reference's own files (tests/golden/make_reference_golden.py: regularization_case)


def _reg_case():
    g = _load("reference_regularization.npz")
    t = lambda k: torch.from_numpy(g[k])       # noqa: E731
    return g, t


def test_regularization_combination_equals_the_reference():
    """torch_losses.regularization_term == DNRegularization() (regularization_strategy.py:146-199, its own defaults) as
    DNSplatterModel.get_loss_dict (dn_model.py:614-729) calls it: gt image clamped at 10/255, the (1 + depth_lambda) factor, L1 +
    TV on the normals, the min-scale term, and the mask products of dn_model.py:646-659.  Value and every gradient."""
    from dn_splatter_amd import torch_losses as tl

    g, t = _reg_case()
    assert np.allclose(g["defaults"], [0.1, 0.2, 0.1])        # depth_tolerance, depth_lambda, normal_lambda (unused by get_loss)
    pd, pn, sc = t("pred_depth").requires_grad_(True), t("pred_normal").requires_grad_(True), t("scales").requires_grad_(True)
    out = {"rgb": t("pred_rgb"), "depth": pd, "normal": pn}
    batch = {"image": t("image"), "mono_depth": t("gt_depth"), "normal": t("gt_normal")}
    v = tl.regularization_term(out, batch, sc)
    assert abs(float(v.detach()) - float(g["reg_value"])) < 2e-6
    assert abs(float(v.detach()) - (float(g["loss_dict_main"]) - float(g["loss_dict_rgb_term"]))) < 2e-6     # main = rgb + regularization
    assert abs(float(g["reg_depth_term"]) + float(g["reg_normal_term"]) + float(g["reg_scale_term"]) - float(g["reg_value"])) < 2e-6
    gd, gn, gs = torch.autograd.grad(v, [pd, pn, sc])
    for got, key in ((gd, "v_depth"), (gn, "v_normal"), (gs, "v_scales")):
        for pre in ("reg_", "loss_dict_"):
            ref = t(pre + key)
            assert float((got - ref).abs().max()) <= 2e-6 * max(1.0, float(ref.abs().max())), pre + key
    # with a mask in the batch
    pd3, pn3 = t("pred_depth").requires_grad_(True), t("pred_normal").requires_grad_(True)
    v3 = tl.regularization_term({"rgb": t("pred_rgb"), "depth": pd3, "normal": pn3}, dict(batch, mask=t("mask")), sc)
    assert abs(float(v3.detach()) - (float(g["masked_main"]) - float(g["loss_dict_rgb_term"]))) < 2e-6
    gd3, gn3 = torch.autograd.grad(v3, [pd3, pn3])
    assert float((gd3 - t("masked_v_depth")).abs().max()) < 2e-6 and float((gn3 - t("masked_v_normal")).abs().max()) < 2e-6
    # dn_loss = rgb term + regularization (dn_model.py:727); the rgb term is nerfstudio's L1 + SSIM (unpinned: SSIM is restated)
    full = tl.dn_loss(out, batch, sc)
    assert abs(float(full.detach()) - float(tl.rgb_term(out, batch).detach()) - float(v.detach())) < 1e-6


@pytest.mark.gpu
def test_hip_fused_loss_equals_the_reference_combination():
    """dnsplat_dn_loss (N2, fused_loss.dn_loss_fused): its depth / normal cotangents and its value minus the rgb term against the
    reference-executed DNRegularization / get_loss_dict vectors."""
    from dn_splatter_amd import fused_loss, torch_losses as tl

    g, t = _reg_case()
    dev = "cuda:0"
    pd, pn = t("pred_depth").to(dev).requires_grad_(True), t("pred_normal").to(dev).requires_grad_(True)
    rgb = t("pred_rgb").to(dev).requires_grad_(True)
    sc = t("scales").to(dev).requires_grad_(True)
    out = {"rgb": rgb, "depth": pd, "normal": pn}
    batch = {"image": t("image").to(dev), "mono_depth": t("gt_depth").to(dev), "normal": t("gt_normal").to(dev)}
    loss = fused_loss.dn_loss_fused(out, batch, sc)
    reg = float(loss.detach()) - float(tl.rgb_term({k: v.detach() for k, v in out.items()}, batch))
    assert abs(reg - float(g["reg_value"])) < 2e-5, (reg, float(g["reg_value"]))
    gd, gn, gs = torch.autograd.grad(loss, [pd, pn, sc])
    for got, key in ((gd, "loss_dict_v_depth"), (gn, "loss_dict_v_normal"), (gs, "loss_dict_v_scales")):
        ref = t(key)
        assert float((got.cpu() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max())), key


# ------------------------------------------------------------------------------------------------------------------
# N3: DNSplatterModel.refinement_after executed from the reference's own text (refinement_case)


def _refinement_scenarios():
    g = _load("reference_refinement.npz")
    names = [str(n) for n in g["scenario_names"]]
    keys = [str(k) for k in g["config_keys"]]
    return g, names, keys


def _scenario_inputs(g, name, keys, device="cpu"):
    from dn_splatter_amd import densify

    t = lambda k: torch.from_numpy(g[k]).to(device)        # noqa: E731
    gp = {k: t("param_" + k) for k in ("means", "scales", "quats", "features_dc", "features_rest", "opacities", "normals")}
    adam = {k: {"exp_avg": t("adam_avg_" + k), "exp_avg_sq": t("adam_sq_" + k)} for k in gp if k != "normals"}
    vals = dict(zip(keys, g[name + "__config"].tolist()))
    ints = ("warmup_length", "refine_every", "reset_alpha_every", "stop_split_at", "stop_screen_size_at", "n_split_samples")
    cfg = densify.RefineConfig(**{k: (int(v) if k in ints else (bool(v) if k == "continue_cull_post_densification" else float(v)))
                                  for k, v in vals.items()})
    stats = densify.DensifyStats(int(g["N"]), device)
    stats.xys_grad_norm, stats.vis_counts, stats.max_2Dsize = t("xys_grad_norm"), t("vis_counts"), t("max_2Dsize")
    return gp, adam, cfg, stats, int(g[name + "__step"])


def _calls(g, name):
    calls = [str(c) for c in g[name + "__calls"]]
    arg = lambda i, j: (None if f"{name}__call{i}_arg{j}_none" in g else g[f"{name}__call{i}_arg{j}"])      # noqa: E731
    return calls, arg


def test_refinement_restatement_equals_the_reference_text():
    """oracle/densify_ref.py's refinement_after (what the GPU kernels are tested against) == dn_model.py:271-386 executed from
    its source with recording stand-ins for the five nerfstudio helpers: same call sequence per branch, same masks, same
    parameters, Adam moments and statistics reset afterwards — all eight scenarios (densify with / without the screen rules,
    opacity reset, cull only, no cull, dn-splatter-big's thresholds, warm-up)."""
    from oracle import densify_ref as ref

    g, names, keys = _refinement_scenarios()
    expected_calls = {"densify": ["split_gaussians", "dup_gaussians", "dup_in_all_optim", "dup_in_all_optim", "cull_gaussians",
                                  "cull_result", "remove_from_all_optim"],
                      "cull": ["cull_gaussians", "cull_result", "remove_from_all_optim"], "none": []}
    for name in names:
        gp, adam, cfg, stats, step = _scenario_inputs(g, name, keys)
        calls, arg = _calls(g, name)
        kind = "densify" if "split_gaussians" in calls else ("cull" if calls else "none")
        assert calls == expected_calls[kind], (name, calls)
        noise = torch.from_numpy(g[name + "__noise"]) if name + "__noise" in g else torch.zeros(0, 3)
        m = ref.Model(gp, cfg, step, int(g["num_train_data"]), tuple(int(v) for v in g["last_size"]), stats.xys_grad_norm.clone(),
                      stats.vis_counts.clone(), stats.max_2Dsize.clone(), adam)
        m.refinement_after(lambda n: noise)
        for k in gp:
            ref_out = torch.from_numpy(g[f"{name}__out_{k}"])
            assert m.gauss_params[k].shape == ref_out.shape, (name, k)
            assert torch.equal(m.gauss_params[k], ref_out), (name, k)
        for k in adam:
            assert torch.equal(m.adam[k]["exp_avg"], torch.from_numpy(g[f"{name}__adam_avg_{k}"])), (name, k)
            assert torch.equal(m.adam[k]["exp_avg_sq"], torch.from_numpy(g[f"{name}__adam_sq_{k}"])), (name, k)
        if step > cfg.warmup_length:
            assert bool(g[name + "__stats_reset"].all()) and m.xys_grad_norm is None and m.max_2Dsize is None
        if kind == "densify":
            # the masks the reference formed (arguments it handed to the helpers) against the flag byte of the product's rule
            flags = ref.classify_torch(gp, stats, cfg, step, tuple(int(v) for v in g["last_size"]), True)
            splits, dups = torch.from_numpy(arg(0, 0)), torch.from_numpy(arg(1, 0))
            assert torch.equal((flags & 1) != 0, splits), name + ": split mask"
            assert torch.equal((flags & 2) != 0, dups), name + ": duplicate mask"
            assert int(arg(0, 1)) == cfg.n_split_samples and int(arg(2, 1)) == cfg.n_split_samples and int(arg(3, 1)) == 1
            assert torch.equal(torch.from_numpy(arg(2, 0)), torch.where(splits)[0]) and torch.equal(torch.from_numpy(arg(3, 0)), torch.where(dups)[0])
            extra = torch.from_numpy(arg(4, 0))           # splits_mask: the parents, then zeros for children and duplicates
            assert extra.shape[0] == splits.shape[0] + cfg.n_split_samples * int(splits.sum()) + int(dups.sum())
            assert torch.equal(extra[: splits.shape[0]], splits) and not bool(extra[splits.shape[0]:].any())
            deleted = torch.from_numpy(arg(5, 0))
            n0 = splits.shape[0]
            nch = cfg.n_split_samples * int(splits.sum())
            assert torch.equal(deleted[:n0], (flags & 4) != 0), name + ": cull of the originals"
            par = torch.where(splits)[0].repeat(cfg.n_split_samples)
            assert torch.equal(deleted[n0:n0 + nch], (flags[par] & 8) != 0), name + ": cull of the split children"
            assert torch.equal(deleted[n0 + nch:], (flags[torch.where(dups)[0]] & 16) != 0), name + ": cull of the duplicates"
        elif kind == "cull":
            assert arg(0, 0) is None
            flags = ref.classify_torch(gp, stats, cfg, step, tuple(int(v) for v in g["last_size"]), False)
            assert torch.equal(torch.from_numpy(arg(1, 0)), (flags & 4) != 0)


def _product_refinement_matches(g, name, keys, device, classify_fn=None, split_fn=None):
    from dn_splatter_amd import densify

    gp, adam, cfg, stats, step = _scenario_inputs(g, name, keys, device)
    noise = torch.from_numpy(g[name + "__noise"]).to(device) if name + "__noise" in g else None

    def split_with_reference_noise(params, parents, _own_noise):
        return (split_fn or densify.split_children)(params, parents, noise)

    new, new_adam, report = densify.refinement_after(gp, stats, cfg, step, int(g["num_train_data"]), tuple(int(v) for v in g["last_size"]),
                                                     adam_state=adam, seed=1, classify_fn=classify_fn,
                                                     split_fn=split_with_reference_noise)
    for k in gp:
        ref_out = torch.from_numpy(g[f"{name}__out_{k}"])
        assert new[k].shape == ref_out.shape, (name, k, new[k].shape, ref_out.shape)
        assert torch.allclose(new[k].cpu(), ref_out, atol=2e-6, rtol=1e-6), (name, k)
    for k in adam:
        assert torch.equal(new_adam[k]["exp_avg"].cpu(), torch.from_numpy(g[f"{name}__adam_avg_{k}"])), (name, k)
        assert torch.equal(new_adam[k]["exp_avg_sq"].cpu(), torch.from_numpy(g[f"{name}__adam_sq_{k}"])), (name, k)
    return report


def test_product_refinement_host_logic_equals_the_reference_text():
    """densify.refinement_after (flag byte -> index gathers; torch stand-ins for the two kernels on the CPU) ends with the same
    Gaussian set, in the same row order, and the same Adam moments as the reference's text in every scenario."""
    from oracle import densify_ref as ref

    g, names, keys = _refinement_scenarios()
    for name in names:
        _product_refinement_matches(g, name, keys, "cpu", classify_fn=ref.classify_torch, split_fn=ref.split_children_torch)


@pytest.mark.gpu
def test_hip_refinement_equals_the_reference_text():
    """The same with dnsplat_densify_classify / dnsplat_densify_split on the GPU."""
    g, names, keys = _refinement_scenarios()
    for name in names:
        _product_refinement_matches(g, name, keys, "cuda:0")
