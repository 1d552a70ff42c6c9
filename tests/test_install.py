"""``dn_splatter_amd.install(model_cls)`` — INTEGRATION.md section B as a function: the fused pass behind
``DNSplatterModel.get_outputs`` (``dn_splatter/dn_model.py:404-612``) without editing the reference file.

CPU part: a stand-in model class with the attributes the reference's method reads (the same stand-in the golden generator executes
the reference's own text against, tests/golden/make_reference_golden.py) and a recording stand-in for the HIP pass: what is handed
to it and what is left on the model must be what the reference's text hands to gsplat / leaves on the model
(tests/golden/reference_get_outputs.npz).  GPU part: the installed method against ``DNSplatterRenderer(fused=True)``."""
import os
import types

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


class StubCameras:
    """What get_outputs reads from nerfstudio.cameras.Cameras (dn_model.py:417-421, 474-479, 585-597)."""

    def __init__(self, c2w, fx, fy, cx, cy, W, H, device="cpu"):
        self.camera_to_worlds = c2w.to(device)
        t = lambda v: torch.tensor([[float(v)]], device=device)     # noqa: E731
        self.fx, self.fy, self.cx, self.cy = t(fx), t(fy), t(cx), t(cy)
        self.width, self.height = torch.tensor([[W]], device=device), torch.tensor([[H]], device=device)
        self.shape = (1,)
        self.metadata = {"cam_idx": 7}
        self.rescaled = []

    def rescale_output_resolution(self, f):
        self.rescaled.append(f)


def _model_class():
    class Model:
        def __init__(self, params, step=2500, **cfg):
            c = dict(use_binary_opacities=False, rasterize_mode="classic", sh_degree=3, sh_degree_interval=1000, predict_normals=True)
            c.update(cfg)
            self.config = types.SimpleNamespace(**c)
            self.training, self.step, self.crop_box = True, step, None
            self.gauss_params = dict(params)
            self.camera_optimizer = types.SimpleNamespace(apply_to_camera=lambda cam: cam.camera_to_worlds)

        means = property(lambda s: s.gauss_params["means"])
        quats = property(lambda s: s.gauss_params["quats"])
        scales = property(lambda s: s.gauss_params["scales"])
        opacities = property(lambda s: s.gauss_params["opacities"])
        features_dc = property(lambda s: s.gauss_params["features_dc"])
        features_rest = property(lambda s: s.gauss_params["features_rest"])

        def _get_downscale_factor(self):
            return 1

        def _get_background_color(self):
            return torch.tensor([0.1490, 0.1647, 0.2157], device=self.means.device)

        def get_outputs(self, camera):
            return {"original": True}

    return Model


def _fixture(device="cpu"):
    g = np.load(os.path.join(HERE, "golden", "reference_get_outputs.npz"))
    params = {k[6:]: torch.from_numpy(g[k]).clone().to(device).requires_grad_(True) for k in g.files if k.startswith("param_")}
    cam = StubCameras(torch.from_numpy(g["c2w"]), float(g["fx"]), float(g["fy"]), float(g["cx"]), float(g["cy"]), int(g["W"]), int(g["H"]), device)
    return g, params, cam


def test_install_replaces_get_outputs_and_hands_the_fused_pass_the_reference_arguments(dns, monkeypatch):
    from dn_splatter_amd import fused

    g, params, cam = _fixture()
    N, W, H = int(g["N"]), int(g["W"]), int(g["H"])
    Model = _model_class()
    original = Model.get_outputs
    assert dns.install(Model) is Model and dns.install(Model) is Model                 # idempotent
    assert Model.get_outputs is not original and Model._dnsplat_original_get_outputs is original
    calls = {}

    def render_dn_outputs(means, quats, scales, opacities, features_dc, features_rest, c2w, fx, fy, cx, cy, width, height, sh_degree,
                          background_rgb, absgrad=True, sigmoid_colors=False, **kw):
        calls.update(dict(means=means, quats=quats, scales=scales, opacities=opacities, features_dc=features_dc, features_rest=features_rest,
                          c2w=c2w, intr=(fx, fy, cx, cy), size=(width, height), sh_degree=sh_degree, background=background_rgb,
                          absgrad=absgrad, sigmoid_colors=sigmoid_colors))
        img = lambda c: torch.zeros(height, width, c)                          # noqa: E731
        out = {"rgb": img(3), "depth": img(1), "normal": img(3), "surface_normal": img(3), "accumulation": img(1)}
        radii = torch.ones(1, N, dtype=torch.int32)
        radii[0, :5] = 0
        info = {"means2d": (means[None, :, :2] * 1.0), "radii": radii, "depths": torch.ones(1, N), "conics": torch.ones(1, N, 3),
                "tiles_per_gauss": torch.ones(1, N, dtype=torch.int32), "normals_world": torch.full((N, 3), 0.25)}
        return out, info

    monkeypatch.setattr(fused, "render_dn_outputs", render_dn_outputs)
    m = Model(params, step=int(g["step"]))
    out = m.get_outputs(cam)
    # the RAW parameters go in (the activations of dn_model.py:497-499 run inside the kernels), the camera as the reference reads it
    for k in ("means", "quats", "scales", "opacities", "features_dc", "features_rest"):
        assert calls[k] is params[k], k
    assert torch.equal(calls["c2w"], cam.camera_to_worlds[0]) and calls["size"] == (W, H)
    assert calls["intr"] == (float(g["fx"]), float(g["fy"]), float(g["cx"]), float(g["cy"]))
    assert calls["sh_degree"] == int(g["call_sh_degree"]) and calls["absgrad"] and not calls["sigmoid_colors"]     # dn_model.py:487-490, :512
    assert torch.allclose(calls["background"], torch.from_numpy(g["out_background"]))
    assert cam.rescaled == [1.0, 1]                                            # dn_model.py:474, :479
    # what nerfstudio's after_train / refinement_after read back (dn_model.py:517-524, :531, :558, :580-583)
    assert m.xys.retains_grad and m.radii.shape == (N,) and m.last_size == (H, W) and m.camera is cam and m.camera_idx == 7
    assert torch.equal(m.vis_indices, torch.arange(5, N)) and torch.equal(m.gauss_params["normals"], torch.full((N, 3), 0.25))
    assert m.num_tiles_hit.shape == (1, N) and m.depths.shape == (1, N) and m.conics.shape == (1, N, 3)
    assert sorted(out) == ["accumulation", "background", "depth", "normal", "rgb", "surface_normal"]      # dn_model.py:605-612
    for k in ("rgb", "depth", "normal", "surface_normal", "accumulation", "background"):
        assert out[k].shape == g["out_" + k].shape, k
    # config.sh_degree == 0: sigmoid(colours), no SH (dn_model.py:491-493)
    m0 = Model(params, sh_degree=0)
    m0.get_outputs(cam)
    assert calls["sigmoid_colors"] and calls["sh_degree"] == 0
    # the one configuration left to the reference's own two-call body; not-a-camera (dn_model.py:416-418)
    assert Model(params, rasterize_mode="antialiased").get_outputs(cam) == {"original": True}
    assert m.get_outputs(object()) == {}
    with pytest.raises(ValueError):
        Model(params, rasterize_mode="blurry").get_outputs(cam)
    # evaluation crop box (dn_model.py:440-464): the cropped parameter rows go in, gauss_params["normals"] is left alone
    m.training = False
    m.crop_box = types.SimpleNamespace(within=lambda means: (torch.arange(means.shape[0]) % 2 == 0)[:, None])
    m.gauss_params["normals"] = "untouched"
    n_crop = (N + 1) // 2

    def render_cropped(means, *a, **kw):
        calls["n_cropped"] = means.shape[0]
        out, info = render_dn_outputs(means, *a, **kw)
        return out, {k: (v[:, :n_crop] if torch.is_tensor(v) and v.dim() >= 2 and v.shape[1] == N else v) for k, v in info.items()}

    monkeypatch.setattr(fused, "render_dn_outputs", render_cropped)
    m.get_outputs(cam)
    assert calls["n_cropped"] == n_crop and m.gauss_params["normals"] == "untouched"
    dns.uninstall(Model)
    assert Model.get_outputs is original and not hasattr(Model, "_dnsplat_original_get_outputs")


def test_install_ssim_swaps_the_module_and_refuses_cpu_tensors(dns):
    """install_ssim(model): model.ssim (dn_model.py:180) becomes fused_loss.SSIM, the old module is kept; the module has no CPU path
    and refuses what dnsplat_ssim does not compute (other windows / ranges, batches, two differentiated arguments)."""
    from dn_splatter_amd.fused_loss import SSIM

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.ssim = torch.nn.Identity()

    m = M()
    old = m.ssim
    assert dns.install_ssim(m) is m and isinstance(m.ssim, SSIM) and m._dnsplat_original_ssim is old
    dns.install_ssim(m)                                                         # idempotent
    assert m._dnsplat_original_ssim is old
    SSIM(data_range=1.0, kernel_size=11)                                       # the reference's constructor call
    for kw in ({"data_range": 255.0}, {"kernel_size": 7}, {"channel": 1}, {"size_average": False}):
        with pytest.raises(NotImplementedError):
            SSIM(**kw)
    a, b = torch.rand(1, 3, 20, 20), torch.rand(1, 3, 20, 20)
    with pytest.raises(dns.DnsplatError):
        m.ssim(a, b)                                                           # no CPU fallback
    with pytest.raises(NotImplementedError):
        m.ssim(torch.rand(2, 3, 20, 20), torch.rand(2, 3, 20, 20))
    with pytest.raises(NotImplementedError):
        m.ssim(a.requires_grad_(True), b.requires_grad_(True))


def test_install_losses_swaps_only_the_modules_it_knows(dns):
    """install_losses(model): the reference's EdgeAwareLogL1("scalar") and TVLoss inside the strategy's holder modules are replaced (by
    class name: the reference need not be importable), other implementations / loss types are left alone, the modules refuse CPU
    tensors (no CPU fallback) and what they do not compute."""
    from dn_splatter_amd import fused_loss

    EdgeAwareLogL1 = type("EdgeAwareLogL1", (torch.nn.Module,), {"implementation": "scalar"})
    PerPixel = type("EdgeAwareLogL1", (torch.nn.Module,), {"implementation": "per-pixel"})
    TVLoss = type("TVLoss", (torch.nn.Module,), {})
    Holder = type("Holder", (torch.nn.Module,), {})

    def model_with(depth_inner, smooth_inner):
        st = torch.nn.Module()
        st.depth_loss, st.normal_smooth_loss = Holder(), Holder()
        st.depth_loss.loss, st.normal_smooth_loss.loss = depth_inner, smooth_inner
        m = torch.nn.Module()
        m.regularization_strategy = st
        return m, st

    m, st = model_with(EdgeAwareLogL1(), TVLoss())
    assert dns.install_losses(m) == ["regularization_strategy.depth_loss.loss", "regularization_strategy.normal_smooth_loss.loss"]
    assert isinstance(st.depth_loss.loss, fused_loss.EdgeAwareLogL1) and isinstance(st.normal_smooth_loss.loss, fused_loss.TVLoss)
    assert dns.install_losses(m) == []
    m2, st2 = model_with(PerPixel(), torch.nn.L1Loss())
    assert dns.install_losses(m2) == [] and isinstance(st2.depth_loss.loss, PerPixel)
    assert dns.install_losses(torch.nn.Module()) == []                         # no strategy, no ssim: nothing to do
    # the scale term is a METHOD of the reference's strategies (regularization_strategy.py:195-199): patched on those classes only
    DNRegularization = type("DNRegularization", (torch.nn.Module,), {"get_scale_loss": lambda self, scales: scales.sum()})
    m3 = torch.nn.Module()
    m3.regularization_strategy = DNRegularization()
    assert dns.install_losses(m3) == ["regularization_strategy.get_scale_loss"] and dns.install_losses(m3) == []
    with pytest.raises(dns.DnsplatError):
        m3.regularization_strategy.get_scale_loss(scales=torch.zeros(4, 3))
    with pytest.raises(dns.DnsplatError):
        st.depth_loss.loss(torch.rand(8, 8, 1), torch.rand(8, 8, 1), torch.rand(8, 8, 3), None)
    with pytest.raises(dns.DnsplatError):
        st.normal_smooth_loss.loss(torch.rand(8, 8, 3))
    with pytest.raises(NotImplementedError):
        st.normal_smooth_loss.loss(torch.rand(2, 8, 8, 3))
    with pytest.raises(NotImplementedError):
        st.depth_loss.loss(torch.rand(8, 8, 1), torch.rand(8, 8, 1).requires_grad_(True), torch.rand(8, 8, 3), None)


@pytest.mark.gpu
def test_installed_get_outputs_equals_the_renderer_mirror_on_the_gpu(dns):
    """The installed method on a stand-in model == DNSplatterRenderer(fused=True).get_outputs (the method the parity suite holds to
    the reference sequence): images bit for bit, the state nerfstudio reads back, and the gradients of a seeded loss."""
    from dn_splatter_amd import synthetic

    dev = "cuda:0"
    N, W, H = 5000, 160, 96
    gp = synthetic.make_gauss_params(N, sh_rest_std=0.1, seed=3)
    cam_r = synthetic.orbit_camera(1, width=W, height=H, focal=100.0).to(dev)
    cam = StubCameras(cam_r.camera_to_worlds.cpu(), cam_r.fx, cam_r.fy, cam_r.cx, cam_r.cy, W, H, dev)
    keys = ("rgb", "depth", "normal", "accumulation")
    gen = torch.Generator().manual_seed(5)
    cot = {k: (torch.rand((H, W, 3 if k in ("rgb", "normal") else 1), generator=gen) * 2 - 1).to(dev) for k in keys}
    Model = dns.install(_model_class())
    try:
        for cfg in (dict(), dict(sh_degree=0)):
            rest = gp["features_rest"] if not cfg else torch.zeros(N, 0, 3)
            leaves = lambda: {k: (rest if k == "features_rest" else v).detach().to(dev).clone().requires_grad_(k != "normals")   # noqa: E731
                              for k, v in gp.items()}
            p_a, p_b = leaves(), leaves()
            m = Model(p_a, **cfg)
            out_a = m.get_outputs(cam)
            r = dns.DNSplatterRenderer(p_b, dns.RendererConfig(**cfg), fused=True)
            r.step = m.step
            out_b = r.get_outputs(cam_r)
            dns.set_deterministic(True)
            try:
                torch.autograd.backward([out_a[k] for k in keys], [cot[k] for k in keys])
                torch.autograd.backward([out_b[k] for k in keys], [cot[k] for k in keys])
            finally:
                dns.set_deterministic(False)
            torch.cuda.synchronize()
            for k in keys + ("surface_normal",):
                assert torch.equal(out_a[k], out_b[k]), (cfg, k)
            assert torch.equal(m.radii, r.radii) and torch.equal(m.xys, r.xys) and torch.equal(m.num_tiles_hit, r.num_tiles_hit)
            assert torch.equal(m.vis_indices, torch.where(r.radii > 0)[0]) and torch.equal(m.gauss_params["normals"], p_b["normals"])
            assert m.xys.grad is not None and torch.equal(m.xys.grad, r.xys.grad) and torch.equal(m.xys.absgrad, r.xys.absgrad)
            for k in ("means", "scales", "quats", "features_dc", "opacities"):
                assert torch.equal(p_a[k].grad, p_b[k].grad), (cfg, k)
    finally:
        dns.uninstall(Model)
