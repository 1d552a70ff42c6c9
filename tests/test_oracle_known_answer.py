"""Pins the CPU oracle (SURVEY.md §4 tiers T0/T1).

The reference ships no tests or golden vectors for this path and gsplat==1.0.0 cannot be imported here
(PARITY UNPINNED — see oracle/dnsplat_oracle.c), so the oracle is pinned by (T0) closed-form answers of the
published algorithm and (T1) an independently written dense torch implementation
differentiated by autograd in fp64.
"""
import math

import pytest
import torch

from _scenes import assert_close, assert_equal_int, cotangents, gsplat_inputs, rel_err, to_leaf
from oracle import dense_ref


def _axis_scene(dtype, f=50.0, z=4.0, s=0.2, o=0.7, W=64, H=64):
    means = torch.tensor([[0.0, 0.0, z]], dtype=dtype)
    quats = torch.tensor([[1.0, 0.0, 0.0, 0.0]], dtype=dtype)
    scales = torch.full((1, 3), s, dtype=dtype)
    opac = torch.tensor([o], dtype=dtype)
    viewmat = torch.eye(4, dtype=dtype)[None]
    K = torch.tensor([[[f, 0, W / 2], [0, f, H / 2], [0, 0, 1.0]]], dtype=dtype)
    return means, quats, scales, opac, viewmat, K


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_t0_single_isotropic_gaussian_on_axis(orc, dtype):
    W = H = 64
    f, z, s, o = 50.0, 4.0, 0.2, 0.7
    means, quats, scales, opac, viewmat, K = _axis_scene(dtype, f, z, s, o, W, H)
    colors = torch.tensor([[0.2, 0.5, 0.9]], dtype=dtype)
    r, a, info = orc.rasterization(means, quats, scales, opac, colors, viewmat, K, W, H, render_mode="RGB+ED")
    var = f * f * s * s / (z * z) + 0.3
    assert torch.allclose(info["means2d"][0, 0], torch.tensor([W / 2, H / 2], dtype=dtype))
    assert torch.allclose(info["conics"][0, 0], torch.tensor([1 / var, 0.0, 1 / var], dtype=dtype), rtol=1e-5)
    assert int(info["radii"][0, 0]) == math.ceil(3 * math.sqrt(var))
    assert float(info["depths"][0, 0]) == z
    # pixel (32,32) has its centre at (32.5,32.5): sigma = 0.5 * (0.25 + 0.25) / var
    alpha = min(0.999, o * math.exp(-0.25 / var))
    assert abs(float(a[0, 32, 32, 0]) - alpha) < 1e-6
    assert torch.allclose(r[0, 32, 32, :3], colors[0] * alpha, atol=1e-6)
    assert abs(float(r[0, 32, 32, 3]) - z) < 1e-5
    # tile bbox: radius 8 around (32,32) -> tiles [1,3) x [1,3) = 4 tiles, row-major emission
    rad = math.ceil(3 * math.sqrt(var))
    n_tiles = (math.ceil((32 + rad) / 16) - math.floor((32 - rad) / 16)) ** 2
    assert int(info["tiles_per_gauss"][0, 0]) == n_tiles
    assert info["flatten_ids"].tolist() == [0] * n_tiles
    tids = sorted(int(k) >> 32 for k in info["isect_ids"].tolist())
    assert tids == [ty * 4 + tx for ty in range(1, 3) for tx in range(1, 3)] if rad <= 16 else True
    depth_bits = torch.tensor([z], dtype=torch.float32).view(torch.int32).item()
    assert all((int(k) & 0xFFFFFFFF) == depth_bits for k in info["isect_ids"].tolist())


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_t0_two_stacked_gaussians_blend_front_to_back(orc, dtype):
    W = H = 32
    f = 40.0
    means = torch.tensor([[0.0, 0.0, 6.0], [0.0, 0.0, 3.0]], dtype=dtype)   # index 1 is nearer
    quats = torch.tensor([[1.0, 0, 0, 0], [1.0, 0, 0, 0]], dtype=dtype)
    scales = torch.full((2, 3), 0.5, dtype=dtype)
    opac = torch.tensor([0.6, 0.4], dtype=dtype)
    colors = torch.tensor([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]], dtype=dtype)
    viewmat = torch.eye(4, dtype=dtype)[None]
    K = torch.tensor([[[f, 0, W / 2], [0, f, H / 2], [0, 0, 1.0]]], dtype=dtype)
    r, a, info = orc.rasterization(means, quats, scales, opac, colors, viewmat, K, W, H, render_mode="RGB")
    va = [f * f * 0.25 / (z * z) + 0.3 for z in (6.0, 3.0)]
    al_far = 0.6 * math.exp(-0.25 / va[0])
    al_near = 0.4 * math.exp(-0.25 / va[1])
    # C = c_near a_near + c_far a_far (1 - a_near)
    assert abs(float(r[0, 16, 16, 1]) - al_near) < 1e-6
    assert abs(float(r[0, 16, 16, 0]) - al_far * (1 - al_near)) < 1e-6
    assert abs(float(a[0, 16, 16, 0]) - (1 - (1 - al_near) * (1 - al_far))) < 1e-6
    # nearer Gaussian first in every tile list
    fid = info["flatten_ids"].tolist()
    assert fid[0] == 1


def test_t0_sh_degree0_colour(orc):
    means, quats, scales, opac, viewmat, K = _axis_scene(torch.float64)
    coeffs = torch.zeros(1, 16, 3, dtype=torch.float64)
    coeffs[0, 0] = torch.tensor([0.3, -0.2, 1.5])
    r, a, _ = orc.rasterization(means, quats, scales, opac, coeffs, viewmat, K, 64, 64, sh_degree=0, render_mode="RGB")
    expect = torch.clamp(0.2820947917738781 * coeffs[0, 0] + 0.5, min=0.0)
    assert torch.allclose(r[0, 32, 32] / a[0, 32, 32], expect, atol=1e-9)


def test_t0_alpha_thresholds_and_saturation(orc):
    """alpha < 1/255 is skipped; T*(1-alpha) <= 1e-4 stops BEFORE applying; alpha is clamped at 0.999."""
    dtype = torch.float64
    W = H = 16
    f = 30.0
    n = 6
    means = torch.tensor([[0.0, 0.0, 2.0 + 0.1 * i] for i in range(n)], dtype=dtype)
    quats = torch.tensor([[1.0, 0, 0, 0]] * n, dtype=dtype)
    scales = torch.full((n, 3), 1.0, dtype=dtype)
    opac = torch.tensor([1.5] * n, dtype=dtype)   # o*vis > 0.999 at the centre -> alpha clamps to 0.999
    colors = torch.eye(n, dtype=dtype)[:, :6]
    viewmat = torch.eye(4, dtype=dtype)[None]
    K = torch.tensor([[[f, 0, W / 2], [0, f, H / 2], [0, 0, 1.0]]], dtype=dtype)
    r, a, info = orc.rasterization(means, quats, scales, opac, colors, viewmat, K, W, H, render_mode="RGB")
    # T after one = 1e-3; the second would give 1e-6 <= 1e-4 -> stop, not applied
    assert abs(float(a[0, 8, 8, 0]) - 0.999) < 1e-9
    assert float(r[0, 8, 8, 1]) == 0.0 and abs(float(r[0, 8, 8, 0]) - 0.999) < 1e-9
    # a faint Gaussian below 1/255 contributes nothing at all
    opac2 = torch.tensor([0.003] + [0.0] * (n - 1), dtype=dtype)
    r2, a2, _ = orc.rasterization(means, quats, scales, opac2, colors, viewmat, K, W, H, render_mode="RGB")
    assert float(a2.abs().max()) == 0.0 and float(r2.abs().max()) == 0.0


def test_t0_culling_rules(orc):
    dtype = torch.float32
    W, H, f = 64, 48, 40.0
    means = torch.tensor([
        [0.0, 0.0, 0.005],    # nearer than near_plane
        [0.0, 0.0, -1.0],     # behind
        [50.0, 0.0, 2.0],     # off-screen right
        [0.0, 0.0, 2.0],      # visible
    ], dtype=dtype)
    quats = torch.tensor([[1.0, 0, 0, 0]] * 4, dtype=dtype)
    scales = torch.full((4, 3), 0.05, dtype=dtype)
    radii, m2d, depths, conics, comp, tiles = orc.project_fwd(means, quats, scales, torch.eye(4), torch.tensor(
        [[f, 0, W / 2], [0, f, H / 2], [0, 0, 1.0]]), W, H)
    assert radii.tolist()[:3] == [0, 0, 0] and radii[3] > 0
    assert tiles.tolist()[:3] == [0, 0, 0] and tiles[3] > 0


def test_t0_quat_and_sh_helpers(orc, dns):
    assert [orc.num_sh_bases(d) for d in range(5)] == [1, 4, 9, 16, 25]
    assert [dns.num_sh_bases(d) for d in range(5)] == [1, 4, 9, 16, 25]
    q = torch.tensor([[2.0, 0, 0, 0], [0.0, 0, 0, 3.0], [1.0, 1.0, 0, 0]])
    R = orc.quat_to_rotmat(q)
    assert torch.allclose(R[0], torch.eye(3))
    assert torch.allclose(R[1], torch.diag(torch.tensor([-1.0, -1.0, 1.0])), atol=1e-6)    # 180 deg about z
    assert torch.allclose(R[2], torch.tensor([[1.0, 0, 0], [0, 0, -1.0], [0, 1.0, 0]]), atol=1e-6)  # 90 deg about x
    assert torch.allclose(dns.quat_to_rotmat(q), R)
    assert torch.allclose(R @ R.transpose(1, 2), torch.eye(3).expand(3, 3, 3), atol=1e-6)


# ------------------------------------------------------------------------------------------------ T1


def _dense_vs_oracle(orc, N, W, H, focal, seed, sh_degree, render_mode, rasterize_mode="classic"):
    inp, viewmat, K, _ = gsplat_inputs(N, W, H, focal=focal, seed=seed, anisotropic=True, sh_rest_std=0.3)
    inp = {k: v.double() for k, v in inp.items()}
    viewmat, K = viewmat.double(), K.double()
    co = to_leaf(inp, "cpu")
    cd = to_leaf(inp, "cpu")
    r_o, a_o, info = orc.rasterization(**co, viewmats=viewmat, Ks=K, width=W, height=H, packed=False,
                                       sh_degree=sh_degree, render_mode=render_mode, rasterize_mode=rasterize_mode)
    r_d, a_d, pr = dense_ref.render(cd["means"], cd["quats"], cd["scales"], cd["opacities"], cd["colors"], viewmat[0], K[0],
                                    W, H, sh_degree=sh_degree, render_mode=render_mode, rasterize_mode=rasterize_mode)
    assert_equal_int(info["radii"][0], pr["radii"], "radii")
    assert_close(r_o[0], r_d, "render", 1e-9)
    assert_close(a_o[0, ..., 0], a_d, "alpha", 1e-9)
    v_r, v_a = cotangents([r_d.shape, a_d.shape], seed + 1)
    ((r_o[0] * v_r.double()).sum() + (a_o[0, ..., 0] * v_a.double()).sum()).backward()
    ((r_d * v_r.double()).sum() + (a_d * v_a.double()).sum()).backward()
    for k in co:
        assert_close(co[k].grad, cd[k].grad, "grad " + k, 1e-7)


@pytest.mark.parametrize("sh_degree,render_mode,mode", [
    (3, "RGB+ED", "classic"), (1, "RGB", "classic"), (3, "RGB+D", "antialiased"), (2, "RGB+ED", "antialiased")])
def test_t1_oracle_backward_equals_autograd_fp64(orc, sh_degree, render_mode, mode):
    _dense_vs_oracle(orc, 160, 48, 40, 35.0, 40 + sh_degree, sh_degree, render_mode, mode)


def test_t1_fp32_oracle_tracks_fp64_oracle(orc):
    inp, viewmat, K, _ = gsplat_inputs(2000, 96, 80, focal=70.0, seed=3, anisotropic=True)
    c32 = to_leaf(inp, "cpu")
    c64 = to_leaf({k: v.double() for k, v in inp.items()}, "cpu")
    r32, a32, i32 = orc.rasterization(**c32, viewmats=viewmat, Ks=K, width=96, height=80, packed=False, sh_degree=3,
                                      render_mode="RGB+ED")
    r64, a64, i64 = orc.rasterization(**c64, viewmats=viewmat.double(), Ks=K.double(), width=96, height=80, packed=False,
                                      sh_degree=3, render_mode="RGB+ED")
    # the fp64 run may flip a handful of ceil()/floor() decisions; the images still agree to fp32 accuracy
    differing = int((i32["radii"] != i64["radii"]).sum())
    assert differing <= 2
    if differing == 0:
        assert rel_err(r32, r64) < 1e-4 and rel_err(a32, a64) < 1e-4


def test_legacy_rasterize_gaussians_defaults(orc):
    """background defaults to ones; return_alpha; empty intersection set returns the background."""
    inp, viewmat, K, _ = gsplat_inputs(500, 64, 48, focal=40.0, seed=5)
    with torch.no_grad():
        _, _, info = orc.rasterization(**inp, viewmats=viewmat, Ks=K, width=64, height=48, packed=False, sh_degree=3)
    cols = torch.rand(500, 3)
    args = (info["means2d"][0], info["depths"][0], info["radii"][0], info["conics"][0], info["tiles_per_gauss"][0], cols,
            inp["opacities"][:, None], 48, 64, 16)
    out, alpha = orc.rasterize_gaussians(*args, return_alpha=True)
    out0 = orc.rasterize_gaussians(*args, background=torch.zeros(3))
    assert torch.allclose(out, out0 + (1 - alpha)[..., None], atol=1e-6)
    none = orc.rasterize_gaussians(info["means2d"][0], info["depths"][0], torch.zeros(500, dtype=torch.int32), info["conics"][0],
                                   torch.zeros(500, dtype=torch.int32), cols, inp["opacities"][:, None], 48, 64, 16)
    assert torch.equal(none, torch.ones(48, 64, 3))


def test_projection_edge_flags_cover_last_place_jitter(orc):
    """orc_project_edge marks the Gaussians whose integer outputs may change when the activated inputs differ in the last
    place (torch's exp / normalisation on the CPU vs the fused kernel's): perturbing scales and quaternions by up to
    +-2 ulp must never change a radius or a tile count of an unmarked Gaussian — and the marked set stays small."""
    import numpy as np
    import torch

    from _scenes import gsplat_inputs

    W, H = 640, 480
    inp, viewmat, K, _ = gsplat_inputs(200_000, W, H, focal=400.0, seed=3, anisotropic=False)
    g = torch.Generator().manual_seed(0)

    def jitter(t):
        a = t.numpy().copy().view(np.int32)
        a += torch.randint(-2, 3, t.shape, generator=g).numpy().astype(np.int32)
        return torch.from_numpy(a.view(np.float32))

    with torch.no_grad():
        edge = orc.project_edge(inp["means"], inp["quats"], inp["scales"], viewmat[0], K[0], W, H)
        ref = orc.project_fwd(inp["means"], inp["quats"], inp["scales"], viewmat[0], K[0], W, H)
        changed = torch.zeros_like(edge)
        for _ in range(3):
            got = orc.project_fwd(inp["means"], jitter(inp["quats"]), jitter(inp["scales"]), viewmat[0], K[0], W, H)
            changed |= (got[0] != ref[0]) | (got[5] != ref[5])
    assert int(changed.sum()) > 0, "the jitter moved no integer at all: the test scene is too easy"
    assert int((changed & ~edge).sum()) == 0
    assert int(edge.sum()) < 0.02 * edge.numel()


def test_tight_tile_boxes_only_leave_out_unreachable_tiles(orc):
    """The tile rule of the product's fused path (dnsplat_camera.tight_tiles, restated in oracle.tight_tile_boxes): inside gsplat's
    3-sigma box, and every tile of gsplat's box it drops has min sigma over the tile's pixel-centre rectangle >= ln(255 opacity)
    (closed form, float64) — no pixel of it can pass the alpha >= 1/255 test of A.5."""
    import math
    import torch
    from _scenes import gsplat_inputs

    W, H = 320, 240
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    kept_total = loose_total = 0
    for seed, aniso in ((3, False), (4, True), (5, True)):
        inp, viewmat, K, _ = gsplat_inputs(6000, W, H, focal=200.0, seed=seed, anisotropic=aniso)
        g = torch.Generator().manual_seed(seed)
        opac = torch.rand(6000, generator=g) ** 3                    # many faint splats, some near 1, some below 1/255
        radii, means2d, depths, conics = orc.project_fwd(inp["means"], inp["quats"], inp["scales"], viewmat[0], K[0], W, H)[:4]
        x0, y0, x1, y1 = orc.tight_tile_boxes(means2d, conics, opac, radii, 16, tw, th)
        r = radii.float()
        lx0 = torch.floor((means2d[:, 0] - r) / 16).clamp(0, tw).long(); lx1 = torch.ceil((means2d[:, 0] + r) / 16).clamp(0, tw).long()
        ly0 = torch.floor((means2d[:, 1] - r) / 16).clamp(0, th).long(); ly1 = torch.ceil((means2d[:, 1] + r) / 16).clamp(0, th).long()
        vis = radii > 0
        assert bool(((x0 >= lx0) & (x1 <= lx1) & (y0 >= ly0) & (y1 <= ly1))[vis].all())
        kept_total += int(((x1 - x0) * (y1 - y0))[vis].sum()); loose_total += int(((lx1 - lx0) * (ly1 - ly0))[vis].sum())
        m, c, o = means2d.double(), conics.double(), opac.double()
        for gi in torch.nonzero(vis).reshape(-1).tolist():
            if o[gi] * 255.0 <= 1.0:
                continue
            tau = math.log(255.0 * o[gi])
            a_, b_, c_ = c[gi].tolist()
            for ty in range(int(ly0[gi]), int(ly1[gi])):
                for tx in range(int(lx0[gi]), int(lx1[gi])):
                    if int(x0[gi]) <= tx < int(x1[gi]) and int(y0[gi]) <= ty < int(y1[gi]):
                        continue
                    lox, hix = tx * 16 + 0.5 - m[gi, 0].item(), tx * 16 + 15.5 - m[gi, 0].item()
                    loy, hiy = ty * 16 + 0.5 - m[gi, 1].item(), ty * 16 + 15.5 - m[gi, 1].item()
                    sig = lambda dx, dy: 0.5 * (a_ * dx * dx + c_ * dy * dy) + b_ * dx * dy   # noqa: E731
                    clip = lambda v, lo, hi: min(max(v, lo), hi)                                 # noqa: E731
                    if lox <= 0 <= hix and loy <= 0 <= hiy:
                        smin = 0.0
                    else:
                        smin = min([sig(dx, clip(-b_ * dx / c_, loy, hiy)) for dx in (lox, hix)] +
                                   [sig(clip(-b_ * dy / a_, lox, hix), dy) for dy in (loy, hiy)])
                    assert smin >= tau - 1e-9, (seed, gi, tx, ty, smin, tau)
    assert kept_total < 0.8 * loose_total, (kept_total, loose_total)
