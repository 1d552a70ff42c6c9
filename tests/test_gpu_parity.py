"""HIP path (through the C ABI, libdnsplat.so) vs the CPU oracle — the parity tests proper.

Written like tests of the two gsplat calls dn-splatter makes (dn_splatter/dn_model.py:495-516 and
:564-575): same argument names, the oracle called exactly the same way on the CPU.  Integers are
compared bit-exactly; floats at the 1e-4 tolerance BASELINE.json states (see _scenes.REL_TOL).
"""
import math
import os

import pytest
import torch

from _scenes import (both_sides_order_independent, check_pixels, check_rows, REL_TOL, assert_close, assert_close_groups, sh_band_groups, fp64_envelope, assert_equal_int, cotangents, gsplat_inputs,
                     keep_mask, rel_err, to_leaf, zero_borderline, render_bounds, flip_bound_linear, flip_bound_alpha, flip_bound_ratio,
                     flip_bound_unit)

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
INT_KEYS = ("radii", "tiles_per_gauss", "flatten_ids", "isect_offsets")
FLOAT_KEYS = ("means2d", "depths", "conics")


def _call_both(dns, orc, inp, viewmat, K, W, H, **kw):
    ci = to_leaf(inp, "cpu")
    gi = to_leaf(inp, DEV)
    r_o, a_o, info_o = orc.rasterization(**ci, viewmats=viewmat, Ks=K, width=W, height=H, packed=False, **kw)
    r_g, a_g, info_g = dns.rasterization(**gi, viewmats=viewmat.to(DEV), Ks=K.to(DEV), width=W, height=H,
                                         packed=False, **kw)
    info_o["_call"] = (orc, inp, viewmat, K, W, H, kw)
    return (r_o, a_o, info_o, ci), (r_g, a_g, info_g, gi)


def _fp64_gradients(info_o, v_r, v_a):
    """The oracle call of _call_both once more in float64 with the same cotangents: {name: gradient}.  Used only for the
    per-entry rounding envelope of the fp32 reference algorithm (_scenes.assert_close); None when the call was not recorded."""
    if "_call" not in info_o:
        return None
    orc, inp, viewmat, K, W, H, kw = info_o["_call"]
    as64 = lambda v: v.double() if torch.is_tensor(v) and v.is_floating_point() else v
    c64 = {k: v.detach().double().requires_grad_(True) for k, v in inp.items()}
    r, a, info = orc.rasterization(**c64, viewmats=viewmat.double(), Ks=K.double(), width=W, height=H, packed=False,
                                   **{k: as64(v) for k, v in kw.items()})
    info["means2d"].retain_grad()
    ((r * v_r.double()).sum() + (a * v_a.double()).sum()).backward()
    out = {k: c64[k].grad for k in c64}
    out["means2d"] = info["means2d"].grad
    out["means2d.absgrad"] = getattr(info["means2d"], "absgrad", None)
    return out


def _keep(o, what="scene"):
    """Pixels without a flagged decision (the oracle's borderline mask, _scenes.py): they carry the cotangents of the backward
    comparison.  The forward comparison covers every pixel, the borderline ones at their flip bound."""
    info_o = o[2]
    if "_keep" not in info_o:
        info_o["_keep"] = keep_mask(info_o["borderline"], what)
    return info_o["_keep"]


def _check_forward(o, g, tol=REL_TOL, what="scene"):
    r_o, a_o, info_o, _ = o
    r_g, a_g, info_g, _ = g
    for k in INT_KEYS:
        assert_equal_int(info_g[k], info_o[k], k)
    assert torch.is_tensor(info_g["isect_ids"]) and info_g["isect_ids"].dtype == torch.int64
    assert_equal_int(info_g["isect_ids"], info_o["isect_ids"], "isect_ids")
    assert info_g["n_isects"] == info_o["flatten_ids"].shape[0]
    for k in FLOAT_KEYS:
        assert_close(info_g[k], info_o[k], k, tol)
    _keep(o, what)            # prints / bounds the number of borderline pixels
    # colour channels and the depth channel against their OWN scales (expected depth ~ 3-13 would let rgb be off by 1e-3)
    nc = r_o.shape[-1]
    mode = info_o.get("_call", (None,) * 7)[6].get("render_mode", "RGB") if "_call" in info_o else "RGB"
    n_depth = 1 if mode in ("RGB+D", "RGB+ED", "D", "ED") else 0
    groups = ([("colour", 0, nc - n_depth)] if nc - n_depth > 0 else []) + ([("depth", nc - n_depth, nc)] if n_depth else [])
    # EVERY pixel is compared: plain tolerance where no decision was flagged, tolerance + the finite flip bound where one was
    rb, ab = render_bounds(info_o, r_o, a_o, mode)
    assert_close_groups(r_g, r_o, "render", groups, tol=tol, bound=rb)
    assert_close(a_g, a_o, "alpha", tol, bound=ab)
    # and per PIXEL, relative to that pixel's own value (borderline pixels stay with their flip bound above)
    keep = _keep(o, what)
    for name, lo, hi in groups:
        check_pixels(r_g[..., lo:hi], r_o[..., lo:hi], f"{what} render[{name}] per pixel", keep=keep, enforce=True)
    check_pixels(a_g, a_o, f"{what} alpha per pixel", keep=keep, enforce=True)


def _check_backward(o, g, tol=REL_TOL, seed=1, absgrad=True, quat_atol=0.0, what="scene"):
    r_o, a_o, info_o, ci = o
    r_g, a_g, info_g, gi = g
    keep = _keep(o, what)
    v_r, v_a = cotangents([r_o.shape, a_o.shape], seed)
    # borderline pixels get zero cotangents on BOTH sides: they contribute exactly nothing to any gradient
    v_r, v_a = zero_borderline(v_r, keep), zero_borderline(v_a[..., 0], keep)[..., None]
    info_o["means2d"].retain_grad()
    info_g["means2d"].retain_grad()
    ((r_o * v_r).sum() + (a_o * v_a).sum()).backward()
    ((r_g * v_r.to(DEV)).sum() + (a_g * v_a.to(DEV)).sum()).backward()
    torch.cuda.synchronize()
    g64 = _fp64_gradients(info_o, v_r, v_a)

    def env(name, g32):
        return fp64_envelope(g32, g64[name]) if g64 is not None and g64.get(name) is not None else None

    for k in ci:
        if ci[k].grad is None:          # e.g. colours in the depth-only render modes
            assert gi[k].grad is None or float(gi[k].grad.abs().max()) == 0.0, k
            continue
        if k == "colors" and ci[k].grad.dim() == 3:
            # SH coefficient gradient band by band (band 0 is an order of magnitude above the others)
            assert_close_groups(gi[k].grad, ci[k].grad, "grad " + k, sh_band_groups(ci[k].grad.shape[1]), dim=1, tol=tol,
                                envelope=env(k, ci[k].grad))
            continue
        assert_close(gi[k].grad, ci[k].grad, "grad " + k, tol, atol=quat_atol if k == "quats" else 0.0, envelope=env(k, ci[k].grad))
    assert_close(info_g["means2d"].grad, info_o["means2d"].grad, "means2d.grad", tol, envelope=env("means2d", info_o["means2d"].grad))
    if absgrad:
        assert_close(info_g["means2d"].absgrad, info_o["means2d"].absgrad, "means2d.absgrad", tol,
                     envelope=env("means2d.absgrad", info_o["means2d"].absgrad))
    # per GAUSSIAN, relative to that Gaussian's own gradient norm (_scenes.check_rows): asserted when neither side's sums depend on
    # the arrival order of atomics (HIP deterministic mode + oracle scatter in double), logged otherwise
    det = both_sides_order_independent(info_o["_call"][0]) if "_call" in info_o else False
    N_ = ci["means"].shape[0]
    for k in ci:
        if ci[k].grad is None or (k == "quats" and quat_atol > 0):
            continue
        check_rows(gi[k].grad, ci[k].grad, f"{what} grad {k} per Gaussian", b64=None if g64 is None else g64.get(k), enforce=True, strict=det, n_rows=N_)
    check_rows(info_g["means2d"].grad, info_o["means2d"].grad, f"{what} means2d.grad per Gaussian",
               b64=None if g64 is None else g64.get("means2d"), enforce=True, strict=det, n_rows=N_)
    if absgrad:
        check_rows(info_g["means2d"].absgrad, info_o["means2d"].absgrad, f"{what} means2d.absgrad per Gaussian",
                   b64=None if g64 is None else g64.get("means2d.absgrad"), enforce=True, strict=det, n_rows=N_)


GRAD_NAMES = ("means", "scales", "quats", "features_dc", "features_rest", "opacities")
OUT_KEYS = ("rgb", "depth", "normal", "accumulation")


def _mirror_pair(dns, orc, gp, cam, hip_kw, cot_seed=2, what="mirror", config=None, step=None, cond=False):
    """DNSplatterModel.get_outputs (dn_model.py:404-612) twice: the reference's own op sequence with the oracle plugged in
    for the two gsplat calls (CPU), and the product (GPU, ``hip_kw`` picks fused / two-call).  The oracle runs first: its
    borderline mask (plus the pixels whose pre-clamp rgb sits within rounding of the clamp(0, 1) corners, where the
    gradient gate of dn_model.py:528 may fall either way) selects the pixels that are compared and zeroes the cotangents of
    the others on BOTH sides.  Returns ((out, params, renderer) for hip, the same for the oracle, keep).
    ``cond``: the oracle side runs under an ``orc.ConditionTrace``; its renderer then carries ``cond_A`` / ``cond_S``, the running
    error bounds of every gradient row (keys: GRAD_NAMES + "xys" + "xys.absgrad") for _scenes.check_rows_conditioned."""
    import contextlib

    captured = {}
    trace = orc.ConditionTrace() if cond else contextlib.nullcontext()

    def rasterization_spy(**kw):
        r, a, info = orc.rasterization(**kw)
        captured["render"], captured["alpha"], captured["info"] = r.detach(), a.detach(), info
        return r, a, info

    def leaves(device):
        return {k: v.detach().to(device).clone().requires_grad_(k != "normals") for k, v in gp.items()}

    def rasterize_gaussians_spy(*a, **kw):
        out = orc.rasterize_gaussians(*a, **kw)
        captured["normal_raw"] = out.detach()
        return out

    p_o = leaves("cpu")
    m_o = dns.DNSplatterRenderer(p_o, config=config, fused=False, rasterization_fn=rasterization_spy,
                                 rasterize_gaussians_fn=rasterize_gaussians_spy)
    if step is not None:
        m_o.step = step
    orc.last_borderline = None
    orc.last_flip_weight = None
    with trace:
        out_o = m_o.get_outputs(cam)
    border = captured["info"]["borderline"].clone()
    flip = captured["info"]["flip_weight"].clone()
    if orc.last_borderline is not None and orc.last_borderline.shape == border.shape:
        border |= orc.last_borderline                       # the second (legacy normal) pass
        flip = torch.maximum(flip, orc.last_flip_weight)
    # finite bounds for the image values of the borderline pixels (0 elsewhere), through the post-ops of dn_model.py:526-537, 577-578
    info_c, bg_c = captured["info"], out_o["background"].detach()
    cmax = info_c["channel_absmax"]
    a_img = captured["alpha"][0, ..., 0]
    m_o.flip_bounds = {
        # rgb = clamp(render + (1 - alpha) bg, 0, 1): 1-Lipschitz in the composite, plus the background's share of d alpha
        "rgb": flip_bound_linear(flip, cmax[:3]) + flip_bound_alpha(flip)[..., None] * bg_c.abs().double(),
        "accumulation": flip_bound_alpha(flip)[..., None],
        "depth": flip_bound_ratio(flip, float(cmax[3]), captured["render"][0, ..., 3], a_img)[..., None],
    }
    if "normal_raw" in captured:
        m_o.flip_bounds["normal"] = flip_bound_unit(flip, 1.0, captured["normal_raw"].norm(dim=-1))
    pre = captured["render"][0, ..., :3] + (1 - captured["alpha"][0]) * out_o["background"]
    border |= ((pre.abs() < 4e-6) | ((pre - 1).abs() < 4e-6)).any(-1)
    keep = keep_mask(border, what)
    gen = torch.Generator().manual_seed(cot_seed)
    cot = {k: zero_borderline(torch.rand(out_o[k].shape, generator=gen) * 2 - 1, keep) for k in OUT_KEYS}
    live = [k for k in OUT_KEYS if out_o[k].requires_grad]          # predict_normals=False hands out a constant normal image
    torch.autograd.backward([out_o[k] for k in live], [cot[k] for k in live], retain_graph=cond)
    if cond:
        m_o.cond_A, m_o.cond_S = trace.param_condition({k: p_o[k] for k in GRAD_NAMES})
        ra, rs = trace.raster_condition(0, "A"), trace.raster_condition(0, "B")      # the first call is the one whose xys carry the gradient
        m_o.cond_A.update({"xys": ra["means2d"], "xys.absgrad": ra["absgrad"]})
        m_o.cond_S.update({"xys": rs["means2d"], "xys.absgrad": rs["absgrad"]})

    # the same sequence once more in float64 (same cotangents): the per-entry rounding envelope of the fp32 reference
    # algorithm for the gradient comparisons (_scenes.assert_close)
    p_64 = {k: v.detach().double().clone().requires_grad_(k != "normals") for k, v in gp.items()}

    def rasterization_64(**kw):
        kw["viewmats"], kw["Ks"] = kw["viewmats"].double(), kw["Ks"].double()
        return orc.rasterization(**kw)

    m_64 = dns.DNSplatterRenderer(p_64, config=config, fused=False, rasterization_fn=rasterization_64,
                                  rasterize_gaussians_fn=orc.rasterize_gaussians)
    if step is not None:
        m_64.step = step
    out_64 = m_64.get_outputs(dns.Camera(cam.camera_to_worlds.double(), cam.fx, cam.fy, cam.cx, cam.cy, cam.width, cam.height))
    torch.autograd.backward([out_64[k] for k in live], [cot[k].double() for k in live])
    m_o.fp64_grads = {k: p_64[k].grad for k in GRAD_NAMES}
    m_o.fp64_grads["xys"] = m_64.xys.grad
    m_o.fp64_grads["xys.absgrad"] = getattr(m_64.xys, "absgrad", None)

    p_g = leaves(DEV)
    m_g = dns.DNSplatterRenderer(p_g, config=config, **hip_kw)
    if step is not None:
        m_g.step = step
    out_g = m_g.get_outputs(cam.to(DEV))
    assert [k for k in OUT_KEYS if out_g[k].requires_grad] == live
    torch.autograd.backward([out_g[k] for k in live], [cot[k].to(DEV) for k in live])
    torch.cuda.synchronize()
    return (out_g, p_g, m_g), (out_o, p_o, m_o), keep


def _check_mirror(hip, ora, keep, what="mirror", quat_atol=0.0, ints=True):
    out_g, p_g, m_g = hip
    out_o, p_o, m_o = ora
    if ints:
        # Both sides were handed RAW parameters: the product activates them inside its projection kernel, the reference
        # sequence in torch on the CPU.  Integer outputs are compared bit for bit for every Gaussian except the few whose
        # radius / tile box / culling decision sits inside the rounding envelope of those activations (oracle-flagged,
        # orc_project_edge); the tile lists are compared with exactly those Gaussians' entries taken out on both sides.
        edge = m_o.last_info["edge_gaussians"]
        solid = ~edge
        print(f"[parity] {what}: {int(edge.sum())} of {edge.numel()} Gaussians rounding-sensitive in projection, their integers compared separately")
        assert int(edge.sum()) <= max(4, 0.01 * edge.numel())
        r_g, r_o = m_g.radii.cpu(), m_o.radii
        t_g, t_o = m_g.num_tiles_hit.reshape(-1).cpu(), m_o.num_tiles_hit.reshape(-1)
        assert_equal_int(r_g[solid], r_o[solid], what + " radii")
        assert int(((r_g[edge] - r_o[edge]).abs() > 1).sum()) == 0 or bool(((r_g[edge] == 0) | (r_o[edge] == 0)).any())
        lists = []
        for m in (m_g, m_o):
            fid = m.last_info["flatten_ids"].cpu().long()
            offs = torch.cat([m.last_info["isect_offsets"].reshape(-1).cpu().long(), torch.tensor([fid.numel()])])
            tile_of = torch.searchsorted(offs, torch.arange(fid.numel()), right=True) - 1
            sel = solid[fid]
            lists.append((fid[sel], tile_of[sel], offs.numel() - 1))
        (fid_g, tile_g, n_t), (fid_o, tile_o, _) = lists
        if m_g.last_info.get("tight_tiles"):
            # The fused path bins over the tight tile boxes (dnsplat_camera.tight_tiles): its lists are the reference's lists with
            # the pairs taken out whose splat cannot reach alpha >= 1/255 at any pixel centre of the tile.  Checked here: they are
            # an order-preserving sub-list of the oracle's, and every pair left out has min sigma over the tile's pixel-centre
            # rectangle >= ln(255 opacity) (float64, closed form), i.e. the per-pixel test of A.5 rejects it at all 256 pixels.
            N_ = solid.numel()
            key_g, key_o = tile_g * N_ + fid_g, tile_o * N_ + fid_o
            kept = torch.isin(key_o, key_g)
            assert_equal_int(fid_o[kept], fid_g, what + " tight tile lists are an order-preserving sub-list of the reference's")
            assert_equal_int(tile_o[kept], tile_g, what + " tight tile lists: tiles")
            # what the caller sees (info["tiles_per_gauss"] / num_tiles_hit, dn_model.py:524) is gsplat's count, bit for bit;
            # the count the binning walked is never larger
            assert_equal_int(t_g[solid], t_o[solid], what + " num_tiles_hit (tight tile boxes are internal)")
            assert (m_g.last_info["tiles_bin"].reshape(-1).cpu()[solid] <= t_o[solid]).all()
            gone_t, gone_g = tile_o[~kept], fid_o[~kept]
            info_o = m_o.last_info
            tw_ = int(info_o["tile_width"]) if "tile_width" in info_o else int(info_o["isect_offsets"].shape[-1])
            xy = info_o["means2d"].detach().reshape(-1, 2).double()[gone_g]
            con = info_o["conics"].detach().reshape(-1, 3).double()[gone_g]
            opa = torch.sigmoid(p_o["opacities"].detach().double()).reshape(-1)[gone_g]
            tx, ty = (gone_t % tw_).double(), (gone_t // tw_).double()
            # d = pixel centre - mean, over [lo, hi] per axis; convex quadratic: the minimum over the rectangle is at the clamped
            # centre if the unconstrained minimum (d = 0) is inside, else on one of the four edges (1-D minima, clamped)
            lox, hix = tx * 16 + 0.5 - xy[:, 0], tx * 16 + 15.5 - xy[:, 0]
            loy, hiy = ty * 16 + 0.5 - xy[:, 1], ty * 16 + 15.5 - xy[:, 1]
            a_, b_, c_ = con[:, 0], con[:, 1], con[:, 2]
            sig = lambda dx, dy: 0.5 * (a_ * dx * dx + c_ * dy * dy) + b_ * dx * dy
            cands = [sig(torch.minimum(torch.maximum(torch.zeros_like(lox), lox), hix), torch.minimum(torch.maximum(torch.zeros_like(loy), loy), hiy))]
            for dx in (lox, hix):
                cands.append(sig(dx, torch.minimum(torch.maximum(-b_ * dx / c_, loy), hiy)))
            for dy in (loy, hiy):
                cands.append(sig(torch.minimum(torch.maximum(-b_ * dy / a_, lox), hix), dy))
            inside = (lox <= 0) & (hix >= 0) & (loy <= 0) & (hiy >= 0)
            smin = torch.where(inside, torch.zeros_like(lox), torch.stack(cands[1:]).min(0).values)
            tau = torch.log(255.0 * opa)
            bad = smin < tau - 1e-9
            print(f"[parity] {what}: tight tile boxes keep {int(kept.sum())} of {kept.numel()} list entries ({100.0 * float(kept.sum()) / max(kept.numel(), 1):.1f} %)")
            assert not bool(bad.any()), f"{what}: {int(bad.sum())} of {bad.numel()} pairs left out of the tight lists could reach alpha >= 1/255"
            assert float(kept.sum()) < 0.95 * kept.numel() or kept.numel() < 1000, "tight tile boxes removed next to nothing"
        else:
            assert_equal_int(t_g[solid], t_o[solid], what + " num_tiles_hit")
            assert_equal_int(fid_g, fid_o, what + " flatten_ids (rounding-sensitive Gaussians taken out)")
            assert_equal_int(torch.bincount(tile_g, minlength=n_t), torch.bincount(tile_o, minlength=n_t),
                             what + " tile list lengths (rounding-sensitive Gaussians taken out)")
    bounds = getattr(m_o, "flip_bounds", {})
    for k in OUT_KEYS:
        assert out_g[k].shape == out_o[k].shape
        # every pixel: plain tolerance, plus the finite flip bound on the borderline ones (no pixel is left out)
        assert_close(out_g[k], out_o[k], what + " " + k, bound=bounds.get(k))
    g64 = getattr(m_o, "fp64_grads", None)

    def env(name, g32):
        return fp64_envelope(g32, g64[name]) if g64 is not None and g64.get(name) is not None else None

    for k in GRAD_NAMES:
        if p_o[k].grad is None or p_o[k].grad.numel() == 0:   # e.g. the (empty) higher-band tensor when sh_degree == 0 feeds sigmoid(colours)
            assert p_g[k].grad is None or p_g[k].grad.numel() == 0 or float(p_g[k].grad.abs().max()) == 0.0, k
            continue
        if k == "features_rest" and p_o[k].grad.dim() == 3 and p_o[k].grad.shape[1] == 15:
            # band by band: each SH band against its own scale
            assert_close_groups(p_g[k].grad, p_o[k].grad, what + " grad " + k, sh_band_groups(15), dim=1, envelope=env(k, p_o[k].grad))
            continue
        assert_close(p_g[k].grad, p_o[k].grad, what + " grad " + k, atol=quat_atol if k == "quats" else 0.0, envelope=env(k, p_o[k].grad))
    assert_close(m_g.xys.grad, m_o.xys.grad, what + " xys.grad (dn_model.py:517-519)", envelope=env("xys", m_o.xys.grad))
    assert_close(m_g.xys.absgrad, m_o.xys.absgrad, what + " xys.absgrad", envelope=env("xys.absgrad", m_o.xys.absgrad))
    # row-relative statistics: per pixel for the images (always asserted), per Gaussian for the gradients (asserted when both sides
    # sum order-independently, logged otherwise) — _scenes.check_rows
    from oracle import oracle as _orc
    det = both_sides_order_independent(_orc)
    for k in OUT_KEYS:
        if k == "normal" and not out_o[k].requires_grad and float(out_o[k].abs().max()) == 0.0:
            continue
        check_pixels(out_g[k], out_o[k], f"{what} {k} per pixel", keep=keep, enforce=True)
    N_ = p_o["means"].shape[0]
    for k in GRAD_NAMES:
        if p_o[k].grad is None or (k == "quats" and quat_atol > 0) or p_o[k].grad.numel() == 0:
            continue
        check_rows(p_g[k].grad, p_o[k].grad, f"{what} grad {k} per Gaussian", b64=None if g64 is None else g64.get(k), enforce=True, strict=det, n_rows=N_)
    check_rows(m_g.xys.grad, m_o.xys.grad, f"{what} xys.grad per Gaussian", b64=None if g64 is None else g64.get("xys"), enforce=True, strict=det, n_rows=N_)
    check_rows(m_g.xys.absgrad, m_o.xys.absgrad, f"{what} xys.absgrad per Gaussian", b64=None if g64 is None else g64.get("xys.absgrad"),
               enforce=True, strict=det, n_rows=N_)


# ------------------------------------------------------------------------------------------------
# BASELINE config C1: 10k Gaussians, 256x256, SH degree 3, RGB+ED, absgrad — the dn-splatter call


def test_c1_rasterization_matches_oracle(dns, orc):
    """BASELINE config C1 verbatim: the reference's random init is ISOTROPIC (scales = log of the mean 3-NN
    distance repeated 3x, dn_model.py:217), so d/d(quats) is mathematically zero; both sides carry ~1e-6 of
    fp32 noise there and that gradient is compared with an absolute tolerance."""
    inp, viewmat, K, _ = gsplat_inputs(10_000, 256, 256, focal=160.0, seed=0)
    o, g = _call_both(dns, orc, inp, viewmat, K, 256, 256, sh_degree=3, render_mode="RGB+ED", absgrad=True)
    # every compared entry is held to 1e-4 of its tensor's scale; only the oracle-flagged borderline pixels (see
    # _scenes.py) are set aside
    _check_forward(o, g)
    _check_backward(o, g, quat_atol=1e-4)


def test_c1_anisotropic_rasterization_matches_oracle(dns, orc):
    """C1 sizes with anisotropic scales and spread opacities, so every gradient (quats included) is exercised."""
    inp, viewmat, K, _ = gsplat_inputs(10_000, 256, 256, focal=160.0, seed=1, anisotropic=True)
    o, g = _call_both(dns, orc, inp, viewmat, K, 256, 256, sh_degree=3, render_mode="RGB+ED", absgrad=True)
    _check_forward(o, g)
    _check_backward(o, g)


@pytest.mark.parametrize("W,H", [(200, 120), (17, 33), (16, 16), (1, 1), (333, 95)])
def test_ragged_image_sizes(dns, orc, W, H):
    inp, viewmat, K, _ = gsplat_inputs(3000, W, H, focal=0.6 * max(W, H), seed=4, anisotropic=True)
    o, g = _call_both(dns, orc, inp, viewmat, K, W, H, sh_degree=3, render_mode="RGB+ED", absgrad=True)
    _check_forward(o, g)
    _check_backward(o, g)


@pytest.mark.parametrize("sh_degree", [0, 1, 2, 3])
def test_sh_degrees(dns, orc, sh_degree):
    inp, viewmat, K, _ = gsplat_inputs(4000, 128, 96, focal=90.0, seed=5, sh_rest_std=0.3, anisotropic=True)
    o, g = _call_both(dns, orc, inp, viewmat, K, 128, 96, sh_degree=sh_degree, render_mode="RGB+ED", absgrad=True)
    _check_forward(o, g)
    _check_backward(o, g)


@pytest.mark.parametrize("render_mode", ["RGB", "D", "ED", "RGB+D", "RGB+ED"])
def test_render_modes(dns, orc, render_mode):
    inp, viewmat, K, _ = gsplat_inputs(3000, 96, 80, focal=70.0, seed=6, anisotropic=True)
    o, g = _call_both(dns, orc, inp, viewmat, K, 96, 80, sh_degree=3, render_mode=render_mode, absgrad=True)
    _check_forward(o, g)
    _check_backward(o, g)


@pytest.mark.parametrize("n_colors", [5, 7])
def test_direct_colors_and_background(dns, orc, n_colors):
    """sh_degree=None path (dn_model.py:491-493) with a background, 5 or 7 feature channels + depth.  7 + depth = 8 channels is the
    record's capacity: the compositing backward then has no free cotangent slot for the next pixel's x and takes the column from
    its pixel counter instead (DNS_BWD_PX_SLOT needs D < 8) — the one instantiation no other test reaches."""
    inp, viewmat, K, _ = gsplat_inputs(3000, 96, 80, focal=70.0, seed=7, anisotropic=True)
    g_ = torch.Generator().manual_seed(9)
    inp["colors"] = torch.rand(3000, n_colors, generator=g_)
    bg = torch.rand(1, n_colors + 1, generator=g_)
    ci = to_leaf(inp, "cpu")
    gi = to_leaf(inp, DEV)
    r_o, a_o, info_o = orc.rasterization(**ci, viewmats=viewmat, Ks=K, width=96, height=80, packed=False,
                                         render_mode="RGB+D", backgrounds=bg, absgrad=True)
    r_g, a_g, info_g = dns.rasterization(**gi, viewmats=viewmat.to(DEV), Ks=K.to(DEV), width=96, height=80,
                                         packed=False, render_mode="RGB+D", backgrounds=bg.to(DEV), absgrad=True)
    _check_forward((r_o, a_o, info_o, ci), (r_g, a_g, info_g, gi))
    _check_backward((r_o, a_o, info_o, ci), (r_g, a_g, info_g, gi))


def test_antialiased_mode(dns, orc):
    inp, viewmat, K, _ = gsplat_inputs(4000, 128, 96, focal=90.0, seed=8, anisotropic=True)
    o, g = _call_both(dns, orc, inp, viewmat, K, 128, 96, sh_degree=3, render_mode="RGB+ED", absgrad=True,
                      rasterize_mode="antialiased")
    _check_forward(o, g)
    _check_backward(o, g)


@pytest.mark.parametrize("near,far,radius_clip,eps2d", [(0.01, 1e10, 0.0, 0.3), (5.0, 9.0, 0.0, 0.3), (0.01, 1e10, 6.0, 0.3),
                                                        (0.5, 50.0, 2.0, 0.1)])
def test_culling_planes_and_extreme_geometry(dns, orc, near, far, radius_clip, eps2d):
    """Near/far planes, radius_clip, eps2d, and a scene built to hit the awkward branches: Gaussians close to the
    camera, far off-axis (the 1.3 tan(fov/2) clamp of the perspective Jacobian), huge and tiny ones."""
    N, W, H = 4000, 160, 128
    inp, viewmat, K, _ = gsplat_inputs(N, W, H, focal=110.0, seed=31, anisotropic=True)
    g_ = torch.Generator().manual_seed(32)
    cam_pos = torch.inverse(viewmat[0])[:3, 3]
    fwd = -cam_pos / cam_pos.norm()
    inp["means"][:300] = cam_pos + fwd * (0.05 + 2.0 * torch.rand(300, 1, generator=g_)) + 0.3 * torch.randn(300, 3, generator=g_)
    inp["scales"][300:500] *= 8.0                     # splats covering a large part of the image
    inp["scales"][500:700] *= 0.05                    # sub-pixel splats (the 0.3 blur dominates)
    side = torch.linalg.cross(fwd, torch.tensor([0.0, 1.0, 0.0]))
    inp["means"][700:900] = cam_pos + fwd * 4.0 + side * (4.0 + 2.0 * torch.rand(200, 1, generator=g_))   # beyond the fov clamp
    inp["scales"][700:900] *= 6.0
    kw = dict(sh_degree=3, render_mode="RGB+ED", absgrad=True, near_plane=near, far_plane=far, radius_clip=radius_clip,
              eps2d=eps2d)
    o, g = _call_both(dns, orc, inp, viewmat, K, W, H, **kw)
    assert int((o[2]["radii"] > 0).sum()) > 100
    _check_forward(o, g)
    _check_backward(o, g)


def test_legacy_rasterize_gaussians_options(dns, orc):
    """return_alpha, explicit background, uint8 colours (legacy gsplat behaviour) of the second-pass drop-in."""
    inp, viewmat, K, _ = gsplat_inputs(3000, 96, 64, focal=60.0, seed=41, anisotropic=True)
    with torch.no_grad():
        _, _, info = orc.rasterization(**inp, viewmats=viewmat, Ks=K, width=96, height=64, packed=False, sh_degree=3)
    cols = torch.rand(3000, 5, generator=torch.Generator().manual_seed(1))
    bg = torch.tensor([0.1, 0.2, 0.3, 0.4, 0.5])
    common = (info["depths"][0], info["radii"][0], info["conics"][0], info["tiles_per_gauss"][0])
    out_o, al_o = orc.rasterize_gaussians(info["means2d"][0], *common, cols, inp["opacities"][:, None], 64, 96, 16,
                                          background=bg, return_alpha=True)
    dev = lambda t: t.to(DEV)   # noqa: E731
    out_g, al_g = dns.rasterize_gaussians(dev(info["means2d"][0]), *[dev(t) for t in common], dev(cols),
                                          dev(inp["opacities"][:, None]), 64, 96, 16, background=dev(bg), return_alpha=True)
    keep_mask(orc.last_borderline, "legacy options")
    cmax5 = torch.maximum(cols.abs().amax(0), bg.abs())
    assert_close(out_g, out_o, "legacy render with background", bound=flip_bound_linear(orc.last_flip_weight, cmax5))
    assert_close(al_g, al_o, "legacy alpha", bound=flip_bound_alpha(orc.last_flip_weight))
    u8 = (cols[:, :3] * 255).to(torch.uint8)
    out_u8 = dns.rasterize_gaussians(dev(info["means2d"][0]), *[dev(t) for t in common], dev(u8), dev(inp["opacities"][:, None]),
                                     64, 96, 16)
    out_f = dns.rasterize_gaussians(dev(info["means2d"][0]), *[dev(t) for t in common], dev(u8).float() / 255,
                                    dev(inp["opacities"][:, None]), 64, 96, 16)
    assert torch.equal(out_u8, out_f)
    with pytest.raises(AssertionError):
        dns.rasterize_gaussians(dev(info["means2d"][0]), *[dev(t) for t in common], dev(cols), dev(inp["opacities"][:, None]),
                                64, 96, 16, background=dev(bg[:3]))
    with pytest.raises(NotImplementedError):
        dns.rasterize_gaussians(dev(info["means2d"][0]), *[dev(t) for t in common], dev(cols), dev(inp["opacities"][:, None]),
                                64, 96, 8)


def test_edge_empty_and_culled(dns, orc):
    """Every Gaussian behind the camera (nothing visible), and N == 0."""
    inp, viewmat, K, _ = gsplat_inputs(500, 64, 48, focal=40.0, seed=2)
    inp["means"] = inp["means"] * 0.01 + torch.tensor([100.0, 0.0, 0.0])  # far behind the orbit camera at +x
    o, g = _call_both(dns, orc, inp, viewmat, K, 64, 48, sh_degree=3, render_mode="RGB+ED", absgrad=True)
    assert int((o[2]["radii"] > 0).sum()) == 0
    _check_forward(o, g)
    r_g, a_g, info_g, gi = g
    assert float(a_g.detach().abs().max()) == 0.0 and float(r_g.detach().abs().max()) == 0.0
    (r_g.sum() + a_g.sum()).backward()
    for k in gi:
        assert float(gi[k].grad.abs().max()) == 0.0, k
    # N == 0
    empty = {k: v[:0] for k, v in inp.items()}
    gi = to_leaf(empty, DEV)
    r, a, info = dns.rasterization(**gi, viewmats=viewmat.to(DEV), Ks=K.to(DEV), width=64, height=48, packed=False,
                                   sh_degree=3, render_mode="RGB+ED")
    assert r.shape == (1, 48, 64, 4) and float(r.detach().abs().max()) == 0.0 and info["n_isects"] == 0


def test_edge_single_gaussian_closed_form(dns):
    """one isotropic Gaussian on the optical axis."""
    W = H = 64
    f, z, s, o = 50.0, 4.0, 0.2, 0.7
    means = torch.tensor([[0.0, 0.0, z]], device=DEV)
    quats = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=DEV)
    scales = torch.full((1, 3), s, device=DEV)
    opac = torch.tensor([o], device=DEV)
    colors = torch.tensor([[0.2, 0.5, 0.9]], device=DEV)
    viewmat = torch.eye(4, device=DEV)[None]
    K = torch.tensor([[[f, 0, W / 2], [0, f, H / 2], [0, 0, 1.0]]], device=DEV)
    r, a, info = dns.rasterization(means, quats, scales, opac, colors, viewmat, K, W, H, packed=False, render_mode="RGB+ED")
    var = f * f * s * s / (z * z) + 0.3
    assert torch.allclose(info["means2d"][0, 0].cpu(), torch.tensor([W / 2, H / 2]))
    assert torch.allclose(info["conics"][0, 0].cpu(), torch.tensor([1 / var, 0.0, 1 / var]), rtol=1e-5)
    assert int(info["radii"][0, 0]) == math.ceil(3 * math.sqrt(var))
    # centre pixel (32,32) has its centre at (32.5, 32.5): sigma = 0.5*(0.25+0.25)/var
    alpha = min(0.999, o * math.exp(-0.25 / var))
    assert abs(float(a[0, 32, 32, 0]) - alpha) < 1e-6
    assert torch.allclose(r[0, 32, 32, :3].cpu(), torch.tensor([0.2, 0.5, 0.9]) * alpha, atol=1e-6)
    assert abs(float(r[0, 32, 32, 3]) - z) < 1e-5   # expected depth = z*alpha/alpha


def test_saturated_opacities_take_the_alpha_clamp(dns, orc):
    """Opacities of 1.0 and 0.9995: alpha = min(0.999, o vis) clamps near the splat centres, where the reference passes no
    gradient to the conic / mean / opacity through alpha.  The backward kernel has a separate instantiation of its step loop
    for buckets that hold such splats; a quarter of the Gaussians here do, the rest keep the clamp-free loop busy."""
    inp, viewmat, K, _ = gsplat_inputs(6000, 160, 128, focal=110.0, seed=12, anisotropic=True)
    inp["opacities"][::4] = 1.0
    inp["opacities"][1::8] = 0.9995
    o, g = _call_both(dns, orc, inp, viewmat, K, 160, 128, sh_degree=3, render_mode="RGB+ED", absgrad=True)
    clamped = (o[2]["radii"][0] > 0) & (inp["opacities"] > 0.999)
    assert int(clamped.sum()) > 500
    _check_forward(o, g)
    _check_backward(o, g)
    # and the fused 7-channel pass (the instantiation the benchmark uses)
    from dn_splatter_amd import synthetic
    gp = synthetic.make_gauss_params(5000, sh_rest_std=0.2, seed=13)
    gp["opacities"] = gp["opacities"].detach().clone()
    gp["opacities"][::3] = 12.0                                   # sigmoid(12) = 0.999994
    cam = synthetic.orbit_camera(3, width=144, height=112, focal=100.0)
    hip, ora, keep = _mirror_pair(dns, orc, gp, cam, dict(fused=True), cot_seed=4, what="clamped fused")
    # the projection reported "some visible opacity above the cap": the backward ran its clamping loop
    assert int(hip[2].last_info["_saturation_flag"].item()) == 1
    _check_mirror(hip, ora, keep, "clamped fused")


def test_occluded_and_transparent_gaussians(dns, orc):
    """T4 properties: zero-opacity Gaussians contribute nothing and get no colour gradient."""
    inp, viewmat, K, _ = gsplat_inputs(3000, 96, 80, focal=70.0, seed=11, anisotropic=True)
    base = to_leaf(inp, DEV)
    r0, a0, _ = dns.rasterization(**base, viewmats=viewmat.to(DEV), Ks=K.to(DEV), width=96, height=80, packed=False,
                                  sh_degree=3, render_mode="RGB+ED")
    # append 500 fully transparent copies: the image must not change at all (bit-exact)
    ext = {k: torch.cat([v, v[:500]]) for k, v in inp.items()}
    ext["opacities"][3000:] = 0.0
    e = to_leaf(ext, DEV)
    r1, a1, _ = dns.rasterization(**e, viewmats=viewmat.to(DEV), Ks=K.to(DEV), width=96, height=80, packed=False,
                                  sh_degree=3, render_mode="RGB+ED")
    assert torch.equal(r0, r1) and torch.equal(a0, a1)
    assert float(a1.detach().min()) >= 0.0 and float(a1.detach().max()) < 1.0
    (r1.sum() + a1.sum()).backward()
    assert float(e["colors"].grad[3000:].abs().max()) == 0.0


def test_permutation_invariance(dns):
    """T4: shuffling Gaussians with distinct depths leaves the image unchanged up to fp32 ordering of equal keys."""
    inp, viewmat, K, _ = gsplat_inputs(3000, 96, 80, focal=70.0, seed=12, anisotropic=True)
    perm = torch.randperm(3000, generator=torch.Generator().manual_seed(0))
    a = {k: v.to(DEV) for k, v in inp.items()}
    b = {k: v[perm].to(DEV) for k, v in inp.items()}
    kw = dict(viewmats=viewmat.to(DEV), Ks=K.to(DEV), width=96, height=80, packed=False, sh_degree=3, render_mode="RGB+ED")
    ra, aa, ia = dns.rasterization(**a, **kw)
    rb, ab, ib = dns.rasterization(**b, **kw)
    assert_equal_int(ib["radii"][0], ia["radii"][0][perm.to(DEV)], "radii under permutation")
    assert_close(rb, ra, "render under permutation", 1e-6)
    assert_close(ab, aa, "alpha under permutation", 1e-6)


# ------------------------------------------------------------------------------------------------
# the legacy normal pass and the get_outputs mirror


def test_rasterize_gaussians_legacy_dropin(dns, orc):
    """gsplat.rasterize_gaussians as called at dn_model.py:564-575 (background defaults to ones)."""
    inp, viewmat, K, _ = gsplat_inputs(5000, 160, 112, focal=110.0, seed=13, anisotropic=True)
    with torch.no_grad():
        _, _, info = orc.rasterization(**inp, viewmats=viewmat, Ks=K, width=160, height=112, packed=False,
                                       sh_degree=3, render_mode="RGB+ED")
    g_ = torch.Generator().manual_seed(3)
    normals = torch.nn.functional.normalize(torch.randn(5000, 3, generator=g_), dim=-1)
    args = dict(xys=info["means2d"][0], conics=info["conics"][0], colors=normals, opacity=inp["opacities"][:, None])
    co = {k: v.detach().clone().requires_grad_(True) for k, v in args.items()}
    cg = {k: v.detach().to(DEV).clone().requires_grad_(True) for k, v in args.items()}
    out_o = orc.rasterize_gaussians(co["xys"], info["depths"][0], info["radii"][0], co["conics"], info["tiles_per_gauss"][0],
                                    co["colors"], co["opacity"], 112, 160, 16)
    out_g = dns.rasterize_gaussians(cg["xys"], info["depths"][0].to(DEV), info["radii"][0].to(DEV), cg["conics"],
                                    info["tiles_per_gauss"][0].to(DEV), cg["colors"], cg["opacity"], 112, 160, 16)
    assert out_g.shape == (112, 160, 3)
    keep = keep_mask(orc.last_borderline, "legacy drop-in")
    assert_close(out_g, out_o, "legacy render", bound=flip_bound_linear(orc.last_flip_weight, torch.ones(3)))
    (v,) = cotangents([out_o.shape], 4)
    v = zero_borderline(v, keep)
    (out_o * v).sum().backward()
    (out_g * v.to(DEV)).sum().backward()
    for k in co:
        assert_close(cg[k].grad, co[k].grad, "legacy grad " + k)


MODES = {"fused_hip_postops": dict(fused=True, fused_postops=True), "fused_torch_postops": dict(fused=True, fused_postops=False),
         "two_call": dict(fused=False)}


@pytest.mark.parametrize("mode", sorted(MODES))
def test_get_outputs_mirror_matches_reference_sequence(dns, orc, mode):
    """DNSplatterModel.get_outputs (dn_model.py:404-612): our fused one-pass renderer and our two-call
    drop-ins against the reference's own op sequence run on the oracle."""
    from dn_splatter_amd import synthetic

    N, W, H = 10_000, 256, 256
    gp = synthetic.make_gauss_params(N, sh_rest_std=0.1, seed=0)
    g_ = torch.Generator().manual_seed(21)
    gp["scales"] = (gp["scales"].detach() + torch.randn(N, 3, generator=g_) * 0.5).requires_grad_(True)
    cam = synthetic.orbit_camera(3, width=W, height=H, focal=160.0)

    hip, ora, keep = _mirror_pair(dns, orc, gp, cam, MODES[mode], what="mirror " + mode)
    out_g, p_g, m_g = hip
    out_o, p_o, m_o = ora
    assert set(out_g) == {"rgb", "depth", "normal", "surface_normal", "accumulation", "background"}
    if "_saturation_flag" in m_g.last_info:      # fused + HIP post-ops: no opacity above the cap here, the clamp-free twin ran
        assert int(m_g.last_info["_saturation_flag"].item()) == 0
    _check_mirror(hip, ora, keep, "mirror " + mode, ints=False)
    assert_equal_int(m_g.radii, m_o.radii, "radii")
    # the fused path bins over tight tile boxes internally (info["tiles_bin"]); what it reports is gsplat's count
    assert_equal_int(m_g.num_tiles_hit.reshape(-1), m_o.num_tiles_hit.reshape(-1), "num_tiles_hit")
    if m_g.last_info.get("tight_tiles"):
        assert bool((m_g.last_info["tiles_bin"].reshape(-1).cpu() <= m_o.num_tiles_hit.reshape(-1)).all())
    # surface_normal is a finite-difference stencil of the depth image (dn_model.py:589-603): depth differences of 1e-5 are
    # amplified by ~fx/d, so the IMAGES of the two sides are not comparable at a fixed tolerance.  What is pinned instead, at a
    # stated tolerance: (1) the product's stencil (dnsplat_dn_depth_normals) applied to the ORACLE's depth image equals the
    # reference sequence's surface_normal of that same depth to 2e-5 absolute (values in [0, 1]); (2) the product's
    # surface_normal IS that stencil of the product's own depth image (2e-5; the fused HIP post-ops produce both in one launch)
    def hip_stencil(depth_hw1, alpha_hw=None, dmax=0.0, want_depth=False):
        from dn_splatter_amd import _lib, _ops
        d = depth_hw1.detach().reshape(H, W).to(DEV).float().contiguous()
        al = torch.ones_like(d) if alpha_hw is None else alpha_hw.detach().reshape(H, W).to(DEV).float().contiguous()
        dmax = torch.full((1,), float(dmax), device=DEV)
        d_out, sn_out = torch.empty_like(d), torch.empty(H, W, 3, device=DEV)
        _lib.run("dnsplat_dn_depth_normals", _lib.lib().dnsplat_dn_depth_normals, W, H, float(cam.fx), float(cam.fy), float(cam.cx),
                 float(cam.cy), _ops._ptr(d), _ops._ptr(al), _ops._ptr(dmax), _ops._ptr(d_out), _ops._ptr(sn_out), _ops._stream())
        torch.cuda.synchronize()
        return (sn_out.cpu(), d_out.cpu()) if want_depth else sn_out.cpu()

    d1 = (hip_stencil(out_o["depth"]) - out_o["surface_normal"].detach()).abs().max().item()
    assert d1 <= 2e-5, f"HIP depth->normal stencil on the oracle's depth vs the reference sequence: {d1:.3e}"
    d2 = (hip_stencil(out_g["depth"]) - out_g["surface_normal"].detach().cpu()).abs().max().item()
    assert d2 <= 2e-5, f"product surface_normal vs the stencil of the product's own depth: {d2:.3e}"
    sn = out_g["surface_normal"].detach().cpu()
    assert torch.equal(sn[0], torch.full_like(sn[0], 0.5)) and torch.equal(sn[:, -1], torch.full_like(sn[:, -1], 0.5))
    # (3) the kernel's alpha <= 0 fill (dn_model.py:533-537: where(alpha > 0, depth, depth.max())) on the taps of the stencil
    # (with alphas = 1 the fill of the neighbour taps is never taken).  The oracle's accumulation and depth with holes
    # punched into them — a block, a single pixel, an image corner, a border run — go through the kernel with the real depth
    # maximum and through the reference sequence's own two steps (the torch mirror of normal_from_depth_image, pinned to the
    # reference in tests/test_reference_golden.py).
    from dn_splatter_amd.model import normal_from_depth_image
    al = out_o["accumulation"].detach().reshape(H, W).clone()
    raw = torch.where(al > 0, out_o["depth"].detach().reshape(H, W), torch.zeros(()))
    for sl in ((slice(40, 90), slice(60, 130)), (slice(100, 101), slice(200, 201)), (slice(0, 9), slice(0, 5)), (slice(H - 1, H), slice(30, 99)),
               (slice(120, 140), slice(W - 3, W))):
        al[sl], raw[sl] = 0.0, 0.0
    dmax = float(raw.detach().max())
    filled = torch.where(al > 0, raw, torch.full((), dmax))
    sn_ref = normal_from_depth_image(filled[..., None], float(cam.fx), float(cam.fy), float(cam.cx), float(cam.cy), (W, H),
                                     torch.eye(4)) @ torch.diag(torch.tensor([1.0, -1.0, -1.0]))
    sn_ref = (1 + sn_ref) / 2
    sn_k, d_k = hip_stencil(raw, al, dmax, want_depth=True)
    assert torch.equal(d_k, filled), "depth fill where(alpha > 0, depth, max) differs from the reference sequence"
    d3 = (sn_k - sn_ref).abs().max().item()
    assert d3 <= 2e-5, f"HIP depth->normal stencil with alpha <= 0 holes vs the reference sequence: {d3:.3e}"
    assert_close(p_g["normals"], p_o["normals"], "gauss_params['normals'] (dn_model.py:558)", 1e-5)


@pytest.mark.parametrize("name,cfg_kw,step,hip_kw", [
    ("sh_degree_0", dict(sh_degree=0), None, dict(fused=True)),                         # dn_model.py:491-493: sigmoid(colours), sh_degree=None
    ("no_normals", dict(predict_normals=False), None, dict(fused=True)),                # dn_model.py:542: zeros for the normal image
    ("sh_schedule", dict(), 1500, dict(fused=True)),                                    # dn_model.py:487-490: min(step // 1000, 3) = 1 active band set
    ("antialiased", dict(rasterize_mode="antialiased"), None, dict(fused=True)),        # compensated opacities in pass 1 only (dn_model.py:571)
])
def test_get_outputs_mirror_config_variants(dns, orc, name, cfg_kw, step, hip_kw):
    """The branches of get_outputs the default configuration does not take, product (it picks the fused or the two-call
    route itself) against the reference sequence on the oracle."""
    from dn_splatter_amd import RendererConfig, synthetic

    N, W, H = 6000, 192, 144
    gp = synthetic.make_gauss_params(N, sh_rest_std=0.2, seed=8)
    g_ = torch.Generator().manual_seed(22)
    gp["scales"] = (gp["scales"].detach() + torch.randn(N, 3, generator=g_) * 0.5).requires_grad_(True)
    if name == "sh_degree_0":
        gp["features_rest"] = torch.zeros(N, 0, 3, requires_grad=True)      # num_sh_bases(0) = 1: no higher bands (dn_model.py:139-154)
    cam = synthetic.orbit_camera(5, width=W, height=H, focal=130.0)
    hip, ora, keep = _mirror_pair(dns, orc, gp, cam, hip_kw, what="mirror " + name, config=RendererConfig(**cfg_kw), step=step)
    _check_mirror(hip, ora, keep, "mirror " + name, ints=False)
    assert_equal_int(hip[2].radii, ora[2].radii, "radii")
    if name == "no_normals":
        assert float(hip[0]["normal"].detach().abs().max()) == 0.0
    if name == "sh_schedule":
        assert float(hip[1]["features_rest"].grad[:, 3:].abs().max()) == 0.0      # bands 2 and 3 are inactive at step 1500


def test_densify_stats_match_nerfstudio_after_train(dns):
    """N3: one kernel == the boolean-mask torch sequence of SplatfactoModel.after_train (see densify.py)."""
    from dn_splatter_amd import synthetic

    N, W, H = 20_000, 320, 240
    gp = synthetic.make_gauss_params(N, sh_rest_std=0.1, seed=3, device=DEV)
    m = dns.DNSplatterRenderer(gp, fused=True)
    stats = dns.DensifyStats(N, DEV)
    ref = dict(g=torch.zeros(N, device=DEV), v=torch.ones(N, device=DEV), m=torch.zeros(N, device=DEV))
    for view in (0, 3):
        cam = synthetic.orbit_camera(view, width=W, height=H, focal=200.0).to(DEV)
        out = m.get_outputs(cam)
        for k in ("means", "scales", "quats", "features_dc", "features_rest", "opacities"):
            gp[k].grad = None
        (out["rgb"].sum() + out["depth"].sum()).backward()
        stats.after_train(m, W, H)
        visible = m.radii > 0
        ref["g"][visible] += m.xys.absgrad[0][visible].norm(dim=-1)
        ref["v"][visible] += 1
        ref["m"][visible] = torch.maximum(ref["m"][visible], m.radii[visible] / float(max(W, H)))
    assert_close(stats.xys_grad_norm, ref["g"], "xys_grad_norm", 1e-6)
    assert torch.equal(stats.vis_counts, ref["v"])
    assert torch.equal(stats.max_2Dsize, ref["m"])
    assert float(stats.vis_counts.max()) == 3.0 and float(stats.max_2Dsize.max()) > 0


@pytest.mark.parametrize("step,kw", [(3500, {}), (2500, {}), (16000, {}), (3500, dict(cull_alpha_thresh=0.005))])
def test_device_refinement_matches_the_reference_sequence(dns, step, kw):
    """N3: dnsplat_densify_classify / dnsplat_densify_split and the whole device-side refinement step fed by the renderer's
    own statistics (two cameras' backward passes through dnsplat_densify_stats), against the reference sequence
    (oracle/densify_ref.py) on the same noise: same decisions, same new Gaussian set, same Adam moments."""
    from dn_splatter_amd import densify, synthetic
    from oracle import densify_ref as ref

    N, W, H = 20_000, 320, 240
    gp = synthetic.make_gauss_params(N, sh_rest_std=0.1, seed=3, device=DEV)
    g = torch.Generator(device=DEV).manual_seed(4)
    with torch.no_grad():
        gp["scales"] += torch.randn(N, 3, device=DEV, generator=g) * 1.2 - 1.5
        gp["opacities"] += torch.randn(N, 1, device=DEV, generator=g) * 2.5
    m = dns.DNSplatterRenderer(gp, fused=True)
    stats = dns.DensifyStats(N, DEV)
    for view in (0, 3):
        cam = synthetic.orbit_camera(view, width=W, height=H, focal=200.0).to(DEV)
        out = m.get_outputs(cam)
        for k in GRAD_NAMES:
            gp[k].grad = None
        (out["rgb"].sum() * 40 + out["depth"].sum()).backward()
        stats.after_train(m, W, H)
    params = {k: v.detach() for k, v in gp.items()}
    adam = {k: {"exp_avg": torch.randn(v.shape, device=DEV, generator=g), "exp_avg_sq": torch.rand(v.shape, device=DEV, generator=g)}
            for k, v in params.items() if k != "normals"}
    cfg = densify.RefineConfig(**kw)
    do_densify = step < cfg.stop_split_at
    # the two kernels against their torch restatements
    flags = densify.classify(params, stats, cfg, step, (H, W), do_densify)
    flags_t = ref.classify_torch(params, stats, cfg, step, (H, W), do_densify)
    assert int((flags != flags_t).sum()) <= 2, int((flags != flags_t).sum())      # a threshold met within an ulp of exp()
    parents = torch.nonzero((flags & 1) != 0).reshape(-1)
    if parents.numel():
        noise = torch.randn(2 * parents.numel(), 3, device=DEV, generator=g)
        cm, cs = densify.split_children(params, parents, noise)
        cm_t, cs_t = ref.split_children_torch(params, parents, noise)
        assert_close(cm, cm_t, "split children means", 1e-6)
        assert_close(cs, cs_t, "split children log-scales", 1e-6)
    # the whole step on device vs the reference-structured sequence (run in torch on the same device, same noise)
    seen = {}

    def split_spy(p, par, noise):
        seen["noise"] = noise
        return densify.split_children(p, par, noise)

    new, new_adam, report = densify.refinement_after(params, stats, cfg, step, 100, (H, W), adam_state=adam, seed=5, split_fn=split_spy)
    model = ref.Model(params, cfg, step, 100, (H, W), stats.xys_grad_norm.clone(), stats.vis_counts.clone(), stats.max_2Dsize.clone(), adam)
    model.refinement_after(lambda n: seen["noise"] if n else torch.zeros(0, 3, device=DEV))
    if int((flags != flags_t).sum()) == 0:
        for k in params:
            assert new[k].shape == model.gauss_params[k].shape, (k, new[k].shape, model.gauss_params[k].shape)
            assert_close(new[k], model.gauss_params[k], "refined " + k, 1e-6)
        for k in adam:
            assert torch.equal(new_adam[k]["exp_avg"], model.adam[k]["exp_avg"]), k
    assert report["n_after"] == new["means"].shape[0]
    if do_densify:
        assert report["n_split"] > 50 and report["n_dup"] > 50 and report["n_culled"] > 50, report
        # the refined set renders: shapes stay consistent through the renderer
        out = dns.DNSplatterRenderer({k: v for k, v in new.items()}, fused=True).get_outputs(synthetic.orbit_camera(1, width=W, height=H, focal=200.0).to(DEV))
        assert bool(torch.isfinite(out["rgb"]).all())
    else:
        assert report["n_split"] == 0 and report["n_culled"] > 50, report


@pytest.mark.parametrize("W,H", [(64, 48), (75, 53), (256, 200), (11, 11)])
def test_fused_loss_matches_the_torch_loss_stack(dns, W, H):
    """N2: dnsplat_dn_loss (value + cotangents) == autograd over the PyTorch restatement of get_loss_dict."""
    from dn_splatter_amd import torch_losses as tl
    from dn_splatter_amd.fused_loss import dn_loss_fused

    batch = tl.synthetic_batch(W, H, DEV, seed=W)
    batch["mono_depth"][: H // 3, : W // 4] = 0.0          # part of the depth map invalid (<= depth_tolerance)
    g = torch.Generator().manual_seed(1000 + H)
    base = {"rgb": torch.rand(H, W, 3, generator=g), "depth": torch.rand(H, W, 1, generator=g) * 9 + 0.2,
            "normal": torch.rand(H, W, 3, generator=g)}
    base["rgb"][0, 0] = batch["image"][0, 0].cpu()          # exact ties: sign(0) must be 0 on both sides
    scales = torch.randn(500, 3, generator=g)

    def run(fn):
        out = {k: v.clone().to(DEV).requires_grad_(True) for k, v in base.items()}
        sc = scales.clone().to(DEV).requires_grad_(True)
        loss = fn(out, batch, sc)
        loss.backward()
        return loss.detach(), {k: v.grad for k, v in out.items()}, sc.grad

    l_t, g_t, s_t = run(tl.dn_loss)
    l_f, g_f, s_f = run(dn_loss_fused)
    assert abs(float(l_f) - float(l_t)) <= 2e-5 * abs(float(l_t)), (float(l_f), float(l_t))
    for k in g_t:
        assert_close(g_f[k], g_t[k], "d loss / d " + k, 2e-4)     # SSIM sensitivities divide by ~1e-4-sized variances
    assert_close(s_f, s_t, "d loss / d scales", 1e-6)
    # without depth / normal supervision
    nb = {"image": batch["image"]}
    l_t, g_t, _ = run(lambda o, b, s: tl.dn_loss(o, nb, s))
    l_f, g_f, _ = run(lambda o, b, s: dn_loss_fused(o, nb, s))
    assert abs(float(l_f) - float(l_t)) <= 2e-5 * abs(float(l_t))
    assert_close(g_f["rgb"], g_t["rgb"], "rgb-only loss gradient", 2e-4)
    assert float(g_f["depth"].abs().max()) == 0.0 and float(g_f["normal"].abs().max()) == 0.0


@pytest.mark.parametrize("W,H", [(64, 48), (75, 53), (256, 200), (11, 11), (333, 17)])
def test_ssim_module_matches_the_torch_ssim(dns, W, H):
    """dnsplat_ssim (ABI 15): splatfacto's SSIM module alone — value and gradient w.r.t. the rendered image == autograd over
    torch_losses.ssim (pytorch_msssim's ten grouped conv2d calls), in fp64 on the torch side so that the comparison measures the
    kernel and not MIOpen's summation order.  Also through the module form (``fused_loss.SSIM``: [1,3,H,W] views, either argument
    order) and inside the otherwise-PyTorch loss stack (``dn_loss(..., ssim_impl="hip")``)."""
    from dn_splatter_amd import torch_losses as tl
    from dn_splatter_amd.fused_loss import SSIM, ssim_hip

    g = torch.Generator().manual_seed(77 + W)
    gt = torch.rand(H, W, 3, generator=g)
    pred = (gt + 0.2 * torch.randn(H, W, 3, generator=g)).clamp(0, 1)
    pred[: H // 2, : W // 2] = gt[: H // 2, : W // 2]                   # identical windows: SSIM == 1, variances cancel exactly
    x64 = pred.double().to(DEV).requires_grad_(True)
    s_t = tl.ssim(x64, gt.double().to(DEV))
    s_t.backward()
    x = pred.to(DEV).requires_grad_(True)
    s_h = ssim_hip(x, gt.to(DEV))
    (3.0 * s_h).backward()                                              # an upstream factor reaches the gradient
    s_h, s_t = s_h.detach(), s_t.detach()
    assert abs(float(s_h) - float(s_t)) <= 2e-5 * abs(float(s_t)), (float(s_h), float(s_t))
    assert_close(x.grad / 3.0, x64.grad.float(), "d SSIM / d pred", 2e-4)
    # value only (no gradient requested: the second launch is skipped)
    with torch.no_grad():
        assert float(ssim_hip(pred.to(DEV), gt.to(DEV))) == float(s_h)
    # module form, as nerfstudio calls it: self.ssim(gt[1,3,H,W], pred[1,3,H,W])
    m = SSIM(data_range=1.0, size_average=True, channel=3)
    x2 = pred.to(DEV).requires_grad_(True)
    s_m = m(gt.to(DEV).permute(2, 0, 1)[None], x2.permute(2, 0, 1)[None])
    s_m.backward()
    assert float(s_m.detach()) == float(s_h)
    assert torch.equal(x2.grad * 3.0, x.grad) or float((x2.grad * 3.0 - x.grad).abs().max()) <= 1e-6 * float(x.grad.abs().max())
    with pytest.raises(NotImplementedError):
        SSIM(data_range=255.0)
    if W > 11:
        # the whole stack, every other term in PyTorch
        batch = tl.synthetic_batch(W, H, DEV, seed=W)
        base = {"rgb": pred, "depth": torch.rand(H, W, 1, generator=g) * 9 + 0.2, "normal": torch.rand(H, W, 3, generator=g)}
        scales = torch.randn(100, 3, generator=g)

        def run(**kw):
            out = {k: v.clone().to(DEV).requires_grad_(True) for k, v in base.items()}
            sc = scales.clone().to(DEV).requires_grad_(True)
            loss = tl.dn_loss(out, batch, sc, **kw)
            loss.backward()
            return float(loss.detach()), {k: v.grad for k, v in out.items()}

        l_t, g_t = run()
        l_h, g_h = run(capturable=True, ssim_impl="hip")
        assert abs(l_h - l_t) <= 2e-5 * abs(l_t)
        for k in g_t:
            assert_close(g_h[k], g_t[k], "hip-SSIM stack d loss / d " + k, 2e-4)
        # ... and with all three swapped modules (install_losses): SSIM, EdgeAwareLogL1, TVLoss on HIP, the rest in PyTorch
        batch["mono_depth"][: H // 3, : W // 4] = 0.0                   # part of the depth map invalid
        l_t, g_t = run()
        l_m, g_m = run(ssim_impl="hip", hip_modules=True)
        assert abs(l_m - l_t) <= 2e-5 * abs(l_t)
        for k in g_t:
            assert_close(g_m[k], g_t[k], "hip-modules stack d loss / d " + k, 2e-4)


def test_spatially_reordered_gaussians_render_the_same_frame(dns):
    """densify.spatial_order / reorder (what refinement_after(spatial_reorder=True) leaves behind): the same Gaussians in another
    row order give the same images — bit for bit in deterministic mode's forward (the depth sort decides the blend order, rows
    only break exact depth ties) — the same per-Gaussian outputs row for row under the permutation, and the same parameter
    gradients up to the order of the fp32 atomics."""
    from dn_splatter_amd import densify, synthetic

    N, W, H = 60_000, 320, 240
    gp = synthetic.make_gauss_params(N, sh_rest_std=0.1, seed=9, device=DEV)
    perm = densify.spatial_order(gp["means"])
    assert perm.device.type == "cuda" and torch.equal(torch.sort(perm)[0], torch.arange(N, device=DEV))
    new, _ = densify.reorder(gp, perm)
    gq = {k: (v.clone().requires_grad_(True) if k != "normals" else v) for k, v in new.items()}
    cam = synthetic.orbit_camera(1, width=W, height=H, focal=200.0).to(DEV)
    g = torch.Generator().manual_seed(3)
    cot = {k: torch.randn(H, W, c, generator=g).to(DEV) for k, c in (("rgb", 3), ("depth", 1), ("normal", 3))}

    def run(params):
        m = dns.DNSplatterRenderer(params, fused=True)
        out = m.get_outputs(cam)
        torch.autograd.backward([out[k] for k in cot], [cot[k] for k in cot])
        return out, m

    out_a, m_a = run(gp)
    out_b, m_b = run(gq)
    for k in ("rgb", "depth", "normal", "surface_normal", "accumulation"):
        assert_close(out_b[k], out_a[k], "reordered " + k, 1e-6)
    assert torch.equal(m_b.radii, m_a.radii[perm]) and torch.equal(m_b.num_tiles_hit.reshape(-1), m_a.num_tiles_hit.reshape(-1)[perm])
    for k in ("means", "quats", "scales", "opacities", "features_dc", "features_rest"):
        assert_close(gq[k].grad, gp[k].grad[perm], "reordered d/d " + k, 1e-4)
    # the culled rows are runs along the curve: far fewer 64-row workgroups hold both kinds
    def mixed(radii):
        b = (radii[: (N // 64) * 64] > 0).view(-1, 64).float().mean(1)
        return float(((b > 0) & (b < 1)).float().mean())
    assert mixed(m_b.radii) < 0.5 * mixed(m_a.radii)


def test_batched_render_loop_equals_sequential(dns):
    """N4: get_outputs_batch — all cameras projected, binned (camera, tile, depth) and composited in ONE launch sequence —
    returns exactly what get_outputs returns per camera."""
    from dn_splatter_amd import synthetic

    N, W, H = 30_000, 320, 240
    gp = synthetic.make_gauss_params(N, sh_rest_std=0.1, seed=5, device=DEV)
    m = dns.DNSplatterRenderer(gp, fused=True)
    cams = [synthetic.orbit_camera(v, width=W, height=H, focal=200.0).to(DEV) for v in range(6)]
    with torch.no_grad():
        ref = [m.get_outputs(c) for c in cams]
    for policy in ("sync", "capacity"):
        dns.set_bin_policy(policy)
        try:
            for max_batch in (6, 4, 2):
                got = m.get_outputs_batch(cams, max_batch=max_batch)
                torch.cuda.synchronize()
                assert len(got) == len(cams)
                for a, b in zip(got, ref):
                    for k in ("rgb", "depth", "normal", "surface_normal", "accumulation"):
                        assert a[k].shape == b[k].shape and torch.equal(a[k], b[k]), (policy, max_batch, k)
                assert m.last_info["n_cameras"] == (max_batch if len(cams) % max_batch == 0 else len(cams) % max_batch)
        finally:
            dns.set_bin_policy("sync")


@pytest.mark.parametrize("W,H", [(160, 128), (200, 120)])
@pytest.mark.usefixtures("hip_deterministic")
def test_multi_camera_rasterization_equals_sequential_calls_and_oracle(dns, orc, W, H):
    """gsplat.rasterization with viewmats [C,4,4] (N4): the C = 4 result equals 4 single-camera calls bit for bit (images,
    projections, per-camera tile lists), its integer outputs — flatten_ids = camera * N + g, isect_offsets [C,th,tw],
    isect_ids with the camera bits of SURVEY.md A.3 — equal the oracle's, and the gradient of a loss over all cameras is the
    sum of the per-camera gradients.  (200 x 120: a frame whose height is not a multiple of the tile size, so the cameras'
    tile grids do not tile a stacked image.)"""
    from dn_splatter_amd import synthetic

    N, C = 4000, 4
    inp, _vm, K, _ = gsplat_inputs(N, W, H, focal=0.7 * W, seed=21, anisotropic=True)
    vms = torch.cat([dns.get_viewmat(synthetic.orbit_camera(v, width=W, height=H, focal=0.7 * W).camera_to_worlds) for v in (0, 2, 3, 5)])
    Ks = K.expand(C, 3, 3).contiguous()
    kw = dict(width=W, height=H, packed=False, sh_degree=3, render_mode="RGB+ED", absgrad=True)
    gi = to_leaf(inp, DEV)
    r_b, a_b, info_b = dns.rasterization(**gi, viewmats=vms.to(DEV), Ks=Ks.to(DEV), **kw)
    assert r_b.shape == (C, H, W, 4) and a_b.shape == (C, H, W, 1) and info_b["n_cameras"] == C
    assert info_b["radii"].shape == (C, N) and info_b["means2d"].shape == (C, N, 2)
    v_r, v_a = cotangents([r_b.shape, a_b.shape], 5)
    info_b["means2d"].retain_grad()
    ((r_b * v_r.to(DEV)).sum() + (a_b * v_a.to(DEV)).sum()).backward()
    # sequential single-camera calls
    gs = to_leaf(inp, DEV)
    base = 0
    th, tw = info_b["tile_height"], info_b["tile_width"]
    for c in range(C):
        r, a, info = dns.rasterization(**gs, viewmats=vms[c:c + 1].to(DEV), Ks=Ks[c:c + 1].to(DEV), **kw)
        assert torch.equal(r[0], r_b[c]) and torch.equal(a[0], a_b[c]), f"camera {c}: batched image differs"
        for k in ("radii", "means2d", "depths", "conics", "tiles_per_gauss"):
            assert torch.equal(info[k][0], info_b[k][c]), (c, k)
        n = info["n_isects"]
        assert torch.equal(info["flatten_ids"] + c * N, info_b["flatten_ids"][base:base + n]), f"camera {c}: tile lists"
        assert torch.equal(info["isect_offsets"][0] + base, info_b["isect_offsets"][c])
        base += n
        info["means2d"].retain_grad()
        ((r * v_r[c:c + 1].to(DEV)).sum() + (a * v_a[c:c + 1].to(DEV)).sum()).backward()     # accumulates over the cameras
        assert_close(info_b["means2d"].grad[c], info["means2d"].grad[0], f"camera {c} means2d.grad", 1e-5)
        assert_close(info_b["means2d"].absgrad[c], info["means2d"].absgrad[0], f"camera {c} means2d.absgrad", 1e-5)
    assert base == info_b["n_isects"]
    for k in gi:
        assert_close(gi[k].grad, gs[k].grad, "batched grad " + k, 1e-5)              # atomics order only
    # the oracle's batch (gsplat key layout)
    with torch.no_grad():
        _r, _a, info_o = orc.rasterization(**inp, viewmats=vms, Ks=Ks, **kw)
    for k in INT_KEYS:
        assert_equal_int(info_b[k], info_o[k], "batch " + k)
    assert_equal_int(info_b["isect_ids"], info_o["isect_ids"], "batch isect_ids (camera bits)")


@pytest.mark.parametrize("layout,deferred", [("split", False), ("split", True), ("cat", False)])
@pytest.mark.usefixtures("hip_deterministic")
def test_sh_factor_exchange_rebuilds_the_multi_camera_gradient(dns, layout, deferred):
    """Data parallel: all-gathering the 3 colour gradients per Gaussian (+ each camera's position) and rebuilding the sum equals
    averaging the 192-byte coefficient gradients of the cameras (dp.ShFactorExchange, dnsplat_sh_grads_from_factors) — checked here
    with three cameras rendered one after the other on one GPU; the geometry gradients are untouched by the mode.
    ``deferred`` (what graph.GraphedDpStep sets): the slab is written by dnsplat_project_bwd itself (dnsplat_proj_grads.sh_factors)
    instead of by dnsplat_sh_factors ahead of it — same slab, same rebuilt gradients."""
    import ctypes

    from dn_splatter_amd import _lib, _ops, dp, synthetic

    N, W, H = 20_000, 320, 240
    gp0 = synthetic.make_gauss_params(N, sh_rest_std=0.2, seed=9)
    gp0["scales"] = gp0["scales"].detach() + torch.randn(N, 3, generator=torch.Generator().manual_seed(10)) * 0.5
    cams = [synthetic.orbit_camera(v, width=W, height=H, focal=200.0).to(DEV) for v in (0, 2, 5)]

    def one(cam, exchange):
        gp = {k: v.detach().to(DEV).clone().requires_grad_(k != "normals") for k, v in gp0.items()}
        dns.set_sh_exchange(exchange)
        try:
            if layout == "split":
                m = dns.DNSplatterRenderer(gp, fused=True)
                out = m.get_outputs(cam)
                loss = (out["rgb"] * torch.linspace(0.5, 1.5, 3, device=DEV)).sum() + out["depth"].sum() + out["normal"].sum()
                sh = None
            else:
                sh = torch.cat([gp["features_dc"][:, None], gp["features_rest"]], 1).detach().requires_grad_(True)
                q = gp["quats"] / gp["quats"].norm(dim=-1, keepdim=True)
                r, a, _ = dns.rasterization(gp["means"], q, torch.exp(gp["scales"]), torch.sigmoid(gp["opacities"]).squeeze(-1), sh,
                                            dns.get_viewmat(cam.camera_to_worlds), cam.get_intrinsics_matrices(), W, H,
                                            packed=False, sh_degree=3, render_mode="RGB+ED")
                loss = (r[..., :3] * torch.linspace(0.5, 1.5, 3, device=DEV)).sum() + r[..., 3].sum() + a.sum()
            loss.backward()
        finally:
            dns.set_sh_exchange(None)
        torch.cuda.synchronize()
        return gp, sh

    dense = [one(c, None) for c in cams]
    ex = dp.ShFactorExchange(own_rows=False)      # the rebuild-every-row form (what runs at world >= 2)
    ex.deferred = deferred
    if layout == "cat":
        # gsplat's concatenated [N,16,3] layout: its gradient is an intermediate autograd tensor that dp.allreduce_gradients
        # cannot reach, so an active exchange must be IGNORED and the kernel must write the coefficient rows itself
        for i, c in enumerate(cams):
            gp_f, sh_f = one(c, ex)
            assert ex.meta is None
            assert_close(sh_f.grad, dense[i][1].grad, "cat layout ignores the factor exchange", 1e-6)
        # the rebuild kernel itself still supports the [N,16,3] destination
        return
    factors = []
    for c in cams:
        gp_f, sh_f = one(c, ex)
        assert ex.meta is not None
        factors.append(ex.mine.clone())
        ex.meta = None
        for k in ("means", "scales", "quats", "opacities"):       # geometry gradients do not depend on the mode
            ref = dense[len(factors) - 1][0][k].grad
            assert_close(gp_f[k].grad, ref, "factor mode grad " + k, 1e-4)    # two runs differ by atomic order only
    fac = torch.stack(factors).contiguous()
    assert fac.shape == (3, 3 * N + 4)                 # one slab per camera: [N,3] colour gradients | camera position | pad
    means_dev = gp0["means"].detach().to(DEV).contiguous()
    if layout == "split":
        v0 = torch.empty(N, 3, device=DEV)
        vN = torch.empty(N, 15, 3, device=DEV)
        _lib.check(_lib.lib().dnsplat_sh_grads_from_factors(N, 3, _ops._ptr(fac), _ops._ptr(means_dev), 3, 16, 1.0 / 3, _ops._ptr(v0), 3,
                                                            _ops._ptr(vN), 45, _ops._stream()), "dnsplat_sh_grads_from_factors")
        ref0 = sum(d[0]["features_dc"].grad for d in dense) / 3
        refN = sum(d[0]["features_rest"].grad for d in dense) / 3
        assert_close(v0, ref0, "rebuilt features_dc gradient", 1e-5)
        assert_close(vN, refN, "rebuilt features_rest gradient", 1e-5)
    else:
        vc = torch.empty(N, 16, 3, device=DEV)
        _lib.check(_lib.lib().dnsplat_sh_grads_from_factors(N, 3, _ops._ptr(fac), _ops._ptr(means_dev), 3, 16, 1.0 / 3, _ops._ptr(vc), 48,
                                                            _ops._ptr(vc.view(-1)[3:]), 48, _ops._stream()),
                   "dnsplat_sh_grads_from_factors")
        ref = sum(d[1].grad for d in dense) / 3
        assert_close(vc, ref, "rebuilt SH coefficient gradient", 1e-5)
    # the single-rank exchange (what DNSPLAT_FORCE_DIST=1 exercises with RCCL) reproduces the dense gradient
    gp_1, sh_1 = one(cams[0], ex)
    if layout == "split":
        assert ex.finish(v_sh0=gp_1["features_dc"].grad, v_shN=gp_1["features_rest"].grad) == 0
    else:
        assert ex.finish(v_coeffs=sh_1.grad) == 0
    torch.cuda.synchronize()
    if layout == "split":
        assert_close(gp_1["features_rest"].grad, dense[0][0]["features_rest"].grad, "single-rank exchange", 1e-6)
        assert_close(gp_1["features_dc"].grad, dense[0][0]["features_dc"].grad, "single-rank exchange dc", 1e-6)
        # the default exchange at world 1: own rows — dnsplat_project_bwd writes the complete rows, finish() launches nothing
        ex1 = dp.ShFactorExchange()
        ex1.deferred = deferred
        gp_2, _ = one(cams[0], ex1)
        assert ex1.use_own_rows() and ex1.finish(v_sh0=gp_2["features_dc"].grad, v_shN=gp_2["features_rest"].grad) == 0
        torch.cuda.synchronize()
        assert torch.equal(gp_2["features_rest"].grad, dense[0][0]["features_rest"].grad), "own rows at world 1 != the single-GPU rows"
        assert torch.equal(gp_2["features_dc"].grad, dense[0][0]["features_dc"].grad)
    else:
        assert_close(sh_1.grad, dense[0][1].grad, "single-rank exchange", 1e-6)


@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.usefixtures("hip_deterministic")
def test_own_rows_and_visible_row_slabs_equal_the_dense_mean(dns, packed):
    """(a) own rows — each camera's coefficient rows written by its own dnsplat_project_bwd, pre-scaled by 1 / W,
    the other W - 1 cameras ADDED from the gathered slabs (dnsplat_sh_grads_add_factors); (b) slabs of the visible rows only (mask +
    block offsets + packed rows: dnsplat_visible_index, dnsplat_proj_grads.sh_packed, dnsplat_sh_grads_from_packed).  Three cameras
    one after the other on one GPU stand in for three ranks; every "rank" must end with the dense mean of the three gradients."""
    from dn_splatter_amd import _lib, _ops, dp, synthetic

    N, W, H = 20_000, 320, 240
    gp0 = synthetic.make_gauss_params(N, sh_rest_std=0.2, seed=9)
    gp0["scales"] = gp0["scales"].detach() + torch.randn(N, 3, generator=torch.Generator().manual_seed(10)) * 0.5
    cams = [synthetic.orbit_camera(v, width=W, height=H, focal=200.0).to(DEV) for v in (0, 2, 5)]
    n_views = len(cams)

    def one(cam, exchange):
        gp = {k: v.detach().to(DEV).clone().requires_grad_(k != "normals") for k, v in gp0.items()}
        dns.set_sh_exchange(exchange)
        try:
            m = dns.DNSplatterRenderer(gp, fused=True)
            out = m.get_outputs(cam)
            ((out["rgb"] * torch.linspace(0.5, 1.5, 3, device=DEV)).sum() + out["depth"].sum() + out["normal"].sum()).backward()
        finally:
            dns.set_sh_exchange(None)
        torch.cuda.synchronize()
        return gp, m

    dense = [one(c, None)[0] for c in cams]
    ref0 = sum(d["features_dc"].grad for d in dense) / n_views
    refN = sum(d["features_rest"].grad for d in dense) / n_views
    n_vis = []
    cap = None
    if packed:
        n_vis = [int((one(c, None)[1].radii > 0).sum()) for c in cams]
        cap = (max(n_vis) + 1023) // 1024 * 1024
        assert cap < N                                  # the slabs are smaller than dense ones
    ex = dp.ShFactorExchange(own_rows=True, packed=packed, capacity=cap)
    ex.scale_override = 1.0 / n_views
    own, slabs = [], []
    for c in cams:
        gp_f, m_f = one(c, ex)
        assert ex.meta is not None
        slabs.append(ex.mine.clone())
        ex.meta = None
        own.append((gp_f["features_dc"].grad.clone(), gp_f["features_rest"].grad.clone()))
    fac = torch.stack(slabs).contiguous()
    if packed:
        assert fac.shape[1] == _lib.lib().dnsplat_packed_slab_floats(N, cap) < 3 * N + 4
        hdr = fac.view(torch.int32)[:, :8].cpu()
        assert hdr[:, 0].tolist() == n_vis and hdr[:, 4].tolist() == [cap] * n_views and not ex.overflowed()
    means_dev = gp0["means"].detach().to(DEV).contiguous()
    for r in range(n_views):                            # "rank" r: its own pre-scaled rows + the others' slabs
        # own rows, pre-scaled: exactly the single-GPU rows x 1 / W
        assert_close(own[r][1], dense[r]["features_rest"].grad / n_views, f"rank {r}: own pre-scaled rows", 1e-6)
        v0, vN = own[r][0].clone(), own[r][1].clone()
        dp._rebuild_hip(fac, means_dev, N, n_views, 3, 16, None, v0, vN, skip_view=r, packed_capacity=cap)
        torch.cuda.synchronize()
        assert_close(v0, ref0, f"rank {r}: own rows + added slabs, band 0", 1e-5)
        assert_close(vN, refN, f"rank {r}: own rows + added slabs, bands 1..3", 1e-5)
    if packed:
        # rebuild-every-row form over packed slabs
        v0 = torch.full((N, 3), float("nan"), device=DEV)
        vN = torch.full((N, 15, 3), float("nan"), device=DEV)
        dp._rebuild_hip(fac, means_dev, N, n_views, 3, 16, None, v0, vN, skip_view=-1, packed_capacity=cap)
        torch.cuda.synchronize()
        assert_close(v0, ref0, "packed slabs, every row rebuilt, band 0", 1e-5)
        assert_close(vN, refN, "packed slabs, every row rebuilt", 1e-5)
        # a capacity below the visible count is REPORTED (rows beyond it are dropped by the sender)
        ex_small = dp.ShFactorExchange(own_rows=False, packed=True, capacity=1024)
        one(cams[0], ex_small)
        ex_small.gathered = ex_small.mine.clone()[None]
        assert ex_small.overflowed()
        ex_small.meta = None


@pytest.mark.parametrize("order", ["random", "morton"])
@pytest.mark.usefixtures("hip_deterministic")
def test_zero_rows_of_persistently_culled_gaussians_are_not_rewritten_and_stay_exact(dns, monkeypatch, order):
    """with the gradients in a dp.GradArena the projection backward skips the zero SH rows of Gaussians that
    were culled before and are culled again (dnsplat_proj_grads.sh_zero_state).  A pose sequence in which Gaussians enter and leave
    the frustum must give, frame by frame, the bits of the run without the bucket — in particular a Gaussian visible in frame k
    and hidden in frame k + 1 gets ZERO gradient in frame k + 1 — and the state words must describe memory after every frame."""
    from dn_splatter_amd import dp, synthetic

    N, W, H = 30_000, 320, 240
    gp = synthetic.make_gauss_params(N, sh_rest_std=0.2, seed=6, device=DEV)
    if order == "morton":
        # rows along a Morton curve (densify.spatial_order): a camera's culled Gaussians are runs of rows, whole 64-row workgroups
        # are culled and — second time round — written by nobody (sh_zero_state + zero_state_geometry: all or nothing per workgroup)
        from dn_splatter_amd import densify

        new, _ = densify.reorder(gp, densify.spatial_order(gp["means"]))
        gp = {k: (v.contiguous().requires_grad_(True) if k != "normals" else v.contiguous()) for k, v in new.items()}
    cams = [synthetic.orbit_camera(v, width=W, height=H, focal=260.0).to(DEV) for v in (0, 3, 0, 5, 3)]
    m = dns.DNSplatterRenderer(gp, fused=True)

    def frame(cam):
        for k in GRAD_NAMES:
            gp[k].grad = None
        out = m.get_outputs(cam)
        (out["rgb"].sum() + out["depth"].sum() + out["normal"].sum()).backward()
        torch.cuda.synchronize()
        return {k: gp[k].grad.clone() for k in GRAD_NAMES}, m.radii.clone()

    plain = [frame(c) for c in cams]
    monkeypatch.setenv("DNSPLAT_SH_ZERO_STATE", "1")
    arena = dp.GradArena(gp)
    assert arena.sh_state is not None and bool((arena.sh_state == -1).all())       # zero-filled bucket: every row known zero
    dns.set_grad_arena(arena)
    try:
        seen = torch.zeros(N, dtype=torch.bool, device=DEV)
        for i, c in enumerate(cams):
            g, radii = frame(c)
            assert arena.holds(gp["features_rest"].grad)
            for k in GRAD_NAMES:
                assert torch.equal(g[k], plain[i][0][k]), f"frame {i}: {k} differs from the run without the bucket"
            vis = radii > 0
            hidden_now = seen & ~vis
            assert int(hidden_now.sum()) > 0 or i == 0
            assert float(g["features_rest"][~vis].abs().max()) == 0.0 and float(g["features_dc"][~vis].abs().max()) == 0.0
            seen |= vis
            bits = torch.zeros(arena.sh_state.numel() * 64, dtype=torch.bool, device=DEV)
            words = arena.sh_state.clone()
            for b in range(64):
                bits[b::64] = ((words >> b) & 1).bool()
            assert torch.equal(bits[:N], ~vis), f"frame {i}: sh_zero_state does not describe the bucket"
            if order == "morton":
                culled_wgs = int((~vis[: (N // 64) * 64]).view(-1, 64).all(dim=1).sum())
                assert culled_wgs >= 8, culled_wgs                    # there ARE wholly culled workgroups (the skip is checked below)
        # an in-place all-reduce of the bucket (or any foreign write) must be followed by invalidate_sh_state(): after it every row is
        # written again
        arena.view("features_rest").fill_(7.0)
        arena.view("means").fill_(7.0)
        arena.invalidate_sh_state()
        g, radii = frame(cams[1])
        for k in GRAD_NAMES:
            assert torch.equal(g[k], plain[1][0][k]), k
        if order == "morton":
            # the same frame again: workgroups that are wholly culled are written by NOBODY now — a mark left in their rows survives
            vis = radii > 0
            wg_culled = (~vis[: (N // 64) * 64]).view(-1, 64).all(dim=1)
            rows = wg_culled.repeat_interleave(64)
            for k in GRAD_NAMES:
                gp[k].grad = None
            arena.view("means")[: rows.numel()][rows] = 5.0
            arena.view("features_rest")[: rows.numel()][rows] = 5.0
            out = m.get_outputs(cams[1])
            (out["rgb"].sum() + out["depth"].sum() + out["normal"].sum()).backward()
            torch.cuda.synchronize()
            assert bool((gp["means"].grad[: rows.numel()][rows] == 5.0).all()) and bool((gp["features_rest"].grad[: rows.numel()][rows] == 5.0).all())
            assert torch.equal(gp["means"].grad[: rows.numel()][~rows], plain[1][0]["means"][: rows.numel()][~rows])
            arena.flat.zero_()
    finally:
        dns.set_grad_arena(None)


@pytest.mark.parametrize("n_views,sh_degree", [(8, 3), (4, 2), (2, 1)])
def test_sh_rebuild_kernel_equals_torch_rebuild_at_node_scale(dns, n_views, sh_degree):
    """dnsplat_sh_grads_from_factors (what dp.ShFactorExchange._rebuild runs on every rank after the all-gather) against the
    torch rebuild the gloo tests substitute for it — autograd through a dense SH evaluation, summed over the views — for the
    8 views of a full node (BASELINE C4 / C5), random factors, odd N."""
    from dn_splatter_amd import dp
    from oracle import dense_ref

    N = 10_007
    g = torch.Generator().manual_seed(40 + n_views)
    means = torch.randn(N, 3, generator=g) * 3
    campos = torch.randn(n_views, 3, generator=g) * 8
    dirs = torch.nn.functional.normalize(means[None] - campos[:, None], dim=-1)      # what every rank re-derives per camera
    cols = torch.randn(n_views, N, 3, generator=g)
    fac = torch.cat([cols.reshape(n_views, -1), campos, torch.zeros(n_views, 1)], -1).contiguous()      # slabs as dnsplat_sh_factors writes them
    tot = torch.zeros(N, 16, 3)
    for v in range(n_views):
        co = torch.zeros(N, 16, 3, requires_grad=True)
        (dense_ref.sh_colors(sh_degree, dirs[v], co) * cols[v]).sum().backward()
        tot += co.grad
    tot /= n_views
    v0 = torch.full((N, 3), float("nan"), device=DEV)
    vN = torch.full((N, 15, 3), float("nan"), device=DEV)
    dp._rebuild_hip(fac.to(DEV), means.to(DEV), N, n_views, sh_degree, 16, None, v0, vN)
    torch.cuda.synchronize()
    assert_close(v0, tot[:, 0], "rebuilt band 0", 1e-5)
    assert_close(vN, tot[:, 1:], "rebuilt bands 1..3 (inactive bands must be zero, not untouched)", 1e-5)


def test_gradient_accumulation_into_the_flat_bucket(dns):
    """with a GradArena installed, a second backward while param.grad still lives in the bucket must give
    g1 + g2 (autograd's in-place +=), not 2 x g2 (which is what writing the new gradient into the aliased bucket gave)."""
    from dn_splatter_amd import dp, synthetic

    N, W, H = 20_000, 320, 240
    gp = synthetic.make_gauss_params(N, sh_rest_std=0.2, seed=6, device=DEV)
    cams = [synthetic.orbit_camera(v, width=W, height=H, focal=200.0).to(DEV) for v in (0, 3)]
    m = dns.DNSplatterRenderer(gp, fused=True)

    def grads(cam):
        for k in GRAD_NAMES:
            gp[k].grad = None
        out = m.get_outputs(cam)
        (out["rgb"].sum() + out["depth"].sum() + out["normal"].sum()).backward()
        return {k: gp[k].grad.clone() for k in GRAD_NAMES}

    g1, g2 = grads(cams[0]), grads(cams[1])
    arena = dp.GradArena(gp)
    dns.set_grad_arena(arena)
    try:
        for k in GRAD_NAMES:
            gp[k].grad = None
        for cam in cams:                                   # no zero_grad in between: accumulate
            out = m.get_outputs(cam)
            (out["rgb"].sum() + out["depth"].sum() + out["normal"].sum()).backward()
        torch.cuda.synchronize()
        for k in GRAD_NAMES:
            assert arena.holds(gp[k].grad), k              # the sum still lives in the bucket
            assert_close(gp[k].grad, g1[k] + g2[k], "accumulated grad " + k, 1e-5)
    finally:
        dns.set_grad_arena(None)


def test_background_width_follows_gsplat(dns, orc):
    """gsplat takes backgrounds [C, colour channels] and gives the depth channel of RGB+ED a zero background
    itself; a row of any other width must raise instead of making the kernel read past it."""
    inp, viewmat, K, _ = gsplat_inputs(2000, 96, 80, focal=70.0, seed=7, anisotropic=True)
    bg3 = torch.tensor([[0.2, 0.4, 0.6]])
    gi = {k: v.to(DEV) for k, v in inp.items()}
    kw = dict(width=96, height=80, packed=False, sh_degree=3, render_mode="RGB+ED")
    with torch.no_grad():
        r_o, a_o, info_o = orc.rasterization(**inp, viewmats=viewmat, Ks=K, backgrounds=bg3, **kw)
        r_g, a_g, _ = dns.rasterization(**gi, viewmats=viewmat.to(DEV), Ks=K.to(DEV), backgrounds=bg3.to(DEV), **kw)
        r_4, _a, _ = dns.rasterization(**gi, viewmats=viewmat.to(DEV), Ks=K.to(DEV),
                                       backgrounds=torch.tensor([[0.2, 0.4, 0.6, 0.0]], device=DEV), **kw)
    keep_mask(info_o["borderline"], "background width")
    assert_close(r_g, r_o, "render with a 3-wide background in RGB+ED", bound=render_bounds(info_o, r_o, a_o, "RGB+ED")[0])
    assert torch.equal(r_g, r_4)
    with pytest.raises(ValueError):
        dns.rasterization(**gi, viewmats=viewmat.to(DEV), Ks=K.to(DEV), backgrounds=torch.zeros(1, 2, device=DEV), **kw)


def test_small_frame_stays_in_the_millisecond_range(dns):
    """Regression guard.  A wave shuffle placed under a lane-dependent branch once made the emit kernel read a neighbour's
    count as 0 and spin through 2^32 / 32 guarded iterations: every result stayed bit-exact, but a 320 x 240 frame (300
    tiles: the smallest two-pass tile sort) took 5.6 s.  Parity tests do not notice that; this one does."""
    import time

    inp, viewmat, K, _ = gsplat_inputs(20_000, 320, 240, focal=200.0, seed=0)
    gi = to_leaf(inp, DEV)
    for it in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r, a, _info = dns.rasterization(**gi, viewmats=viewmat.to(DEV), Ks=K.to(DEV), width=320, height=240, packed=False,
                                        sh_degree=3, render_mode="RGB+ED", absgrad=True)
        (r.sum() + a.sum()).backward()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    assert dt < 0.25, f"fwd+bwd of a 320x240 / 20k-Gaussian frame took {dt * 1e3:.0f} ms"


def test_capacity_policy_recovers_from_an_overflowing_guess(dns):
    """'capacity' mode sizes the intersection buffers from earlier frames and enqueues everything without a host
    round-trip; when the guess is too small the emit + composite must be redone with the exact size."""
    from dn_splatter_amd import _ops

    inp, viewmat, K, _ = gsplat_inputs(20_000, 320, 240, focal=200.0, seed=15)
    gi = {k: v.to(DEV) for k, v in inp.items()}
    kw = dict(viewmats=viewmat.to(DEV), Ks=K.to(DEV), width=320, height=240, packed=False, sh_degree=3, render_mode="RGB+ED")
    try:
        dns.set_bin_policy("sync")
        r0, a0, i0 = dns.rasterization(**gi, **kw)
        dns.set_bin_policy("capacity")
        key = (torch.device(DEV), 20_000, 320, 240)
        assert key in _ops.BUFFERS.capacity_hint
        _ops.BUFFERS.capacity_hint[key] = 1000            # far below the real n_isects
        r1, a1, i1 = dns.rasterization(**gi, **kw)
        assert i1["n_isects"] == i0["n_isects"] > 1000
        assert torch.equal(r0, r1) and torch.equal(a0, a1)
        assert_equal_int(i1["flatten_ids"], i0["flatten_ids"], "flatten_ids after overflow")
        assert _ops.BUFFERS.capacity_hint[key] >= i0["n_isects"]
    finally:
        dns.set_bin_policy("sync")


def test_bin_policy_capacity_equals_sync(dns):
    inp, viewmat, K, _ = gsplat_inputs(20_000, 320, 240, focal=200.0, seed=14)
    gi = {k: v.to(DEV) for k, v in inp.items()}
    kw = dict(viewmats=viewmat.to(DEV), Ks=K.to(DEV), width=320, height=240, packed=False, sh_degree=3, render_mode="RGB+ED")
    try:
        dns.set_bin_policy("sync")
        r0, a0, i0 = dns.rasterization(**gi, **kw)
        dns.set_bin_policy("capacity")
        for _ in range(3):
            r1, a1, i1 = dns.rasterization(**gi, **kw)
            assert torch.equal(r0, r1) and torch.equal(a0, a1)
            assert_equal_int(i1["flatten_ids"], i0["flatten_ids"], "flatten_ids")
    finally:
        dns.set_bin_policy("sync")


# ------------------------------------------------------------------------------------------------
# committed golden vectors (builder-authored from the oracle: tests/golden/make_golden.py)


def test_against_golden_fixture(dns):
    import numpy as np
    import os

    path = os.path.join(os.path.dirname(__file__), "golden", "c1_small.npz")
    gold = np.load(path)
    N, W, H, focal, seed = (int(gold[k]) for k in ("N", "W", "H", "focal", "seed"))
    inp, viewmat, K, _ = gsplat_inputs(N, W, H, focal=float(focal), seed=seed, anisotropic=True)
    gi = to_leaf(inp, DEV)
    r, a, info = dns.rasterization(**gi, viewmats=viewmat.to(DEV), Ks=K.to(DEV), width=W, height=H, packed=False,
                                   sh_degree=3, render_mode="RGB+ED", absgrad=True)
    for k in INT_KEYS:
        assert_equal_int(info[k], torch.from_numpy(gold[k]), "golden " + k)
    keep = keep_mask(torch.from_numpy(gold["borderline"]), "golden c1_small")
    g_r, g_a = torch.from_numpy(gold["render"]), torch.from_numpy(gold["alpha"])
    rb, ab = render_bounds({"flip_weight": torch.from_numpy(gold["flip_weight"]), "channel_absmax": torch.from_numpy(gold["channel_absmax"])},
                           g_r, g_a, "RGB+ED")
    assert_close_groups(r, g_r, "golden render", [("colour", 0, 3), ("depth", 3, 4)], bound=rb)
    assert_close(a, g_a, "golden alpha", bound=ab)
    v_r, v_a = cotangents([r.shape, a.shape], int(gold["cot_seed"]))
    v_r, v_a = zero_borderline(v_r, keep), zero_borderline(v_a[..., 0], keep)[..., None]
    ((r * v_r.to(DEV)).sum() + (a * v_a.to(DEV)).sum()).backward()
    for k in gi:
        assert_close(gi[k].grad, torch.from_numpy(gold["grad_" + k]), "golden grad " + k)


# ------------------------------------------------------------------------------------------------
# BASELINE configs at full size (C2: 1 M Gaussians 1920x1080; C3: 3 M 1600x1200; C5: 5 M 1600x1200 per GPU)

FULL = {"c2": (1_000_000, 1920, 1080), "c3": (3_000_000, 1600, 1200), "c5": (5_000_000, 1600, 1200)}


def _oracle_threads():
    # the oracle's gradient scatter uses omp atomics: beyond ~32 threads it only gets slower (bench.py cpu_baseline)
    torch.set_num_threads(min(32, os.cpu_count() or 1))


@pytest.fixture(scope="module", params=sorted(FULL))
def full_scene(request, dns):
    from dn_splatter_amd import synthetic

    N, W, H = FULL[request.param]
    gp = synthetic.make_gauss_params(N, sh_rest_std=0.1, seed=0, device=DEV)
    cam = synthetic.orbit_camera(0, width=W, height=H).to(DEV)
    m = dns.DNSplatterRenderer(gp, fused=True)
    out = m.get_outputs(cam)
    yield gp, cam, m, out
    del gp, m, out
    torch.cuda.empty_cache()


def test_c2_full_frame_fused_pass_matches_oracle(dns, orc):
    """The benchmark step itself: BASELINE C2 — all 1 M Gaussians, the whole 1920 x 1080 frame, fused colour + depth +
    normal pass with the HIP post-ops — against the reference's two-call sequence (dn_model.py:495-578) run on the oracle.
    Every integer output bit for bit, every image and every gradient within 1e-4 (borderline pixels set aside, count
    printed).  The oracle needs about half a minute on 32 threads for its two forward and two backward passes."""
    from dn_splatter_amd import synthetic

    _oracle_threads()
    N, W, H = FULL["c2"]
    gp = synthetic.make_gauss_params(N, sh_rest_std=0.1, seed=0)
    cam = synthetic.orbit_camera(0, width=W, height=H)
    hip, ora, keep = _mirror_pair(dns, orc, gp, cam, dict(fused=True), cot_seed=1, what="C2 full frame")
    assert hip[2].last_info["n_isects"] > 12_000_000      # tight tile boxes; 21 M with gsplat's
    assert float(hip[0]["accumulation"].detach().min()) > 0.999                # every pixel saturates, as in the benchmark
    _check_mirror(hip, ora, keep, "C2 full frame", quat_atol=1e-4)    # isotropic init: d/d(quats) is rounding noise


def test_c3_full_frame_fused_pass_matches_oracle(dns, orc):
    """BASELINE C3 at FULL size (dn-splatter-big: 3 M Gaussians, the whole 1600 x 1200 frame) through the fused pass + HIP
    post-ops, against the reference's two-call sequence on the oracle, as test_c2_full_frame_fused_pass_matches_oracle: all
    images and all gradients at 1e-4 over every non-borderline pixel (floats were only compared on a 384^2
    window at this size).  The oracle's four passes take a couple of minutes on the host."""
    from dn_splatter_amd import synthetic

    _oracle_threads()
    N, W, H = FULL["c3"]
    gp = synthetic.make_gauss_params(N, sh_rest_std=0.1, seed=0)
    cam = synthetic.orbit_camera(0, width=W, height=H)
    hip, ora, keep = _mirror_pair(dns, orc, gp, cam, dict(fused=True), cot_seed=4, what="C3 full frame")
    assert hip[2].last_info["n_isects"] > 20_000_000
    _check_mirror(hip, ora, keep, "C3 full frame", quat_atol=1e-4)


@pytest.mark.skipif(os.environ.get("DNSPLAT_SKIP_C5_FULL", "0") == "1", reason="DNSPLAT_SKIP_C5_FULL=1")
def test_c5_full_frame_fused_pass_matches_oracle(dns, orc):
    """BASELINE C5's per-GPU share at FULL size (5 M Gaussians, 1600 x 1200), as the C3 test above."""
    from dn_splatter_amd import synthetic

    _oracle_threads()
    N, W, H = FULL["c5"]
    gp = synthetic.make_gauss_params(N, sh_rest_std=0.1, seed=0)
    cam = synthetic.orbit_camera(0, width=W, height=H)
    hip, ora, keep = _mirror_pair(dns, orc, gp, cam, dict(fused=True), cot_seed=4, what="C5 full frame")
    _check_mirror(hip, ora, keep, "C5 full frame", quat_atol=1e-4)


def test_c2_full_frame_rasterization_drop_in_matches_oracle(dns, orc):
    """The whole C2 frame (1 M Gaussians, 1920 x 1080) through the drop-in ``rasterization()`` with the inputs dn-splatter hands
    it (dn_model.py:495-516: RGB+ED, absgrad): gsplat's own tile boxes, so EVERY integer output — radii, tiles_per_gauss,
    flatten_ids, isect_offsets, isect_ids of all 21 M intersections — is compared bit for bit with no carve-out, and the
    render / alpha / all gradients at 1e-4 over the whole frame."""
    N, W, H = FULL["c2"]
    _oracle_threads()
    inp, viewmat, K, _ = gsplat_inputs(N, W, H, focal=1200.0, seed=0)
    o, g = _call_both(dns, orc, inp, viewmat, K, W, H, sh_degree=3, render_mode="RGB+ED", absgrad=True)
    assert g[2]["n_isects"] > 20_000_000
    _check_forward(o, g, what="C2 full frame, rasterization()")
    _check_backward(o, g, quat_atol=1e-4, what="C2 full frame, rasterization()")


@pytest.mark.parametrize("workload,crop", [("c2", 384), ("c3", 384)])
def test_full_scene_centre_crop_matches_oracle(dns, orc, workload, crop):
    """The BASELINE C2 / C3 scenes (1 M Gaussians at 1080p, 3 M at 1600 x 1200; fx = fy = 1200), rendered through a
    384 x 384 window at the image centre so that the CPU oracle finishes in seconds: same Gaussians, same depth
    complexity (thousands of entries per tile list, every pixel saturating) as the benchmark.  gsplat.rasterization as
    dn-splatter calls it (RGB+ED, absgrad); both sides get the same activated inputs, so every integer output is compared
    bit for bit."""
    N, W, H = FULL[workload]
    C = crop
    _oracle_threads()
    inp, viewmat, K, _ = gsplat_inputs(N, W, H, focal=1200.0, seed=0)
    K = K.clone()
    K[..., 0, 2] -= (W - C) // 2
    K[..., 1, 2] -= (H - C) // 2
    o, g = _call_both(dns, orc, inp, viewmat, K, C, C, sh_degree=3, render_mode="RGB+ED", absgrad=True)
    assert g[2]["n_isects"] > 1_000_000
    assert float(g[1].detach().min()) > 0.999                            # saturating pixels, like the full frame
    _check_forward(o, g, what=workload + " centre crop")
    _check_backward(o, g, quat_atol=1e-4, what=workload + " centre crop")


@pytest.mark.parametrize("workload", ["c3", "c5"])
def test_big_scene_centre_crop_fused_pass_matches_oracle(dns, orc, workload):
    """BASELINE C3 / C5's per-GPU share (dn-splatter-big: 3 M / 5 M Gaussians, 1600 x 1200) through the fused pass + HIP
    post-ops — the path the C3 / C5 bench lines time — on a 384 x 384 centre window, against the reference sequence on the
    oracle (twice: fp32, and fp64 for the rounding envelope of the gradient comparisons)."""
    from dn_splatter_amd import synthetic
    from dn_splatter_amd.model import Camera

    _oracle_threads()
    N, W, H = FULL[workload]
    C = 384
    gp = synthetic.make_gauss_params(N, sh_rest_std=0.1, seed=0)
    cam = synthetic.orbit_camera(0, width=W, height=H)
    ccam = Camera(cam.camera_to_worlds, cam.fx, cam.fy, cam.cx - (W - C) // 2, cam.cy - (H - C) // 2, C, C)
    what = workload.upper() + " centre crop (fused)"
    hip, ora, keep = _mirror_pair(dns, orc, gp, ccam, dict(fused=True), cot_seed=3, what=what)
    assert hip[2].last_info["n_isects"] > 1_000_000
    _check_mirror(hip, ora, keep, what, quat_atol=1e-4)


@pytest.mark.parametrize("workload", ["c3", "c5"])
def test_full_size_projection_and_binning_match_oracle(dns, orc, workload):
    """C3 / C5 at FULL size, integer half of the path: projection (radii), tile counts, the sorted tile lists and the tile
    offsets of all 3 M / 5 M Gaussians over the whole 1600 x 1200 frame, bit for bit against the oracle's projection +
    isect_tiles + 64-bit stable sort (the single-threaded sort of ~10^8 pairs takes the oracle some ten seconds)."""
    N, W, H = FULL[workload]
    _oracle_threads()
    inp, viewmat, K, _ = gsplat_inputs(N, W, H, focal=1200.0, seed=0)
    with torch.no_grad():
        gi = {k: v.to(DEV) for k, v in inp.items()}
        _r, _a, info = dns.rasterization(**gi, viewmats=viewmat.to(DEV), Ks=K.to(DEV), width=W, height=H, packed=False,
                                         sh_degree=3, render_mode="RGB+ED")
        radii, means2d, depths, conics, _c, tiles = orc.project_fwd(inp["means"], inp["quats"], inp["scales"], viewmat[0], K[0], W, H)
        tw, th = math.ceil(W / 16), math.ceil(H / 16)
        _t, isect_ids, flatten_ids = orc.isect_tiles(means2d, radii, depths, 16, tw, th)
        offsets = orc.isect_offset_encode(isect_ids, tw, th)
    from dn_splatter_amd import _ops
    assert _ops.binning_status(info["_binning"], N) == 0, "a look-back wait of the tile sort timed out"
    assert_equal_int(info["radii"][0], radii, workload + " radii")
    assert_equal_int(info["tiles_per_gauss"][0], tiles, workload + " tiles_per_gauss")
    assert info["n_isects"] == flatten_ids.shape[0] > 30_000_000
    assert_equal_int(info["flatten_ids"], flatten_ids, workload + " flatten_ids")
    assert_equal_int(info["isect_offsets"][0], offsets, workload + " isect_offsets")
    for k, ref in (("means2d", means2d), ("depths", depths), ("conics", conics)):
        assert_close(info[k][0], ref, workload + " " + k)


def test_full_size_binning_properties(dns, full_scene):
    from dn_splatter_amd import _ops

    gp, cam, m, out = full_scene
    info = m.last_info
    assert _ops.binning_status(info["_binning"], info["radii"].numel()) == 0, "a look-back wait of the tile sort timed out"
    n = info["n_isects"]
    tiles = info["tiles_bin"][0].long()          # the count the binning walked (== tiles_per_gauss unless tight tile boxes)
    assert int(tiles.sum()) == n, "sum(tiles_bin) != n_isects"
    if info.get("tight_tiles"):     # a visible Gaussian may reach alpha >= 1/255 at no pixel centre at all
        assert int(((tiles > 0) & ~(info["radii"][0] > 0)).sum()) == 0
    else:
        assert int(((info["radii"][0] > 0) != (tiles > 0)).sum()) == 0
    offs = info["isect_offsets"].reshape(-1).long()
    assert int(offs[0]) == 0 and bool((offs[1:] >= offs[:-1]).all()) and int(offs[-1]) <= n
    # every tile list is depth-sorted, ties broken by Gaussian index (stable sort, Appendix A.3)
    fid = info["flatten_ids"].long()
    d = info["depths"][0][fid]
    tile_of = torch.searchsorted(offs, torch.arange(n, device=DEV), right=True) - 1
    same = tile_of[1:] == tile_of[:-1]
    ok = (d[1:] > d[:-1]) | ((d[1:] == d[:-1]) & (fid[1:] > fid[:-1]))
    assert bool((ok | ~same).all()), "a tile list is not (depth, index)-sorted"
    # each Gaussian appears exactly tiles_per_gauss times
    cnt = torch.bincount(fid, minlength=tiles.shape[0])
    assert torch.equal(cnt, tiles)
    # and only in tiles of its bounding box (A.3)
    xy = info["means2d"][0][fid]
    r = info["radii"][0][fid].float()
    tw = info["tile_width"]
    tx, ty = (tile_of % tw).float(), (tile_of // tw).float()
    inside = (tx >= torch.floor((xy[:, 0] - r) / 16)) & (tx < torch.ceil((xy[:, 0] + r) / 16)) & \
             (ty >= torch.floor((xy[:, 1] - r) / 16)) & (ty < torch.ceil((xy[:, 1] + r) / 16))
    assert bool(inside.all())


def test_full_size_image_properties_and_linearity(dns, full_scene):
    gp, cam, m, out = full_scene
    acc = out["accumulation"]
    assert float(acc.detach().min()) >= 0.0 and float(acc.detach().max()) < 1.0
    for k in ("rgb", "depth", "normal"):
        assert bool(torch.isfinite(out[k]).all()), k
    assert float(out["rgb"].detach().min()) >= 0.0 and float(out["rgb"].detach().max()) <= 1.0
    # determinism of the forward (no atomics on the forward data path): bit-identical re-render
    out2 = m.get_outputs(cam)
    for k in ("rgb", "depth", "normal", "accumulation"):
        assert torch.equal(out[k], out2[k]), k
    # backward is linear in the cotangent: grad(v1 + v2) == grad(v1) + grad(v2) (atomics => 1e-4 tolerance)
    keys = ("rgb", "depth", "normal", "accumulation")
    gen = torch.Generator(device=DEV).manual_seed(7)
    v1 = {k: torch.rand(out2[k].shape, device=DEV, generator=gen) * 2 - 1 for k in keys}
    v2 = {k: torch.rand(out2[k].shape, device=DEV, generator=gen) * 2 - 1 for k in keys}
    names = ("means", "scales", "quats", "features_dc", "features_rest", "opacities")
    params = [gp[k] for k in names]

    def grads(vs):
        loss = sum((out2[k] * vs[k]).sum() for k in keys)
        return torch.autograd.grad(loss, params, retain_graph=True)

    g1, g2 = grads(v1), grads(v2)
    g12 = grads({k: v1[k] + v2[k] for k in keys})
    for name, a, b, c in zip(names, g1, g2, g12):
        assert bool(torch.isfinite(c).all()), name
        assert_close(a + b, c, "linearity of grad " + name)
    # Gaussians that hit no tile get exactly zero gradient
    hidden = m.radii <= 0
    if bool(hidden.any()):
        for name, c in zip(names, g12):
            assert float(c[hidden].abs().max()) == 0.0, name


@pytest.mark.usefixtures("hip_deterministic")
def test_tight_tile_boxes_change_the_lists_not_the_images(dns):
    """dnsplat_camera.tight_tiles (the fused path's default): fewer (tile, Gaussian) pairs, the same composited numbers —
    forward images bit for bit (the pairs left out were skipped at every pixel anyway, the others are blended in the same
    order), gradients up to the order of the atomic sums."""
    from dn_splatter_amd import _ops, synthetic

    N, W, H = 20_000, 320, 240
    gp = synthetic.make_gauss_params(N, sh_rest_std=0.1, seed=5, device=DEV)
    g_ = torch.Generator().manual_seed(31)
    gp["scales"] = (gp["scales"].detach() + (torch.randn(N, 3, generator=g_) * 0.6).to(DEV)).requires_grad_(True)
    gp["opacities"] = (gp["opacities"].detach() + (torch.randn(N, 1, generator=g_) * 2.0).to(DEV)).requires_grad_(True)
    cam = synthetic.orbit_camera(2, width=W, height=H, focal=200.0).to(DEV)
    keys = ("rgb", "depth", "normal", "accumulation")
    gen = torch.Generator(device=DEV).manual_seed(3)
    cot = None
    res = {}
    old = _ops.TIGHT_TILES
    try:
        for tight in (False, True):
            _ops.TIGHT_TILES = tight
            for v in gp.values():
                v.grad = None
            r = dns.DNSplatterRenderer(gp, fused=True)
            out = r.get_outputs(cam)
            assert bool(r.last_info.get("tight_tiles")) == tight
            if cot is None:
                cot = [torch.rand(out[k].shape, device=DEV, generator=gen) * 2 - 1 for k in keys]
            torch.autograd.backward([out[k] for k in keys], cot)
            res[tight] = ({k: out[k].detach().clone() for k in keys}, {k: v.grad.clone() for k, v in gp.items() if v.grad is not None},
                          int(r.last_info["n_isects"]), r.last_info["tiles_bin"].clone(), r.radii.clone())
            if tight:   # the reported count stays gsplat's
                assert torch.equal(r.last_info["tiles_per_gauss"], res[False][3])
    finally:
        _ops.TIGHT_TILES = old
    (o0, g0, n0, t0, r0), (o1, g1, n1, t1, r1) = res[False], res[True]
    print(f"[parity] tight tile boxes: {n1} of {n0} intersections ({100.0 * n1 / n0:.1f} %)")
    # the kernel's boxes against the CPU restatement of the rule (oracle.tight_tile_boxes) on the kernel's own projection: equal
    # up to the last place of logf / sqrtf / sigmoid on the two sides (a handful of Gaussians may round to the neighbouring tile)
    from oracle import oracle as orc_
    info = r.last_info
    bx = orc_.tight_tile_boxes(info["means2d"][0].detach().cpu(), info["conics"][0].detach().cpu(),
                               torch.sigmoid(gp["opacities"].detach().cpu()).reshape(-1), r1.cpu(), 16,
                               info["tile_width"], info["tile_height"])
    cpu_tiles = (bx[2] - bx[0]) * (bx[3] - bx[1])
    differ = int((cpu_tiles != t1.reshape(-1).cpu().long()).sum())
    print(f"[parity] tight tile counts: {differ} of {N} Gaussians differ from the CPU restatement of the rule")
    assert differ <= max(2, N // 2000)
    assert n1 < 0.9 * n0 and bool((t1 <= t0).all()) and torch.equal(r0, r1)
    for k in keys:
        assert torch.equal(o0[k], o1[k]), k
    for k in g0:
        # two runs of the same atomic sums in different orders (the shorter lists shift which wave adds first): a few 1e-6
        assert_close(g1[k], g0[k], "tight vs gsplat boxes: grad " + k, tol=1e-5)


@pytest.mark.usefixtures("hip_deterministic")
def test_extra_terms_on_means2d_add_to_the_compositing_gradient(dns, orc):
    """info["means2d"] is an autograd tensor like any other: a loss term hung on it directly must ADD to the screen-space
    gradient the compositing backward leaves in the records (which reaches `.grad` without autograd's clone, _ops.py)."""
    inp, viewmat, K, _ = gsplat_inputs(4000, 160, 128, focal=110.0, seed=9, anisotropic=True)
    ci, gi = to_leaf(inp, "cpu"), to_leaf(inp, DEV)
    kw = dict(width=160, height=128, packed=False, sh_degree=3, render_mode="RGB+ED", absgrad=True)
    r_o, a_o, info_o = orc.rasterization(**ci, viewmats=viewmat, Ks=K, **kw)
    r_g, a_g, info_g = dns.rasterization(**gi, viewmats=viewmat.to(DEV), Ks=K.to(DEV), **kw)
    keep = keep_mask(info_o["borderline"], "extra means2d term")
    v_r, = cotangents([r_o.shape], 5)
    v_r = zero_borderline(v_r, keep)
    w = cotangents([info_o["means2d"].shape], 6)[0] * 1e-3
    info_o["means2d"].retain_grad(); info_g["means2d"].retain_grad()
    ((r_o * v_r).sum() + (info_o["means2d"] * w).sum()).backward()
    ((r_g * v_r.to(DEV)).sum() + (info_g["means2d"] * w.to(DEV)).sum()).backward()
    assert_close(info_g["means2d"].grad, info_o["means2d"].grad, "means2d.grad with an extra term")
    for k in ("means", "scales", "quats"):
        assert_close(gi[k].grad, ci[k].grad, "grad " + k + " with an extra term on means2d")
    # and without retain_grad() no .grad appears
    gj = to_leaf(inp, DEV)
    r2, _, info2 = dns.rasterization(**gj, viewmats=viewmat.to(DEV), Ks=K.to(DEV), **kw)
    r2.sum().backward()
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert info2["means2d"].grad is None
    # ... but the tensor still takes part in autograd like any other: torch.autograd.grad on it and a hook registered on it see
    # the rasterizer's term (+ the extra one), and the parameters receive the same gradients as with retain_grad()
    gk = to_leaf(inp, DEV)
    r3, _, info3 = dns.rasterization(**gk, viewmats=viewmat.to(DEV), Ks=K.to(DEV), **kw)
    seen = {}
    info3["means2d"].register_hook(lambda g: seen.__setitem__("g", g.detach().clone()))
    loss3 = (r3 * v_r.to(DEV)).sum() + (info3["means2d"] * w.to(DEV)).sum()
    g_m2d, = torch.autograd.grad(loss3, info3["means2d"], retain_graph=True)
    assert_close(g_m2d, info_o["means2d"].grad, "autograd.grad(loss, means2d) without retain_grad")
    loss3.backward()
    assert_close(seen["g"], info_o["means2d"].grad, "hook on means2d without retain_grad")
    for k in ("means", "scales", "quats"):
        assert_close(gk[k].grad, ci[k].grad, "grad " + k + " with an extra term on means2d, no retain_grad")


def test_second_backward_through_the_same_frame_gets_fresh_gradient_records(dns):
    """The fused forward clears the gradient records its backward accumulates into (dnsplat_raster_args.zero_fill); a second
    backward through the same graph must not add onto the first one's sums."""
    from dn_splatter_amd import synthetic

    gp = synthetic.make_gauss_params(5000, sh_rest_std=0.1, seed=2, device=DEV)
    cam = synthetic.orbit_camera(1, width=160, height=112, focal=100.0).to(DEV)
    r = dns.DNSplatterRenderer(gp, fused=True)
    out = r.get_outputs(cam)
    loss = out["rgb"].sum() + out["depth"].mean() + out["normal"].sum()
    loss.backward(retain_graph=True)
    g1 = {k: v.grad.clone() for k, v in gp.items() if v.grad is not None}
    for v in gp.values():
        v.grad = None
    loss.backward()
    for k in g1:
        assert_close(gp[k].grad, g1[k], "second backward: grad " + k, tol=1e-5)


def test_deferred_bin_policy_same_frames_and_a_loud_overflow(dns):
    """'deferred' (bench.py's policy for eager launches): the host never waits for a frame's intersection count, it verifies it
    once it has arrived.  Same images, gradients and lists as 'sync'; a capacity guess that turns out too small RAISES (at the
    latest when the count is asked for) instead of leaving truncated lists behind, and enlarges the guess."""
    from dn_splatter_amd import _lib, _ops, synthetic

    gp = synthetic.make_gauss_params(20_000, sh_rest_std=0.1, seed=4, device=DEV)
    cam = synthetic.orbit_camera(2, width=320, height=240, focal=200.0).to(DEV)
    keys = ("rgb", "depth", "normal", "accumulation")
    gen = torch.Generator(device=DEV).manual_seed(3)
    cot = None

    def frame(policy):
        nonlocal cot
        dns.set_bin_policy(policy)
        for v in gp.values():
            v.grad = None
        r = dns.DNSplatterRenderer(gp, fused=True)
        out = r.get_outputs(cam)
        if cot is None:
            cot = [torch.rand(out[k].shape, device=DEV, generator=gen) * 2 - 1 for k in keys]
        torch.autograd.backward([out[k] for k in keys], cot)
        return ({k: out[k].detach().clone() for k in keys}, {k: v.grad.clone() for k, v in gp.items() if v.grad is not None}, r)

    try:
        o0, g0, r0 = frame("sync")
        n0 = int(r0.last_info["n_isects"])
        o1, g1, r1 = frame("deferred")
        assert r1.last_info["_binning"].pending is not None or r1.last_info["_binning"]._n is not None
        assert int(r1.last_info["n_isects"]) == n0                       # reading it is what waits
        for k in keys:
            assert torch.equal(o0[k], o1[k]), k
        for k in g0:
            assert_close(g1[k], g0[k], "deferred vs sync: grad " + k, tol=1e-5)
        assert_equal_int(r1.last_info["flatten_ids"], r0.last_info["flatten_ids"], "flatten_ids deferred vs sync")
        assert_equal_int(r1.last_info["isect_offsets"], r0.last_info["isect_offsets"], "isect_offsets deferred vs sync")
        key = (torch.device(DEV), 20_000, 320, 240)
        _ops.BUFFERS.capacity_hint[key] = 1000                           # far below the real count
        with pytest.raises(_lib.DnsplatError, match="exceed the capacity"):
            _, _, r2 = frame("deferred")
            _ops.verify_pending_counts(torch.device(DEV), block=True)
        assert _ops.BUFFERS.capacity_hint[key] >= n0                     # enlarged: the repeated step goes through
        o3, g3, r3 = frame("deferred")
        _ops.verify_pending_counts(torch.device(DEV), block=True)
        for k in keys:
            assert torch.equal(o0[k], o3[k]), k
    finally:
        dns.set_bin_policy("sync")


def test_graphed_step_replays_equal_eager_frames(dns):
    """graph.GraphedStep: get_outputs + backward captured into a HIP graph ('static' bin policy, no host code between the
    kernels) and replayed — same outputs and gradients as eager frames, also after the camera pose was changed in place, and
    check() notices a frame that outgrew the captured buffers."""
    from dn_splatter_amd import _lib, _ops, dp, synthetic
    from dn_splatter_amd.graph import GraphedStep

    gp = synthetic.make_gauss_params(20_000, sh_rest_std=0.1, seed=6, device=DEV)
    cams = [synthetic.orbit_camera(i, width=320, height=240, focal=200.0).to(DEV) for i in (1, 3)]
    cam = synthetic.orbit_camera(1, width=320, height=240, focal=200.0).to(DEV)      # its pose tensor is what the graph reads
    keys = ("rgb", "depth", "normal", "accumulation")
    gen = torch.Generator(device=DEV).manual_seed(8)
    shapes = {"rgb": (240, 320, 3), "depth": (240, 320, 1), "normal": (240, 320, 3), "accumulation": (240, 320, 1)}
    cot = [torch.rand(shapes[k], device=DEV, generator=gen) * 2 - 1 for k in keys]
    r = dns.DNSplatterRenderer(gp, fused=True)

    def compute():
        out = r.get_outputs(cam)
        torch.autograd.backward([out[k] for k in keys], cot)
        return out

    def eager(c):
        dns.set_bin_policy("sync")
        for v in gp.values():
            v.grad = None
        r.forget()
        out = r.get_outputs(c)
        torch.autograd.backward([out[k] for k in keys], cot)
        res = ({k: out[k].detach().clone() for k in keys}, {k: v.grad.clone() for k, v in gp.items() if v.grad is not None})
        r.forget()
        return res

    try:
        ref = [eager(c) for c in cams]
        for v in gp.values():
            v.grad = None
        step = GraphedStep(compute, params={k: gp[k] for k in dp.GRAD_KEYS})
        for i in (0, 1, 0):
            cam.camera_to_worlds.copy_(cams[i].camera_to_worlds)          # a new pose, in place
            out = step()
            torch.cuda.synchronize()
            for k in keys:
                assert torch.equal(out[k], ref[i][0][k]), (i, k)
            for k, g in ref[i][1].items():
                assert_close(gp[k].grad, g, f"graph replay, camera {i}: grad {k}", tol=1e-5)
        step.check()
        # a frame that needs more intersections than the captured buffers hold is noticed (here: by shrinking what the check
        # believes the capacity to be)
        for kk in list(_ops.BUFFERS.static_cap):
            if kk[3:] == (20_000, 320, 240):
                _ops.BUFFERS.static_cap[kk] = 10
        with pytest.raises(_lib.DnsplatError, match="more than the captured buffers hold"):
            step.check()
        with pytest.raises(_lib.DnsplatError, match="more than the captured buffers hold"):
            step.check()                      # sticky: the step stays invalid
        step.close()
    finally:
        dns.set_bin_policy("sync")
        for kk in list(_ops.BUFFERS.static_cap):
            if kk[3:] == (20_000, 320, 240):
                del _ops.BUFFERS.static_cap[kk]


@pytest.mark.filterwarnings("ignore:The AccumulateGrad node's stream does not match")       # the re-capture runs on a new side stream by design
def test_graphed_step_overflow_is_reported_and_a_recapture_fits(dns):
    """a pose copied into a captured step may need more intersections than the buffers captured at the
    warm-up pose hold.  check() must say so, and a NEW capture after the error must get buffers that fit (the capacity guess is
    raised from the device's running maximum) instead of overflowing again in its warm-up."""
    from dn_splatter_amd import _lib, _ops, dp, synthetic
    from dn_splatter_amd.graph import GraphedStep

    N, W, H = 30_000, 256, 192
    gp = synthetic.make_gauss_params(N, sh_rest_std=0.1, seed=11, device=DEV)
    far = synthetic.orbit_camera(0, width=W, height=H, focal=60.0).to(DEV)         # wide view: small splats, few pairs
    near = synthetic.orbit_camera(0, width=W, height=H, focal=60.0).to(DEV)
    near.camera_to_worlds[..., :3, 3] *= 0.35                                       # inside the cloud: several times the pairs
    cam = synthetic.orbit_camera(0, width=W, height=H, focal=60.0).to(DEV)
    keys = ("rgb", "depth", "normal", "accumulation")
    gen = torch.Generator(device=DEV).manual_seed(3)
    cot = [torch.rand((H, W, c), device=DEV, generator=gen) * 2 - 1 for c in (3, 1, 3, 1)]
    r = dns.DNSplatterRenderer(gp, fused=True)

    def compute():
        out = r.get_outputs(cam)
        torch.autograd.backward([out[k] for k in keys], cot)
        return out

    def count(c):
        dns.set_bin_policy("sync")
        r.forget()
        with torch.no_grad():
            r.get_outputs(c)
        n = int(r.last_info["n_isects"])
        r.forget()
        return n

    hkey = (torch.device(DEV), N, W, H)
    try:
        n_far, n_near = count(far), count(near)
        assert n_near > 1.6 * n_far, (n_far, n_near)
        _ops.BUFFERS.capacity_hint.pop(hkey, None)                                   # the guesses of the two counting frames
        cam.camera_to_worlds.copy_(far.camera_to_worlds)
        for v in gp.values():
            v.grad = None
        step = GraphedStep(compute, params={k: gp[k] for k in dp.GRAD_KEYS})
        step(); step.check()
        cam.camera_to_worlds.copy_(near.camera_to_worlds)
        step()
        with pytest.raises(_lib.DnsplatError, match="more than the captured buffers hold"):
            step.check()
        assert _ops.BUFFERS.capacity_hint[hkey] >= n_near
        step2 = step.recapture()                                                     # warm-up at the near pose, enlarged buffers
        assert step.graphs == [] and step2.graphs
        out = step2()
        step2.check()
        torch.cuda.synchronize()
        assert int(r.last_info["_binning"]._n_dev.item()) == n_near
        step2.close()
    finally:
        dns.set_bin_policy("sync")
        _ops.BUFFERS.capacity_hint.pop(hkey, None)


# ------------------------------------------------------------------------------------------------
# The backward of the BORDERLINE pixels.  Everywhere else in this file their cotangents are zero on both
# sides; here they are the ONLY pixels with a cotangent, and the kernels' gradients are held to what the rule set admits there.


def _hip_raster_level_grads(s, v_r, v_a, deterministic=False):
    """rasterize_to_pixels of the raster-level scene ``s`` (tests/_scenes.raster_level_scene) through the product: records packed by
    dnsplat_pack_splats, binned by dnsplat_bin_*, composited forward and backward by dnsplat_raster_fwd / _bwd.  Returns the five
    raster-level gradients as CPU tensors."""
    from dn_splatter_amd import _ops

    prev = _ops.DETERMINISTIC["on"]
    _ops.set_deterministic(deterministic)
    try:
        leaf = {k: s[k].to(DEV).clone().requires_grad_(True) for k in ("xys", "conics", "colors", "opacities")}
        splats = _ops._PackFn.apply(leaf["xys"], leaf["conics"], leaf["opacities"], leaf["colors"])
        # the tensor the caller reads .grad / .absgrad on (gsplat's info["means2d"]); the kernels take xy from the records
        m2d = leaf["xys"].detach()[None].clone().requires_grad_(True)
        render, alphas = _ops.rasterize(m2d, splats, s["depths"].to(DEV), s["radii"].to(DEV), s["tiles"].to(DEV),
                                        background=s["background"].to(DEV), width=s["W"], height=s["H"], tile_size=16, D=s["D"],
                                        absgrad=True)
        torch.autograd.backward([render, alphas], [v_r.to(DEV)[None], v_a.to(DEV)[None]])
        torch.cuda.synchronize()
        # the screen-space gradient arrives through means2d; what is left in the records' xy columns (nothing) goes back to xys
        return {"means2d": (m2d.grad[0] + leaf["xys"].grad).cpu(), "absgrad": m2d.absgrad.reshape(-1, 2).cpu(), "conics": leaf["conics"].grad.cpu(),
                "colors": leaf["colors"].grad.cpu(), "opacities": leaf["opacities"].grad.reshape(-1).cpu()}, render[0].detach().cpu()
    finally:
        _ops.set_deterministic(prev)


@pytest.mark.parametrize("seed,D,aniso", [(1, 4, True), (2, 7, True), (0, 4, False)])
def test_backward_of_borderline_pixels_lies_in_the_hull_of_admissible_decisions(dns, orc, seed, D, aniso):
    """C1 sizes, cotangents on the oracle-flagged borderline pixels only.  A decision taken the other way changes WHICH (pixel, splat)
    pairs exist, so there is no tolerance around one reference gradient; what the rule set admits is the set of gradients under
    every combination of outcomes of the flagged decisions (orc_rasterize_bwd_hull evaluates them all).  Checked: (1) every entry of
    the kernels' five raster-level gradients lies between the smallest and the largest admissible value, up to 1e-4 of the tensor's
    scale; (2) for single pixels, the kernels' gradient equals ONE admissible combination in all entries at once."""
    from _scenes import assert_in_hull, raster_level_scene

    s = raster_level_scene(orc, seed, D=D, anisotropic=aniso, view=seed)
    m = s["borderline"]
    n_b = int(m.sum())
    print(f"[parity] {n_b} of {m.numel()} pixels borderline; they alone carry cotangents")
    assert 0 < n_b <= 0.01 * m.numel()
    g = s["gen"]
    v_r = (torch.rand(s["H"], s["W"], D, generator=g) * 2 - 1) * m[..., None]
    v_a = (torch.rand(s["H"], s["W"], generator=g) * 2 - 1) * m
    args = (s["xys"], s["conics"], s["colors"], s["opacities"], s["background"], s["W"], s["H"], 16, s["offsets"], s["flatten_ids"])
    lo, hi, st = orc.rasterize_bwd_hull(*args, m, v_r, v_a)
    assert st["pixels"] == n_b and st["pixels_over_cap"] == 0 and st["pixels_incomplete"] == 0, st
    hip, _ = _hip_raster_level_grads(s, v_r, v_a)
    assert_in_hull(hip, lo, hi, f"borderline-only backward (seed {seed}, {D} channels)")
    hip_d, _ = _hip_raster_level_grads(s, v_r, v_a, deterministic=True)
    assert_in_hull(hip_d, lo, hi, f"borderline-only backward, deterministic mode (seed {seed}, {D} channels)")

    # (2) one pixel at a time
    ys, xs = torch.nonzero(m, as_tuple=True)
    other = 0
    for y, x in list(zip(ys.tolist(), xs.tolist()))[:: max(1, n_b // 16)][:16]:
        one_px = torch.zeros_like(m)
        one_px[y, x] = True
        vr1, va1 = v_r * one_px[..., None], v_a * one_px
        h1, _ = _hip_raster_level_grads(s, vr1, va1)
        _l, _h, st1 = orc.rasterize_bwd_hull(*args, one_px, vr1, va1)
        k = st1["max_flags_seen"]
        assert 1 <= k <= 10
        t_fin = max(1.0 - float(s["alphas"][y, x]), 1e-30)
        dlt = min(1.0, 2.0 * 2.0 ** -24 / t_fin)        # T_final = 1 - alpha image, 2 ulp of the image value (orc_rasterize_bwd_hull)
        best, best_c = None, None
        for c in range(1 << k):
            one, _same, _ = orc.rasterize_bwd_hull(*args, one_px, vr1, va1, combo=c)
            w = max(float(((h1[n].double().reshape(one[n].shape) - one[n]).abs()
                           / (REL_TOL * max(float(one[n].abs().max()), float(h1[n].abs().max()), 1e-30) + dlt * one[n].abs())).max()) for n in one)
            if best is None or w < best:
                best, best_c = w, c
        assert best <= 1.0, f"pixel ({y}, {x}): the kernels' gradient matches none of the {1 << k} admissible combinations (closest {best:.2f})"
        base = orc.rasterize_bwd(*args, s["alphas"], s["last_ids"], vr1, va1, absgrad=True)
        nat, _s2, _ = orc.rasterize_bwd_hull(*args, one_px, vr1, va1, combo=best_c)
        if any(float((b.double().reshape(nat[n].shape) - nat[n]).abs().max()) > REL_TOL * max(float(nat[n].abs().max()), 1e-30)
               for n, b in zip(("means2d", "absgrad", "conics", "colors", "opacities"), base)):
            other += 1
    print(f"[parity] single-pixel check: {other} of the sampled borderline pixels were decided the other way by the kernels — and match that combination")


def _fused_hull_case(dns, orc, monkeypatch, gp, cam, what, max_share=0.01):
    """The same statement as above for the BENCHMARK instantiation — the fused 7-channel pass with the dn epilogue / prologue, keep
    masks and tight tile boxes (raster_bwd_kernel<7,4,DN,.,MASKS>): image cotangents on the borderline pixels only; the kernels'
    gradient records (v_xy | v_conic | v_opacity | v_channels | |v_xy|) against the hull the oracle builds for the reference's two
    passes — rgb + depth with the screen-space gradient, normals without it (dn_model.py:562 feeds xys.detach()) — from the
    composite-level cotangents autograd hands its own backward calls.  The two passes share every decision; their hulls are added
    (a superset of the admissible set: a necessary condition)."""
    from _scenes import assert_in_hull
    from dn_splatter_amd import _ops

    N, W, H = gp["means"].shape[0], int(cam.width), int(cam.height)

    # ---- the reference sequence on the oracle, its raster-level calls recorded
    fwd_calls, bwd_calls = [], []
    real_fwd, real_bwd = orc.rasterize_fwd, orc.rasterize_bwd

    def spy_fwd(*a, **kw):
        fwd_calls.append(a[:10])
        return real_fwd(*a, **kw)

    def spy_bwd(*a, **kw):
        bwd_calls.append((a[12], a[13]))                     # v_render, v_alphas
        return real_bwd(*a, **kw)

    monkeypatch.setattr(orc, "rasterize_fwd", spy_fwd)
    monkeypatch.setattr(orc, "rasterize_bwd", spy_bwd)
    p_o = {k: v.detach().clone().requires_grad_(k != "normals") for k, v in gp.items()}
    m_o = dns.DNSplatterRenderer(p_o, fused=False, rasterization_fn=orc.rasterization, rasterize_gaussians_fn=orc.rasterize_gaussians)
    out_o = m_o.get_outputs(cam)
    border = m_o.last_info["borderline"].reshape(H, W) | orc.last_borderline
    pre = out_o["rgb"].detach()
    assert len(fwd_calls) == 2 and fwd_calls[0][2].shape[1] == 4 and fwd_calls[1][2].shape[1] == 3
    n_b = int(border.sum())
    print(f"[parity] {what}: {n_b} of {border.numel()} pixels borderline; they alone carry cotangents")
    assert 0 < n_b <= max_share * border.numel()
    # pixels whose rgb sits on a clamp(0, 1) corner have an undefined gate for the rgb cotangent: none of them may be selected
    border &= ~(((pre - 0.0).abs() < 4e-6) | ((pre - 1.0).abs() < 4e-6)).any(-1)
    gen = torch.Generator().manual_seed(5)
    cot = {k: (torch.rand(out_o[k].shape, generator=gen) * 2 - 1) * border[..., None] for k in OUT_KEYS}
    torch.autograd.backward([out_o[k] for k in OUT_KEYS], [cot[k] for k in OUT_KEYS])
    assert len(bwd_calls) == 2
    monkeypatch.undo()
    # autograd runs the two backward calls in reverse order of the forward calls
    (vr2, va2), (vr1, va1) = bwd_calls
    assert vr1.shape[-1] == 4 and vr2.shape[-1] == 3
    lo1, hi1, st1 = orc.rasterize_bwd_hull(*fwd_calls[0], border, vr1, va1)
    lo2, hi2, st2 = orc.rasterize_bwd_hull(*fwd_calls[1], border, vr2, va2)
    for st in (st1, st2):
        assert st["pixels_over_cap"] == 0 and st["pixels_incomplete"] == 0, st
    lo = {"means2d": lo1["means2d"], "absgrad": lo1["absgrad"], "conics": lo1["conics"] + lo2["conics"],
          "opacities": lo1["opacities"] + lo2["opacities"], "colors": torch.cat([lo1["colors"], lo2["colors"]], 1)}
    hi = {"means2d": hi1["means2d"], "absgrad": hi1["absgrad"], "conics": hi1["conics"] + hi2["conics"],
          "opacities": hi1["opacities"] + hi2["opacities"], "colors": torch.cat([hi1["colors"], hi2["colors"]], 1)}

    # ---- the product: the fused path's own launch sequence, with the gradient records kept
    p_g = {k: v.detach().to(DEV).clone().requires_grad_(k != "normals") for k, v in gp.items()}
    bg = torch.tensor(dns.RendererConfig().background_color, device=DEV)
    camg = cam.to(DEV)
    viewmat, K, nf, flag = _ops.camera_prepare(camg.camera_to_worlds[0], camg.fx, camg.fy, camg.cx, camg.cy, with_flag=True, n_depth_max=1)
    cfg = _ops.ProjCfg(width=W, height=H, scales_are_log=True, opacities_are_logit=True, sh_degree=3, with_depth=True, with_normals=True,
                       want_normals_world=True, tight_tiles=_ops.TIGHT_TILES)
    pr = _ops.project(p_g["means"], p_g["quats"], p_g["scales"], p_g["opacities"].reshape(N), sh0=p_g["features_dc"], shN=p_g["features_rest"],
                      viewmat=viewmat[None], K=K[None], normal_frame=nf[None], cfg=cfg, saturation_flag=flag)
    pr["splats"].retain_grad()
    pr["means2d"].retain_grad()
    rgb, depth, normal, acc, _sn = _ops.rasterize_dn(pr["means2d"], pr["splats"], pr["depths"], pr["radii"], pr["tiles_bin"], background_rgb=bg,
                                                    width=W, height=H, intrinsics=[(camg.fx, camg.fy, camg.cx, camg.cy)], absgrad=True,
                                                    holder={"saturation_flag": flag}, tight=pr["tight_tiles"], tile_boxes=pr["tile_boxes"])
    outs = {"rgb": rgb[0], "depth": depth[0], "normal": normal[0], "accumulation": acc[0]}
    torch.autograd.backward([outs[k] for k in OUT_KEYS], [cot[k].to(DEV) for k in OUT_KEYS])
    torch.cuda.synchronize()
    rec = pr["splats"].grad.cpu()
    hip = {"means2d": rec[:, 0:2], "absgrad": rec[:, 14:16], "conics": rec[:, 2:5], "opacities": rec[:, 5], "colors": rec[:, 6:13]}
    assert_in_hull(hip, lo, hi, what + ", borderline-only backward (gradient records)")


def test_fused_pass_backward_of_borderline_pixels_lies_in_the_hull(dns, orc, monkeypatch):
    """C1 size, anisotropic scales and spread opacities."""
    from dn_splatter_amd import synthetic

    N = 10_000
    gp = synthetic.make_gauss_params(N, sh_rest_std=0.1, seed=0)
    g_ = torch.Generator().manual_seed(21)
    gp["scales"] = (gp["scales"].detach() + torch.randn(N, 3, generator=g_) * 0.5).requires_grad_(True)
    gp["opacities"] = (gp["opacities"].detach() + torch.randn(N, 1, generator=g_) * 1.5).requires_grad_(True)
    _fused_hull_case(dns, orc, monkeypatch, gp, synthetic.orbit_camera(3, width=256, height=256, focal=160.0), "fused pass")


def test_c2_full_frame_backward_of_borderline_pixels_lies_in_the_hull(dns, orc, monkeypatch):
    """The benchmark frame (1 M Gaussians, 1920 x 1080, the reference's initialisation): its ~10 000 borderline pixels — the ones every
    other backward comparison of that frame leaves without a cotangent — carry the only cotangents here."""
    from dn_splatter_amd import synthetic

    _oracle_threads()
    N, W, H = FULL["c2"]
    gp = synthetic.make_gauss_params(N, sh_rest_std=0.1, seed=0)
    _fused_hull_case(dns, orc, monkeypatch, gp, synthetic.orbit_camera(0, width=W, height=H), "C2 full frame, fused pass")
