"""Seeded scenes and comparison helpers shared by the CPU and GPU test files."""
from __future__ import annotations

import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The parity bar BASELINE.json states: "tile/bin indices bit-exact, rendered RGB/depth/normal and grads
# within 1e-4 rel fp32".  "rel" is taken relative to the tensor's scale (max |reference|): gradients are
# sums of thousands of signed terms whose fp32 summation order differs between the oracle (pixel-major,
# omp atomics) and the GPU (splat-major registers + one atomic per tile), so an element-wise relative
# error is unbounded at cancelling entries while the scale-relative error stays ~1e-6.
REL_TOL = 1e-4


def rel_err(a, b) -> float:
    a = a.detach().double().cpu().reshape(-1)
    b = b.detach().double().cpu().reshape(-1)
    assert a.shape == b.shape, (a.shape, b.shape)
    if b.numel() == 0:
        return 0.0
    scale = b.abs().max().item()
    return (a - b).abs().max().item() / (scale + 1e-30)


# The compositing rule has hard cut-offs (skip a pair if sigma < 0 or alpha < 1/255; stop a pixel when T(1-alpha) <=
# 1e-4; pass no gradient through alpha where o vis > 0.999 — SURVEY.md A.6/A.7).  Two fp32 implementations whose exp()
# differ in the last place (libm expf in the oracle, v_exp_f32 on the GPU, ex2.approx in gsplat's CUDA) take the other
# side of a cut-off for a handful of (pixel, splat) pairs per frame.  The oracle therefore reports, per pixel, whether
# any of its decisions fell inside the fp32 rounding envelope of its threshold (info["borderline"], see
# orc_rasterize_fwd in oracle/oracle_impl.inc), and how much compositing weight those decisions can move at the pixel
# (info["flip_weight"]).  What the tests do with it:
#   * FORWARD: every pixel is compared.  A pixel without a flagged decision is held to REL_TOL; a borderline pixel is held to
#     REL_TOL + the finite bound that follows from its flip weight (flip_bound_* below: one flipped skip moves at most
#     2 alpha_k T_k max|c|, i.e. <= (2/255) T max|c|; a flipped stop at most what is left, ~1e-4 max|c|).  A kernel that wrote
#     garbage there fails.
#   * BACKWARD: the cotangents fed to BOTH backward passes are zero on the borderline pixels, so they contribute exactly
#     nothing to any gradient on either side; every gradient entry is held to REL_TOL (+ the fp64 envelope, assert_close).
EXCLUDED = []    # (what, n_borderline, n_pixels) of every masked comparison, printed by the tests


def keep_mask(borderline, what=""):
    """bool [H,W] of the pixels WITHOUT a flagged decision (they carry the cotangents of the backward comparison); records and
    prints how many are borderline."""
    b = borderline.reshape(borderline.shape[-2], borderline.shape[-1]).bool().cpu()
    n, tot = int(b.sum()), b.numel()
    EXCLUDED.append((what, n, tot))
    print(f"[parity] {what}: {n} of {tot} pixels borderline ({100.0 * n / max(tot, 1):.4f} %): images held to the flip bound there, "
          "cotangents zero on both sides")
    # a mask that swallowed a visible share of the image would make the comparison meaningless
    assert n <= max(8, 0.01 * tot), f"{what}: {n} of {tot} pixels flagged borderline"
    return ~b


# ---- finite bounds for the image values of borderline pixels (all float64, shaped like the image they bound) -------------
FLIP_SLACK = 1.05    # the weights are evaluated on the oracle's own trajectory; the other side's T differs by rounding


def flip_bound_linear(flip, cmax):
    """raw composite channels out_c = sum_i w_i c_ic (+ T bg_c): [..., H, W] flip weights, [C] per-channel max |c| -> [..., H, W, C]"""
    return FLIP_SLACK * 2.0 * flip.double()[..., None] * cmax.double().reshape(*([1] * flip.dim()), -1)


def flip_bound_alpha(flip):
    """alpha image = sum_i w_i"""
    return FLIP_SLACK * flip.double()


def flip_bound_ratio(flip, cmax, value, alpha):
    """expected depth = acc / max(alpha, 1e-10): |d| <= (|d acc| + |value| |d alpha|) / (alpha - |d alpha|); unbounded (inf) where
    the flagged weight is the whole of the pixel's alpha.  All [..., H, W]; cmax a number."""
    f = FLIP_SLACK * flip.double()
    den = alpha.double() - f
    b = (2.0 * f * float(cmax) + value.double().abs() * f) / den.clamp_min(1e-300)
    return torch.where(f > 0, torch.where(den > 0, b, torch.full_like(b, float("inf"))), torch.zeros_like(b))


def flip_bound_unit(flip, cmax, norm):
    """(n / |n| + 1) / 2 of a composited vector n [..., H, W, 3] with |n| = ``norm`` [..., H, W]: |d n|_2 <= sqrt(3) 2 w cmax and
    |d (n/|n|)| <= 2 |d n| / max(|n|, |n'|)"""
    dn = FLIP_SLACK * 2.0 * (3.0 ** 0.5) * float(cmax) * flip.double()
    b = torch.minimum(torch.ones_like(dn), dn / (norm.double() - dn).clamp_min(1e-300))
    return torch.where(dn > 0, torch.where(norm.double() > dn, b, torch.ones_like(b)), torch.zeros_like(b))[..., None].expand(*flip.shape, 3)


def render_bounds(info, render, alphas, render_mode):
    """Bounds for the outputs of a rasterization() call (oracle info dict): (bound like ``render``, bound like ``alphas``)."""
    flip = info["flip_weight"].reshape(render.shape[:-1])
    cmax = info["channel_absmax"]
    rb = flip_bound_linear(flip, cmax)
    if render_mode in ("ED", "RGB+ED"):
        rb = rb.clone()
        rb[..., -1] = flip_bound_ratio(flip, float(cmax[-1]), render.detach()[..., -1], alphas.detach().reshape(flip.shape))
    return rb, flip_bound_alpha(flip).reshape(alphas.shape)


def assert_borderline_bounded(r_g, r_o, info_o, what, alpha=False, render_mode="RGB+ED", alphas_o=None):
    """The forward comparison of one output of a rasterization() call over ALL pixels (tools/parity_seed_sweep.py)."""
    if alpha:
        assert_close(r_g, r_o, what, bound=flip_bound_alpha(info_o["flip_weight"].reshape(r_o.shape[:-1]))[..., None])
        return
    a_o = alphas_o if alphas_o is not None else info_o["_alphas"]
    rb, _ = render_bounds(info_o, r_o, a_o, render_mode)
    nc = r_o.shape[-1]
    assert_close_groups(r_g, r_o, what, [("colour", 0, nc - 1), ("depth", nc - 1, nc)], bound=rb)


def image_pixels(t, keep):
    """[..., H, W, C] or [..., H, W] image -> the rows of the kept pixels."""
    H, W = keep.shape
    t = t.detach().cpu()
    if t.shape[-2:] == (H, W):
        t = t.reshape(-1, H, W)[0][..., None]
    else:
        t = t.reshape(-1, H, W, t.shape[-1])[0]
    return t[keep]


def zero_borderline(v, keep):
    """Cotangent image with the borderline pixels zeroed (same shape as ``v``)."""
    H, W = keep.shape
    k = keep.to(v.dtype)
    if v.shape[-2:] == (H, W):
        return v * k
    return v * k[..., None]


# gradient entries per comparison that may need the fp64 envelope (beyond the plain tolerance): the suite's own scenes need none
# (worst plain ratio 0.77, gpurun_out margins), unseen anisotropic scenes a handful (tools/parity_seed_sweep.py)
ENVELOPE_MAX_ENTRIES = 8
ENVELOPE_MAX_FRACTION = 2e-5


def assert_close_groups(a, b, what, groups, dim=-1, **kw):
    """assert_close on slices of ``dim`` taken separately, each against the scale of ITS slice of the reference: rgb and expected
    depth of a render (depth ~ 3-13 would otherwise set the scale for the colours), the SH bands of a coefficient gradient
    (band 0 is an order of magnitude above the rest).  ``groups``: [(name, lo, hi), ...]."""
    env, keep, bound = kw.pop("envelope", None), kw.get("keep"), kw.pop("bound", None)
    for name, lo, hi in groups:
        idx = [slice(None)] * a.dim()
        idx[dim] = slice(lo, hi)
        if lo >= a.shape[dim]:
            continue
        assert_close(a[tuple(idx)], b[tuple(idx)], f"{what}[{name}]", envelope=None if env is None else env[tuple(idx)],
                     bound=None if bound is None else bound[tuple(idx)], **kw)


def sh_band_groups(K):
    """bands of a [N, K, 3] coefficient tensor (K = 16: full, 15: features_rest)"""
    if K == 16:
        return [("band 0", 0, 1), ("band 1", 1, 4), ("band 2", 4, 9), ("band 3", 9, 16)]
    if K == 15:
        return [("band 1", 0, 3), ("band 2", 3, 8), ("band 3", 8, 15)]
    return [("all", 0, K)]


def assert_close(a, b, what, tol=REL_TOL, atol=0.0, keep=None, envelope=None, bound=None):
    """|a-b| <= tol * max|b| + atol (+ envelope) (+ bound) at every entry (over the pixels selected by ``keep`` for images).
    ``bound``: per-entry finite bound for the image values of borderline pixels (flip_bound_*; 0 everywhere else), shaped like ``b``.
    ``atol`` is only for tensors that are mathematically zero (e.g. the quaternion gradient of isotropic
    Gaussians), where both sides hold nothing but fp32 rounding noise.
    ``envelope``: per-entry rounding envelope of the REFERENCE algorithm itself, FP32_ENVELOPE x |oracle in fp32 - the same
    oracle in fp64| (fp64_envelope below): on anisotropic scenes the fp32 evaluation of a few gradient entries (sums with
    heavy cancellation through an ill-conditioned 2x2 inverse) is itself 1-2.4 tolerances away from exact arithmetic
    (tools/parity_seed_sweep.py), so no fp32 implementation with a different summation order can be expected closer than that
    THERE.  Example (seed 103 of the sweep): d/d(scale_0) of a disc with scales (0.021, 0.30, 0.40) covering 140 tiles is
    -1.65686 in fp64, -1.65977 in the fp32 oracle, -1.66212 / -1.66300 in two HIP runs (the atomics' order differs): a
    relative error of 2e-3 in EVERY fp32 evaluation.  Everywhere else the envelope is ~1e-7 of the value and the comparison
    stays at the plain tolerance; with the envelope the suite's worst comparison sits at 0.52 of its allowance and 60 unseen
    seeds of the C1 scene pass (without it: 0.77, and 3 of 60 fail).
    DNSPLAT_MARGIN_LOG=<file>: append "what, worst error / allowance" for every comparison (how close the suite runs to its limits)."""
    a_ = a.detach().cpu()
    b_ = b.detach().cpu()
    assert a_.shape == b_.shape, f"{what}: shape {tuple(a_.shape)} vs {tuple(b_.shape)}"
    assert torch.isfinite(a_.float()).all(), f"{what}: non-finite values"
    if b_.numel() == 0:
        return
    scale = b_.double().abs().max().item()
    if keep is not None:
        a_, b_ = image_pixels(a_, keep), image_pixels(b_, keep)
    d = (a_.double() - b_.double()).abs().reshape(-1)
    allow = torch.full_like(d, tol * scale + atol)
    if envelope is not None:
        assert keep is None and envelope.shape == b.shape
        allow = allow + envelope.detach().cpu().double().reshape(-1)
    if bound is not None:
        assert keep is None and tuple(bound.shape) == tuple(b.shape), (what, tuple(bound.shape), tuple(b.shape))
        bd = bound.detach().cpu().double().reshape(-1)
        n_b = int((bd > 0).sum())
        if n_b:
            over = d[bd > 0] > (tol * scale + atol)
            print(f"[parity] {what}: {n_b} entries of borderline pixels held to their flip bound (median {float(bd[bd > 0].median()):.2e}); "
                  f"{int(over.sum())} of them differ by more than the plain tolerance, worst error / bound = "
                  f"{float((d[bd > 0] / (allow[bd > 0] + bd[bd > 0])).max()):.3f}")
        allow = allow + bd
    ratio = (d / allow.clamp_min(1e-300)) if d.numel() else d
    worst = ratio.max().item() if d.numel() else 0.0
    log = os.environ.get("DNSPLAT_MARGIN_LOG")
    if log:
        strict = (d.max().item() / (tol * scale + atol)) if d.numel() and tol * scale + atol > 0 else 0.0
        with open(log, "a") as f:
            f.write(f"{os.environ.get('PYTEST_CURRENT_TEST', '?').split(' ')[0]}\t{what}\t{worst:.3f}\t{strict:.3f}\n")
    if envelope is not None and d.numel():
        # how many entries the plain tolerance alone would have failed: printed, and bounded — the envelope is for a handful of
        # ill-conditioned sums, not a second tolerance
        need = int((d > tol * scale + atol).sum())
        if need:
            print(f"[parity] {what}: {need} of {d.numel()} entries needed the fp64 rounding envelope")
        assert need <= max(ENVELOPE_MAX_ENTRIES, ENVELOPE_MAX_FRACTION * d.numel()), \
            f"{what}: {need} of {d.numel()} entries are beyond the plain tolerance (allowed: the fp64 envelope for at most " \
            f"{max(ENVELOPE_MAX_ENTRIES, int(ENVELOPE_MAX_FRACTION * d.numel()))})"
    if worst > 1.0:
        n_bad = int((ratio > 1.0).sum())
        raise AssertionError(f"{what}: max abs error {d.max().item():.3e} vs {tol:.1e} * scale {scale:.3e} + {atol:.1e}"
                             f"{' + fp64 envelope' if envelope is not None else ''}: worst error / allowance = {worst:.2f} "
                             f"({n_bad} of {d.numel()} entries beyond it)")


FP32_ENVELOPE = 4.0


def fp64_envelope(g32, g64):
    """FP32_ENVELOPE x |fp32 oracle - fp64 oracle| per entry: how far the reference algorithm's own fp32 evaluation is from
    exact arithmetic at that entry (assert_close)."""
    return FP32_ENVELOPE * (g32.detach().double() - g64.detach().double()).abs()


def assert_equal_int(a, b, what):
    a_ = a.detach().cpu()
    b_ = b.detach().cpu()
    assert a_.shape == b_.shape, f"{what}: shape {tuple(a_.shape)} vs {tuple(b_.shape)}"
    if not torch.equal(a_, b_):
        bad = (a_ != b_).nonzero()
        raise AssertionError(f"{what}: {bad.shape[0]} of {a_.numel()} integers differ, first at {bad[0].tolist()}: "
                             f"{a_[tuple(bad[0])].item()} vs {b_[tuple(bad[0])].item()}")


def gsplat_inputs(N, W, H, focal, seed=0, sh_rest_std=0.1, view=0, anisotropic=False):
    """Activated inputs in the form dn_model.py:496-504 hands to gsplat.rasterization (CPU tensors)."""
    import dn_splatter_amd as dns
    from dn_splatter_amd import synthetic

    gp = synthetic.make_gauss_params(N, sh_rest_std=sh_rest_std, seed=seed)
    cam = synthetic.orbit_camera(view, width=W, height=H, focal=focal)
    g = torch.Generator().manual_seed(seed + 1000)
    scales_log = gp["scales"].detach().clone()
    if anisotropic:
        scales_log = scales_log + torch.randn(N, 3, generator=g) * 0.6
    opac_logit = gp["opacities"].detach().clone()
    if anisotropic:
        opac_logit = opac_logit + torch.randn(N, 1, generator=g) * 2.0
    quats = gp["quats"].detach()
    inp = dict(
        means=gp["means"].detach().clone(),
        quats=(quats / quats.norm(dim=-1, keepdim=True)).clone(),
        scales=torch.exp(scales_log),
        opacities=torch.sigmoid(opac_logit).squeeze(-1),
        colors=torch.cat([gp["features_dc"].detach()[:, None], gp["features_rest"].detach()], 1).clone(),
    )
    viewmat = dns.get_viewmat(cam.camera_to_worlds)
    K = cam.get_intrinsics_matrices()
    return inp, viewmat, K, cam


def to_leaf(inp, device):
    return {k: v.detach().to(device).clone().requires_grad_(True) for k, v in inp.items()}


def cotangents(shapes, seed=1):
    g = torch.Generator().manual_seed(seed)
    return [torch.rand(s, generator=g) * 2 - 1 for s in shapes]


# ---- row-relative statistics ("1e-4 rel" held per Gaussian / per pixel, not only against the tensor's maximum) -----
# For a gradient tensor with one row per Gaussian: r_g = ||a_g - b_g||_2 / ||b_g||_2 over the rows whose reference norm is at
# least ROW_FLOOR x the largest row norm (rows below that are dominated by what cancels in them; they stay covered by the
# scale-relative comparison of assert_close).  Asserted where BOTH sides sum order-independently (HIP deterministic mode, oracle
# scatter accumulated in double), so that what is measured is the arithmetic, not the arrival order of atomics:
#   p99 <= ROW_P99 and max <= ROW_MAX of  max(0, ||d_g|| - FP32_ENVELOPE x ||oracle32_g - oracle64_g||) / ||ref_g||,
# i.e. outside the row's fp64 envelope.  The envelope is needed per row for the same reason as per entry (assert_close): the
# reference algorithm's OWN fp32 evaluation is that far from exact arithmetic — measured on the C1 scenes, oracle fp32 against oracle
# fp64, per Gaussian: p50 6e-6 / p99 2e-4 / max 2e-3 isotropic, p50 1.3e-4 / p99 2e-3 / max 8e-3 anisotropic (the backward starts
# from T_final = 1 - alpha image, a multiple of 2^-24: 6e-4 relative at a saturated pixel; 1 / (1 - alpha) at opacities near the
# cap) — so no second fp32 implementation can be held to 1e-4 per row against it, only to 1e-4 beyond that envelope.  The raw
# statistics are printed and logged beside the asserted ones.
# In the default (atomics) mode the statistics are logged only.  DNSPLAT_ROWREL_LOG=<file> appends one line per comparison.
# Measured (gpurun_out/r05a -> profiles/r05_rowrel.tsv): gradients, deterministic mode, outside the envelope: p99 <= 2.6e-5 everywhere
# (C2 full frame: 4.9e-6), max 1.6e-4 over the 20-scene sweep and 1.17e-3 for ONE of the 29 584 counted opacity rows of the full C2
# frame (an extreme-value statistic over 1 M rows: the kernels add a half tile's 128 terms in fp32 before the double sum, the oracle
# adds every term in double — 1e-6 x the row's cancellation factor, which the fp32-vs-fp64 envelope of the ORACLE does not contain);
# images, every mode: p99 <= 6.5e-7, max <= 1.4e-6 per pixel.  Hence: gradients p99 <= 1e-4, max <= 2e-3; images p99 <= 1e-5, max <= 1e-4.
ROW_FLOOR = 1e-3
ROW_P99 = 1e-4
ROW_MAX = 2e-3
PIX_P99 = 1e-5
PIX_MAX = 1e-4
# Default (atomics) mode, where the arrival order of ~30 atomic rows per Gaussian adds its own noise: measured over the suite
# (profiles/r05c_rowrel.tsv) outside the envelope p99 <= 9.0e-5, max <= 1.3e-3; where a test computes no fp64 run (no envelope) the
# raw statistics are p99 <= 5.0e-4, max <= 8.0e-3 (C5 frame, opacities).  Asserted with a factor ~2-4 of room — a guard against a
# kernel that corrupts whole rows which the tensor-scale comparison would not see, not a statement about rounding.
ROW_P99_DEFAULT, ROW_MAX_DEFAULT = 2e-4, 5e-3
ROW_P99_RAW, ROW_MAX_RAW = 1e-3, 2e-2


def row_rel_stats(a, b, b64=None, floor=ROW_FLOOR):
    """(p50, p99, max, max outside the fp64 envelope, rows counted, rows total) of the row-relative L2 error; rows = all leading
    dimensions but the last (a 1-D tensor is one entry per row)."""
    a_ = a.detach().double().cpu()
    b_ = b.detach().double().cpu()
    assert a_.shape == b_.shape, (a_.shape, b_.shape)
    if b_.dim() == 1:
        a_, b_ = a_[:, None], b_[:, None]
    a_, b_ = a_.reshape(-1, b_.shape[-1]), b_.reshape(-1, b_.shape[-1])
    nb = b_.norm(dim=1)
    if nb.numel() == 0 or float(nb.max()) == 0.0:
        return None
    sel = nb >= floor * nb.max()
    err = (a_ - b_).norm(dim=1)[sel]
    r = err / nb[sel]
    r_out = r
    if b64 is not None:
        e64 = b64.detach().double().cpu()
        e64 = (e64[:, None] if e64.dim() == 1 else e64).reshape(-1, b_.shape[-1])
        env = FP32_ENVELOPE * (b_ - e64).norm(dim=1)[sel]
        r_out = (err - env).clamp_min(0) / nb[sel]

    def q(t, f):
        if not t.numel():
            return 0.0
        k = min(t.numel() - 1, int(f * t.numel()))
        return float(torch.sort(t).values[k])       # torch.quantile refuses more than 16 M entries
    return q(r, 0.5), q(r, 0.99), float(r.max()), float(r_out.max()), int(sel.sum()), int(nb.numel()), q(r_out, 0.99)


ROWREL_ENFORCE = os.environ.get("DNSPLAT_ROWREL_ENFORCE", "1") != "0"      # 0: log the statistics, assert nothing (exploration runs)


def both_sides_order_independent(orc) -> bool:
    """True when the HIP path runs its deterministic gradient mode AND the oracle accumulates its scatter in double: the only
    configuration in which a row-relative bound on gradients measures arithmetic rather than the arrival order of atomics."""
    from dn_splatter_amd import _ops
    return bool(_ops.DETERMINISTIC["on"]) and bool(orc.lib().orc_get_exact_accum())


def check_rows(a, b, what, b64=None, enforce=False, p99=None, rmax=None, n_rows=None, strict=True):
    """``n_rows``: reshape to [n_rows, -1] first (one row per Gaussian whatever the trailing shape).  ``strict``: both sides sum
    order-independently (the deterministic-mode bounds ROW_P99 / ROW_MAX); otherwise the default-mode guards (with / without an fp64
    envelope)."""
    enforce = enforce and ROWREL_ENFORCE
    if p99 is None:
        p99 = ROW_P99 if strict else (ROW_P99_DEFAULT if b64 is not None else ROW_P99_RAW)
    if rmax is None:
        rmax = ROW_MAX if strict else (ROW_MAX_DEFAULT if b64 is not None else ROW_MAX_RAW)
    if n_rows is not None:
        a, b = a.detach().reshape(n_rows, -1), b.detach().reshape(n_rows, -1)
        b64 = None if b64 is None else b64.detach().reshape(n_rows, -1)
    st = row_rel_stats(a, b, b64)
    if st is None:
        return None
    p50, p99_, mx, mx_out, n, tot, p99_out = st
    print(f"[parity] {what}: row-relative error over {n} of {tot} rows: p50 {p50:.2e}  p99 {p99_:.2e}  max {mx:.2e}"
          + (f"  (outside the fp64 envelope: p99 {p99_out:.2e}  max {mx_out:.2e})" if b64 is not None else ""))
    log = os.environ.get("DNSPLAT_ROWREL_LOG")
    if log:
        with open(log, "a") as f:
            f.write(f"{os.environ.get('PYTEST_CURRENT_TEST', '?').split(' ')[0]}\t{what}\t{n}\t{tot}\t{p50:.3e}\t{p99_:.3e}\t{mx:.3e}\t{p99_out:.3e}\t"
                    f"{mx_out:.3e}\t{('asserted <= %.0e / %.0e' % (p99, rmax)) if enforce else 'logged'}\n")
    if enforce:
        if p99_out > p99 or mx_out > rmax:
            print(f"[parity] {what}: ROW-RELATIVE BOUND EXCEEDED (p99 {p99_out:.3e} vs {p99:.1e}, max {mx_out:.3e} vs {rmax:.1e})")
        assert p99_out <= p99, f"{what}: p99 of the row-relative error{' outside the fp64 envelope' if b64 is not None else ''} {p99_out:.3e} > {p99:.1e}"
        assert mx_out <= rmax, f"{what}: largest row-relative error{' outside the fp64 envelope' if b64 is not None else ''} {mx_out:.3e} > {rmax:.1e}"
    return st


def check_pixels(a, b, what, keep=None, enforce=False, p99=PIX_P99, rmax=PIX_MAX):
    """The same statistic per PIXEL of an image [.., H, W, C] (over its C channels), borderline pixels left to their flip bound."""
    a_, b_ = a.detach().cpu(), b.detach().cpu()
    if keep is not None:
        a_, b_ = image_pixels(a_, keep), image_pixels(b_, keep)
    return check_rows(a_, b_, what, enforce=enforce, p99=p99, rmax=rmax)


# ---- condition-aware bound for EVERY visible Gaussian --------------------------------------------------------
# The floor-based statistic above counts the rows whose reference norm is >= 1e-3 x the largest (about 5 % of the visible Gaussians
# of the C2 frame).  This one covers ALL rows with radii > 0, each against its OWN running error bound from the oracle's
# ConditionTrace (oracle/oracle.py, orc_rasterize_bwd_cond in oracle_impl.inc), chained to the parameters through |J|:
#     worst case            ||d_g||_2 <= COND_C      x 2^-24 x ||A_g||_2     A = sum of term magnitudes x rounding count kappa
#     independent roundings ||d_g||_2 <= COND_LAMBDA x 2^-24 x ||S_g||_2     S = sqrt(sum of squared magnitudes x variance count)
# d = HIP (deterministic mode) - oracle fp32 (scatter in double): two fp32 evaluations of the same rule set from the same inputs.
# A row whose bound is zero (no pair blended it) must be exactly equal.  The constants are ONE pair for every tensor, scene and size;
# measured (profiles/r06_rowrel.tsv): see the numbers printed by check_rows_conditioned.  In the default (atomics) mode the same
# statistic is asserted with COND_*_DEFAULT (the arrival order of ~30 fp32 atomic rows per Gaussian adds roundings the model of the
# double-accumulated reduction does not count).
U24 = 2.0 ** -24
# c = 4.  The bound is on ONE evaluation's distance from exact arithmetic; HIP and the oracle's fp32 run are each within ~1 x of it and may
# sit on opposite sides of the exact value (2 x), with a factor 2 of room.  Worst row seen: 1.45 x A (seed 104 anisotropic, an SH-coefficient
# row of a 48-pixel-radius splat, tools/cond_outlier.py; 1.96 before kappa counted the splat's own rounded conic / centre, which shift the
# alphas of all its pixels the same way: oracle.input_perturbation); everything else <= 0.60 A over the suite and the sweep's other 59
# scenes (median 0.08), <= 2.0 S everywhere; C2 frame 0.22 A / 0.77 S.
# With atomics the arrival order adds the roundings of ~30 sequential fp32 adds per Gaussian: c = 8 there.
COND_C, COND_LAMBDA = 4.0, 8.0
COND_C_DEFAULT, COND_LAMBDA_DEFAULT = 8.0, 16.0


def check_rows_conditioned(a, b, cond_a, cond_s, visible, what, enforce=True, strict=True):
    """``a`` (HIP) vs ``b`` (oracle) gradient tensors with one row per Gaussian; ``cond_a`` / ``cond_s``: the A / S bounds of the same
    shape (units of one rounding); ``visible`` bool [N].  Returns (rows counted, worst ratio vs A, worst ratio vs S)."""
    N = visible.numel()
    a_ = a.detach().double().cpu().reshape(N, -1)[visible]
    b_ = b.detach().double().cpu().reshape(N, -1)[visible]
    ca = cond_a.detach().double().reshape(N, -1)[visible]
    cs = cond_s.detach().double().reshape(N, -1)[visible]
    d = (a_ - b_).norm(dim=1)
    na, ns, nb = ca.norm(dim=1), cs.norm(dim=1), b_.norm(dim=1)
    dead = na == 0
    n_dead_bad = int((d[dead] != 0).sum())
    live = ~dead
    ra = d[live] / (U24 * na[live])
    rs = d[live] / (U24 * ns[live]).clamp_min(1e-300)
    rel_bound = (U24 * ns[live] / nb[live].clamp_min(1e-300))
    rel_err = d[live] / nb[live].clamp_min(1e-300)

    def q(t, f):
        if not t.numel():
            return 0.0
        k = min(t.numel() - 1, int(f * t.numel()))
        return float(torch.sort(t).values[k])
    c_lim, l_lim = (COND_C, COND_LAMBDA) if strict else (COND_C_DEFAULT, COND_LAMBDA_DEFAULT)
    n = int(visible.sum())
    wa, ws = (float(ra.max()) if ra.numel() else 0.0), (float(rs.max()) if rs.numel() else 0.0)
    print(f"[parity] {what}: ALL {n} visible rows ({int(live.sum())} with a non-zero bound, {int(dead.sum())} exactly equal): "
          f"error / (2^-24 A): p50 {q(ra, .5):.2e} p99 {q(ra, .99):.2e} max {wa:.3f} (<= {c_lim});  error / (2^-24 S): p50 {q(rs, .5):.2e} "
          f"p99 {q(rs, .99):.2e} max {ws:.3f} (<= {l_lim});  row-relative error p50 {q(rel_err, .5):.1e} p99 {q(rel_err, .99):.1e} max "
          f"{(float(rel_err.max()) if rel_err.numel() else 0.0):.1e};  bound 2^-24 S / |ref| p50 {q(rel_bound, .5):.1e} p99 {q(rel_bound, .99):.1e}")
    log = os.environ.get("DNSPLAT_CONDREL_LOG")
    if log:
        with open(log, "a") as f:
            f.write(f"{os.environ.get('PYTEST_CURRENT_TEST', '?').split(' ')[0]}\t{what}\t{n}\t{N}\t{int(live.sum())}\t{q(ra, .5):.3e}\t{q(ra, .99):.3e}\t{wa:.3e}\t"
                    f"{q(rs, .5):.3e}\t{q(rs, .99):.3e}\t{ws:.3e}\t{q(rel_err, .5):.3e}\t{q(rel_err, .99):.3e}\t"
                    f"{(float(rel_err.max()) if rel_err.numel() else 0.0):.3e}\t{q(rel_bound, .5):.3e}\t{q(rel_bound, .99):.3e}\t"
                    f"{('asserted <= %g A, <= %g S' % (c_lim, l_lim)) if enforce else 'logged'}\n")
    if enforce:
        assert n_dead_bad == 0, f"{what}: {n_dead_bad} rows differ although no pair contributes to them"
        assert wa <= c_lim, f"{what}: a row is {wa:.3f} x its worst-case running error bound (allowed {c_lim})"
        assert ws <= l_lim, f"{what}: a row is {ws:.3f} standard deviations of the independent-roundings model off (allowed {l_lim})"
    return n, wa, ws


# ---- raster-level scene for the backward of borderline pixels (tests/test_borderline_bounds.py, tests/test_gpu_parity.py) ---------


def raster_level_scene(orc, seed, N=10_000, W=256, H=256, focal=160.0, D=4, anisotropic=True, view=0):
    """Projected Gaussians (the oracle's own projection), their sorted tile lists, random per-Gaussian channels and a background:
    everything gsplat's rasterize_to_pixels takes, as CPU tensors, plus the oracle's forward with its borderline mask."""
    inp, viewmat, K, _ = gsplat_inputs(N, W, H, focal=focal, seed=seed, anisotropic=anisotropic, view=view)
    radii, xys, depths, conics, _comp, tiles = orc.project_fwd(inp["means"], inp["quats"], inp["scales"], viewmat[0], K[0], W, H)
    tw, th = (W + 15) // 16, (H + 15) // 16
    _t, ids, fid = orc.isect_tiles(xys, radii, depths, 16, tw, th)
    offs = orc.isect_offset_encode(ids, tw, th)
    g = torch.Generator().manual_seed(seed + 77)
    cols = torch.rand(N, D, generator=g)
    bg = torch.rand(D, generator=g)
    border = torch.zeros(H, W, dtype=torch.uint8)
    flip = torch.zeros(H, W, dtype=torch.float32)
    render, alphas, last = orc.rasterize_fwd(xys, conics, cols, inp["opacities"], bg, W, H, 16, offs, fid, border, flip)
    return dict(W=W, H=H, D=D, N=N, xys=xys, conics=conics, colors=cols, opacities=inp["opacities"], background=bg, radii=radii,
                depths=depths, tiles=tiles, offsets=offs, flatten_ids=fid, render=render, alphas=alphas, last_ids=last,
                borderline=border.bool(), gen=g)


HULL_KEYS = ("means2d", "absgrad", "conics", "colors", "opacities")


def assert_in_hull(grads, lo, hi, what, tol=REL_TOL, scales=None):
    """lo - tol * scale <= g <= hi + tol * scale at every entry of every tensor of HULL_KEYS; ``scales``: per-key scale (default: the
    largest |lo|, |hi| of that tensor).  Returns the worst violation in units of tol * scale."""
    worst = 0.0
    for k in HULL_KEYS:
        g = grads[k].detach().double().cpu().reshape(lo[k].shape)
        scale = float(scales[k]) if scales is not None else float(torch.maximum(lo[k].abs(), hi[k].abs()).max())
        if scale == 0.0:
            assert float(g.abs().max()) == 0.0, f"{what} {k}: non-zero gradient where every admissible evaluation gives zero"
            continue
        viol = torch.maximum(lo[k] - g, g - hi[k]).clamp_min(0)
        w = float(viol.max()) / (tol * scale)
        width = float((hi[k] - lo[k]).max()) / scale
        print(f"[parity] {what} {k}: worst distance from the hull of admissible decisions = {w:.3f} x (1e-4 x scale {scale:.3e}); "
              f"{int((hi[k] > lo[k]).sum())} entries with a non-degenerate interval, widest {width:.2e} of scale")
        log = os.environ.get("DNSPLAT_MARGIN_LOG")
        if log:
            with open(log, "a") as f:
                f.write(f"{os.environ.get('PYTEST_CURRENT_TEST', '?').split(' ')[0]}\t{what} {k} (hull)\t{w:.3f}\t{w:.3f}\n")
        worst = max(worst, w)
        assert w <= 1.0, f"{what} {k}: {int((viol > tol * scale).sum())} entries outside the hull by more than 1e-4 x scale, worst {w:.2f}"
    return worst
