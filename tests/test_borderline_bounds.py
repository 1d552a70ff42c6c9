"""The finite bound the parity tests hold borderline pixels to (tests/_scenes.py flip_bound_*, oracle/oracle_impl.inc
``flip_weight``), exercised on the CPU: the oracle in fp32 against the SAME oracle in fp64.  The two evaluations take the other
side of a skip / stop decision at some of the flagged pixels — exactly what a second fp32 implementation (the HIP kernels) does —
so their difference there must stay inside the bound, while everywhere else it stays inside the plain tolerance."""
import pytest
import torch

from _scenes import REL_TOL, assert_close, assert_close_groups, gsplat_inputs, render_bounds


def _both(orc, seed, N=4000, W=128, H=96, aniso=True, mode="RGB+ED"):
    inp, viewmat, K, _ = gsplat_inputs(N, W, H, focal=90.0, seed=seed, anisotropic=aniso)
    kw = dict(width=W, height=H, packed=False, sh_degree=3, render_mode=mode)
    with torch.no_grad():
        r32, a32, i32 = orc.rasterization(**inp, viewmats=viewmat, Ks=K, **kw)
        r64, a64, _ = orc.rasterization(**{k: v.double() for k, v in inp.items()}, viewmats=viewmat.double(), Ks=K.double(), **kw)
    return r32, a32, i32, r64.float(), a64.float()


@pytest.mark.parametrize("mode", ["RGB+ED", "RGB"])
def test_flipped_decisions_stay_inside_the_flip_bound(orc, mode):
    flipped = flagged = 0
    for seed in range(300, 312):
        r32, a32, info, r64, a64 = _both(orc, seed, mode=mode)
        rb, ab = render_bounds(info, r32, a32, mode)
        assert rb.shape == r32.shape and ab.shape == a32.shape
        # bound is zero exactly on the pixels without a flagged skip / stop decision
        fw = info["flip_weight"]
        assert bool(((fw > 0) <= info["borderline"]).all())
        nc = r32.shape[-1]
        groups = [("colour", 0, 3)] + ([("depth", 3, 4)] if nc == 4 else [])
        assert_close_groups(r64, r32, f"seed {seed} render fp64 vs fp32", groups, bound=rb)
        assert_close(a64, a32, f"seed {seed} alpha fp64 vs fp32", bound=ab)
        flagged += int((fw > 0).sum())
        over = (a64 - a32).abs()[0, ..., 0] > REL_TOL * float(a32.abs().max())
        for _n, lo, hi in groups:
            over |= ((r64 - r32).abs()[0, ..., lo:hi] > REL_TOL * float(r32[..., lo:hi].abs().max())).any(-1)
        assert bool((over <= (fw > 0)).all())            # beyond the plain tolerance only where a decision was flagged
        flipped += int(over.sum())
    # the case the bound exists for does occur: some flagged pixels really differ by more than the plain tolerance
    assert flagged > 0
    print(f"[parity] {flagged} flagged pixels over 12 scenes, {flipped} of them differ by more than the plain tolerance between fp32 and fp64")
    assert flipped > 0, "no decision flipped between fp32 and fp64 on any of the scenes: the test does not exercise the bound"


def test_garbage_on_a_borderline_pixel_fails(orc):
    """What the exclusion of round 3 could not see: a wrong value written to a borderline pixel."""
    r32, a32, info, _, _ = _both(orc, 300)
    fw = info["flip_weight"]
    assert int((fw > 0).sum()) > 0
    y, x = (fw > 0).nonzero()[0].tolist()
    rb, ab = render_bounds(info, r32, a32, "RGB+ED")
    bad = r32.clone()
    bad[0, y, x, 0] += 0.05
    with pytest.raises(AssertionError):
        assert_close_groups(bad, r32, "render", [("colour", 0, 3), ("depth", 3, 4)], bound=rb)
    bad_a = a32.clone()
    bad_a[0, y, x, 0] = 0.0
    with pytest.raises(AssertionError):
        assert_close(bad_a, a32, "alpha", bound=ab)


# ---- the BACKWARD of borderline pixels: cotangents ONLY on them, result held to the hull of admissible decisions


def _hull_case(orc, seed, D=4, **kw):
    from _scenes import raster_level_scene

    s = raster_level_scene(orc, seed, N=4000, W=128, H=96, focal=90.0, D=D, **kw)
    m = s["borderline"]
    g = s["gen"]
    s["v_r"] = (torch.rand(s["H"], s["W"], D, generator=g) * 2 - 1) * m[..., None]
    s["v_a"] = (torch.rand(s["H"], s["W"], generator=g) * 2 - 1) * m
    args = (s["xys"], s["conics"], s["colors"], s["opacities"], s["background"], s["W"], s["H"], 16, s["offsets"], s["flatten_ids"])
    s["args"] = args
    s["lo"], s["hi"], s["status"] = orc.rasterize_bwd_hull(*args, m, s["v_r"], s["v_a"])
    return s


def _as_dict(t5):
    return dict(zip(("means2d", "absgrad", "conics", "colors", "opacities"), t5))


def test_backward_of_borderline_pixels_lies_in_the_hull_of_admissible_decisions(orc):
    """The oracle's own fp32 backward (the natural outcome of every decision) is inside the hull to the last bits of the double
    accumulation; the SAME oracle in fp64 — which takes the other side of some flagged decisions, as a second fp32 implementation
    does — is inside it to the plain tolerance; a wrong gradient on a borderline pixel is not."""
    from _scenes import assert_in_hull

    prev = orc.set_exact_accumulation(True)
    try:
        wide = 0
        for seed in range(300, 306):
            s = _hull_case(orc, seed)
            st = s["status"]
            assert st["pixels"] == int(s["borderline"].sum()) > 0
            assert st["pixels_over_cap"] == 0 and st["pixels_incomplete"] == 0, st
            base = _as_dict(orc.rasterize_bwd(*s["args"], s["alphas"], s["last_ids"], s["v_r"], s["v_a"], absgrad=True))
            assert assert_in_hull(base, s["lo"], s["hi"], f"seed {seed} oracle fp32") <= 1e-3
            d = lambda t: t.double() if t.is_floating_point() else t      # noqa: E731
            a64 = tuple(d(t) if torch.is_tensor(t) else t for t in s["args"])
            r64, al64, last64 = orc.rasterize_fwd(*a64)
            g64 = _as_dict(orc.rasterize_bwd(*a64, al64, last64, s["v_r"].double(), s["v_a"].double(), absgrad=True))
            assert_in_hull(g64, s["lo"], s["hi"], f"seed {seed} oracle fp64")
            wide += sum(int((s["hi"][k] > s["lo"][k]).sum()) for k in s["lo"])
            # garbage: one entry of a touched Gaussian moved by 1e-2 of the tensor's scale
            bad = {k: v.clone() for k, v in base.items()}
            gi = int(bad["opacities"].abs().argmax())
            bad["opacities"][gi] += 1e-2 * float(bad["opacities"].abs().max())
            with pytest.raises(AssertionError):
                assert_in_hull(bad, s["lo"], s["hi"], f"seed {seed} garbage")
        assert wide > 0, "no flagged decision changed any gradient entry: the hull is a point everywhere"
    finally:
        orc.set_exact_accumulation(prev)


def test_single_borderline_pixel_gradient_equals_one_admissible_combination(orc):
    """Sharper than the hull: with the cotangent on ONE borderline pixel, the fp64 oracle's gradient must equal the gradient of one
    of the 2^k combinations of outcomes of that pixel's k flagged decisions (all entries inside the plain tolerance at once)."""
    s = _hull_case(orc, 301)
    ys, xs = torch.nonzero(s["borderline"], as_tuple=True)
    d = lambda t: t.double() if t.is_floating_point() else t      # noqa: E731
    a64 = tuple(d(t) if torch.is_tensor(t) else t for t in s["args"])
    r64, al64, last64 = orc.rasterize_fwd(*a64)
    prev = orc.set_exact_accumulation(True)
    try:
        matched_other = 0
        for y, x in list(zip(ys.tolist(), xs.tolist()))[:24]:
            m = torch.zeros_like(s["borderline"])
            m[y, x] = True
            v_r, v_a = s["v_r"] * m[..., None], s["v_a"] * m
            g64 = _as_dict(orc.rasterize_bwd(*a64, al64, last64, v_r.double(), v_a.double(), absgrad=True))
            _lo, _hi, st = orc.rasterize_bwd_hull(*s["args"], m, v_r, v_a)
            k = st["max_flags_seen"]
            assert 1 <= k <= 10
            best, best_c = None, None
            for c in range(1 << k):
                one, _same, _ = orc.rasterize_bwd_hull(*s["args"], m, v_r, v_a, combo=c)
                # the backward starts from T_final = 1 - alpha image in fp32 (a multiple of 2^-24): 2 ulp / T_final of relative shift
                # on everything the pixel contributes, on top of the plain tolerance (oracle_impl.inc, orc_rasterize_bwd_hull)
                t_fin = max(1.0 - float(s["alphas"][y, x]), 1e-30)
                dlt = min(1.0, 2.0 * 2.0 ** -24 / t_fin)
                w = max(float(((g64[n] - one[n]).abs() / (1e-4 * max(float(one[n].abs().max()), float(g64[n].abs().max()), 1e-30)
                                                        + dlt * one[n].abs())).max()) for n in one)
                if best is None or w < best:
                    best, best_c = w, c
            assert best <= 1.0, f"pixel ({y}, {x}): the fp64 gradient matches none of the {1 << k} admissible combinations (closest {best:.2f} tolerances)"
            nat, _s, _ = orc.rasterize_bwd_hull(*s["args"], m, v_r, v_a, combo=best_c)
            base = _as_dict(orc.rasterize_bwd(*s["args"], s["alphas"], s["last_ids"], v_r, v_a, absgrad=True))
            if any(float((base[n].double() - nat[n]).abs().max()) > 1e-4 * max(float(nat[n].abs().max()), 1e-30) for n in nat):
                matched_other += 1
        print(f"[parity] {matched_other} of 24 borderline pixels: the fp64 evaluation takes another admissible combination than the fp32 one")
    finally:
        orc.set_exact_accumulation(prev)
