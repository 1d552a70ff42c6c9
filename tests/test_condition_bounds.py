"""CPU twin of the condition-aware gradient bound (tests/test_gpu_determinism.py): the oracle's own fp32
evaluation against its fp64 evaluation must lie inside the running error bound of EVERY visible Gaussian's gradient rows, and a
gradient that is wrong by 1e-3 of a row must not.  No GPU involved: this pins the bound itself (oracle/oracle.py ConditionTrace,
orc_rasterize_bwd_cond in oracle/oracle_impl.inc; the reference call site is dn_splatter/dn_model.py:495-524)."""
import pytest
import torch

from _scenes import check_rows_conditioned, cotangents, gsplat_inputs, to_leaf, zero_borderline


@pytest.mark.parametrize("aniso", [False, True])
def test_oracle_fp32_lies_inside_its_own_running_error_bound(orc, aniso):
    W = H = 128
    N = 3000
    inp, viewmat, K, _ = gsplat_inputs(N, W, H, focal=80.0, seed=11, anisotropic=aniso)
    ci = to_leaf(inp, "cpu")
    kw = dict(width=W, height=H, packed=False, sh_degree=3, render_mode="RGB+ED", absgrad=True)
    prev = orc.set_exact_accumulation(True)
    try:
        with orc.ConditionTrace() as tr:
            r, a, info = orc.rasterization(**ci, viewmats=viewmat, Ks=K, **kw)
            info["means2d"].retain_grad()
            keep = ~info["borderline"]
            v_r, v_a = cotangents([r.shape, a.shape], 1)
            v_r, v_a = zero_borderline(v_r, keep), zero_borderline(v_a[..., 0], keep)[..., None]
            ((r * v_r).sum() + (a * v_a).sum()).backward(retain_graph=True)
            c_a, c_s = tr.param_condition(ci)
            r_a, r_s = tr.raster_condition(0, "A"), tr.raster_condition(0, "B")
        c64 = {k: v.detach().double().requires_grad_(True) for k, v in inp.items()}
        r_d, a_d, info_d = orc.rasterization(**c64, viewmats=viewmat.double(), Ks=K.double(), **kw)
        info_d["means2d"].retain_grad()
        ((r_d * v_r.double()).sum() + (a_d * v_a.double()).sum()).backward()
    finally:
        orc.set_exact_accumulation(prev)
    visible = info["radii"][0] > 0
    assert int(visible.sum()) > N // 2
    # the bound dominates the row's own magnitude term by term: A >= S >= 0, A >= |gradient| (kappa >= 8)
    for k in ci:
        assert bool((c_a[k] + 1e-300 >= c_s[k]).all()) and bool((c_s[k] >= 0).all()), k
        assert bool((c_a[k].reshape(N, -1)[visible] * 1.0001 >= ci[k].grad.double().abs().reshape(N, -1)[visible]).all()), k
    for k in ci:
        n, wa, ws = check_rows_conditioned(ci[k].grad, c64[k].grad, c_a[k], c_s[k], visible, f"oracle fp32 vs fp64 grad {k}")
        assert n == int(visible.sum())
    check_rows_conditioned(info["means2d"].grad, info_d["means2d"].grad, r_a["means2d"], r_s["means2d"], visible, "oracle fp32 vs fp64 means2d.grad")
    # a gradient off by 1e-3 of each row's norm is NOT inside the bound of most rows: the bound is not vacuous
    k = "colors"
    g = ci[k].grad.double().reshape(N, -1)
    bad = (g + 1e-3 * g.norm(dim=1, keepdim=True) * torch.ones_like(g) / g.shape[1] ** 0.5).reshape(ci[k].grad.shape)
    with pytest.raises(AssertionError):
        check_rows_conditioned(bad, c64[k].grad, c_a[k], c_s[k], visible, "perturbed gradient (must fail)")
