"""Which row of which tensor is the worst of a scene under the condition-aware bound (tests/_scenes.check_rows_conditioned)?
    DNSPLAT_DETERMINISTIC=1 python tools/cond_outlier.py <seed> <aniso 0|1>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import dn_splatter_amd as dns
from _scenes import U24, cotangents, gsplat_inputs, to_leaf, zero_borderline
from oracle import oracle as orc

seed, aniso = int(sys.argv[1]), bool(int(sys.argv[2]))
orc.set_exact_accumulation(True)
DEV = "cuda:0"
inp, viewmat, K, _ = gsplat_inputs(10_000, 256, 256, focal=160.0, seed=seed, anisotropic=aniso, view=seed % 8)
ci = to_leaf(inp, "cpu")
kw = dict(width=256, height=256, packed=False, sh_degree=3, render_mode="RGB+ED", absgrad=True)
with orc.ConditionTrace() as tr:
    r_o, a_o, info_o = orc.rasterization(**ci, viewmats=viewmat, Ks=K, **kw)
    keep = ~info_o["borderline"]
    v_r, v_a = cotangents([r_o.shape, a_o.shape], seed)
    v_r, v_a = zero_borderline(v_r, keep), zero_borderline(v_a[..., 0], keep)[..., None]
    ((r_o * v_r).sum() + (a_o * v_a).sum()).backward(retain_graph=True)
    cA, cS = tr.param_condition(ci)
    rA = tr.raster_condition(0, "A")
c64 = {k: v.detach().double().requires_grad_(True) for k, v in inp.items()}
r_d, a_d, _ = orc.rasterization(**c64, viewmats=viewmat.double(), Ks=K.double(), **kw)
((r_d * v_r.double()).sum() + (a_d * v_a.double()).sum()).backward()
gi = to_leaf(inp, DEV)
r_g, a_g, info_g = dns.rasterization(**gi, viewmats=viewmat.to(DEV), Ks=K.to(DEV), **kw)
((r_g * v_r.to(DEV)).sum() + (a_g * v_a.to(DEV)).sum()).backward()
vis = info_o["radii"][0] > 0
N = vis.numel()
for k in ci:
    d = (gi[k].grad.cpu().double() - ci[k].grad.double()).reshape(N, -1)
    d64 = (ci[k].grad.double() - c64[k].grad).reshape(N, -1)
    dh64 = (gi[k].grad.cpu().double() - c64[k].grad).reshape(N, -1)
    A = cA[k].reshape(N, -1)
    ratio = d.norm(dim=1) / (U24 * A.norm(dim=1)).clamp_min(1e-300)
    ratio[~vis] = 0
    g = int(ratio.argmax())
    print(f"{k}: worst row {g}: |hip-o32| {float(d[g].norm()):.3e}  |o32-o64| {float(d64[g].norm()):.3e}  |hip-o64| {float(dh64[g].norm()):.3e}  u|A| {float(U24 * A[g].norm()):.3e}  "
          f"|grad| {float(c64[k].grad.reshape(N, -1)[g].norm()):.3e}  ratio {float(ratio[g]):.3f}  opacity {float(inp['opacities'][g]):.5f}  radius {int(info_o['radii'][0][g])} "
          f"raster A row {[f'{float(x):.2e}' for x in rA['conics'][g]]}")
