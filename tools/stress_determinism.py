"""Hunt for races: repeat fwd+bwd on the GPU and compare every run with the first one."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import dn_splatter_amd as dns
from _scenes import gsplat_inputs, to_leaf, cotangents
DEV = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
worst = {}
for case, (N, W, H, f, aniso) in enumerate([(10_000, 256, 256, 160.0, False), (10_000, 256, 256, 160.0, True), (60_000, 640, 360, 400.0, True)]):
    inp, viewmat, K, _ = gsplat_inputs(N, W, H, focal=f, seed=case, anisotropic=aniso)
    ref = None
    for it in range(reps):
        g = to_leaf(inp, DEV)
        r, a, info = dns.rasterization(**g, viewmats=viewmat.to(DEV), Ks=K.to(DEV), width=W, height=H, packed=False,
                                       sh_degree=3, render_mode="RGB+ED", absgrad=True)
        v_r, v_a = cotangents([r.shape, a.shape], 1)
        info["means2d"].retain_grad()
        ((r * v_r.to(DEV)).sum() + (a * v_a.to(DEV)).sum()).backward()
        cur = {"render": r.detach(), "alpha": a.detach(), "m2d": info["means2d"].grad, "abs": info["means2d"].absgrad,
               **{k: g[k].grad for k in g}}
        cur = {k: v.clone() for k, v in cur.items()}
        if ref is None:
            ref = cur
            continue
        for k in cur:
            scale = ref[k].abs().max().item() + 1e-30
            e = (cur[k] - ref[k]).abs().max().item() / scale
            worst[(case, k)] = max(worst.get((case, k), 0.0), e)
            if e > 1e-4:
                print("OUTLIER case", case, "iter", it, k, e)
for k, v in sorted(worst.items()):
    print(k, "%.2e" % v)
