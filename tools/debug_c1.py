import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import dn_splatter_amd as dns
from oracle import oracle as orc
from _scenes import gsplat_inputs, to_leaf, cotangents, rel_err
torch.set_num_threads(16)
DEV = "cuda:0"
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
inp, viewmat, K, _ = gsplat_inputs(10_000, 256, 256, focal=160.0, seed=seed, anisotropic=True)
kw = dict(width=256, height=256, packed=False, sh_degree=3, render_mode="RGB+ED", absgrad=True)
def run(fn, ii, vm, k_, dev):
    r, a, info = fn(**ii, viewmats=vm, Ks=k_, **kw)
    v_r, v_a = cotangents([r.shape, a.shape], 1)
    info["means2d"].retain_grad()
    ((r * v_r.to(dev).to(r.dtype)).sum() + (a * v_a.to(dev).to(r.dtype)).sum()).backward()
    return r, a, info
c32 = to_leaf(inp, "cpu"); g = to_leaf(inp, DEV)
o32 = run(orc.rasterization, c32, viewmat, K, "cpu")
gg = run(dns.rasterization, g, viewmat.to(DEV), K.to(DEV), DEV)
torch.cuda.synchronize()
print("render", rel_err(gg[0], o32[0]), "alpha", rel_err(gg[1], o32[1]))
dr = (gg[0].cpu() - o32[0]).abs().amax(-1)[0]
print("pixels with render err > 1e-5:", int((dr > 1e-5 * o32[0].abs().max()).sum()))
for k in c32:
    a, b = g[k].grad.cpu(), c32[k].grad
    scale = b.abs().max().item()
    d = (a - b).abs().reshape(a.shape[0], -1).amax(1)
    bad = (d > 1e-4 * scale).nonzero().flatten()
    print(k, "rel %.3e" % rel_err(a, b), "n_bad", bad.numel(), "of", a.shape[0])
    if k == "opacities":
        for i in bad[:12].tolist():
            print("   g", i, "opac %.5f" % inp["opacities"][i].item(), "gpu %.5f o32 %.5f" % (a[i].item(), b[i].item()), "radius", o32[2]["radii"][0, i].item())
