import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import dn_splatter_amd as dns
from oracle import oracle as orc
from _scenes import gsplat_inputs, to_leaf, cotangents, rel_err
torch.set_num_threads(16)
DEV = "cuda:0"
inp, viewmat, K, _ = gsplat_inputs(10_000, 256, 256, focal=160.0, seed=0)
kw = dict(width=256, height=256, packed=False, sh_degree=3, render_mode="RGB+ED", absgrad=True)
def run(fn, ii, vm, k_, dev):
    r, a, info = fn(**ii, viewmats=vm, Ks=k_, **kw)
    v_r, v_a = cotangents([r.shape, a.shape], 1)
    info["means2d"].retain_grad()
    ((r * v_r.to(dev).to(r.dtype)).sum() + (a * v_a.to(dev).to(r.dtype)).sum()).backward()
    return r, a, info
c32 = to_leaf(inp, "cpu"); c64 = to_leaf({k: v.double() for k, v in inp.items()}, "cpu"); g = to_leaf(inp, DEV)
o32 = run(orc.rasterization, c32, viewmat, K, "cpu")
o64 = run(orc.rasterization, c64, viewmat.double(), K.double(), "cpu")
gg = run(dns.rasterization, g, viewmat.to(DEV), K.to(DEV), DEV)
torch.cuda.synchronize()
for k in c32:
    print(k, "gpu-vs-o32 %.3e  gpu-vs-o64 %.3e  o32-vs-o64 %.3e" % (rel_err(g[k].grad, c32[k].grad), rel_err(g[k].grad, c64[k].grad), rel_err(c32[k].grad, c64[k].grad)))
d = (g["quats"].grad.cpu() - c32["quats"].grad).abs()
idx = d.max(dim=1).values.argmax().item()
print("worst gaussian", idx, "radii", o32[2]["radii"][0, idx].item(), gg[2]["radii"][0, idx].item(), "depth", o32[2]["depths"][0, idx].item(),
      "m2d", o32[2]["means2d"][0, idx].tolist(), "conic", o32[2]["conics"][0, idx].tolist())
for k in c32:
    print(k, "gpu", g[k].grad[idx].flatten()[:6].tolist(), "\n   o32", c32[k].grad[idx].flatten()[:6].tolist(), "\n   o64", c64[k].grad[idx].flatten()[:6].tolist())
print("m2d.grad gpu", gg[2]["means2d"].grad[0, idx].tolist(), "o32", o32[2]["means2d"].grad[0, idx].tolist())
print("scale/quat", inp["scales"][idx].tolist(), inp["quats"][idx].tolist())
# how many gaussians differ materially
big = (d.max(dim=1).values > 1e-3 * c32["quats"].grad.abs().max()).sum().item()
print("n gaussians with quats grad err > 1e-3 scale:", big, "max |grad|", c32["quats"].grad.abs().max().item())
