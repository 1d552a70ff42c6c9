"""Debug: per-tile start/end clocks of raster_bwd (instrumented build) -> concurrency profile."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dbg = torch.zeros(3 * 8160 + 16, dtype=torch.int64, device="cuda:0")
os.environ["DNS_DBG_PTR"] = str(dbg.data_ptr())
import dn_splatter_amd as dns
from dn_splatter_amd import synthetic
dev = "cuda:0"
gp = synthetic.make_gauss_params(1_000_000, sh_rest_std=0.1, seed=0, device=dev)
cam = synthetic.orbit_camera(0, width=1920, height=1080).to(dev)
m = dns.DNSplatterRenderer(gp, fused=True)
for it in range(3):
    out = m.get_outputs(cam)
    keys = ("rgb", "depth", "normal", "accumulation")
    torch.autograd.backward([out[k] for k in keys], [torch.ones_like(out[k]) for k in keys])
    torch.cuda.synchronize()
d = dbg[: 3 * 8160].view(8160, 3).cpu()
t0 = d[:, 0].min().item()
start = (d[:, 0] - t0).double(); end = (d[:, 1] - t0).double()
npass = (d[:, 2] >> 48) & 0xffff
xcc = (d[:, 2] >> 32) & 0xf
hwid = d[:, 2] & 0xffffffff
total = end.max().item()
dur = end - start
print("wall_clock ticks total", total, "n blocks", len(d))
print("dur mean %.0f p50 %.0f p90 %.0f max %.0f" % (dur.mean(), dur.median(), dur.quantile(0.9), dur.max()))
print("passes mean %.2f max %d" % (npass.double().mean(), npass.max()))
# concurrency over time
import numpy as np
edges = np.linspace(0, total, 41)
s_np, e_np = start.numpy(), end.numpy()
conc = [(np.minimum(e_np, edges[i + 1]) - np.maximum(s_np, edges[i])).clip(min=0).sum() / (edges[i + 1] - edges[i]) for i in range(40)]
print("concurrency (waves) per 2.5% slice:", [int(c) for c in conc])
print("mean concurrency", sum(conc) / 40)
# ticks per pass
pp = (dur / npass.clamp(min=1).double())
print("ticks per pass mean %.0f p10 %.0f p90 %.0f" % (pp.mean(), pp.quantile(0.1), pp.quantile(0.9)))
for x in range(8):
    msk = xcc == x
    print("xcc", x, "blocks", int(msk.sum()), "last end %.0f" % end[msk].max().item(), "sum dur %.0f" % dur[msk].sum().item())
