"""Forward-only frames/s of the offline render loop (N4): sequential get_outputs vs get_outputs_batch (C cameras per launch sequence)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dn_splatter_amd as dns
from dn_splatter_amd import synthetic
dev = "cuda:0"
gp = synthetic.make_gauss_params(1_000_000, sh_rest_std=0.1, seed=0, device=dev)
m = dns.DNSplatterRenderer(gp, fused=True)
cams = [synthetic.orbit_camera(v, width=1920, height=1080).to(dev) for v in range(8)] * 4
dns.set_bin_policy("capacity")
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return len(cams) * reps / (time.perf_counter() - t)
with torch.no_grad():
    print("sequential get_outputs        : %.1f frames/s" % timed(lambda: [m.get_outputs(c) for c in cams]))
for mb in (1, 2, 4, 8):
    print("get_outputs_batch max_batch=%d : %.1f frames/s" % (mb, timed(lambda: m.get_outputs_batch(cams, max_batch=mb))))
