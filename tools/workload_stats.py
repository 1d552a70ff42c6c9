"""Workload statistics of a bench scene on the GPU: list lengths, termination depth, pair counts.

    python tools/workload_stats.py [c2|c3|c1]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import dn_splatter_amd as dns  # noqa: E402
from bench import WORKLOADS  # noqa: E402
from dn_splatter_amd import synthetic  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
N, W, H, focal = WORKLOADS[wl]
dev = "cuda:0"
gp = synthetic.make_gauss_params(N, sh_rest_std=0.1, seed=0, device=dev)
cam = synthetic.orbit_camera(0, width=W, height=H, focal=focal).to(dev)
m = dns.DNSplatterRenderer(gp, fused=True)
out = m.get_outputs(cam)
info = m.last_info
I = info["n_isects"]
tw, th = info["tile_width"], info["tile_height"]
T = tw * th
offs = torch.cat([info["isect_offsets"].reshape(-1).long(), torch.tensor([I], device=dev)])
lens = offs[1:] - offs[:-1]

# find the compositing node of the autograd graph: its saved tensors hold last_ids
seen = set()
stack = [out["accumulation"].grad_fn]
node = None
while stack:
    f = stack.pop()
    if f is None or f in seen:
        continue
    seen.add(f)
    if "_RasterFn" in type(f).__name__:
        node = f
        break
    stack.extend(n for n, _ in f.next_functions)
means2d, splats, flatten_ids, tile_offsets, render, alphas, last_ids = node.saved_tensors
pad_h, pad_w = th * 16, tw * 16
lp = torch.full((pad_h, pad_w), -1, device=dev, dtype=torch.long)
lp[:H, :W] = last_ids.long()
ap = torch.zeros((pad_h, pad_w), device=dev)
ap[:H, :W] = alphas
tiles = lp.view(th, 16, tw, 16).permute(0, 2, 1, 3).reshape(T, 256)
atile = ap.view(th, 16, tw, 16).permute(0, 2, 1, 3).reshape(T, 256)
start = offs[:-1]
depth_px = (tiles - start[:, None] + 1).clamp(min=0)   # list entries a pixel walks in the backward
depth_px = torch.where(atile > 0, depth_px, torch.zeros_like(depth_px))
hi = depth_px.max(dim=1).values
half = depth_px.view(T, 2, 128).max(dim=2).values        # forward waves = 16x8 half tiles
res = {
    "workload": wl, "N": N, "Nv": int((m.radii > 0).sum()), "I": I, "tiles": T,
    "list_len_mean": float(lens.float().mean()), "list_len_max": int(lens.max()), "list_len_p50": float(lens.float().median()),
    "term_depth_px_mean": float(depth_px.float().mean()), "term_depth_tile_max_mean": float(hi.float().mean()),
    "term_depth_tile_max_max": int(hi.max()),
    "pairs_pixel_exact": int(depth_px.sum()), "pairs_tile_max": int((hi * 256).sum()), "pairs_full_lists": int((lens * 256).sum()),
    "pairs_systolic": int((((hi + 63) // 64) * 64 * 319).sum()), "pairs_fwd_halfwave_lower_bound": int((half * 128).sum()),
    "alpha_mean": float(alphas.mean()), "alpha_sat_frac": float((alphas > 0.9998).float().mean()),
}
g = torch.Generator(device="cpu").manual_seed(0)
sample = torch.randperm(T, generator=g)[:128].tolist()
tot = val = wave_any = wave_tot = q_any = q_tot = 0
for t in sample:
    n = int(hi[t])
    if n == 0:
        continue
    ids = flatten_ids[int(start[t]): int(start[t]) + n].long()
    rec = splats[ids]
    ty, tx = divmod(t, tw)
    px = (torch.arange(16, device=dev) + tx * 16 + 0.5)[None, :].expand(16, 16).reshape(-1)
    py = (torch.arange(16, device=dev) + ty * 16 + 0.5)[:, None].expand(16, 16).reshape(-1)
    dx = rec[:, 0:1] - px[None]
    dy = rec[:, 1:2] - py[None]
    sig = 0.5 * (rec[:, 2:3] * dx * dx + rec[:, 4:5] * dy * dy) + rec[:, 3:4] * dx * dy
    al = torch.clamp(rec[:, 5:6] * torch.exp(-sig), max=0.999)
    ok = (sig >= 0) & (al >= 1 / 255) & (torch.arange(n, device=dev)[:, None] < depth_px[t][None, :])
    tot += ok.numel()
    val += int(ok.sum())
    wave_any += int(ok.view(n, 4, 64).any(dim=2).sum())
    wave_tot += n * 4
    q_any += int(ok.any(dim=1).sum())
    q_tot += n
res["valid_pair_frac_of_tile_max"] = val / max(tot, 1)
res["wave64rows_any_valid_frac"] = wave_any / max(wave_tot, 1)
res["splat_any_valid_in_tile_frac"] = q_any / max(q_tot, 1)
print(json.dumps(res, indent=1))
