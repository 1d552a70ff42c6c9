"""Where the (pixel, splat) slots of the fused pass go, from the forward's own keep masks (GPU):
per 16x8 half tile, of the entries the rectangle test keeps — how many are blended by no pixel at all, by pixels of only
one 16x4 quarter, and what share of the kept x 128 slots is a blended pair.

    python tools/pair_structure.py [c2|c3|c1] [n_tiles_sampled]
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
n_sample = int(sys.argv[2]) if len(sys.argv) > 2 else 256
N, W, H, focal = WORKLOADS[wl]
dev = "cuda:0"
gp = synthetic.make_gauss_params(N, sh_rest_std=0.1, seed=0, device=dev)
cam = synthetic.orbit_camera(0, width=W, height=H, focal=focal).to(dev)
m = dns.DNSplatterRenderer(gp, fused=True)
out = m.get_outputs(cam)
info = m.last_info
tw, th = info["tile_width"], info["tile_height"]
T = tw * th
seen, stack, node = set(), [out["accumulation"].grad_fn], None
while stack:
    f = stack.pop()
    if f is None or f in seen:
        continue
    seen.add(f)
    if "_RasterDnFn" in type(f).__name__:
        node = f
        break
    stack.extend(n for n, _ in f.next_functions)
means2d, splats, flatten_ids, tile_offsets, render, alphas, last_ids, bg = node.saved_tensors
masks, stride = node.keep["masks"], node.keep["stride"]
offs = tile_offsets.long()
alphas = alphas[0]
last_ids = last_ids[0].long()

g = torch.Generator(device="cpu").manual_seed(0)
sample = torch.randperm(T, generator=g)[:n_sample].tolist()
acc = dict(entries=0, kept=0, touched=0, one_quarter=0, slots=0, blended=0, q_slots=0, rows_any=0, rows_tot=0, e8_slots=0)
for t in sample:
    s, e = int(offs[t]), int(offs[t + 1])
    if e <= s:
        continue
    ty, tx = divmod(t, tw)
    for part in range(2):
        y0 = ty * 16 + part * 8
        ys = torch.arange(y0, y0 + 8, device=dev)
        xs = torch.arange(tx * 16, tx * 16 + 16, device=dev)
        inside = (ys[:, None] < H) & (xs[None, :] < W)
        yy = ys.clamp(max=H - 1)[:, None].expand(8, 16)
        xx = xs.clamp(max=W - 1)[None, :].expand(8, 16)
        lid = torch.where(inside & (alphas[yy, xx] > 0), last_ids[yy, xx], torch.full_like(yy, -1)).reshape(-1)
        hi = int(lid.max())
        if hi < s:
            continue
        nb = ((hi - s) >> 6) + 1
        mw = masks[part * stride + (s >> 6) + t: part * stride + (s >> 6) + t + nb]
        bits = ((mw[:, None] >> torch.arange(64, device=dev)[None, :]) & 1).bool().reshape(-1)
        idx = s + torch.arange(nb * 64, device=dev)
        bits &= idx <= hi
        kept = idx[bits]
        acc["entries"] += hi - s + 1
        acc["kept"] += int(kept.numel())
        if kept.numel() == 0:
            continue
        rec = splats[flatten_ids[kept].long()]
        px = (xs.float() + 0.5)[None, :].expand(8, 16).reshape(-1)
        py = (ys.float() + 0.5)[:, None].expand(8, 16).reshape(-1)
        dx = rec[:, 0:1] - px[None]
        dy = rec[:, 1:2] - py[None]
        sig = 0.5 * (rec[:, 2:3] * dx * dx + rec[:, 4:5] * dy * dy) + rec[:, 3:4] * dx * dy
        al = rec[:, 5:6] * torch.exp(-sig)
        ok = (sig >= 0) & (al >= 1 / 255) & (kept[:, None] <= lid[None, :])
        anyp = ok.any(dim=1)
        q0 = ok[:, :64].any(dim=1)
        q1 = ok[:, 64:].any(dim=1)
        acc["touched"] += int(anyp.sum())
        acc["one_quarter"] += int((q0 ^ q1).sum())
        acc["slots"] += int(kept.numel()) * 128
        acc["blended"] += int(ok.sum())
        acc["q_slots"] += int(q0.sum() + q1.sum()) * 64
        rows = ok.view(-1, 8, 16).any(dim=2)
        acc["rows_any"] += int(rows.sum())
        acc["rows_tot"] += rows.numel()
        e8 = ok.view(-1, 8, 2, 8).permute(0, 2, 1, 3).reshape(-1, 2, 64).any(dim=2)     # 8x8 quarters
        acc["e8_slots"] += int(e8.sum()) * 64
res = {"workload": wl, "tiles_sampled": len(sample), **acc,
       "kept_of_entries": acc["kept"] / max(acc["entries"], 1),
       "touched_of_kept": acc["touched"] / max(acc["kept"], 1),
       "one_quarter_of_kept": acc["one_quarter"] / max(acc["kept"], 1),
       "blended_of_slots": acc["blended"] / max(acc["slots"], 1),
       "slots_if_16x4_quarters": acc["q_slots"] / max(acc["slots"], 1),
       "slots_if_8x8_quarters": acc["e8_slots"] / max(acc["slots"], 1),
       "slots_if_pixel_rows": acc["rows_any"] * 16 / max(acc["slots"], 1)}
print(json.dumps(res, indent=1))
