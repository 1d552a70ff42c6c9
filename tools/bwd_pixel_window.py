"""CPU estimate (oracle forward, no GPU) of what a per-bucket PIXEL WINDOW would save in the systolic compositing backward.

Today a bucket of 128 splats streams all 128 pixels of its half tile (143 steps).  A pixel whose last blended list index lies
below the bucket's lowest index meets no splat of the bucket.  With the unit's pixel rows stored in order of descending last index,
the pixels a bucket has to stream are a PREFIX of that order: steps(bucket) = n_active + 15 instead of 143.

    python tools/bwd_pixel_window.py [c2|c1] [kept_fraction]

Prints slots issued today / with the window (relative), for exact last ids and for last ids rounded up to the end of the 64-entry
batch (what the HIP forward stores for pixels that never saturate).
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import dn_splatter_amd as dns  # noqa: E402
from bench import WORKLOADS  # noqa: E402
from dn_splatter_amd import synthetic  # noqa: E402
from oracle import oracle as orc  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
kappa = float(sys.argv[2]) if len(sys.argv) > 2 else 0.767
N, W, H, focal = WORKLOADS[wl]
torch.set_num_threads(os.cpu_count())
gp = synthetic.make_gauss_params(N, sh_rest_std=0.1, seed=0)
cam = synthetic.orbit_camera(0, width=W, height=H, focal=focal)
t0 = time.time()
q = gp["quats"].detach()
viewmat = dns.get_viewmat(cam.camera_to_worlds)
K = cam.get_intrinsics_matrices()
means, quats, scales = gp["means"].detach(), q / q.norm(dim=-1, keepdim=True), torch.exp(gp["scales"].detach())
opac = torch.sigmoid(gp["opacities"].detach()).squeeze(-1)
radii, means2d, depths, conics, comp, tiles = orc.project_fwd(means, quats, scales, viewmat[0], K[0], W, H, 0.3, 0.01, 1e10, 0.0, 16, False)
tw, th = (W + 15) // 16, (H + 15) // 16
_t, isect_ids, flatten_ids = orc.isect_tiles(means2d, radii, depths, 16, tw, th)
offsets = orc.isect_offset_encode(isect_ids, tw, th).reshape(-1).long()
print(f"projected + binned: {flatten_ids.numel()} pairs, {time.time() - t0:.1f} s", flush=True)
# tight tile boxes: drop the pairs outside (what the fused path bins)
x0, y0, x1, y1 = orc.tight_tile_boxes(means2d, conics, opac, radii, 16, tw, th)
T = tw * th
offs_all = torch.cat([offsets, torch.tensor([flatten_ids.numel()])])
tile_of = torch.searchsorted(offs_all, torch.arange(flatten_ids.numel()), right=True) - 1
g = flatten_ids.long()
tx, ty = tile_of % tw, tile_of // tw
keep = (tx >= x0[g]) & (tx < x1[g]) & (ty >= y0[g]) & (ty < y1[g])
fid = flatten_ids[keep].contiguous()
tile_k = tile_of[keep]
offs_k = torch.searchsorted(tile_k, torch.arange(T + 1)).int()
print(f"tight lists: {fid.numel()} pairs", flush=True)
cols = torch.rand(N, 1)
render, alphas, last_ids = orc.rasterize_fwd(means2d, conics, cols, opac, None, W, H, 16, offs_k[:-1].contiguous(), fid)
print(f"forward done {time.time() - t0:.1f} s", flush=True)
last = last_ids.long().numpy()
al = alphas.numpy()
offs = offs_k.long().numpy()
span = 128.0 / kappa            # list entries one bucket of 128 kept splats spans

res = {}
for label in ("exact", "batch64"):
    old = new = useful_px = 0
    nb_tot = 0
    hist = np.zeros(129, dtype=np.int64)
    for t in range(T):
        s, e = offs[t], offs[t + 1]
        if e <= s:
            continue
        tyy, txx = divmod(t, tw)
        for part in range(2):
            ya, xa = tyy * 16 + part * 8, txx * 16
            l = last[ya:ya + 8, xa:xa + 16].reshape(-1).copy()
            if l.size == 0:
                continue
            a = al[ya:ya + 8, xa:xa + 16].reshape(-1)
            l[a <= 0] = -1
            if label == "batch64":
                # an open pixel stores the end of the 64-entry batch (aligned to the list start) its last blended entry sits in
                opn = (a > 0) & (a < 1 - 1e-4 * 1.0001)
                l = np.where(opn, np.minimum(s + ((l - s) // 64 + 1) * 64 - 1, e - 1), l)
            hi = l.max()
            if hi < s:
                continue
            depth = hi - s + 1
            nb = int(np.ceil(depth / span))
            ls = np.sort(l)[::-1]
            for b in range(nb):
                lo_b = hi - (b + 1) * span + 1
                n_b = int((ls >= lo_b).sum())
                hist[n_b] += 1
                old += 143
                new += n_b + 15
            nb_tot += nb
    res[label] = (old, new, nb_tot, hist)
    print(f"{label}: buckets {nb_tot}, steps today {old}, with pixel window {new} ({new / old:.3f}); "
          f"buckets by active pixels: <=32: {hist[:33].sum()}, <=64: {hist[:65].sum()}, <=96: {hist[:97].sum()}, <128: {hist[:128].sum()}, =128: {hist[128]}",
          flush=True)
tile_len = (offs[1:] - offs[:-1])
pix_t = (np.arange(H)[:, None] // 16) * tw + (np.arange(W)[None, :] // 16)
rel = (last - offs[pix_t]) / np.maximum(tile_len[pix_t], 1)
sat = al >= 1 - 1.0001e-4
print("saturated pixels:", sat.mean(), "mean alpha", al.mean(), "rel last id: mean", rel.mean(), "p10/p50/p90", np.percentile(rel, [10, 50, 90]))
print("rel last of saturated:", rel[sat].mean() if sat.any() else None, " of open:", rel[~sat].mean())
