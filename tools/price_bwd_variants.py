"""Slot counts of the compositing backward's systolic stream under three re-cuts of its work, priced BEFORE writing any ISA
(VERDICT r04 item 5).  CPU only: the C2 benchmark scene is projected and binned by the CPU restatement (analysis tooling, like
tools/parity_seed_sweep.py — nothing here is product code), a sample of tiles is composited in numpy with the rule set of SURVEY.md
A.6, and the kernel's own accounting (raster_bwd.hip: unit = 16x8 half tile, bucket = 128 kept splats = 64 lanes x 2, PERIOD = pixels +
15 steps per bucket, the unit's last bucket folded: <= 32 splats 79 steps, <= 64 splats 111 steps) is applied to

  current   what the kernel issues today (reproduces bench.py's bwd_slots_issued / bwd_pairs_replayed: useful_pair_fraction 0.60-0.62)
  (a)       16x4 strips as units (64 pixels, PERIOD 79; last bucket 47 / 71 / 79 steps), each strip with its own keep set and its own
            highest list index: fewer slots per kept splat, twice the units, and one atomic row per (strip, splat)
  (b)       no quantisation at all: every bucket of every unit as dense as a full one (143 steps per 128 splats pro rata) — the bound
            on ANY re-ordering of a unit's splats over buckets / folds
  (c)       every bucket streams only the pixels whose last blended index reaches into it (pixels sorted by last index, a prefix
            per bucket): bucket steps = n_pixels + 15

    python tools/price_bwd_variants.py [n_tiles_sampled] [gsplat|tight]
"""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dn_splatter_amd as dns  # noqa: E402
from dn_splatter_amd import synthetic  # noqa: E402
from oracle import oracle as orc  # noqa: E402

n_sample = int(sys.argv[1]) if len(sys.argv) > 1 else 200
boxes = sys.argv[2] if len(sys.argv) > 2 else "tight"
N, W, H, focal = 1_000_000, 1920, 1080, 1200.0
gp = synthetic.make_gauss_params(N, sh_rest_std=0.1, seed=0)
cam = synthetic.orbit_camera(0, width=W, height=H, focal=focal)
viewmat, K = dns.get_viewmat(cam.camera_to_worlds), cam.get_intrinsics_matrices()
q = gp["quats"].detach()
radii, xys, depths, conics, _c, _t = orc.project_fwd(gp["means"].detach(), q / q.norm(dim=-1, keepdim=True), torch.exp(gp["scales"].detach()),
                                                     viewmat[0], K[0], W, H)
opac = torch.sigmoid(gp["opacities"].detach()).reshape(-1)
tw, th = (W + 15) // 16, (H + 15) // 16
_t, ids, fid = orc.isect_tiles(xys, radii, depths, 16, tw, th)
offs = orc.isect_offset_encode(ids, tw, th).reshape(-1).numpy()
fid, xy, con, op = fid.numpy(), xys.numpy(), conics.numpy(), opac.numpy()
T = tw * th
sample = np.random.default_rng(0).choice(T, n_sample, replace=False)


def last_steps(take, npix):
    """steps of a unit's last bucket (raster_bwd.hip: fold 4 / fold 2 / plain), npix pixels per unit"""
    if take <= 32:
        return 15 + 64 * npix // 128
    if take <= 64:
        return 31 + 80 * npix // 128
    return npix + 15


def unit_slots(blended, valid, npix_unit):
    """blended / valid: [pixels of the unit, list entries] -> (kept splats, useful pairs, slots issued, slots variant c)"""
    last = np.where(blended.any(1), blended.shape[1] - 1 - blended[:, ::-1].argmax(1), -1)
    hi = last.max()
    if hi < 0:
        return 0, 0, 0, 0
    kept = np.where(valid.any(0) & (np.arange(blended.shape[1]) <= hi))[0]     # proxy for the forward's rectangle test (99 % of its
    kc = len(kept)                                                             # keeps blend into some pixel: tools/pair_structure.py)
    if kc == 0:
        return 0, 0, 0, 0
    kb = kept[::-1]
    nb = math.ceil(kc / 128)
    cur = c = 0
    for b in range(nb):
        chunk = kb[b * 128:(b + 1) * 128]
        n_b = int((last >= chunk.min()).sum())
        st = (npix_unit + 15) if len(chunk) == 128 else last_steps(len(chunk), npix_unit)
        cur += st
        c += min(st, max(n_b, 1) + 15 + (0 if len(chunk) == 128 else 48))
    return kc, int(blended[:, kept].sum()), cur * 128, c * 128


acc = {k: 0 for k in ("units", "kept", "useful", "cur", "c", "a_units", "a_kept", "a_slots")}
for t in sample:
    s, e = offs[t], (offs[t + 1] if t + 1 < T else len(fid))
    if e <= s:
        continue
    g = fid[s:e]
    ty, tx = divmod(int(t), tw)
    ys, xs = np.arange(ty * 16, ty * 16 + 16) + 0.5, np.arange(tx * 16, tx * 16 + 16) + 0.5
    py, px = (a.reshape(-1) for a in np.meshgrid(ys, xs, indexing="ij"))
    ok = (py < H) & (px < W)
    dx, dy = xy[g, 0][None, :] - px[:, None], xy[g, 1][None, :] - py[:, None]
    sig = 0.5 * (con[g, 0] * dx * dx + con[g, 2] * dy * dy) + con[g, 1] * dx * dy
    al = np.minimum(0.999, op[g][None, :] * np.exp(-sig))
    valid = (sig >= 0) & (al >= 1 / 255) & ok[:, None]
    a = np.where(valid, al, 0.0)
    stop = np.cumprod(1 - a, axis=1) <= 1e-4
    first_stop = np.where(stop.any(1), stop.argmax(1), a.shape[1])
    blended = valid & (np.arange(a.shape[1])[None, :] < first_stop[:, None])
    for part in range(2):                       # today's units: 16x8 half tiles
        rows = slice(part * 128, part * 128 + 128)
        kc, useful, cur, c = unit_slots(blended[rows], valid[rows], 128)
        acc["units"] += kc > 0; acc["kept"] += kc; acc["useful"] += useful; acc["cur"] += cur; acc["c"] += c
    for part in range(4):                       # (a): 16x4 strips
        rows = slice(part * 64, part * 64 + 64)
        kc, _u, cur, _c = unit_slots(blended[rows], valid[rows], 64)
        acc["a_units"] += kc > 0; acc["a_kept"] += kc; acc["a_slots"] += cur

cur = acc["cur"]
dense = acc["kept"] * 143.0                     # (b): 143 steps x 128 slots per 128 splats, pro rata
print(f"C2 scene, {n_sample} of {T} tiles sampled, gsplat tile boxes for the lists (the kept set does not depend on the boxes)")
print(f"half-tile units {acc['units']}, kept splats per unit {acc['kept'] / acc['units']:.1f}, blended pairs {acc['useful']}")
print(f"{'variant':44s} {'slots':>12s} {'vs current':>10s} {'useful fraction':>16s} {'units':>8s} {'atomic rows':>12s}")
rows = [("current (16x8 units, last bucket folded)", cur, acc["units"], acc["kept"]),
        ("(a) 16x4 strips as units", acc["a_slots"], acc["a_units"], acc["a_kept"]),
        ("(b) no bucket quantisation (bound)", dense, acc["units"], acc["kept"]),
        ("(c) buckets stream only the pixels reaching them", acc["c"], acc["units"], acc["kept"])]
for name, slots, units, rows_ in rows:
    print(f"{name:44s} {slots:12.0f} {slots / cur:10.3f} {acc['useful'] / slots:16.3f} {units:8d} {rows_:12d}")
