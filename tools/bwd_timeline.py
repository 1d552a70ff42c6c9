"""Per-work-unit timeline of raster_bwd on the benchmark frame (instrumented build, see tools/bwd_timeline.sh):
waves in flight over time, unit duration against its start time and its list depth.

What it showed at the end of round 1: with whole-tile units the chip stayed full for 1.4 ms and then drained for 0.66 ms at
39 % occupancy (units take ~570 us and do not speed up much while their SIMDs empty), 20 % of the launch — which is why a wave
now owns half a tile.  Read it together with the PMC pass (SIMDs 95 % busy with vector instructions while the chip is full)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import dn_splatter_amd as dns  # noqa: E402
from dn_splatter_amd import _lib, synthetic  # noqa: E402

dev = "cuda:0"
N, W, H, focal = 1_000_000, 1920, 1080, 1200.0
gp = synthetic.make_gauss_params(N, sh_rest_std=0.1, seed=0, device=dev)
cam = synthetic.orbit_camera(0, n_views=8, width=W, height=H, focal=focal).to(dev)
r = dns.DNSplatterRenderer(gp, fused=True)
units = ((W + 15) // 16) * ((H + 15) // 16) * int(os.environ.get("PARTS", "2"))
dbg = torch.zeros(4 * 2 * units,   # a tapered launch has more blocks than units
                   dtype=torch.int64, device=dev)
L = _lib.lib()
orig = L.dnsplat_raster_bwd


def wrapped(argp, stream):
    argp._obj.v_alphas = dbg.data_ptr()
    return orig(argp, stream)


gen = torch.Generator(device=dev).manual_seed(1)
keys = ("rgb", "depth", "normal", "accumulation")
for it in range(3):
    for k in gp:
        gp[k].grad = None
    out = r.get_outputs(cam)
    cot = [torch.rand(out[k].shape, device=dev, generator=gen) * 2 - 1 for k in keys]
    if it == 2:
        L.dnsplat_raster_bwd = wrapped
    torch.autograd.backward([out[k] for k in keys], cot)
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(2 * units, 4)
ok = d[:, 1] > 0
t0, t1 = d[ok, 0].astype(np.float64), d[ok, 1].astype(np.float64)
depth, steps, splats = [((d[ok, 2] >> sh) & 0xFFFFF).astype(np.float64) for sh in (0, 20, 40)]
base = t0.min()
t0, t1 = (t0 - base) / 100.0, (t1 - base) / 100.0          # 100 MHz ticks -> us
dur = t1 - t0
print(f"units {ok.sum()}  kernel span {t1.max():.0f} us  unit duration mean {dur.mean():.0f} p10 {np.quantile(dur, 0.1):.0f} "
      f"p90 {np.quantile(dur, 0.9):.0f} max {dur.max():.0f} us  perfectly packed: {dur.sum() / 2816:.0f}+ us")
edges = np.linspace(0, t1.max(), 21)
for a, b in zip(edges[:-1], edges[1:]):
    inflight = np.clip(np.minimum(t1, b) - np.maximum(t0, a), 0, None).sum() / (b - a)
    m = (t0 >= a) & (t0 < b)
    print(f"{a:7.0f}-{b:7.0f} us: {inflight:6.0f} in flight, {m.sum():5d} started" + (f", their mean duration {dur[m].mean():.0f} us" if m.any() else ""))
print("corr(duration, list depth) =", round(float(np.corrcoef(dur, depth)[0, 1]), 3),
      " corr(duration, steps streamed) =", round(float(np.corrcoef(dur, steps)[0, 1]), 3),
      " corr(duration, splats in buckets) =", round(float(np.corrcoef(dur, splats)[0, 1]), 3))
full = t0 < 0.7 * t1.max()                     # units that ran while the chip was full
fit = np.polyfit(steps[full], dur[full], 1)
res = dur[full] - np.polyval(fit, steps[full])
print(f"while the chip is full: duration = {fit[0]:.3f} us/step x steps + {fit[1]:.0f} us, residual std {res.std():.1f} us "
      f"({100 * res.std() / dur[full].mean():.1f} % of the mean); steps mean {steps.mean():.0f} p10 {np.quantile(steps, 0.1):.0f} p90 {np.quantile(steps, 0.9):.0f}")
print(f"list depth mean {depth.mean():.0f}, splats in buckets mean {splats.mean():.0f}")
hw, xcc = d[ok, 3] & 0xFFFFFFFF, (d[ok, 3] >> 32) & 0xF
simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
cu_key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
simd_key = cu_key * 4 + simd
for name, key in (("XCC", xcc), ("CU", cu_key), ("SIMD", simd_key)):
    ks = np.unique(key)
    cnt = np.array([(key == k).sum() for k in ks]); mean = np.array([dur[key == k].mean() for k in ks])
    busy = np.array([dur[key == k].sum() for k in ks])
    print(f"per {name}: {len(ks)} distinct, units each mean {cnt.mean():.1f} min {cnt.min()} max {cnt.max()}; mean unit duration per {name}: "
          f"min {mean.min():.0f} max {mean.max():.0f} std {mean.std():.1f} us; sum of durations min {busy.min():.0f} max {busy.max():.0f}")
    if name == "XCC":
        print("   per XCC mean duration:", [round(float(m)) for m in mean], "units:", cnt.tolist())
np.save(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "bwd_timeline.npy"), np.stack([t0, t1, depth, steps, splats, simd_key.astype(np.float64)]))
