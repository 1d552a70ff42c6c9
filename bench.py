#!/usr/bin/env python
"""bench.py — fwd+bwd frames/s of the dn-splatter rendering hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c1] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one camera per GPU: ``get_outputs`` of the host mirror of
``DNSplatterModel.get_outputs`` (dn_splatter/dn_model.py:404-612 — activations, projection, SH, per-Gaussian
normals, binning, fused colour+depth+normal compositing, the reference's torch post-ops) followed by the full
backward from dense random cotangents on rgb/depth/normal/accumulation to the six optimised tensors, and, for
N > 1, the RCCL all-reduce (mean) of those gradients — 236 B per Gaussian in one flat bucket.  Inputs are
synthetic random-init Gaussians (the reference's own initialisation) already resident in HBM.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      — the dominant stage's algorithmic HBM bytes per launch / its mean HIP-event duration in the
                  timed region, against the 8 TB/s HBM3E peak (guides/MI355X_MICROARCH.md).  Only that stage is
                  bracketed with events inside the timed region (an event pair costs a ~6 us bubble on the stream);
                  the other entries of `stages` come from PROBE_STEPS fully instrumented steps run between the
                  warm-up and the timed region;
  cpu_baseline  — the CPU oracle (a port: the reference has no CPU rasterizer and gsplat's needs CUDA) timed on
                  the host cores on a bounded crop of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, guides/MI355X_MICROARCH.md "Chip-level parameters"

WORKLOADS = {
    # name: (N Gaussians, width, height, focal)  — BASELINE.json configs[0..2]; focal is builder-chosen
    "c1": (10_000, 256, 256, 160.0),
    "c2": (1_000_000, 1920, 1080, 1200.0),
    "c3": (3_000_000, 1600, 1200, 1200.0),
    "c5": (5_000_000, 1600, 1200, 1200.0),   # configs[4]: per GPU of the 8-GPU run, normally with --losses
}

OUT_KEYS = ("rgb", "depth", "normal", "accumulation")
STAGE_NAMES = ("dnsplat_project_fwd", "binning", "dnsplat_raster_fwd", "dnsplat_raster_bwd", "dnsplat_project_bwd")
PROBE_STEPS = 3
SH_K = 16
D_CH = 7


def stage_bytes(N, Nv, I, P, T):
    """Algorithmic HBM bytes per launch of each stage (SURVEY.md §8(d) terms re-grouped by our five stages;
    they sum to B = 84 N + 796 Nv + 216 I + 76 P + 12 T).  DESIGN.md "Roofline accounting" derives each."""
    return {
        "dnsplat_project_fwd": 76 * N + 228 * Nv,
        "binning": 8 * N + 20 * Nv + 44 * I + 12 * T,
        "dnsplat_raster_fwd": 56 * I + 36 * P,
        "dnsplat_raster_bwd": 116 * I + 40 * P,
        "dnsplat_project_bwd": 548 * Nv,
    }


def cpu_baseline(workload, crop=896):
    """Times the CPU oracle — the reference's own two-call sequence (rasterization + legacy
    rasterize_gaussians, dn_model.py:495-575) through the same host mirror — on a centre crop of the same
    scene and scales by the pixel ratio.  Test infrastructure used as the checker/baseline only."""
    import dn_splatter_amd as dns
    from dn_splatter_amd import synthetic
    from dn_splatter_amd.model import Camera
    from oracle import oracle as orc

    N, W, H, focal = WORKLOADS[workload]
    # the crop has a few thousand tiles (the oracle parallelises over tiles) and its gradient scatter uses omp atomics:
    # beyond ~32 threads it only gets slower (measured: 49 s on 256 threads vs 14 s on 8), so cap the team there
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)     # torch and the oracle share the process' OpenMP runtime (libgomp)
    gp = synthetic.make_gauss_params(N, sh_rest_std=0.1, seed=0)
    cam = synthetic.orbit_camera(0, width=W, height=H, focal=focal)
    cw, ch = min(crop, W), min(crop, H)
    x0, y0 = (W - cw) // 2, (H - ch) // 2
    ccam = Camera(cam.camera_to_worlds, cam.fx, cam.fy, cam.cx - x0, cam.cy - y0, cw, ch)
    params = {k: v.detach().requires_grad_(k != "normals") for k, v in gp.items()}
    m = dns.DNSplatterRenderer(params, fused=False, rasterization_fn=orc.rasterization,
                               rasterize_gaussians_fn=orc.rasterize_gaussians)
    gen = torch.Generator().manual_seed(1)
    t0 = time.perf_counter()
    out = m.get_outputs(ccam)
    loss = sum((out[k] * (torch.rand(out[k].shape, generator=gen) * 2 - 1)).sum() for k in OUT_KEYS)
    loss.backward()
    dt = time.perf_counter() - t0
    scale = (W * H) / float(cw * ch)
    return {
        "value": 1.0 / (dt * scale),
        "unit": "frames/s",
        "cores": cores,
        "kind": "port",
        "sample": (f"oracle fwd+bwd of a {cw}x{ch} centre crop of the {workload} scene (all {N} Gaussians projected, "
                   f"{int(m.last_info['flatten_ids'].shape[0])} intersections) took {dt:.2f} s on {cores} threads; "
                   f"value = 1/(t x {scale:.1f} pixel ratio)"),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--bin-policy", default="capacity", choices=["sync", "capacity"])
    ap.add_argument("--dense-allreduce", action="store_true",
                    help="all-reduce the full 236 B/Gaussian bucket instead of exchanging the SH gradients as factors")
    ap.add_argument("--two-call", action="store_true", help="reference's two-pass sequence instead of the fused pass")
    ap.add_argument("--torch-postops", action="store_true", help="keep dn_model.py:526-603 in torch instead of the HIP epilogue")
    ap.add_argument("--losses", nargs="?", const="torch", default=None, choices=["torch", "fused"],
                    help="time dn-splatter's loss stack (L1+SSIM, EdgeAwareLogL1 depth, normal L1+TV, scale) instead of feeding "
                         "random cotangents (BASELINE config C5): 'torch' = as the reference does, 'fused' = dnsplat_dn_loss")
    args = ap.parse_args()

    import dn_splatter_amd as dns
    from dn_splatter_amd import _lib, dp, synthetic

    rank, world, local, dev = dp.init_from_env("cuda")
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)
    N, W, H, focal = WORKLOADS[args.workload]
    P = W * H
    T = ((W + 15) // 16) * ((H + 15) // 16)

    # identical parameters on every rank (seed 0), one camera per rank (8-view orbit)
    gp = synthetic.make_gauss_params(N, sh_rest_std=0.1, seed=0, device=dev)
    cam = synthetic.orbit_camera(rank % 8, n_views=8, width=W, height=H, focal=focal).to(dev)
    renderer = dns.DNSplatterRenderer(gp, fused=not args.two_call, fused_postops=not args.torch_postops)
    dns.set_bin_policy(args.bin_policy)
    arena = dp.GradArena(gp)
    dns.set_grad_arena(arena)
    exchange = None
    if (world > 1 or os.environ.get("DNSPLAT_FORCE_DIST", "0") == "1") and not args.dense_allreduce and not args.two_call:
        exchange = dp.ShFactorExchange()
        dns.set_sh_exchange(exchange)
    gen = torch.Generator(device=dev).manual_seed(1 + rank)
    shapes = {"rgb": (H, W, 3), "depth": (H, W, 1), "normal": (H, W, 3), "accumulation": (H, W, 1)}
    cot = {k: torch.rand(shapes[k], device=dev, generator=gen) * 2 - 1 for k in OUT_KEYS}

    batch = None
    if args.losses:
        from dn_splatter_amd import torch_losses
        batch = torch_losses.synthetic_batch(W, H, dev, seed=rank)
        if args.losses == "fused":
            from dn_splatter_amd import fused_loss
            loss_counts = fused_loss.depth_counts(batch["mono_depth"])

    def step():
        for k in dp.GRAD_KEYS:
            gp[k].grad = None
        out = renderer.get_outputs(cam)
        if batch is not None and args.losses == "fused":
            fused_loss.dn_loss_fused(out, batch, gp["scales"], counts=loss_counts).backward()
        elif batch is not None:
            torch_losses.dn_loss(out, batch, gp["scales"]).backward()
        else:
            # the losses stay in PyTorch (north star); their result is a dense cotangent per output image
            torch.autograd.backward([out[k] for k in OUT_KEYS], [cot[k] for k in OUT_KEYS])
        return dp.allreduce_gradients(gp, arena, exchange=exchange)

    for _ in range(args.warmup):
        step()
    # Stage breakdown: PROBE_STEPS fully instrumented steps OUTSIDE the timed region.  Bracketing all six stages with
    # HIP events costs ~80 us of stream time per frame (a ~6 us bubble per event pair, seen in the rocprofv3 timeline),
    # so the timed region below only brackets the dominant stage, whose live duration the roofline figure needs.
    torch.cuda.synchronize()
    probe = _lib.StageTimer()
    _lib.TIMER = probe
    for _ in range(PROBE_STEPS):
        step()
    torch.cuda.synchronize()
    _lib.TIMER = None
    pstats = probe.summary()

    def stage_ms(st, name, steps):
        if name == "binning":
            return sum(st[k][2] for k in ("dnsplat_bin_prepare", "dnsplat_bin_emit_sort") if k in st) / max(steps, 1)
        return st[name][1] if name in st else None

    probe_ms = {name: stage_ms(pstats, name, PROBE_STEPS) for name in STAGE_NAMES}
    dominant = max((n for n in probe_ms if probe_ms[n] is not None), key=lambda n: probe_ms[n])
    live = ("dnsplat_bin_prepare", "dnsplat_bin_emit_sort") if dominant == "binning" else (dominant,)
    dp.barrier()
    torch.cuda.synchronize()
    timer = _lib.StageTimer(only=live)
    _lib.TIMER = timer
    t0 = time.perf_counter()
    wire = 0
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    marks[0].record()
    for i in range(args.steps):
        wire = step()
        marks[i + 1].record()
    torch.cuda.synchronize()
    dp.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    _lib.TIMER = None
    elapsed = dp.max_over_ranks(elapsed, dev)

    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    q = lambda f: round(per_step[min(len(per_step) - 1, int(f * len(per_step)))], 4)   # noqa: E731
    info = renderer.last_info
    I = int(info["n_isects"])
    Nv = int((renderer.radii > 0).sum())
    stats = timer.summary()
    stages = {}
    sb = stage_bytes(N, Nv, I, P, T)
    for name, b in sb.items():
        ms = stage_ms(stats, name, args.steps) if name == dominant else probe_ms.get(name)
        if ms is None:
            continue
        stages[name] = {"ms": round(ms, 4), "alg_bytes": b, "GBps": round(b / (ms * 1e-3) / 1e9, 1) if ms > 0 else None,
                        "measured": "timed region" if name == dominant else f"{PROBE_STEPS} instrumented steps before it"}
    pmc_traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path):
        try:
            pmc = json.load(open(pmc_path))
            ent = pmc.get(args.workload, {}).get(dominant)
            if ent:
                pmc_traffic = ent["hbm_bytes_per_launch"]
        except Exception:
            pmc_traffic = None
    ach = stages[dominant]["GBps"]
    roofline = {"bound": "hbm", "kernel": dominant, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": pmc_traffic,
                "alg_bytes_per_launch": sb[dominant], "ms_per_launch": stages[dominant]["ms"]}
    B = sum(sb.values())
    fps_total = world * args.steps / elapsed
    frame_roofline = {"alg_bytes_per_frame": B, "achieved_GBps": round(B * (args.steps / elapsed) / 1e9, 1),
                      "frac": round(B * (args.steps / elapsed) / 1e9 / HBM_PEAK_GBS, 4)}
    gpu_stage_ms = sum(v["ms"] for v in stages.values())
    i_all = dp.sum_over_ranks([float(I)], dev)[0]

    if rank == 0:
        res = {
            "metric": "fwd+bwd frames/sec @1M Gaussians 1080p; HBM GB/s vs roofline",
            "value": round(fps_total, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "gpu_ms_per_step_p10_p50_p90": [q(0.1), q(0.5), q(0.9)], "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {N} random-init Gaussians, 1 camera/GPU {W}x{H}, SH degree 3 + "
                                   f"expected depth + per-Gaussian normals ({D_CH} channels, "
                                   f"{'two-call' if args.two_call else 'fused one-pass'}, post-ops in {'torch' if (args.torch_postops or args.two_call) else 'HIP'}), fx=fy={focal}, orbit r=8, "
                                   f"closed-form 3-NN scale init, {('dn-splatter loss stack (' + args.losses + ')') if args.losses else 'random dense cotangents'}",
                       "N": N, "Nv": Nv, "n_isects": I, "mean_isects_per_rank": i_all / world,
                       "mean_tile_list_len": round(I / T, 1), "pixels": P, "bin_policy": args.bin_policy,
                       "allreduce_bytes_per_step": wire,
                       "grads_in_flat_bucket": bool(all(arena.holds(gp[k].grad) for k in dp.GRAD_KEYS)),
                       "parallelism": (f"dp{world} (camera per GPU; per step {wire} B exchanged per GPU over RCCL: "
                                       + ("geometry grads all-reduced, SH grads all-gathered as factors)" if exchange is not None
                                          else "one all-reduce of the flat gradient bucket)")) if world > 1 else "single GPU"},
            "roofline": roofline,
            "frame_roofline": frame_roofline,
            "stages": stages,
            "other_ms_torch_postops_autograd_host": round(1e3 * elapsed / args.steps - gpu_stage_ms, 4),
        }
        if world == 1 and not args.no_cpu_baseline:
            dns.set_grad_arena(None)
            dns.set_sh_exchange(None)
            try:
                res["cpu_baseline"] = cpu_baseline(args.workload)
            except Exception as e:  # the baseline is reported, never required for the GPU number
                res["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"failed: {e!r}"}
    else:
        res = None
    dp.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    if res is not None:
        # The JSON line must be the last thing on stdout: RCCL writes its version banner through C stdio, which a pipe
        # only sees when that buffer is flushed — after Python's own prints unless it is flushed first.
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
