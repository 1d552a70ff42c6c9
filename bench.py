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
  roofline_valu — the roofline that actually binds the two compositing kernels: (pixel, splat) pairs counted by the
                  measurement instantiation of the kernels in ONE extra step outside the timed region, x flops per pair
                  (counted from the kernel source), / the stage times, against 157.3 TFLOP/s fp32 vector;
  cpu_baseline  — the CPU oracle (a port: the reference has no CPU rasterizer and gsplat's needs CUDA) timed on
                  the host cores on one full frame of the same workload (a centre crop with --cpu-crop).
For N > 1 also `multi_gpu`: exchange time exposed / hidden, bytes per GPU and per xGMI link, max / mean intersections per rank.
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# multi-process GPU work on this pool: the host driver only supports dmabuf IPC (without it RCCL fails with
# `hipIpcGetMemHandle: invalid argument`).  Already exported on the boxes; a default here covers a bare shell.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import faulthandler  # noqa: E402

import torch  # noqa: E402

faulthandler.enable()    # a crash inside a native call leaves the Python stack on stderr instead of a bare exit code

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, guides/MI355X_MICROARCH.md "Chip-level parameters"
VALU_PEAK_TFLOPS = 157.3  # fp32 vector peak (256 CUs x 4 SIMD32 x 2 flops x 2.4 GHz x packed 2), cdna_hip_programming.md
# flops per (pixel, splat) pair, counted from the kernel sources for the fused 7-channel pass (FMA = 2):
#   forward  dx, dy (2) + exponent (u, w: 2 FMA; dy*w; FMA = 7) + exp2, x opacity, min (3) + T(1-alpha) (2) + alpha T (1) + 7 channel FMAs (14)
#   backward exponent (9) + exp2, x opacity (2) + 1-alpha, rcp, T chain (3) + alpha T (1) + 14 channel FMAs (28) + v_alpha of the two
#            groups (10) + v_sigma (2) + conic sums (2 mul + 3 FMA = 8) + mean sums (2 mul + 2 add + 2 |.| add = 6) + opacity sum (2) + S chains (4)
FLOPS_PER_PAIR = {"dnsplat_raster_fwd": 29, "dnsplat_raster_bwd": 75}

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
# team size of the cpu_baseline leg, from profiles/r05b_cpu_baseline_scaling.txt: one C2 frame takes 5.4 / 5.6 / 6.1 / 9.0 / 60 s on
# 16 / 32 / 64 / 128 / 256 threads of the GPU box's host (the tile-parallel compositing no longer limits it: what is left is the
# single-threaded 64-bit sort of the 21 M pairs and torch's own ops; 256 threads oversubscribe the cores torch's pool also uses)
CPU_BASELINE_THREADS = 32
# a side section (another workload / the train loop, in a process of its own) normally takes 5-10 s on the GPU box, a fresh box's
# first torch import up to two minutes: a child that exceeds this is reported as an error in its entry, the headline line is not held up
CHILD_TIMEOUT_S = 240
FIRST_TOUCH_STEPS = 2
# The chip clocks down within milliseconds of idling and needs ~30 ms of load to come back (first steps after an idle gap:
# 2.9, 3.06, 2.84, 2.79 ... ms against 2.53 ms from the twelfth on).  Everything untimed that makes the GPU wait for the host
# (instrumented probe steps, the graph capture, gc.collect) therefore comes BEFORE this many untimed pre-roll steps, which are
# followed without a gap by the W warm-up steps, the barrier + synchronize and the timed region.
PREROLL_STEPS = 12
SH_K = 16
D_CH = 7


def _strip_comments(src: str) -> str:
    """C / C++ source without comments and without blank space at line ends: what the compiler sees."""
    import re

    def keep_strings(m):
        t = m.group(0)
        return t if t[0] in "\"'" else " "
    src = re.sub(r'//[^\n]*|/\*.*?\*/|"(?:\\.|[^"\\])*"|\'(?:\\.|[^\'\\])*\'', keep_strings, src, flags=re.S)
    return "\n".join(l.rstrip() for l in src.splitlines() if l.strip())


def kernel_source_sha16(root=None):
    """Hash of everything that determines the device code (csrc/*.hip, *.h, include/*.h, the build script), comments and blank
    lines left out: rewording a comment does not un-stamp the PMC traffic file."""
    import hashlib

    root = root or ROOT
    h = hashlib.sha256()
    csrc = os.path.join(root, "dn-splatter_amd", "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h", ".sh")))
    files += sorted(os.path.join(root, "include", f) for f in os.listdir(os.path.join(root, "include")))
    for f in files:
        h.update(os.path.basename(f).encode())
        text = open(f, "r", encoding="utf-8").read()
        h.update((text if f.endswith(".sh") else _strip_comments(text)).encode())
    return h.hexdigest()[:16]


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


def cpu_baseline(workload, crop=None):
    """Times the CPU oracle — the reference's own two-call sequence (rasterization + legacy
    rasterize_gaussians, dn_model.py:495-575) through the same host mirror — on ONE FULL FRAME of the same scene
    (~20 s on 32 threads at C2); with ``crop`` on a centre crop scaled by the pixel ratio (hosts with few cores).
    Test infrastructure used as the checker/baseline only."""
    import dn_splatter_amd as dns
    from dn_splatter_amd import synthetic
    from dn_splatter_amd.model import Camera
    from oracle import oracle as orc

    N, W, H, focal = WORKLOADS[workload]
    # the oracle parallelises over tiles (8160 at C2); since round 5 its gradient scatter adds up a (tile, splat) row locally and
    # issues ONE atomic row per (tile, splat) instead of one atomic per (pixel, splat, component), which had capped the useful team
    # at ~32 threads (49 s on 256 threads vs 14 s on 8).  DNSPLAT_CPU_THREADS overrides the team size (scaling runs).
    cores = int(os.environ.get("DNSPLAT_CPU_THREADS", "0")) or min(os.cpu_count() or 1, CPU_BASELINE_THREADS)
    torch.set_num_threads(cores)     # torch and the oracle share the process' OpenMP runtime (libgomp)
    gp = synthetic.make_gauss_params(N, sh_rest_std=0.1, seed=0)
    cam = synthetic.orbit_camera(0, width=W, height=H, focal=focal)
    if crop is None and (os.cpu_count() or 1) < 16:
        crop = 512                                     # a full 1080p frame would take minutes on a small host
    cw, ch = (W, H) if crop is None else (min(crop, W), min(crop, H))
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
        "host_cpu_count": os.cpu_count(),
        "kind": "port",
        "sample": (f"oracle fwd+bwd of {'one full ' + str(cw) + 'x' + str(ch) + ' frame' if scale == 1.0 else 'a ' + str(cw) + 'x' + str(ch) + ' centre crop'} "
                   f"of the {workload} scene (all {N} Gaussians projected, "
                   f"{int(m.last_info['flatten_ids'].shape[0])} intersections) took {dt:.2f} s on {cores} of the host's "
                   f"{os.cpu_count()} hardware threads (one process, OpenMP over tiles; team size from profiles/r05b_cpu_baseline_scaling.txt)"
                   + ("" if scale == 1.0 else f"; value = 1/(t x {scale:.1f} pixel ratio)")),
    }


def torch_loss_kwargs(losses):
    """--losses mode -> keyword arguments of torch_losses.dn_loss."""
    return {"capturable": losses in ("torch_capturable", "torch_hip_ssim"),
            "ssim_impl": "hip" if losses in ("torch_hip_ssim", "torch_eager_hip_ssim", "torch_hip_modules") else None,
            "hip_modules": losses == "torch_hip_modules"}


def side_section(workload, steps, losses=None, tight=True, shared=None, rank=0):
    """One more configuration measured AFTER the timed region of the headline workload, by the same method at a smaller scale:
    parameters resident in HBM, the whole step (get_outputs + backward) captured into a HIP graph, PREROLL_STEPS untimed replays,
    then ``steps`` replays timed between two synchronisations.  Per-stage times come from PROBE_STEPS instrumented eager steps
    before the capture.  ``shared`` = (gp, renderer, cam, cot) re-uses the headline scene (the index-exact configuration);
    otherwise the BASELINE workload ``workload`` is built here and freed afterwards.  ``tight=False``: gsplat's own tile boxes
    (DNSPLAT_TIGHT_TILES=0), i.e. flatten_ids / isect_offsets are the reference's bit for bit."""
    import dn_splatter_amd as dns
    from dn_splatter_amd import _lib, _ops, dp, synthetic
    from dn_splatter_amd.graph import GraphedStep

    N, W, H, focal = WORKLOADS[workload]
    P, T = W * H, ((W + 15) // 16) * ((H + 15) // 16)
    dev = torch.device("cuda", torch.cuda.current_device())
    prev = dict(arena=_ops.GRAD_ARENA, tight=_ops.TIGHT_TILES, policy=_ops.BIN_POLICY["mode"])
    gstep = None
    try:
        _ops.TIGHT_TILES = bool(tight)
        if shared is not None:
            gp, renderer, cam, cot = shared
            arena = prev["arena"]
        else:
            gp = synthetic.make_gauss_params(N, sh_rest_std=0.1, seed=0, device=dev)
            cam = synthetic.orbit_camera(rank % 8, n_views=8, width=W, height=H, focal=focal).to(dev)
            renderer = dns.DNSplatterRenderer(gp, fused=True)
            arena = dp.GradArena(gp)
            dns.set_grad_arena(arena)
            gen = torch.Generator(device=dev).manual_seed(1 + rank)
            shapes = {"rgb": (H, W, 3), "depth": (H, W, 1), "normal": (H, W, 3), "accumulation": (H, W, 1)}
            cot = {k: torch.rand(shapes[k], device=dev, generator=gen) * 2 - 1 for k in OUT_KEYS}
        batch = loss_counts = None
        if losses:
            from dn_splatter_amd import fused_loss, torch_losses
            batch = torch_losses.synthetic_batch(W, H, dev, seed=rank)
            if losses == "fused":
                loss_counts = fused_loss.depth_counts(batch["mono_depth"])

        def compute():
            for k in dp.GRAD_KEYS:
                gp[k].grad = None
            out = renderer.get_outputs(cam)
            if batch is not None and losses == "fused":
                fused_loss.dn_loss_fused(out, batch, gp["scales"], counts=loss_counts).backward()
            elif batch is not None:
                torch_losses.dn_loss(out, batch, gp["scales"], **torch_loss_kwargs(losses)).backward()
            else:
                torch.autograd.backward([out[k] for k in OUT_KEYS], [cot[k] for k in OUT_KEYS])

        renderer.forget()
        dns.set_bin_policy("capacity")          # a first frame that outgrows an older capacity guess is repaired in place
        for _ in range(FIRST_TOUCH_STEPS + 1):
            compute()
        torch.cuda.synchronize()
        for _ in range(PREROLL_STEPS):
            compute()
        torch.cuda.synchronize()
        probe = _lib.StageTimer()
        _lib.TIMER = probe
        for _ in range(PROBE_STEPS):
            compute()
        torch.cuda.synchronize()
        _lib.TIMER = None
        pst = probe.summary()
        I_sorted = int(renderer.last_info["n_isects"])
        Nv = int((renderer.radii > 0).sum())
        # nothing may keep the last eager frame's autograd graph alive (its AccumulateGrad nodes belong to the eager stream; a
        # capture on another stream that meets them drags that stream into the capture)
        renderer.forget()
        gc.collect()
        launch = "one HIP graph replay per step"
        try:
            gstep = GraphedStep(compute, params={k: gp[k] for k in dp.GRAD_KEYS})
            run = gstep
        except Exception as e:
            launch = f"eager launches from Python (graph capture failed: {e!r})"
            dns.set_bin_policy("deferred")
            run = compute
        for _ in range(PREROLL_STEPS):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            run()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if gstep is not None:
            gstep.check()
        else:
            _ops.verify_pending_counts(dev, block=True)

        def sms(name):
            if name == "binning":
                return sum(pst[k][2] for k in ("dnsplat_bin_prepare", "dnsplat_bin_emit_sort") if k in pst) / PROBE_STEPS
            return pst[name][1] if name in pst else None

        sb = stage_bytes(N, Nv, I_sorted, P, T)
        B = sum(sb.values())
        ms_bwd = sms("dnsplat_raster_bwd")
        res = {"workload": workload, "value": round(steps / dt, 3), "unit": "frames/s", "ms_per_step": round(1e3 * dt / steps, 4),
               "steps": steps, "launch": launch, "losses": losses or "random dense cotangents",
               "tile_boxes": "tight (fused path)" if tight else "gsplat 3-sigma (index-exact)",
               "N": N, "Nv": Nv, "n_isects_sorted": I_sorted, "pixels": P,
               "stages_ms": {n: round(sms(n), 4) for n in STAGE_NAMES if sms(n) is not None},
               "stages_measured": f"{PROBE_STEPS} instrumented eager steps before the capture",
               "frame_roofline_frac": round(B * (steps / dt) / 1e9 / HBM_PEAK_GBS, 4),
               "raster_bwd_roofline_frac": (round(sb["dnsplat_raster_bwd"] / (ms_bwd * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms_bwd else None)}
        return res
    except Exception as e:
        return {"workload": workload, "value": None, "error": repr(e)}
    finally:
        _lib.TIMER = None
        if gstep is not None:
            gstep.close()
        _ops.TIGHT_TILES = prev["tight"]
        dns.set_grad_arena(prev["arena"])
        dns.set_bin_policy(prev["policy"])
        if shared is None:
            _ops.forget_capacity_guesses(dev)
        gc.collect()
        torch.cuda.empty_cache()


def make_scene(N, dev, scene="reference_init", seed=0):
    """The benchmark scene.  "reference_init": the reference's own random initialisation (isotropic 3-NN scales, opacity 0.1,
    BASELINE.md section 4).  "anisotropic": the parity suite's harder scene at full size (tests/_scenes.gsplat_inputs(anisotropic=
    True)): log-scales + N(0, 0.6) per axis, opacity logits + N(0, 2) — long thin footprints, opacities from ~0 to beyond the 0.999
    cap (the clamping twin of the compositing backward runs), shorter useful runs per half tile."""
    from dn_splatter_amd import synthetic

    gp = synthetic.make_gauss_params(N, sh_rest_std=0.1, seed=seed, device=dev)
    if scene == "anisotropic":
        g = torch.Generator(device=dev).manual_seed(1000 + seed)
        with torch.no_grad():
            gp["scales"] += torch.randn(N, 3, device=dev, generator=g) * 0.6
            gp["opacities"] += torch.randn(N, 1, device=dev, generator=g) * 2.0
    elif scene == "morton":
        # the reference's initialisation with its rows laid out along a Morton curve (densify.spatial_order: what
        # refinement_after(spatial_reorder=True) leaves behind) — the same Gaussians, the same images, other memory order
        from dn_splatter_amd import densify

        new, _ = densify.reorder(gp, densify.spatial_order(gp["means"]))
        gp = {k: (v.contiguous().requires_grad_(True) if k != "normals" else v.contiguous()) for k, v in new.items()}
    elif scene != "reference_init":
        raise ValueError(scene)
    return gp


def train_loop_section(workload, steps, rank=0):
    """A train-like timed section (the reference loop: dn_model.py:271-386 refinement, :938-950 callbacks,
    dn_datamanager.py:90-96 sequential cameras): the whole step — get_outputs, the fused dn-splatter loss stack, backward,
    DensifyStats.after_train — captured ONCE into a HIP graph and replayed with a NEW POSE each step (8 orbit poses cycled by
    camera_to_worlds.copy_), check() every 10 steps, and in the middle one refinement_after + after_refinement + re-capture on the
    refined Gaussian set.  Prints one JSON line after the first half and one at the end (a crash in the re-capture leaves the first)."""
    import dn_splatter_amd as dns
    from dn_splatter_amd import _ops, densify, dp, fused_loss, synthetic, torch_losses
    from dn_splatter_amd.graph import GraphedStep

    N, W, H, focal = WORKLOADS[workload]
    dev = torch.device("cuda", torch.cuda.current_device())
    gp = make_scene(N, dev)
    poses = [synthetic.orbit_camera(i, n_views=8, width=W, height=H, focal=focal).camera_to_worlds.to(dev) for i in range(8)]
    cam = synthetic.orbit_camera(0, n_views=8, width=W, height=H, focal=focal).to(dev)      # its pose tensor is what the graph reads
    batch = torch_losses.synthetic_batch(W, H, dev, seed=rank)
    counts = fused_loss.depth_counts(batch["mono_depth"])
    half = max(steps // 2, 10)
    res = {"workload": workload, "what": "8 orbit poses cycled under one captured step (fused loss, densify statistics in the step), "
           "check() every 10 steps, one refinement + re-capture in the middle", "unit": "frames/s", "steps": 2 * half, "phases": []}

    def build(gp):
        renderer = dns.DNSplatterRenderer(gp, fused=True)
        arena = dp.GradArena(gp)
        dns.set_grad_arena(arena)
        stats = densify.DensifyStats(gp["means"].shape[0], dev)

        def compute():
            for k in dp.GRAD_KEYS:
                gp[k].grad = None
            out = renderer.get_outputs(cam)
            fused_loss.dn_loss_fused(out, batch, gp["scales"], counts=counts).backward()
            stats.after_train(renderer, W, H)

        # every pose once, eagerly: the capacity guess becomes 1.25 x the largest count any pose of the cycle produces — unless a
        # guess for this size exists already (densify.after_refinement carries the old size's over in proportion): then the capture's
        # own two warm-up frames are all that runs eagerly
        dns.set_bin_policy("capacity")
        isects = None
        if (dev, int(gp["means"].shape[0]), W, H) not in _ops.BUFFERS.capacity_hint:
            isects = []
            for p in poses:
                cam.camera_to_worlds.copy_(p)
                compute()
                isects.append(int(renderer.last_info["n_isects"]))
        torch.cuda.synchronize()

        def reset_stats():      # IN PLACE: the captured dnsplat_densify_stats launch writes into these very tensors
            stats.xys_grad_norm.zero_(); stats.vis_counts.fill_(1.0); stats.max_2Dsize.zero_()

        renderer.forget()
        gc.collect(1)       # the last eager frame's autograd graph is young garbage: a full collection costs ~35 ms of this process
        t0 = time.perf_counter()
        step = GraphedStep(compute, params={k: gp[k] for k in dp.GRAD_KEYS})
        torch.cuda.synchronize()
        capture_ms = 1e3 * (time.perf_counter() - t0)
        cap = min([v for k, v in _ops.BUFFERS.static_cap.items() if k[:2] == (dev, step.stream.cuda_stream)] or [0])
        reset_stats()
        return renderer, stats, step, isects, capture_ms, cap

    def run_phase(step, n, first):
        for i in range(PREROLL_STEPS):
            cam.camera_to_worlds.copy_(poses[i % 8])
            step()
        torch.cuda.synchronize()
        checks = 0
        t0 = time.perf_counter()
        for i in range(n):
            cam.camera_to_worlds.copy_(poses[(first + i) % 8])
            step()
            if (i + 1) % 10 == 0:
                step.check()                    # synchronises: part of what a training loop pays
                checks += 1
        torch.cuda.synchronize()
        return time.perf_counter() - t0, checks

    # the refinement code path once before anything is timed: the FIRST call in a process pays ~0.2-0.3 s of one-time initialisation
    # (torch's nonzero / index_select / randn kernels, the two densify kernels), a training loop pays it once in ~140 refinements
    t0 = time.perf_counter()
    warm = densify.DensifyStats(N, dev)
    warm.xys_grad_norm[::7] = 1.0                 # every seventh Gaussian over the gradient threshold: split / duplicate paths are taken
    densify.refinement_after({k: v.detach() for k, v in gp.items()}, warm, densify.RefineConfig(), 3500, 8, (H, W), seed=1)
    torch.cuda.synchronize()
    first_refine_ms = 1e3 * (time.perf_counter() - t0)
    del warm
    renderer, stats, step, isects, capture_ms, cap = build(gp)
    dt1, checks1 = run_phase(step, half, 0)
    ph = {"N": int(gp["means"].shape[0]), "steps": half, "value": round(half / dt1, 3), "ms_per_step": round(1e3 * dt1 / half, 4),
          "capture_ms": round(capture_ms, 1), "n_isects_per_pose": isects, "n_isects_max_over_mean": round(max(isects) / (sum(isects) / 8), 4),
          "capacity": cap, "checks": checks1, "overflows": 0, "n_isects_max_seen_by_the_replays": int(max(t.item() for t in step._n_max.values()))}
    res["phases"].append(ph)
    res.update(value=ph["value"], ms_per_step=ph["ms_per_step"], note="first half only (the line after the refinement replaces this one)")
    print(json.dumps(res), flush=True)

    # ---- the refinement step (dn_model.py:271-386) on the statistics the replays accumulated, then a new capture
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks = {}

    def mark(name):
        torch.cuda.synchronize()
        marks[name] = round(1e3 * (time.perf_counter() - t0) - sum(marks.values()), 2)

    step.close()
    mark("close_old_step")
    cfg = densify.RefineConfig()
    params = {k: v.detach() for k, v in gp.items()}
    new, _adam, report = densify.refinement_after(params, stats, cfg, 3500, 8, (H, W), adam_state=None, seed=5)
    mark("refinement_after")
    gp2 = {k: (v.requires_grad_(True) if k != "normals" else v) for k, v in new.items()}      # the gathered rows ARE the new leaves
    del gp, params, new, renderer, stats, step
    densify.after_refinement(gp2, report=report)          # new bucket; the capacity guesses carried over to the new size
    mark("new_leaves_and_bucket")
    gc.collect(1)
    mark("gc_collect_young")
    refine_ms = 1e3 * (time.perf_counter() - t0)
    launch = "one HIP graph replay per step"
    t1 = time.perf_counter()
    try:
        renderer, stats, step, isects2, capture_ms2, cap2 = build(gp2)
    except Exception as e:      # the eager path is always there
        res["recapture_error"] = repr(e)
        raise
    torch.cuda.synchronize()
    rebuild_ms = 1e3 * (time.perf_counter() - t1)      # the 8 eager pose frames (capacity), the warm-up frames and the capture
    dt2, checks2 = run_phase(step, half, half)
    step.check()
    ph2 = {"N": int(gp2["means"].shape[0]), "steps": half, "value": round(half / dt2, 3), "ms_per_step": round(1e3 * dt2 / half, 4),
           "capture_ms": round(capture_ms2, 1), "n_isects_per_pose": isects2 or "not probed: capacity carried over from the old size",
           "capacity": cap2, "checks": checks2, "overflows": 0,
           "n_isects_max_seen_by_the_replays": int(max(t.item() for t in step._n_max.values()))}
    res["phases"].append(ph2)
    total = dt1 + dt2 + (refine_ms + rebuild_ms) * 1e-3
    res.update(value=round(2 * half / total, 3), ms_per_step=round(1e3 * total / (2 * half), 4), launch=launch,
               refinement={"ms": round(refine_ms, 1), "breakdown_ms": marks, "rebuild_ms": round(rebuild_ms, 1),
                           "first_call_in_the_process_ms": round(first_refine_ms, 1),
                           "of_it_capture_ms": round(capture_ms2, 1), **{k: report[k] for k in
                           ("n_before", "n_after", "n_split", "n_dup", "n_culled")}},
               note="value = all steps / (both phases + refinement + rebuild: the 8 eager pose frames that size the buffers, the warm-up "
                    "frames and the capture); phases[i].value = replays of one capture alone")
    print(json.dumps(res), flush=True)


def child_section(section, steps, timeout=None):
    """Runs ``bench.py --section <section>`` in a process of its own (a second capture with other buffer sizes inside one process has
    crashed the HIP runtime before, see child_workload) and returns the LAST JSON line it printed."""
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--section", section, "--steps", str(steps)]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                            "DNSPLAT_FORCE_DIST")}
    out = ""
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout or CHILD_TIMEOUT_S, env=env)
        out = r.stdout
        lines = [l for l in out.strip().splitlines() if l.startswith("{")]
        d = json.loads(lines[-1])
        if r.returncode != 0:
            d["child_exit_code"] = r.returncode
            d["child_stderr_tail"] = r.stderr[-400:]
        return d
    except Exception as e:
        return {"section": section, "value": None, "error": repr(e), "stdout_tail": out[-300:]}


def child_workload(workload, losses, steps, scene=None, extra_args=()):
    """Another BASELINE workload measured by THIS script in a process of its own (same method as the headline: graph replay,
    pre-roll, device stamps around the dominant stage, counting step), started after the headline's timed region; returns the
    fields of its JSON line that BASELINE.md section 2 asks for.  A process of its own, because a second and third HIP-graph
    capture with other buffer sizes inside one process crashed the HIP runtime in hipStreamEndCapture on ROCm 7.2 (round 4,
    gpurun_out/r04b) and a crash there must not take the headline line with it."""
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--workload", workload, "--steps", str(steps), "--warmup", "3",
           "--no-cpu-baseline", "--no-strict", "--no-extra-workloads"] + (["--losses", losses] if losses else [])
    if scene:
        cmd += ["--scene", scene]
    cmd += list(extra_args)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                            "DNSPLAT_FORCE_DIST")}
    if scene == "morton":
        # rows along the curve AND the bucket's zero rows tracked: wholly culled workgroups of the projection backward write nothing
        env["DNSPLAT_SH_ZERO_STATE"] = "1"
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=CHILD_TIMEOUT_S, env=env)
        d = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:
        return {"workload": workload, "value": None, "error": repr(e)}
    cfg, rv = d.get("config", {}), d.get("roofline_valu") or {}
    return {"workload": cfg.get("workload"), "value": d.get("value"), "unit": d.get("unit"), "ms_per_step": d.get("ms_per_step"),
            "steps": d.get("steps"), "launch": d.get("launch"), "losses": losses or "random dense cotangents",
            "N": cfg.get("N"), "Nv": cfg.get("Nv"), "n_isects": cfg.get("n_isects"), "n_isects_sorted": cfg.get("n_isects_sorted"),
            "mean_blended_gaussians_per_pixel": cfg.get("mean_blended_gaussians_per_pixel"),
            "stages_ms": {k: v.get("ms") for k, v in (d.get("stages") or {}).items()},
            "stages_GBps": {k: v.get("GBps") for k, v in (d.get("stages") or {}).items()},
            "roofline": {k: (d.get("roofline") or {}).get(k) for k in ("kernel", "achieved", "frac", "ms_per_launch")},
            "frame_roofline_frac": (d.get("frame_roofline") or {}).get("frac"),
            "useful_pair_fraction": {k: v.get("useful_pair_fraction") for k, v in rv.items() if isinstance(v, dict) and "useful_pair_fraction" in v},
            "scene": cfg.get("scene"), "alpha_clamp_twin_active": cfg.get("alpha_clamp_twin_active"),
            "eager_drop_in": d.get("eager_drop_in"),
            "measured": "a child process running this script on that workload, after the headline's timed region"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-crop", type=int, default=None, help="time the CPU baseline on a centre crop of this size instead of a full frame")
    ap.add_argument("--bin-policy", default="deferred", choices=["sync", "capacity", "deferred"],
                    help="how the host learns the frame's intersection count (dn-splatter_amd/_ops.py BIN_POLICY): 'deferred' never "
                         "waits for it in the steady state (verified as soon as it has arrived, an overflow raises)")
    ap.add_argument("--dense-allreduce", action="store_true",
                    help="all-reduce the full 236 B/Gaussian bucket instead of exchanging the SH gradients as factors")
    ap.add_argument("--exchange", default="auto", choices=["auto", "rebuild", "own", "packed", "own+packed"],
                    help="N > 1 ranks (or DNSPLAT_FORCE_DIST=1), the SH part of the exchange step (dp.ShFactorExchange): 'auto' = own-camera "
                         "rows at world 1 (nothing to rebuild), every row rebuilt from dense slabs at world > 1; 'rebuild' = every row rebuilt at "
                         "any world size (round 5); 'packed' = slabs of the visible rows only (84 -> ~61 B / Gaussian received at 8 GPUs; "
                         "+0.03 / +0.13 ms of kernels at C2 / C5 on one rank, profiles/r06_exchange_single_rank_rccl.txt); 'own' / "
                         "'own+packed' force own-camera rows at any world size")
    ap.add_argument("--reduction", default="sum", choices=["sum", "mean"],
                    help="N > 1 ranks: 'sum' = every rank scales its image cotangents by 1 / world and the collectives ADD (dp.set_reduction: "
                         "RCCL's AVG is pre-multiply + sum, which on one rank still runs a kernel over the whole bucket prefix); 'mean' = "
                         "unscaled cotangents, averaging collectives (round 5)")
    ap.add_argument("--slices", type=int, default=1,
                    help="N > 1 ranks (or DNSPLAT_FORCE_DIST=1): the projection backward as this many slices of Gaussians, slice k's "
                         "colour-gradient slab all-gathered while slices k+1.. compute (dp.SlicedShExchange); 1 (default) = one launch, "
                         "one slab: measured on one rank through RCCL every extra slice costs ~0.05 ms of launches and stream "
                         "hand-overs (profiles/r05b_*rccl*), more than the earlier start of the links buys at 1 M Gaussians")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="replay the step (get_outputs + backward) as a captured HIP graph (dn-splatter_amd/graph.py): 'auto' = when the "
                         "step has no collective in it (one rank) and runs the fused path; falls back to eager launches if the capture fails")
    ap.add_argument("--two-call", action="store_true", help="reference's two-pass sequence instead of the fused pass")
    ap.add_argument("--torch-postops", action="store_true", help="keep dn_model.py:526-603 in torch instead of the HIP epilogue")
    ap.add_argument("--losses", nargs="?", const="torch", default=None, choices=["torch", "torch_capturable", "torch_hip_ssim", "torch_eager_hip_ssim", "torch_hip_modules", "fused"],
                    help="time dn-splatter's loss stack (L1+SSIM, EdgeAwareLogL1 depth, normal L1+TV, scale) instead of feeding "
                         "random cotangents (BASELINE config C5): 'torch' = as the reference does (its boolean-mask gathers need the "
                         "host: no graph capture, eager launches), 'torch_capturable' = the same PyTorch stack with the masked means as "
                         "sum / count (capturable), 'torch_hip_ssim' = the capturable PyTorch stack with splatfacto's SSIM module on dnsplat_ssim "
                         "(fused_loss.SSIM; every other term in PyTorch), 'torch_eager_hip_ssim' = the reference's stack AS IT IS (boolean-mask "
                         "gathers, eager) with only that module swapped: what install() + install_ssim() give without touching the loss "
                         "code, 'torch_hip_modules' = the reference's stack with the three modules install_losses(model) swaps (SSIM, "
                         "EdgeAwareLogL1, TVLoss) on HIP and everything else — L1 terms, masks, weights, the scale term — in PyTorch, "
                         "'fused' = dnsplat_dn_loss")
    ap.add_argument("--lean", action="store_true",
                    help="profiling runs (rocprofv3 --kernel-trace / --pmc serialise every launch): eager launches, no pre-roll, no "
                         "counting step, no strict-index-parity section")
    ap.add_argument("--no-strict", action="store_true", help="skip the index-exact (gsplat tile boxes) section after the timed region")
    ap.add_argument("--no-extra-workloads", action="store_true",
                    help="skip the short C3 / C5 / C5-with-fused-losses sections a default C2 run appends (extra_workloads in the JSON line)")
    ap.add_argument("--scene", default="reference_init", choices=["reference_init", "anisotropic", "morton"],
                    help="reference_init: the reference's random initialisation (BASELINE.md); anisotropic: the parity suite's "
                         "anisotropic / spread-opacity scene at the workload's size (make_scene)")
    ap.add_argument("--section", default=None, choices=["train_loop"],
                    help="run one of the side sections a default C2 run starts as child processes, print its JSON line(s) and exit")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="start the ranks, count them with one all-reduce and print {n_gpus, ranks_seen} without rendering (works "
                         "without a GPU over gloo: the CPU test of the --gpus N self-launch)")
    args = ap.parse_args()
    if args.lean:
        args.graph = "off"

    # `python bench.py --gpus N` without a launcher: become `python -m torch.distributed.run --nproc-per-node N bench.py ...`
    # (one process per GPU over RCCL), so that the line printed is never a single-GPU run labelled as N
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        import socket

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)

    import dn_splatter_amd as dns
    from dn_splatter_amd import _lib, dp, synthetic

    rank, world, local, dev = dp.init_from_env(None if args.rendezvous_only else "cuda")
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if torch.distributed.is_initialized():
        assert torch.distributed.get_world_size() == args.gpus
    if args.rendezvous_only:
        seen = int(dp.sum_over_ranks([1.0], dev)[0])
        dp.barrier()
        if rank == 0:
            print(json.dumps({"metric": "fwd+bwd frames/sec @1M Gaussians 1080p; HBM GB/s vs roofline", "value": None,
                              "unit": "frames/s", "n_gpus": world, "multi_gpu": {"ranks_seen": seen},
                              "note": "rendezvous only: the ranks were started and counted, nothing was rendered"}), flush=True)
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
        return
    if args.section == "train_loop":
        train_loop_section(args.workload, args.steps, rank)
        return
    N, W, H, focal = WORKLOADS[args.workload]
    P = W * H
    T = ((W + 15) // 16) * ((H + 15) // 16)

    # identical parameters on every rank (seed 0), one camera per rank (8-view orbit)
    gp = make_scene(N, dev, args.scene, seed=0)
    cam = synthetic.orbit_camera(rank % 8, n_views=8, width=W, height=H, focal=focal).to(dev)
    renderer = dns.DNSplatterRenderer(gp, fused=not args.two_call, fused_postops=not args.torch_postops)
    dns.set_bin_policy(args.bin_policy)
    arena = dp.GradArena(gp)
    dns.set_grad_arena(arena)
    exchange = None
    if (world > 1 or os.environ.get("DNSPLAT_FORCE_DIST", "0") == "1") and not args.dense_allreduce and not args.two_call:
        if args.slices > 1:
            exchange = dp.SlicedShExchange(args.slices)
        else:
            mode = args.exchange
            own = {"auto": None, "rebuild": False, "own": True, "packed": False, "own+packed": True}[mode]
            packed = mode in ("packed", "own+packed")
            exchange = dp.ShFactorExchange(own_rows=own, packed=packed)
        dns.set_sh_exchange(exchange)
    gen = torch.Generator(device=dev).manual_seed(1 + rank)
    shapes = {"rgb": (H, W, 3), "depth": (H, W, 1), "normal": (H, W, 3), "accumulation": (H, W, 1)}
    cot = {k: torch.rand(shapes[k], device=dev, generator=gen) * 2 - 1 for k in OUT_KEYS}
    if exchange is not None or world > 1:
        dp.set_reduction(args.reduction)
        if args.reduction == "sum" and world > 1:
            cot = {k: v / world for k, v in cot.items()}      # the loss of a data-parallel step is the mean over the ranks' cameras

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
            loss = fused_loss.dn_loss_fused(out, batch, gp["scales"], counts=loss_counts)
            (loss * loss_scale if loss_scale != 1.0 else loss).backward()
        elif batch is not None:
            loss = torch_losses.dn_loss(out, batch, gp["scales"], **torch_loss_kwargs(args.losses))
            (loss * loss_scale if loss_scale != 1.0 else loss).backward()
        else:
            # the losses stay in PyTorch (north star); their result is a dense cotangent per output image
            torch.autograd.backward([out[k] for k in OUT_KEYS], [cot[k] for k in OUT_KEYS])
        return dp.allreduce_gradients(gp, arena, exchange=exchange)

    # "sum" reduction: the step's loss is the MEAN over the ranks' cameras, so every rank scales its own by 1 / world
    loss_scale = (1.0 / world) if (world > 1 and dp.REDUCTION["mode"] == "sum") else 1.0
    step_eager = step
    for _ in range(FIRST_TOUCH_STEPS):     # allocations, capacity guesses, code objects (the W warm-up steps come right before the timed region)
        step()
    if exchange is not None and getattr(exchange, "packed", False):
        # slabs of visible rows only: every rank's capacity = 1.1 x the largest visible count of the ranks' cameras (one host sync and
        # one tiny collective, here, outside every timed region); a step that overflowed it anyway falls back to dense slabs
        exchange.calibrate(renderer.radii)
        step(); step()
        torch.cuda.synchronize()
        if dp.max_over_ranks(float(exchange.overflowed()), dev) > 0:
            exchange.packed = False
            exchange.mine = exchange.gathered = None
            step()
    # Stage breakdown: PROBE_STEPS fully instrumented steps OUTSIDE the timed region.  Bracketing all six stages with
    # HIP events costs ~80 us of stream time per frame (a ~6 us bubble per event pair, seen in the rocprofv3 timeline),
    # so the timed region below only brackets the dominant stage, whose live duration the roofline figure needs.
    for _ in range(0 if args.lean else PREROLL_STEPS):      # the probes, too, are taken at operating clocks (see PREROLL_STEPS)
        step()
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
    # ---- the EAGER drop-in figure (VERDICT r04 missing 4): what a maintainer gets who applies INTEGRATION.md under nerfstudio's
    # trainer — the same fused step launched kernel by kernel from Python through ctypes and autograd (~35 launches), no graph,
    # bin policy as given (default "deferred": the host never waits for the intersection count).  Pre-rolled like everything else.
    eager_drop_in = None
    if world == 1 and not args.lean and not args.two_call and not args.torch_postops:
        from dn_splatter_amd import _ops as _ops_e
        for _ in range(PREROLL_STEPS):
            step()
        torch.cuda.synchronize()
        gc.collect(); gc.disable()
        te, host_e = time.perf_counter(), 0.0
        for _ in range(args.steps):
            h0 = time.perf_counter()
            step()
            host_e += time.perf_counter() - h0
        torch.cuda.synchronize()
        _ops_e.verify_pending_counts(dev, block=True)
        dte = time.perf_counter() - te
        gc.enable()
        eager_drop_in = {"value": round(args.steps / dte, 3), "unit": "frames/s", "ms_per_step": round(1e3 * dte / args.steps, 4),
                         "host_enqueue_ms_per_step": round(1e3 * host_e / args.steps, 4), "steps": args.steps,
                         "launch": f"eager launches from Python (fused path, bin policy '{args.bin_policy}', no HIP graph)",
                         "host_cpu_count": os.cpu_count()}
    # one step through the COUNTING instantiation of the compositing kernels (outside the timed region)
    counts = None
    if not args.two_call and not args.torch_postops and not args.lean:
        from dn_splatter_amd import _ops
        _ops.PAIR_COUNTERS = torch.zeros(8, dtype=torch.int64, device=dev)
        step()
        torch.cuda.synchronize()
        counts = _ops.PAIR_COUNTERS.tolist()
        _ops.PAIR_COUNTERS = None
    dominant = max((n for n in probe_ms if probe_ms[n] is not None), key=lambda n: probe_ms[n])
    live = ("dnsplat_bin_prepare", "dnsplat_bin_emit_sort") if dominant == "binning" else (dominant,)
    dp.barrier()
    torch.cuda.synchronize()
    # ---- the step as a replayed HIP graph -------------------------------------------------------------------------------
    # The bracket around the dominant stage is part of the graph: two one-thread kernels that append the device's wall clock to a
    # ring (HIP events cannot be recorded into a graph under capture on ROCm 7.2).  Every replay of the timed region leaves its
    # pair; the tick is calibrated below against HIP events around a stamped interval.
    gstep, graph_note, stamps, tick_ms = None, None, None, None
    # with an exchange step (N > 1 ranks, or DNSPLAT_FORCE_DIST=1): the compute is one graph, the collectives follow each replay
    # eagerly (graph.GraphedDpStep); a dense all-reduce (--dense-allreduce) keeps the eager step
    want_graph = args.graph == "on" or (args.graph == "auto" and (world == 1 or exchange is not None) and not args.two_call
                                        and not args.torch_postops)
    gdp = None
    if want_graph:
        from dn_splatter_amd.graph import GraphedDpStep, GraphedStep

        def compute():
            for k in dp.GRAD_KEYS:
                gp[k].grad = None
            out = renderer.get_outputs(cam)
            if batch is not None and args.losses == "fused":
                fused_loss.dn_loss_fused(out, batch, gp["scales"], counts=loss_counts).backward()
            elif batch is not None:
                torch_losses.dn_loss(out, batch, gp["scales"], **torch_loss_kwargs(args.losses)).backward()
            else:
                torch.autograd.backward([out[k] for k in OUT_KEYS], [cot[k] for k in OUT_KEYS])

        stamps = _lib.StampTimer(live, dev)

        def before_capture(c):
            _lib.TIMER = stamps

        try:
            renderer.forget()          # the autograd graph of the last eager frame (its AccumulateGrad nodes live on the eager stream)
            if exchange is not None:
                gdp = GraphedDpStep(compute, gp, arena, exchange=exchange, before_capture=before_capture)
                gstep = gdp.step
            else:
                gstep = GraphedStep(compute, params={k: gp[k] for k in dp.GRAD_KEYS}, before_capture=before_capture)
        except Exception as e:      # the eager path is always there
            if args.graph == "on":
                raise
            gstep, graph_note = None, f"graph capture failed, eager launches instead: {e!r}"
            dns.set_bin_policy(args.bin_policy)
        finally:
            _lib.TIMER = None
        if gdp is not None:
            def step():     # noqa: F811
                gdp()
                return gdp.wire
        elif gstep is not None:
            def step():     # noqa: F811
                gstep()
                return 0
    timer = _lib.StageTimer(only=live)
    _lib.TIMER = None
    wire = 0
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    # Python's cyclic garbage collector off for the timed region (as timeit does): a generation-2 pass over the interpreter's
    # objects stalls the host for ~10 ms, i.e. four frames' worth of launches — seen as single 10 ms steps in a 30-step run
    gc.collect()
    gc.disable()
    # pre-roll (see PREROLL_STEPS); with the graph it doubles as the calibration interval of the stamp clock: HIP events around
    # an interval that two extra stamps delimit
    if gstep is not None:
        torch.cuda.synchronize()
        stamps.reset()
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        stamps.stamp(); c0.record()
    for _ in range(0 if args.lean else PREROLL_STEPS):
        step()
    if gstep is not None:
        c1.record(); stamps.stamp()
        torch.cuda.synchronize()
        r = stamps.ring[:int(stamps.cursor.item())].tolist()
        tick_ms = c0.elapsed_time(c1) / float(r[-1] - r[0])
    _lib.TIMER = None
    for _ in range(args.warmup):
        step()
    if gstep is not None:
        stamps.reset()
    _lib.TIMER = timer if gstep is None else None
    dp.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks[0].record()
    host_s = 0.0
    for i in range(args.steps):
        h0 = time.perf_counter()
        wire = step()
        host_s += time.perf_counter() - h0
        marks[i + 1].record()
    torch.cuda.synchronize()
    from dn_splatter_amd import _ops as _ops_mod
    _ops_mod.verify_pending_counts(dev, block=True)     # "deferred" bin policy: every frame's intersection count checked inside the timed region
    if gstep is not None:
        gstep.check()                                   # graph replays: no frame of the timed region overflowed its buffers
    dp.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gc.enable()
    _lib.TIMER = None
    elapsed = dp.max_over_ranks(elapsed, dev)

    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    q = lambda f: round(per_step[min(len(per_step) - 1, int(f * len(per_step)))], 4)   # noqa: E731
    info = renderer.last_info
    I_sorted = int(info["n_isects"])               # what this build bins: the fused path's tight tile boxes
    Nv = int((renderer.radii > 0).sum())
    # I of SURVEY.md 8(d) under the reference's rule (gsplat's 3-sigma tile box, A.3) is reported beside it: the bytes a stage is
    # credited with ("alg_bytes", what `achieved` / `frac` are computed from) are priced on the pairs the launch PROCESSES
    # (n_isects_sorted), the same formula on the reference-rule count is "alg_bytes_reference_rule"
    with torch.no_grad():
        xy_, r_ = renderer.xys.detach().reshape(-1, 2), renderer.radii.reshape(-1).float()
        tw_, th_ = (W + 15) // 16, (H + 15) // 16
        bx0 = torch.floor((xy_[:, 0] - r_) / 16).clamp(0, tw_); bx1 = torch.ceil((xy_[:, 0] + r_) / 16).clamp(0, tw_)
        by0 = torch.floor((xy_[:, 1] - r_) / 16).clamp(0, th_); by1 = torch.ceil((xy_[:, 1] + r_) / 16).clamp(0, th_)
        I = int((((bx1 - bx0) * (by1 - by0)) * (r_ > 0)).double().sum().item())
    stats = timer.summary()
    live_ms = []
    if gstep is not None:
        live_ms = [t * tick_ms for t in stamps.intervals_ticks()]     # one bracket per entry point of the dominant stage and step
    stages = {}
    sb = stage_bytes(N, Nv, I_sorted, P, T)
    sb_ref = stage_bytes(N, Nv, I, P, T)
    for name, b in sb.items():
        if name == dominant and gstep is not None:
            ms = sum(live_ms) / max(args.steps, 1)
        else:
            ms = stage_ms(stats, name, args.steps) if name == dominant else probe_ms.get(name)
        if ms is None:
            continue
        stages[name] = {"ms": round(ms, 4), "alg_bytes": b, "alg_bytes_reference_rule": sb_ref[name],
                        "GBps": round(b / (ms * 1e-3) / 1e9, 1) if ms > 0 else None,
                        "measured": "timed region" if name == dominant else f"{PROBE_STEPS} instrumented steps before it"}
    # HBM bytes per launch from the PMC passes (tools/pmc_traffic.sh -> profiles/pmc_traffic.json: separate FETCH_SIZE and
    # WRITE_SIZE runs of this very command, as MI355X_MICROARCH.md prescribes).  The file records the hash of the kernel
    # sources it was measured on: a figure taken on other kernels is not reported.
    pmc_traffic, traffic_note, stage_traffic = None, None, {}
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path):
        try:
            traffic_key = args.workload if args.scene == "reference_init" else f"{args.workload}_{args.scene}"
            ent_w = json.load(open(pmc_path)).get(traffic_key, {})
            if ent_w.get("source_sha16") == kernel_source_sha16():
                stage_traffic = {k: v["hbm_bytes_per_launch"] for k, v in ent_w.items() if isinstance(v, dict)}
                pmc_traffic = stage_traffic.get(dominant)
            else:
                traffic_note = (f"profiles/pmc_traffic.json[{traffic_key}] was measured on kernel sources "
                                f"{ent_w.get('source_sha16')}, this build is {kernel_source_sha16()}: re-run tools/pmc_traffic.sh")
        except Exception as e:
            traffic_note = f"profiles/pmc_traffic.json unreadable: {e!r}"
    for name, t in stage_traffic.items():
        if name in stages:
            stages[name]["hbm_traffic"] = t
            if stages[name]["ms"] > 0:      # what the memory system actually moved per second, beside the algorithmic figure (GBps)
                stages[name]["hbm_traffic_GBps"] = round(t / (stages[name]["ms"] * 1e-3) / 1e9, 1)
                stages[name]["hbm_traffic_frac"] = round(t / (stages[name]["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                stages[name]["traffic_over_alg_bytes"] = round(t / max(stages[name]["alg_bytes"], 1), 3)
    for name in stages:
        if stages[name].get("GBps") is not None:
            stages[name]["alg_frac"] = round(stages[name]["GBps"] / HBM_PEAK_GBS, 4)
    if "dnsplat_project_fwd_colours" in pstats and "dnsplat_project_fwd" in stages:
        # the SH colour half of the projection runs on a side stream beside the binning kernels (ProjCfg.split_colours)
        stages["dnsplat_project_fwd"]["colour_phase_on_side_stream_ms"] = round(pstats["dnsplat_project_fwd_colours"][1], 4)
    if "binning" in stages:      # its two entry points (depth sort + counts | pair generation + tile sort), from the instrumented steps
        for k in ("dnsplat_bin_prepare", "dnsplat_bin_emit_sort"):
            if k in pstats:
                stages["binning"][k + "_ms"] = round(pstats[k][1], 4)
    ach = stages[dominant]["GBps"]
    roofline = {"bound": "hbm", "kernel": dominant, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": pmc_traffic,
                "alg_bytes_per_launch": sb[dominant], "alg_bytes_reference_rule": sb_ref[dominant],
                "ms_per_launch": stages[dominant]["ms"],
                # the metric asks for the HBM figure; what BINDS this kernel is vector issue (roofline_valu)
                "binding_roofline": "roofline_valu" if dominant in ("dnsplat_raster_bwd", "dnsplat_raster_fwd") else "hbm"}
    if traffic_note:
        roofline["traffic_note"] = traffic_note
    # VALU roofline of the two compositing kernels (what binds them: DESIGN.md §4)
    roofline_valu = None
    if counts is not None:
        c = [int(x) for x in counts]
        roofline_valu = {"bound": "valu", "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "pairs": {"list_entries_examined": c[0], "splats_walked": c[1], "fwd_pairs_evaluated": c[2],
                                   "pairs_blended": c[3], "bwd_slots_issued": c[4], "bwd_pairs_replayed": c[5]},
                         "flops_per_pair": FLOPS_PER_PAIR}
        for name, issued, useful in (("dnsplat_raster_fwd", c[2], c[3]), ("dnsplat_raster_bwd", c[4], c[5])):
            if name in stages and stages[name]["ms"] > 0:
                t = stages[name]["ms"] * 1e-3
                f = FLOPS_PER_PAIR[name]
                roofline_valu[name] = {"achieved": round(issued * f / t / 1e12, 2), "frac": round(issued * f / t / 1e12 / VALU_PEAK_TFLOPS, 4),
                                       "useful_achieved": round(useful * f / t / 1e12, 2),
                                       "useful_frac": round(useful * f / t / 1e12 / VALU_PEAK_TFLOPS, 4),
                                       "useful_pair_fraction": round(useful / max(issued, 1), 4)}
    B = sum(sb.values())
    fps_total = world * args.steps / elapsed
    frame_roofline = {"alg_bytes_per_frame": B, "alg_bytes_reference_rule": sum(sb_ref.values()),
                      "achieved_GBps": round(B * (args.steps / elapsed) / 1e9, 1),
                      "frac": round(B * (args.steps / elapsed) / 1e9 / HBM_PEAK_GBS, 4)}
    gpu_stage_ms = sum(v["ms"] for v in stages.values())
    i_all = dp.sum_over_ranks([float(I)], dev)[0]

    # ---- the index-exact configuration, after the timed region ------------------------------------------------------------
    # The fused path bins over tight tile boxes: its sorted lists are a sub-list of gsplat's (the pairs left out cannot reach
    # alpha >= 1/255 anywhere in their tile; images and gradients are the same numbers).  With gsplat's own boxes
    # (DNSPLAT_TIGHT_TILES=0) flatten_ids / isect_offsets are the reference's bit for bit — that configuration is timed here.
    strict, extras = None, None
    from dn_splatter_amd import _ops as _ops_mod2
    # (not when the multi-GPU accounting below runs, DNSPLAT_FORCE_DIST=1: it replays the headline's graph, which this section closes)
    if (world == 1 and exchange is None and not args.two_call and not args.torch_postops and _ops_mod2.TIGHT_TILES and not args.lean
            and not args.no_strict and os.environ.get("DNSPLAT_FORCE_DIST", "0") != "1"):
        if gstep is not None:
            gstep.close()
        K3 = max(5, min(20, args.steps))
        strict = side_section(args.workload, K3, losses=args.losses, tight=False, shared=(gp, renderer, cam, cot), rank=rank)
        strict["what"] = ("same step with gsplat's 3-sigma tile boxes (DNSPLAT_TIGHT_TILES=0): flatten_ids / isect_offsets / "
                          "tiles_per_gauss are the reference's bit for bit (tests/test_gpu_parity.py)")
        # ---- the other BASELINE configurations, so that the driver's record carries them
        if args.workload == "c2" and not args.no_extra_workloads and not args.losses:
            extras = {}
            for name, wl, ls in (("c3", "c3", None), ("c5", "c5", None), ("c5_fused_loss", "c5", "fused"), ("c5_torch_loss", "c5", "torch"),
                                 ("c5_torch_loss_capturable", "c5", "torch_capturable"), ("c5_torch_loss_hip_ssim", "c5", "torch_hip_ssim"),
                                 ("c5_torch_loss_eager_hip_ssim", "c5", "torch_eager_hip_ssim"), ("c5_torch_loss_hip_modules", "c5", "torch_hip_modules")):
                extras[name] = child_workload(wl, ls, max(5, min(10, args.steps)))
            # the north star's C5 ("depth + mono-normal loss enabled", losses in PyTorch-ROCm: regularization_strategy.py:146-199,
            # losses.py:187-224) is c5_torch_loss; c5_fused_loss is the same loss stack as two HIP launches (N2)
            extras["c2_anisotropic"] = child_workload("c2", None, max(5, min(10, args.steps)), scene="anisotropic")
            # the SAME Gaussians as the headline / c5 with their rows laid out along a Morton curve (densify.spatial_order, what
            # refinement_after(spatial_reorder=True) leaves behind) and DNSPLAT_SH_ZERO_STATE=1: the per-Gaussian kernels then move
            # close to their algorithmic bytes (a camera's culled Gaussians are whole workgroups)
            extras["c2_morton"] = child_workload("c2", None, max(5, min(10, args.steps)), scene="morton")
            extras["c5_morton"] = child_workload("c5", None, max(5, min(10, args.steps)), scene="morton")
            # INTEGRATION.md section A as it stands: the reference's two passes through the two drop-in symbols plus its own ~40 torch
            # kernels, launched eagerly (where the time goes: profiles/r06_two_call_breakdown.txt)
            extras["c2_two_call"] = child_workload("c2", None, max(5, min(10, args.steps)), extra_args=["--two-call"])
            extras["c2_train_loop"] = child_section("train_loop", 200)

    # ---- multi-GPU accounting, all outside the timed region ----------------------------------------
    multi = None
    if world > 1 or os.environ.get("DNSPLAT_FORCE_DIST", "0") == "1":
        K2 = max(3, min(10, args.steps))

        def timed(fn, n):
            for _ in range(PREROLL_STEPS):       # operating clocks (see PREROLL_STEPS): each of the three figures is taken the same way
                fn()
            dp.barrier(); torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize(); dp.barrier()
            return dp.max_over_ranks((time.perf_counter() - t) / n, dev) * 1e3

        def compute_only():
            for k in dp.GRAD_KEYS:
                gp[k].grad = None
            out = renderer.get_outputs(cam)
            torch.autograd.backward([out[k] for k in OUT_KEYS], [cot[k] for k in OUT_KEYS])

        def exchange_only():
            # the same collectives on the same buffers, nothing to hide behind
            if exchange is not None:
                exchange.begin(N, dev, 3, 16)
                exchange.launch()
            dp.allreduce_gradients(gp, arena, exchange=exchange)

        if gdp is not None:
            # the graphed step: the same replay with and without the exchange behind it, and the exchange alone on the same buffers
            t_compute = timed(gdp.compute_only, K2)
            step(); step()
            t_step = timed(step, K2)
            t_comm = timed(gdp.exchange_only, K2)
        else:
            dns.set_sh_exchange(None)                # kernels write the SH rows themselves, no collective is started
            t_compute = timed(compute_only, K2)
            dns.set_sh_exchange(exchange)
            step(); step()
            t_step = timed(step, K2)
            t_comm = timed(exchange_only, K2)
        exposed = max(0.0, t_step - t_compute)
        # one rank through RCCL (development boxes): the same step captured WITHOUT any exchange — what a single-GPU run replays,
        # dnsplat_project_bwd writing the coefficient rows itself — so that the cost of the whole exchange machinery (slab, rebuild
        # kernel, stream hand-overs, RCCL's one-rank kernels) is a number.  Not on real multi-rank runs: a third capture beside
        # two live graphs is not something to try for the first time on the driver's 8-GPU node.
        t_single = None
        if world == 1 and gdp is not None:
            g1 = None
            try:
                dns.set_sh_exchange(None)
                renderer.forget(); gc.collect()
                g1 = GraphedStep(compute, params={k: gp[k] for k in dp.GRAD_KEYS})
                t_single = timed(g1, K2)
            except Exception as e:
                t_single = None
                graph_note = (graph_note or "") + f" single-GPU comparison capture failed: {e!r}"
            finally:
                if g1 is not None:
                    g1.close()
                dns.set_sh_exchange(exchange)
        per_rank_I = dp.gather_over_ranks(float(I), dev)
        multi = {"step_ms": round(t_step, 4), "compute_only_ms": round(t_compute, 4), "exchange_alone_ms": round(t_comm, 4),
                 "exchange_exposed_ms": round(exposed, 4), "exchange_hidden_ms": round(max(0.0, t_comm - exposed), 4),
                 "single_gpu_graphed_step_ms": (round(t_single, 4) if t_single is not None else None),
                 "exchange_exposed_vs_single_gpu_step_ms": (round(t_step - t_single, 4) if t_single is not None else None),
                 "exchange_slices": getattr(exchange, "slices", 1) if exchange is not None else None,
                 "reduction": dp.REDUCTION["mode"] + (" (cotangents pre-scaled by 1 / world, collectives add)" if dp.REDUCTION["mode"] == "sum" else ""),
                 "exchange_sh_rows": (None if exchange is None else
                                      ("own camera's rows written by dnsplat_project_bwd (x 1/W), the other W-1 added from the slabs"
                                       if exchange.use_own_rows() else "every row rebuilt from the W gathered slabs")),
                 "exchange_slabs": (None if exchange is None else
                                    (f"packed: mask + block offsets + the colour gradients of the visible Gaussians only, capacity "
                                     f"{exchange.packed_capacity(N)} of {N} rows" if getattr(exchange, "packed", False)
                                     else "dense: 12 B per Gaussian")),
                 "slab_bytes_per_rank": (None if exchange is None or getattr(exchange, "slices", 1) > 1 else int(exchange.slab_floats(N) * 4)),
                 "launch": (("everything up to the projection backward = one HIP graph replay per step; its K slice launches, each "
                             "followed by the all-gather of its slab, the geometry all-reduce and the rebuilds issued eagerly behind it "
                             "(graph.GraphedDpStep + dp.SlicedShExchange)") if (gdp is not None and gdp.sliced)
                            else "compute = one HIP graph replay per step, the collectives issued eagerly behind it (graph.GraphedDpStep)"
                            if gdp is not None else "eager launches from Python"),
                 "bytes_exchanged_per_gpu_per_step": int(wire),
                 # what the collectives sustained when nothing else ran (bytes received + sent per direction / exchange_alone_ms);
                 # DESIGN.md 6 prices the 8-GPU step on an ASSUMED 400 GB/s per GPU and direction: this is the number to compare
                 "exchange_alone_GBps_per_gpu": (round(wire / (t_comm * 1e-3) / 1e9, 1) if (world > 1 and t_comm > 0) else None),
                 "exchange_alone_includes_project_bwd_slices": bool(gdp is not None and gdp.sliced),
                 "bytes_per_xgmi_link_per_step": int(wire / (world - 1)) if world > 1 else None,
                 "link_note": ("per-GPU bytes spread evenly over the W-1 direct xGMI links of the fully connected node" if world > 1
                               else "one rank: the collectives ran through RCCL but nothing crossed a link"),
                 "ranks_seen": world,
                 "isects_per_rank": [int(x) for x in per_rank_I],
                 "isects_max_over_mean": round(max(per_rank_I) / (sum(per_rank_I) / len(per_rank_I)), 4)}

    if rank == 0:
        res = {
            "metric": "fwd+bwd frames/sec @1M Gaussians 1080p; HBM GB/s vs roofline",
            "value": round(fps_total, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "gpu_ms_per_step_p10_p50_p90": [q(0.1), q(0.5), q(0.9)], "gpu_ms_per_step_max": round(per_step[-1], 4),
            "host_enqueue_ms_per_step": round(1e3 * host_s / args.steps, 4),
            "gpu_ms_per_step_series": [round(marks[i].elapsed_time(marks[i + 1]), 3) for i in range(args.steps)],
            "dominant_stage_ms_series": [round(x, 3) for x in live_ms] if live_ms else None,
            "launch": ("one HIP graph replay per step; the dominant stage is bracketed inside the graph by device wall-clock stamps "
                       f"(s_memrealtime, {1e6 * tick_ms:.3f} ns per tick calibrated against HIP events; {len(live_ms)} brackets in the "
                       f"{args.steps} timed steps)") if gstep is not None
                      else ("eager launches from Python" + (f" ({graph_note})" if graph_note else "")),
            "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {N} random-init Gaussians, 1 camera/GPU {W}x{H}, SH degree 3 + "
                                   f"expected depth + per-Gaussian normals ({D_CH} channels, "
                                   f"{'two-call' if args.two_call else 'fused one-pass'}, post-ops in {'torch' if (args.torch_postops or args.two_call) else 'HIP'}), fx=fy={focal}, orbit r=8, "
                                   f"closed-form 3-NN scale init{'' if args.scene == 'reference_init' else ' perturbed: log-scales + N(0, 0.6), opacity logits + N(0, 2)'}, "
                                   f"{('dn-splatter loss stack (' + args.losses + ')') if args.losses else 'random dense cotangents'}",
                       "scene": args.scene,
                       "alpha_clamp_twin_active": (bool(int(info["_saturation_flag"].item())) if info.get("_saturation_flag") is not None else None),
                       "N": N, "Nv": Nv, "n_isects": I, "n_isects_sorted": I_sorted, "mean_isects_per_rank": i_all / world,
                       "mean_tile_list_len": round(I / T, 1), "pixels": P, "bin_policy": args.bin_policy,
                       # every result row reports Nv, I and the mean number of Gaussians blended per pixel
                       "mean_blended_gaussians_per_pixel": (round(counts[3] / P, 1) if counts else None),
                       "allreduce_bytes_per_step": wire,
                       "grads_in_flat_bucket": bool(all(arena.holds(gp[k].grad) for k in dp.GRAD_KEYS)),
                       "parallelism": (f"dp{world} (camera per GPU; per step {wire} B exchanged per GPU over RCCL: "
                                       + ("geometry grads all-reduced, SH grads all-gathered as factors)" if exchange is not None
                                          else "one all-reduce of the flat gradient bucket)")) if world > 1 else "single GPU"},
            "roofline": roofline,
            "roofline_valu": roofline_valu,
            "frame_roofline": frame_roofline,
            "strict_index_parity": strict,
            "eager_drop_in": eager_drop_in,
            "extra_workloads": extras,
            "multi_gpu": multi,
            "stages": stages,
            "other_ms_torch_postops_autograd_host": round(1e3 * elapsed / args.steps - gpu_stage_ms, 4),
        }
        if world == 1 and not args.no_cpu_baseline:
            dns.set_grad_arena(None)
            dns.set_sh_exchange(None)
            try:
                res["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_crop)
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
