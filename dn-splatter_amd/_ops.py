"""Tensor-level wrappers over the C ABI and the two autograd nodes everything else is built from.

Graph shape (same cut as gsplat 1.0.0, so the tensors dn-splatter reads back keep their meaning):

    params --_ProjectFn--> means2d, depths, conics, splats --_RasterFn--> render, alphas
              (stage 1/5)        ^ info["means2d"]: .grad / .absgrad            (stages 2-4)

``_ProjectFn`` fuses what gsplat splits into fully_fused_projection + spherical_harmonics and what
dn_model.py:543-560 does in torch; ``_RasterFn`` fuses isect_tiles + sort + rasterize_to_pixels.
"""
from __future__ import annotations

import ctypes
import math
import os
from dataclasses import dataclass
from typing import Dict, Optional

import torch
from torch import Tensor

from . import _lib
from ._lib import BinArgs, Camera, DnPost, ProjGrads, ProjOut, RasterArgs, Scene, RECORD_FLOATS


def _ptr(t: Optional[Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream() -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_gpu(t: Tensor, name: str) -> None:
    if not t.is_cuda:
        raise _lib.DnsplatError(
            f"{name} is on {t.device}: the dn-splatter_amd renderer runs only on the GPU through libdnsplat.so "
            "(there is no CPU fallback; the CPU oracle under oracle/ is test infrastructure)")


def _f32c(t: Tensor, name: str) -> Tensor:
    _need_gpu(t, name)
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    return t.contiguous()


# --------------------------------------------------------------------------------------------------
# configuration records


@dataclass(frozen=True)
class ProjCfg:
    width: int
    height: int
    tile_size: int = 16
    eps2d: float = 0.3
    near_plane: float = 0.01
    far_plane: float = 1e10
    radius_clip: float = 0.0
    antialiased: bool = False
    scales_are_log: bool = False
    opacities_are_logit: bool = False
    sh_degree: int = -1          # -1: direct colours
    with_depth: bool = False
    with_normals: bool = False
    want_normals_world: bool = False
    colors_are_logit: bool = False   # direct colours: sigmoid() inside the kernel (dn_model.py:491-492, config.sh_degree == 0)
    skip_culled_records: bool = False   # dnsplat_proj_out.skip_culled_records: culled Gaussians' records are left unwritten
    tight_tiles: bool = False    # dnsplat_camera.tight_tiles: tile counts over the alpha >= 1/255 box instead of gsplat's 3-sigma box
    split_colours: bool = False  # dnsplat_proj_out.phase 1 + 2: the SH colours on a side stream, beside the binning kernels

    @property
    def tiles(self):
        return math.ceil(self.width / self.tile_size), math.ceil(self.height / self.tile_size)


class _Buffers:
    """Grow-only per-device scratch (binning workspace, pinned counters) so steady-state frames do not
    touch the allocator."""

    def __init__(self):
        self.ws: Dict[torch.device, Tensor] = {}
        self.pinned: Dict[torch.device, Tensor] = {}
        self.ring: Dict[tuple, dict] = {}
        self.side: Dict[tuple, "torch.cuda.Stream"] = {}
        self.static_cap: Dict[tuple, int] = {}
        self.n_max: Dict[tuple, Tensor] = {}     # "static" bin policy: running maximum of n_isects per (device, stream) + frame size key
        self.capacity_hint: Dict[tuple, int] = {}

    @staticmethod
    def _key(device):
        # one set of scratch per (device, stream): frames rendered concurrently on different HIP streams
        # (model.get_outputs_batch) must not share the binning workspace or the pinned counter
        return (device, torch.cuda.current_stream(device).cuda_stream)

    def side_stream(self, device) -> "torch.cuda.Stream":
        """The stream work that may overlap the current stream's is put on (one per device and current stream)."""
        key = self._key(device)
        st = self.side.get(key)
        if st is None:
            # lowest priority: what runs there is meant to fill the gaps the current stream leaves, not to compete with it
            prio = int(os.environ.get("DNSPLAT_SIDE_PRIORITY", "1"))
            st = torch.cuda.Stream(device, priority=prio)
            self.side[key] = st
        return st

    def workspace(self, device, nbytes: int) -> Tensor:
        key = self._key(device)
        cur = self.ws.get(key)
        if cur is None or cur.numel() < nbytes:
            cur = torch.empty(int(nbytes * 1.25) + 1024, dtype=torch.uint8, device=device)
            self.ws[key] = cur
        return cur

    def pinned_i64(self, device) -> Tensor:
        key = self._key(device)
        cur = self.pinned.get(key)
        if cur is None:
            cur = torch.zeros(1, dtype=torch.int64).pin_memory()
            self.pinned[key] = cur
        return cur

    RING = 8

    def ring_slot(self, device):
        """("deferred" bin policy) the next (pinned int64 [1], event) pair of a small per-(device, stream) ring: the count of a
        frame is copied into its own slot, so several frames can be in flight before the host looks at any of them."""
        key = self._key(device)
        ring = self.ring.get(key)
        if ring is None:
            mem = torch.zeros(self.RING, dtype=torch.int64).pin_memory()
            ring = {"mem": mem, "events": [torch.cuda.Event() for _ in range(self.RING)], "next": 0, "pending": []}
            self.ring[key] = ring
        i = ring["next"]
        ring["next"] = (i + 1) % self.RING
        slot = ring["mem"][i:i + 1]
        # the slot about to be reused must have been looked at — whoever owns it, also when counts were resolved out of order
        # through Binning.n_isects — and at most RING - 1 frames are ever unverified
        for p in [p for p in ring["pending"] if p.slot.data_ptr() == slot.data_ptr()]:
            p.resolve()
        while len(ring["pending"]) >= self.RING - 1:
            ring["pending"][0].resolve()
        return ring, slot, ring["events"][i]


BUFFERS = _Buffers()

# Optional flat gradient bucket (dp.GradArena): when set, stage 5 writes parameter gradients straight into
# it so that data-parallel training all-reduces ONE buffer without flatten copies.
GRAD_ARENA = None


def set_grad_arena(arena) -> None:
    global GRAD_ARENA
    GRAD_ARENA = arena


# Optional compact SH-gradient exchange (dp.ShFactorExchange): when set, stage 5 hands out the 6 factors per
# Gaussian instead of the 48 coefficient gradients; dp.allreduce_gradients all-gathers them and rebuilds the sum.
SH_EXCHANGE = None


def set_sh_exchange(exchange) -> None:
    global SH_EXCHANGE
    SH_EXCHANGE = exchange


def _grad_like(param: Tensor) -> Tensor:
    """Where stage 5 writes the gradient of ``param``: its slice of the flat bucket, unless ``param.grad`` already IS that
    slice (zero_grad(set_to_none=False), or a second camera accumulated into the same step).  In that case autograd will
    run ``param.grad += new``: had the kernel written into the bucket, old and new gradient would be the same memory and
    the sum would come out as 2 x new.  A fresh tensor keeps accumulation correct (the sum still lands in the bucket)."""
    a = GRAD_ARENA
    if a is not None and not a.in_use(param):
        t = a.take(param)
        if t is not None:
            return t.view(param.shape)
    if a is not None and a.in_use(param):
        a.invalidate_sh_state()        # autograd is about to `+=` into the bucket: its rows are no longer what the kernel last wrote
    return torch.empty_like(param)

# "sync": read n_isects back before emitting (one host round-trip per frame, what gsplat does).
# "capacity": size the intersection buffers from the previous frames (x1.25), enqueue everything,
#             then verify n_isects <= capacity while the compositing kernel already runs; an overflow
#             re-runs the emit+composite with exact sizes.  No GPU bubble, always correct.
# "deferred": as "capacity", but the host does not wait for the count at all: it is copied into a pinned ring slot and
#             verified as soon as it has arrived (the frame's own backward, the next frame's binning, or whoever asks for
#             n_isects first — whichever comes first, none of them blocks in the steady state).  The host can run a whole
#             frame ahead of the GPU.  An overflow cannot be repaired after the fact: it RAISES (never silently wrong) and
#             enlarges the capacity, so that re-running the step succeeds.  The first frame of a size runs in "sync" mode.
# "static":   for frames captured into a HIP graph (graph.GraphedStep): the capacity is whatever earlier eager frames established
#             and NOTHING on the host looks at the count; the device keeps a running maximum (dnsplat_bin_args.n_isects_max)
#             that static_overflow() reads on demand.
BIN_POLICY = {"mode": "sync"}


def set_bin_policy(mode: str) -> None:
    if mode not in ("sync", "capacity", "deferred", "static"):
        raise ValueError(mode)
    BIN_POLICY["mode"] = mode


class _PendingCount:
    """The intersection count of one frame binned under the "deferred" policy, until the host has looked at it."""

    def __init__(self, ring, slot, event, capacity, key):
        self.ring, self.slot, self.event, self.capacity, self.key = ring, slot, event, capacity, key
        self.n: Optional[int] = None
        self.overflowed = False

    def ready(self) -> bool:
        return self.n is not None or self.event.query()

    def resolve(self) -> int:
        """Waits for the count if it has not arrived yet, verifies it, returns it.  Raises on overflow — on EVERY call for a frame
        that overflowed (its lists are truncated: nobody may read its count or its list as if it were complete)."""
        if self.n is None:
            self.event.synchronize()
            self.n = int(self.slot.item())
            if self in self.ring["pending"]:
                self.ring["pending"].remove(self)
            hint = BUFFERS.capacity_hint.get(self.key, 0)
            BUFFERS.capacity_hint[self.key] = max(hint, int(self.n * 1.25) + 4096)
            self.overflowed = self.n > self.capacity
        if self.overflowed:
            raise _lib.DnsplatError(
                f"tile binning: {self.n} intersections exceed the capacity {self.capacity} guessed from earlier frames "
                "(bin policy 'deferred' verifies the count after the fact): the outputs and gradients of that frame are "
                "invalid. The capacity has been enlarged; run the step again (or use set_bin_policy('capacity'), which "
                "repairs an overflow itself at the price of one host wait per frame)")
        return self.n


def static_overflow(device, stream=None) -> Optional[int]:
    """ "static" bin policy: the largest intersection count a frame binned on ``stream`` (default: the current one) produced, if
    it exceeded the capacity the frames of that size ran with (their lists were then truncated), else None.  Synchronises.
    An overflow is REPORTED ONCE and repaired for whoever captures next: the capacity guess of that frame size is raised to
    1.25 x the count seen, the sticky device maximum and the recorded static capacity are cleared — a re-capture after the error
    gets buffers that fit (before, a re-capture reused the same guess and overflowed again)."""
    skey = (device, (torch.cuda.current_stream(device) if stream is None else stream).cuda_stream)
    worst = None
    for k, t in list(BUFFERS.n_max.items()):
        if k[:2] != skey:
            continue
        n = int(t.item())
        if n > BUFFERS.static_cap.get(k, 0):
            worst = n if worst is None else max(worst, n)
            hkey = k[2:]                                   # (device, N, width, height): the key of the capacity guesses
            BUFFERS.capacity_hint[hkey] = max(BUFFERS.capacity_hint.get(hkey, 0), int(n * 1.25) + 4096)
            t.zero_()
            BUFFERS.static_cap.pop(k, None)
    return worst


def forget_capacity_guesses(device=None) -> None:
    """Drops the capacity guesses earlier frames left behind — for ``device`` or for all: the Gaussian set changed size
    (densify.after_refinement), the next frame of a size starts in "sync".  The "static" policy's per-stream records (running
    maxima, static capacities) are NOT touched: the running maximum is a device word that the captured bin kernels of a live
    graph.GraphedStep still write on every replay, and this dict holds the reference that keeps it allocated — they go
    with their step (GraphedStep.close -> forget_static)."""
    for k in [k for k in BUFFERS.capacity_hint if device is None or k[0] == device]:
        del BUFFERS.capacity_hint[k]


def carry_capacity_guesses(device, n_old: int, n_new: int, slack: float = 1.1) -> int:
    """The Gaussian set changed size (densify.after_refinement): instead of forgetting what the frames of the old size needed —
    which costs the next capture a round of eager frames over every pose just to size its buffers again — scale each guess by
    n_new / n_old x ``slack`` and file it under the new size.  A guess is a capacity, not a promise: a frame that outgrows it is
    reported by the bin policy in force (``GraphedStep.check()`` raises and enlarges it).  Returns the number of guesses carried."""
    carried = 0
    for k in [k for k in BUFFERS.capacity_hint if (device is None or k[0] == device) and k[1] == n_old]:
        hint = BUFFERS.capacity_hint.pop(k)
        nk = (k[0], n_new) + tuple(k[2:])
        # never scaled DOWN: culling removes low-opacity, small-footprint Gaussians, so the intersections fall less than N does, and
        # an under-estimate would make the next capture's warm-up frame overflow instead of sizing itself
        ratio = max(1.0, n_new / max(n_old, 1))
        BUFFERS.capacity_hint[nk] = max(BUFFERS.capacity_hint.get(nk, 0), int(hint * ratio * slack) + 4096)
        carried += 1
    return carried


def forget_static(device, stream) -> None:
    """Drops the "static" policy's per-stream records (running maxima, capacities) of ``stream``: called when the GraphedStep that
    owned the stream is discarded, so that they do not accumulate across re-captures."""
    skey = (device, stream.cuda_stream)
    for d in (BUFFERS.n_max, BUFFERS.static_cap):
        for k in [k for k in d if k[:2] == skey]:
            del d[k]
    for d in (BUFFERS.ws, BUFFERS.pinned, BUFFERS.ring, BUFFERS.side):
        d.pop(skey, None)


def verify_pending_counts(device, block: bool = False) -> None:
    """Looks at every deferred intersection count of the current stream that has arrived (all of them with ``block``)."""
    ring = BUFFERS.ring.get(_Buffers._key(device))
    if ring is None:
        return
    for p in list(ring["pending"]):
        if block or p.ready():
            p.resolve()


# --------------------------------------------------------------------------------------------------
# stage 1 / 5


def _scene_struct(N, means, quats, scales, opacities, cfg: ProjCfg, sh0, sh0_stride, shN, shN_stride, sh_K, colors):
    s = Scene()
    s.N = N
    s.means, s.quats, s.scales, s.opacities = _ptr(means), _ptr(quats), _ptr(scales), _ptr(opacities)
    s.scales_are_log = int(cfg.scales_are_log)
    s.opacities_are_logit = int(cfg.opacities_are_logit)
    s.sh_degree = cfg.sh_degree
    s.sh_K = sh_K
    s.sh0, s.sh0_stride = _ptr(sh0), sh0_stride
    s.shN, s.shN_stride = _ptr(shN), shN_stride
    s.colors = _ptr(colors)
    s.n_colors = 0 if colors is None else colors.shape[-1]
    s.colors_are_logit = int(cfg.colors_are_logit and colors is not None)
    return s


def _camera_struct(viewmat, K, normal_frame, cfg: ProjCfg):
    c = Camera()
    c.viewmat, c.K, c.normal_frame = _ptr(viewmat), _ptr(K), _ptr(normal_frame)
    c.width, c.height, c.tile_size = cfg.width, cfg.height, cfg.tile_size
    c.eps2d, c.near_plane, c.far_plane, c.radius_clip = cfg.eps2d, cfg.near_plane, cfg.far_plane, cfg.radius_clip
    c.antialiased = int(cfg.antialiased)
    c.tight_tiles = int(cfg.tight_tiles)
    return c


class _ProjectFn(torch.autograd.Function):
    """Stage 1 forward / stage 5 backward.  SH coefficients come either as one tensor ``coeffs``
    [N,K,3] (gsplat layout, dn_model.py:466-468 concatenates it) or split ``sh0`` [N,3] +
    ``shN`` [N,K-1,3] (the model's own features_dc / features_rest — no 192 B/Gaussian cat copy)."""

    @staticmethod
    def forward(ctx, means, quats, scales, opacities, coeffs, sh0, shN, colors, viewmat, K, normal_frame, cfg: ProjCfg,
                saturation_flag=None, side=None):
        means = _f32c(means, "means"); quats = _f32c(quats, "quats"); scales = _f32c(scales, "scales")
        opacities = _f32c(opacities, "opacities")
        viewmat = _f32c(viewmat, "viewmats"); K = _f32c(K, "Ks")
        N = means.shape[0]
        dev = means.device
        sh_K = 0
        p_sh0 = p_shN = None
        s0 = sN = 0
        if cfg.sh_degree >= 0:
            if coeffs is not None:
                coeffs = _f32c(coeffs, "colors")
                sh_K = coeffs.shape[1]
                p_sh0, s0 = coeffs, 3 * sh_K
                p_shN, sN = coeffs.view(-1)[3:], 3 * sh_K
            else:
                sh0 = _f32c(sh0, "features_dc")
                sh_K = 1
                p_sh0, s0 = sh0, 3
                if shN is not None and shN.shape[1] > 0:
                    shN = _f32c(shN, "features_rest")
                    sh_K = 1 + shN.shape[1]
                    p_shN, sN = shN, 3 * (sh_K - 1)
            if sh_K < (cfg.sh_degree + 1) ** 2:
                raise ValueError(f"sh_degree={cfg.sh_degree} needs {(cfg.sh_degree + 1) ** 2} bases, got {sh_K}")
        elif colors is not None:
            colors = _f32c(colors, "colors")
        if normal_frame is not None:
            normal_frame = _f32c(normal_frame, "normal_frame")

        # one launch per camera of the batch (C = 1 in training, dn_model.py:421; C > 1 for the batched render loops of the
        # offline consumers): per-camera outputs are the rows of [C,N,...] tensors, the records of camera c are rows
        # c*N .. (c+1)*N of one [C*N,16] buffer — the layout dnsplat_bin_* / dnsplat_raster_* take for a batch
        viewmat = viewmat.reshape(-1, 4, 4)
        K = K.reshape(-1, 3, 3)
        C = viewmat.shape[0]
        if normal_frame is not None:
            normal_frame = normal_frame.reshape(-1, 12)
        radii = torch.empty(C, N, dtype=torch.int32, device=dev)
        tiles = torch.empty(C, N, dtype=torch.int32, device=dev)
        # tight tile boxes: gsplat's count for the caller (info["tiles_per_gauss"]), the tight one for the binning
        tiles_bin = torch.empty(C, N, dtype=torch.int32, device=dev) if cfg.tight_tiles else None
        tile_boxes = torch.empty(C, N, 2, dtype=torch.int32, device=dev)      # (first tile, width) of the box the binning walks
        means2d = torch.empty(C, N, 2, dtype=torch.float32, device=dev)
        depths = torch.empty(C, N, dtype=torch.float32, device=dev)
        conics = torch.empty(C, N, 3, dtype=torch.float32, device=dev)
        comp = torch.empty(C, N, dtype=torch.float32, device=dev) if cfg.antialiased else None
        splats = torch.empty(C * N, RECORD_FLOATS, dtype=torch.float32, device=dev)
        nworld = torch.empty(C, N, 3, dtype=torch.float32, device=dev) if cfg.want_normals_world else None

        scene = _scene_struct(N, means, quats, scales, opacities, cfg, p_sh0, s0, p_shN, sN, sh_K, colors)
        # phase 1 here, phase 2 (the SH colours: most of the bytes) on a side stream while this stream goes on to the binning
        # kernels, which leave most of the chip idle; whoever reads the colour channels of the records waits for side["colours_ready"]
        split = bool(cfg.split_colours and side is not None and cfg.sh_degree >= 0)
        outs = []
        for c in range(C):
            cam = _camera_struct(viewmat[c], K[c], None if normal_frame is None else normal_frame[c], cfg)
            out = ProjOut()
            out.radii, out.means2d, out.depths, out.conics = _ptr(radii[c]), _ptr(means2d[c]), _ptr(depths[c]), _ptr(conics[c])
            out.compensations = _ptr(comp[c]) if comp is not None else None
            out.tiles_per_gauss, out.splats = _ptr(tiles[c]), _ptr(splats[c * N:])
            out.tiles_bin = _ptr(tiles_bin[c]) if tiles_bin is not None else None
            out.tile_boxes = _ptr(tile_boxes[c])
            out.normals_world = _ptr(nworld[c]) if nworld is not None else None
            out.with_depth_channel = int(cfg.with_depth)
            out.with_normal_channels = int(cfg.with_normals)
            out.saturation_flag = _ptr(saturation_flag)      # one word for all cameras of the batch (zeroed by camera_prepare)
            out.skip_culled_records = int(cfg.skip_culled_records)
            out.phase = 1 if split else 0
            _lib.run("dnsplat_project_fwd", _lib.lib().dnsplat_project_fwd, ctypes.byref(scene), ctypes.byref(cam), ctypes.byref(out), _stream())
            outs.append((cam, out))
        if split:
            main = torch.cuda.current_stream(dev)
            s2 = BUFFERS.side_stream(dev)
            s2.wait_stream(main)
            with torch.cuda.stream(s2):
                for cam, out in outs:
                    out.phase = 2
                    _lib.run("dnsplat_project_fwd_colours", _lib.lib().dnsplat_project_fwd, ctypes.byref(scene), ctypes.byref(cam),
                             ctypes.byref(out), _stream())
                ev = torch.cuda.Event()
                ev.record(s2)
            side["colours_ready"] = ev

        ctx.cfg = cfg
        ctx.sh_K = sh_K
        ctx.layout = "cat" if coeffs is not None else ("split" if cfg.sh_degree >= 0 else "colors")
        ctx.save_for_backward(means, quats, scales, opacities, coeffs, sh0, shN, colors, viewmat, K, normal_frame, radii, splats)
        ctx.set_materialize_grads(False)
        empty = torch.empty(0, device=dev)
        if tiles_bin is None:
            tiles_bin = torch.empty(0, dtype=torch.int32, device=dev)
        # ONE call: every call of mark_non_differentiable replaces the set of the previous one (normals_world with a grad_fn
        # would keep the frame's autograd graph alive through gauss_params["normals"])
        ctx.mark_non_differentiable(*([radii, tiles, tiles_bin, tile_boxes] + ([nworld] if nworld is not None else [])))
        return (means2d, depths, conics, comp if comp is not None else empty, splats, radii, tiles,
                nworld if nworld is not None else empty, tiles_bin, tile_boxes)

    @staticmethod
    def backward(ctx, v_means2d, v_depths, v_conics, v_comp, v_splats, _r, _t, _n, _tb, _bx):
        means, quats, scales, opacities, coeffs, sh0, shN, colors, viewmat, K, normal_frame, radii, splats_fwd = ctx.saved_tensors
        cfg: ProjCfg = ctx.cfg
        N = means.shape[0]
        C = viewmat.shape[0]
        dev = means.device
        if v_splats is None:
            v_splats = torch.zeros(C * N, RECORD_FLOATS, dtype=torch.float32, device=dev)
        v_splats = v_splats.contiguous()
        # The screen-space gradient reaches this node in two parts that are ALWAYS added: columns 0-1 of the gradient records (what
        # the compositing backward accumulated and did not hand to autograd separately, _hand_over_means2d_grad) and whatever
        # autograd delivers for means2d itself (terms the caller hung on info["means2d"], the legacy pass fed with non-detached
        # xys, or — when means2d was not retained — the compositing term, whose record columns were then cleared).  The kernel
        # takes g.v_means2d INSTEAD of the record's columns, hence the sum here.  No state outside the autograd graph is involved:
        # records summed over two compositing calls, or a backward that stops at means2d, cannot be double counted.
        v_m2d = None
        if v_means2d is not None:
            v_m2d = (v_means2d.reshape(C * N, 2) + v_splats[:, 0:2]).reshape(C, N, 2).contiguous()
        v_dep = v_depths.reshape(C, N).contiguous() if v_depths is not None else None
        v_con = v_conics.reshape(C, N, 3).contiguous() if v_conics is not None else None
        v_cmp = v_comp.reshape(C, N).contiguous() if (v_comp is not None and cfg.antialiased) else None
        sh_K = ctx.sh_K
        need = ctx.needs_input_grad
        total = None      # running sum over the cameras of a batch (C > 1); a single camera writes its outputs directly

        for c in range(C):
            v_means = _grad_like(means) if C == 1 else torch.empty_like(means)
            v_quats = _grad_like(quats) if C == 1 else torch.empty_like(quats)
            v_scales = _grad_like(scales) if C == 1 else torch.empty_like(scales)
            v_opac = _grad_like(opacities) if C == 1 else torch.empty_like(opacities)
            p_sh0 = p_shN = None
            s0 = sN = 0
            g = ProjGrads()
            v_coeffs = v_sh0 = v_shN = v_colors = None
            if ctx.layout == "cat":
                p_sh0, s0 = coeffs, 3 * sh_K
                p_shN, sN = coeffs.view(-1)[3:], 3 * sh_K
                v_coeffs = torch.empty_like(coeffs)
                g.v_sh0, g.v_sh0_stride = _ptr(v_coeffs), 3 * sh_K
                g.v_shN, g.v_shN_stride = _ptr(v_coeffs.view(-1)[3:]), 3 * sh_K
            elif ctx.layout == "split":
                p_sh0, s0 = sh0, 3
                v_sh0 = _grad_like(sh0) if C == 1 else torch.empty_like(sh0)
                g.v_sh0, g.v_sh0_stride = _ptr(v_sh0), 3
                if sh_K > 1:
                    p_shN, sN = shN, 3 * (sh_K - 1)
                    v_shN = _grad_like(shN) if C == 1 else torch.empty_like(shN)
                    g.v_shN, g.v_shN_stride = _ptr(v_shN), 3 * (sh_K - 1)
            elif colors is not None:
                v_colors = torch.empty_like(colors)
                g.v_colors = _ptr(v_colors)

            scene = _scene_struct(N, means, quats, scales, opacities, cfg, p_sh0, s0, p_shN, sN, sh_K, colors)
            cam = _camera_struct(viewmat[c], K[c], None if normal_frame is None else normal_frame[c], cfg)
            fwd = ProjOut()
            fwd.with_depth_channel = int(cfg.with_depth)
            fwd.with_normal_channels = int(cfg.with_normals)
            vs_c = v_splats[c * N:(c + 1) * N]
            g.radii, g.v_splats = _ptr(radii[c]), _ptr(vs_c)
            g.v_means2d = _ptr(v_m2d[c]) if v_m2d is not None else None
            g.v_depths = _ptr(v_dep[c]) if v_dep is not None else None
            g.v_conics = _ptr(v_con[c]) if v_con is not None else None
            g.v_compensations = _ptr(v_cmp[c]) if v_cmp is not None else None
            g.v_means, g.v_quats, g.v_scales, g.v_opacities = _ptr(v_means), _ptr(v_quats), _ptr(v_scales), _ptr(v_opac)
            ex = SH_EXCHANGE
            # Only the model's own split layout (features_dc / features_rest are the leaf parameters dp.allreduce_gradients
            # rebuilds into), one camera per rank.  With the concatenated gsplat layout the coefficient gradient is an
            # intermediate autograd tensor that nobody could fill in afterwards, so the kernel writes the rows itself.
            if ex is not None and ctx.layout == "split" and sh_K == 16 and C == 1 and getattr(ex, "slices", 1) > 1:
                # dp.SlicedShExchange: the same entry point on K slices of the Gaussians (every row pointer advanced by g0, N = n_k),
                # each with its own mini slab.  Under capture (record_only) nothing is launched here: the argument structs stay with
                # the exchange, graph.GraphedDpStep issues launch k + all-gather k behind each replay; the geometry gradients are
                # then NOT handed to autograd (None): the launches write the bucket slices, which GraphedDpStep installs as .grad.
                slabs = ex.begin(N, dev, cfg.sh_degree, sh_K, means=means)
                # what the recorded launches read must stay allocated (graph-pool tensors of the captured backward).  NOT the gradient
                # tensors: they are slices of the bucket, which outlives the step — and autograd adopts a returned gradient as .grad
                # only while nobody else holds it (a second reference here made it clone v_sh0 / v_shN out of the bucket)
                launches, keep = [], [means, quats, scales, opacities, sh0, shN, viewmat, K, normal_frame, radii, vs_c, v_m2d, v_dep, v_con,
                                      v_cmp, slabs]
                if ex.record_only and (GRAD_ARENA is None or not all(GRAD_ARENA.holds(t) for t in (v_means, v_quats, v_scales, v_opac, v_sh0, v_shN))):
                    raise _lib.DnsplatError("the sliced exchange in recorded mode needs the gradients in a dp.GradArena (set_grad_arena) "
                                            "and .grad = None when the captured backward starts")
                for k, (g0, g1) in enumerate(ex.bounds):
                    sc_k = _scene_struct(g1 - g0, means[g0:g1], quats[g0:g1], scales[g0:g1], opacities[g0:g1], cfg, sh0[g0:g1], 3,
                                         shN[g0:g1], 3 * (sh_K - 1), sh_K, None)
                    g_k = ProjGrads()
                    g_k.radii, g_k.v_splats = _ptr(radii[c][g0:g1]), _ptr(vs_c[g0:g1])
                    g_k.v_means2d = _ptr(v_m2d[c][g0:g1]) if v_m2d is not None else None
                    g_k.v_depths = _ptr(v_dep[c][g0:g1]) if v_dep is not None else None
                    g_k.v_conics = _ptr(v_con[c][g0:g1]) if v_con is not None else None
                    g_k.v_compensations = _ptr(v_cmp[c][g0:g1]) if v_cmp is not None else None
                    g_k.v_means, g_k.v_quats = _ptr(v_means[g0:g1]), _ptr(v_quats[g0:g1])
                    g_k.v_scales, g_k.v_opacities = _ptr(v_scales[g0:g1]), _ptr(v_opac[g0:g1])
                    g_k.v_sh0, g_k.v_sh0_stride = _ptr(v_sh0[g0:g1]), 3
                    g_k.v_shN, g_k.v_shN_stride = _ptr(v_shN[g0:g1]), 3 * (sh_K - 1)
                    g_k.sh_factors, g_k.sh_grads_skip = _ptr(slabs[k]), 1
                    launches.append((ctypes.byref(sc_k), ctypes.byref(cam), ctypes.byref(fwd), ctypes.byref(g_k)))
                    keep += [sc_k, g_k]
                keep += [cam, fwd, scene]
                if ex.record_only:
                    ex.record(launches, keep)
                    v_means = v_quats = v_scales = v_opac = None
                else:
                    # eager: slab k's all-gather is queued right behind launch k, so it travels while slices k+1.. compute
                    works = []
                    for k, args in enumerate(launches):
                        _lib.run("dnsplat_project_bwd", _lib.lib().dnsplat_project_bwd, *args, _stream())
                        works.append(ex.gather_slice(k))
                    ex.works = works
                outs = [v_means, v_quats, v_scales, v_opac, v_coeffs, v_sh0, v_shN, v_colors]
                total = outs
                continue
            if ex is not None and ctx.layout == "split" and sh_K == 16 and C == 1:
                # The coefficient-gradient tensors are handed to autograd unwritten; dp.allreduce_gradients fills them.  The
                # factors come from their own small kernel so that their all-gather is already under way while the geometry
                # gradients are computed below.
                fac = ex.begin(N, dev, cfg.sh_degree, sh_K, means=means)
                own = ex.use_own_rows()
                if ex.packed:
                    # slabs of visible rows only: header, masks and block offsets from the forward's radii (two small launches),
                    # the rows from dnsplat_project_bwd itself
                    if ex.scratch is None or ex.scratch.numel() < (N + 63) // 64 or ex.scratch.device != dev:
                        ex.scratch = torch.empty((N + 63) // 64, dtype=torch.int32, device=dev)
                    _lib.run("dnsplat_visible_index", _lib.lib().dnsplat_visible_index, N, ex.packed_capacity(N), _ptr(radii[c]),
                             _ptr(viewmat[c]), _ptr(fac), _ptr(ex.scratch), _stream())
                    g.sh_packed = _ptr(fac)
                elif ex.deferred or own:
                    # a captured step (graph.GraphedDpStep): the exchange only starts behind the replay, so nothing is gained by
                    # having the slab early — dnsplat_project_bwd writes it from the values it holds anyway (one launch less)
                    g.sh_factors = _ptr(fac)
                else:
                    _lib.run("dnsplat_sh_factors", _lib.lib().dnsplat_sh_factors, N, _ptr(radii[c]), _ptr(viewmat[c]),
                             _ptr(splats_fwd), _ptr(vs_c), _ptr(fac), _stream())
                    ex.launch()
                if own:
                    # this camera's rows as on a single GPU, pre-scaled by 1 / world; the exchange adds the other cameras' shares
                    from . import dp as _dp
                    g.sh_grad_scale = ex.scale_override if ex.scale_override is not None else _dp.reduction_scale(_dp.world_size(ex.group))
                else:
                    g.sh_grads_skip = 1
            elif (ex is None and GRAD_ARENA is not None and GRAD_ARENA.sh_state is not None and ctx.layout == "split" and sh_K == 16
                  and C == 1 and GRAD_ARENA.holds(v_sh0) and GRAD_ARENA.holds(v_shN)):
                # the rows land in the flat bucket, whose zero rows are tracked: a Gaussian that is culled again is not re-zeroed
                g.sh_zero_state = _ptr(GRAD_ARENA.sh_state)
                g.zero_state_geometry = int(all(GRAD_ARENA.holds(t) for t in (v_means, v_quats, v_scales, v_opac)))
            _lib.run("dnsplat_project_bwd", _lib.lib().dnsplat_project_bwd, ctypes.byref(scene), ctypes.byref(cam), ctypes.byref(fwd),
                     ctypes.byref(g), _stream())
            outs = [v_means, v_quats, v_scales, v_opac, v_coeffs, v_sh0, v_shN, v_colors]
            if total is None:
                total = outs
            else:
                for acc, t in zip(total, outs):
                    if acc is not None:
                        acc.add_(t)
        return tuple(t if (t is not None and need[i]) else None for i, t in enumerate(total)) + (None, None, None, None, None, None)


def project(means, quats, scales, opacities, *, coeffs=None, sh0=None, shN=None, colors=None, viewmat, K,
            normal_frame=None, cfg: ProjCfg, saturation_flag: Optional[Tensor] = None, side: Optional[dict] = None):
    """``viewmat`` [4,4] or [C,4,4] (``K``, ``normal_frame`` alike) -> dict(means2d[C,N,2], depths[C,N], conics[C,N,3],
    compensations[C,N] | None, splats[C*N,16], radii[C,N], tiles_per_gauss[C,N], normals_world[C,N,3] | None)"""
    m2d, dep, con, comp, splats, radii, tiles, nworld, tiles_bin, tile_boxes = _ProjectFn.apply(
        means, quats, scales, opacities, coeffs, sh0, shN, colors, viewmat, K, normal_frame, cfg, saturation_flag, side)
    # tiles_per_gauss: gsplat's count (A.3), always.  tiles_bin: what dnsplat_bin_* must be given — the same tensor, or the count
    # over the tight boxes when cfg.tight_tiles (the flag travels WITH the counts: rasterize* take both from here)
    # "was it asked for", not "has it elements": with N == 0 every output is empty and still has to be there
    return dict(means2d=m2d, depths=dep, conics=con, compensations=comp if cfg.antialiased else None, splats=splats,
                radii=radii, tiles_per_gauss=tiles, normals_world=nworld if cfg.want_normals_world else None,
                tiles_bin=tiles_bin if cfg.tight_tiles else tiles, tight_tiles=bool(cfg.tight_tiles), tile_boxes=tile_boxes)


# --------------------------------------------------------------------------------------------------
# stage 2: binning


class Binning:
    """flatten_ids [capacity] int32 (first n_isects valid; entry = camera * N + gaussian), tile_offsets [C*T+1] int32.
    ``n_isects`` is known on return under the "sync" / "capacity" policies; under "deferred" reading it is what waits."""

    def __init__(self, flatten_ids: Tensor, tile_offsets: Tensor, n_isects, tile_width: int, tile_height: int,
                 n_cameras: int = 1, pending: Optional["_PendingCount"] = None, tile_ends: Optional[Tensor] = None,
                 n_dev: Optional[Tensor] = None):
        self.flatten_ids, self.tile_offsets = flatten_ids, tile_offsets
        # tile_ends [C*T] (fused path): the list of tile t is [tile_offsets[t], tile_ends[t]); tile_offsets of EMPTY tiles are then
        # not gsplat's (filled_offsets() gives those)
        self.tile_ends, self._n_dev = tile_ends, n_dev
        self._n, self.pending = n_isects, pending
        self.tile_width, self.tile_height, self.n_cameras = tile_width, tile_height, n_cameras

    @property
    def n_isects(self) -> int:
        if self.pending is not None:
            self._n = self.pending.resolve()      # raises (every time) if that frame overflowed: self.pending stays
            self.pending = None
        if self._n is None:                      # "static" policy: nobody has asked yet — read the device scalar (synchronises)
            self._n = int(self._n_dev.item())
            if self._n > self.flatten_ids.numel():
                raise _lib.DnsplatError(f"tile binning: {self._n} intersections exceed the static capacity {self.flatten_ids.numel()}")
        return self._n

    def filled_offsets(self) -> Tensor:
        """gsplat's isect_offsets [C*T+1]: an empty tile carries the offset of the next non-empty one."""
        if self.tile_ends is None:
            return self.tile_offsets
        # the kernels left n_isects in the entries of the empty tiles: a suffix minimum is the same number
        return torch.flip(torch.cummin(torch.flip(self.tile_offsets, [0]), 0).values, [0])

    @n_isects.setter
    def n_isects(self, v) -> None:
        self._n, self.pending = v, None

    def verify_if_ready(self) -> None:
        if self.pending is not None and self.pending.ready():
            _ = self.n_isects


def bin_tiles(means2d: Tensor, radii: Tensor, depths: Tensor, tiles: Tensor, width: int, height: int,
              tile_size: int, after_emit=None, n_cameras: int = 1, tight_splats: Optional[Tensor] = None,
              defer_ok: bool = False, want_ends: bool = False, tile_boxes: Optional[Tensor] = None) -> Binning:
    """Stage 2 over ``n_cameras`` stacked projections (inputs flattened to [C*N, ...]).  ``after_emit(binning)`` (optional)
    is called right after the emit/sort kernels are enqueued and BEFORE any host wait, so the caller can queue the
    compositing kernel behind them; in "capacity" mode it is called again if the capacity guess turned out too small."""
    lib = _lib.lib()
    dev = means2d.device
    N = radii.numel()
    tw, th = math.ceil(width / tile_size), math.ceil(height / tile_size)
    T = tw * th * n_cameras
    n_dev = torch.empty(1, dtype=torch.int64, device=dev)
    n_host = BUFFERS.pinned_i64(dev)
    key = (dev, N, width, height)

    def make_args(capacity, flatten_ids, tile_offsets):
        nbytes = lib.dnsplat_bin_workspace_bytes(N, capacity, T)
        ws = BUFFERS.workspace(dev, nbytes)
        a = BinArgs()
        a.N, a.n_cameras, a.width, a.height, a.tile_size = N, n_cameras, width, height, tile_size
        a.means2d, a.radii, a.depths, a.tiles_per_gauss = _ptr(means2d), _ptr(radii), _ptr(depths), _ptr(tiles)
        a.isect_capacity = capacity
        a.flatten_ids, a.tile_offsets = _ptr(flatten_ids), _ptr(tile_offsets)
        a.n_isects, a.n_isects_host = _ptr(n_dev), ctypes.c_void_p(n_host.data_ptr())
        a.workspace, a.workspace_bytes = _ptr(ws), ws.numel()
        # ``tiles`` were counted over the tight boxes (ProjCfg.tight_tiles): the emit kernel rebuilds them from the records
        a.splats, a.tight_tiles = _ptr(tight_splats), int(tight_splats is not None)
        # the fused path keeps its lists to itself: [start, end) per tile instead of gsplat's offsets (no fill launch)
        a.tile_ends, a.skip_offsets_fill = _ptr(tile_ends), int(tile_ends is not None)
        a.tile_boxes = _ptr(tile_boxes)          # the boxes ``tiles`` were counted over, from the same projection
        a.n_isects_max = _ptr(BUFFERS.n_max.get(_Buffers._key(dev) + key))
        return a, ws

    tile_offsets = torch.empty(T + 1, dtype=torch.int32, device=dev)
    tile_ends = torch.empty(T, dtype=torch.int32, device=dev) if want_ends else None
    mode = BIN_POLICY["mode"]
    if mode == "deferred" and not defer_ok:
        mode = "capacity"      # callers that hand n_isects / flatten_ids[:n] straight back (the drop-in calls) gain nothing from deferring
    hint = BUFFERS.capacity_hint.get(key, 0)
    if mode == "static" and not defer_ok:
        mode = "capacity"
    if mode == "static" and hint:
        # nothing on the host depends on the count (a frame captured into a HIP graph): the capacity is the hint, and the
        # device keeps a running maximum of n_isects that static_overflow() compares with it whenever the caller likes
        if _Buffers._key(dev) + key not in BUFFERS.n_max:
            BUFFERS.n_max[_Buffers._key(dev) + key] = torch.zeros(1, dtype=torch.int64, device=dev)
        capacity = hint
        sk = _Buffers._key(dev) + key
        BUFFERS.static_cap[sk] = min(BUFFERS.static_cap.get(sk, capacity), capacity)      # the smallest buffers any such frame got
        flatten_ids = torch.empty(max(capacity, 1), dtype=torch.int32, device=dev)
        args, _ = make_args(capacity, flatten_ids, tile_offsets)
        args.n_isects_host = None
        _lib.run("dnsplat_bin_prepare", _lib.lib().dnsplat_bin_prepare, ctypes.byref(args), _stream())
        _lib.run("dnsplat_bin_emit_sort", _lib.lib().dnsplat_bin_emit_sort, ctypes.byref(args), _stream())
        b = Binning(flatten_ids, tile_offsets, None, tw, th, n_cameras, tile_ends=tile_ends, n_dev=n_dev)
        if after_emit is not None:
            after_emit(b)
        return b
    if mode == "deferred" and hint:
        verify_pending_counts(dev)                     # earlier frames whose count has arrived meanwhile (no wait)
        ring, slot, ev = BUFFERS.ring_slot(dev)
        n_host = slot
        capacity = hint
        flatten_ids = torch.empty(max(capacity, 1), dtype=torch.int32, device=dev)
        args, _ = make_args(capacity, flatten_ids, tile_offsets)
        _lib.run("dnsplat_bin_prepare", _lib.lib().dnsplat_bin_prepare, ctypes.byref(args), _stream())
        ev.record()
        _lib.run("dnsplat_bin_emit_sort", _lib.lib().dnsplat_bin_emit_sort, ctypes.byref(args), _stream())
        pend = _PendingCount(ring, slot, ev, capacity, key)
        ring["pending"].append(pend)
        b = Binning(flatten_ids, tile_offsets, None, tw, th, n_cameras, pending=pend, tile_ends=tile_ends, n_dev=n_dev)
        if after_emit is not None:
            after_emit(b)
        return b
    if mode == "capacity" and hint:
        capacity = hint
        flatten_ids = torch.empty(max(capacity, 1), dtype=torch.int32, device=dev)
        args, ws0 = make_args(capacity, flatten_ids, tile_offsets)
        _lib.run("dnsplat_bin_prepare", _lib.lib().dnsplat_bin_prepare, ctypes.byref(args), _stream())
        ev = torch.cuda.Event()
        ev.record()
        _lib.run("dnsplat_bin_emit_sort", _lib.lib().dnsplat_bin_emit_sort, ctypes.byref(args), _stream())
        b = Binning(flatten_ids, tile_offsets, -1, tw, th, n_cameras, tile_ends=tile_ends, n_dev=n_dev)
        if after_emit is not None:
            after_emit(b)
        ev.synchronize()  # waits for the (early) depth sort only; compositing keeps running
        n = int(n_host.item())
        if n <= capacity:
            b.n_isects = n
            BUFFERS.capacity_hint[key] = max(hint, int(n * 1.25) + 4096)
            return b
        # guess too small: fall through and redo with the exact size
    else:
        # the N-sized front of the workspace has the same layout for every capacity, so the depth sort
        # done here stays valid for the emit below as long as the buffer is not re-allocated
        args, ws0 = make_args(hint, None, tile_offsets)
        _lib.run("dnsplat_bin_prepare", _lib.lib().dnsplat_bin_prepare, ctypes.byref(args), _stream())
        torch.cuda.current_stream().synchronize()
        n = int(n_host.item())
    capacity = n
    BUFFERS.capacity_hint[key] = max(hint, int(n * 1.25) + 4096)
    flatten_ids = torch.empty(max(capacity, 1), dtype=torch.int32, device=dev)
    args, ws1 = make_args(capacity, flatten_ids, tile_offsets)
    if ws1.data_ptr() != ws0.data_ptr():
        _lib.run("dnsplat_bin_prepare", _lib.lib().dnsplat_bin_prepare, ctypes.byref(args), _stream())
    _lib.run("dnsplat_bin_emit_sort", _lib.lib().dnsplat_bin_emit_sort, ctypes.byref(args), _stream())
    b = Binning(flatten_ids, tile_offsets, n, tw, th, n_cameras, tile_ends=tile_ends, n_dev=n_dev)
    if after_emit is not None:
        after_emit(b)
    return b


class LazyInfo(dict):
    """An ``info`` dict some of whose entries are only computed when read (``lazy``: key -> thunk).  The fused path under the
    "deferred" bin policy uses it for ``n_isects`` / ``flatten_ids[:n_isects]``, which need the intersection count on the
    host: reading them is what waits for it, rendering a frame is not."""

    def __init__(self, eager: dict, lazy: dict):
        super().__init__(eager)
        self._lazy = dict(lazy)
        for k in lazy:
            super().__setitem__(k, None)

    def _force(self, k):
        thunk = self._lazy.pop(k, None)
        if thunk is not None:
            super().__setitem__(k, thunk())

    def __getitem__(self, k):
        self._force(k)
        return super().__getitem__(k)

    def get(self, k, default=None):
        if k in self:
            return self[k]
        return default

    def items(self):
        for k in list(self._lazy):
            self._force(k)
        return super().items()

    def values(self):
        for k in list(self._lazy):
            self._force(k)
        return super().values()

    # dict(info), {**info}, info.copy() and pickling take CPython's dict-merge fast path for a dict subclass whose __iter__ is
    # dict's own, which reads the stored None of an entry nobody has asked for yet.  With __iter__ overridden they go
    # through keys() + __getitem__, i.e. through _force.
    def __iter__(self):
        return iter(list(super().keys()))

    def keys(self):
        return super().keys()          # a view, as for any dict (set operations on it keep working)

    def _forced(self) -> dict:
        for k in list(self._lazy):
            self._force(k)
        return {k: dict.__getitem__(self, k) for k in super().keys()}

    def copy(self):
        return self._forced()

    def __reduce__(self):
        return (dict, (self._forced(),))


def binning_status(b: Binning, n_entries: int) -> int:
    """The tile sort's status word of the frame that produced ``b`` (0 = fine; see dnsplat_bin_status_offset).  Reads the
    workspace of the current stream, i.e. call it before the next frame is binned there; synchronises."""
    ws = BUFFERS.ws.get(_Buffers._key(b.flatten_ids.device))
    if ws is None or b.n_isects <= 0:
        return 0
    off = _lib.lib().dnsplat_bin_status_offset(n_entries, b.flatten_ids.numel())
    return int(ws[off:off + 4].view(torch.int32).item())


def isect_ids(b: Binning, depths: Tensor) -> Tensor:
    """gsplat's 64-bit sorted keys (camera | tile << 32 | depth bits), rebuilt on demand for the info dict."""
    out = torch.empty(max(b.n_isects, 1), dtype=torch.int64, device=depths.device)
    _lib.run("dnsplat_bin_isect_ids", _lib.lib().dnsplat_bin_isect_ids, b.tile_width * b.tile_height, b.n_cameras, _ptr(b.tile_offsets),
             _ptr(b.flatten_ids), _ptr(depths.reshape(-1).contiguous()), _ptr(out), b.n_isects, _stream())
    return out[: b.n_isects]


# --------------------------------------------------------------------------------------------------
# stages 3/4: compositing



# Deterministic gradient scatter (DNSPLAT_DETERMINISTIC=1 / set_deterministic(True); a test / debug mode, about 3x the backward's time
# and 128 B per list entry of scratch): the compositing backward stores the row of every (half tile, splat) in its own slot
# (dnsplat_raster_args.det_partials) and dnsplat_det_reduce adds each Gaussian's rows up in list order in float64 — the same bits in
# every run, where the default path's fp32 atomics land in arrival order ("atomics-vs-deterministic-reduction mode").
DETERMINISTIC = {"on": os.environ.get("DNSPLAT_DETERMINISTIC", "0") == "1"}


def set_deterministic(on: bool) -> None:
    DETERMINISTIC["on"] = bool(on)


def _det_begin(a: "RasterArgs", b: "Binning", dev):
    """Deterministic mode: the zero-filled slot buffer for this launch (None when the mode is off)."""
    if not DETERMINISTIC["on"]:
        return None
    cap = max(int(b.flatten_ids.numel()), 1)
    part = torch.zeros(2, cap, RECORD_FLOATS, dtype=torch.float32, device=dev)
    a.det_partials, a.det_capacity = _ptr(part), cap
    return part


def _det_finish(part: Optional[Tensor], b: "Binning", v_splats: Tensor) -> None:
    if part is None:
        return
    dev = v_splats.device
    cap = part.shape[1]
    n_dev = b._n_dev if b._n_dev is not None else torch.tensor([b.n_isects], dtype=torch.int64, device=dev)
    ws = torch.empty(_lib.lib().dnsplat_det_workspace_bytes(cap), dtype=torch.uint8, device=dev)
    d = _lib.DetArgs()
    d.n_records, d.capacity = v_splats.shape[0], cap
    d.n_isects, d.flatten_ids, d.partials, d.v_splats = _ptr(n_dev), _ptr(b.flatten_ids), _ptr(part), _ptr(v_splats)
    d.workspace, d.workspace_bytes = _ptr(ws), ws.numel()
    _lib.run("dnsplat_det_reduce", _lib.lib().dnsplat_det_reduce, ctypes.byref(d), _stream())


def _hand_over_means2d_grad(means2d: Tensor, v_splats: Tensor):
    """gsplat's contract (dn_model.py:517-519): after ``info["means2d"].retain_grad()`` the screen-space gradient is found in
    ``info["means2d"].grad``.  Returning columns 0-1 of the gradient records as means2d's autograd gradient would make the
    retain_grad hook CLONE that strided view (a 13 us copy kernel per frame at 1 M Gaussians) although the projection
    backward reads the same numbers from the records anyway.  So with retain_grad() autograd gets no separate gradient for
    means2d (None: the records carry it) and the view itself is stored in ``.grad`` — added to whatever is already there, as
    the hook would.  WITHOUT retain_grad() the term goes back through autograd as usual, so that torch.autograd.grad(loss,
    info["means2d"]) and hooks registered on the tensor see it: a copy of the columns is returned and the columns themselves
    are cleared, because the projection backward always ADDS what autograd delivers for means2d to the records' columns
    (_ProjectFn.backward) — each term travels exactly one way, whatever else feeds the two tensors.
    Returns the gradient to hand to autograd for ``means2d``."""
    view = v_splats[:, 0:2].view(means2d.shape)
    if not means2d.retains_grad:
        if means2d.requires_grad:
            g = view.clone()
            v_splats[:, 0:2].zero_()
            return g
        return None
    means2d.grad = view if means2d.grad is None else means2d.grad + view
    return None


class _RasterFn(torch.autograd.Function):
    """Bins, then composites D channels.  ``means2d`` is an input only so that autograd routes the
    xy-gradient through the tensor dn-splatter calls retain_grad() on (dn_model.py:517-519); the
    kernels read xy from ``splats``."""

    @staticmethod
    def forward(ctx, means2d, splats, depths, radii, tiles, background, width, height, tile_size, D, ed_channel,
                xy_split, absgrad, holder):
        dev = splats.device
        C = means2d.shape[0] if means2d.dim() == 3 else 1      # cameras of the batch: means2d [C,N,2], splats [C*N,16]
        render = torch.empty(C, height, width, D, dtype=torch.float32, device=dev)
        alphas = torch.empty(C, height, width, dtype=torch.float32, device=dev)
        last_ids = torch.empty(C, height, width, dtype=torch.int32, device=dev)
        bg = _f32c(background, "background") if background is not None else None

        def composite(b: Binning):
            a = RasterArgs()
            a.n_cameras = C
            a.width, a.height, a.tile_size, a.D = width, height, tile_size, D
            a.splats, a.flatten_ids, a.tile_offsets = _ptr(splats), _ptr(b.flatten_ids), _ptr(b.tile_offsets)
            a.background = _ptr(bg)
            a.ed_channel = ed_channel
            a.render, a.alphas, a.last_ids = _ptr(render), _ptr(alphas), _ptr(last_ids)
            _lib.run("dnsplat_raster_fwd", _lib.lib().dnsplat_raster_fwd, ctypes.byref(a), _stream())

        tight = bool(holder is not None and holder.get("tight"))      # ``tiles`` were counted over the tight boxes
        b = bin_tiles(means2d.detach().reshape(-1, 2), radii.reshape(-1), depths.detach().reshape(-1), tiles.reshape(-1), width,
                      height, tile_size, after_emit=composite, n_cameras=C, tight_splats=splats.detach() if tight else None,
                      tile_boxes=holder.get("tile_boxes") if holder is not None else None)
        if holder is not None:
            holder["binning"] = b
        ctx.save_for_backward(means2d, splats, b.flatten_ids, b.tile_offsets, render, alphas, last_ids)
        ctx.bg = bg
        ctx.binning = b
        ctx.cfg = (width, height, tile_size, D, ed_channel, xy_split, absgrad, C)
        ctx.set_materialize_grads(False)
        return render, alphas

    @staticmethod
    def backward(ctx, v_render, v_alphas):
        means2d, splats, flatten_ids, tile_offsets, render, alphas, last_ids = ctx.saved_tensors
        width, height, tile_size, D, ed_channel, xy_split, absgrad, C = ctx.cfg
        N = splats.shape[0]                      # records of the whole batch (cameras x Gaussians)
        dev = splats.device
        v_splats = torch.zeros(N, RECORD_FLOATS, dtype=torch.float32, device=dev)
        if v_render is None and v_alphas is None:
            return (None, v_splats) + (None,) * 12
        if v_render is None:
            v_render = torch.zeros_like(render)
        v_render = v_render.contiguous()
        v_alphas = v_alphas.contiguous() if v_alphas is not None else None
        a = RasterArgs()
        a.n_cameras = C
        a.width, a.height, a.tile_size, a.D = width, height, tile_size, D
        a.splats, a.flatten_ids, a.tile_offsets = _ptr(splats), _ptr(flatten_ids), _ptr(tile_offsets)
        a.background = _ptr(ctx.bg)
        a.ed_channel = ed_channel
        a.render, a.alphas, a.last_ids = _ptr(render), _ptr(alphas), _ptr(last_ids)
        a.v_render, a.v_alphas = _ptr(v_render), _ptr(v_alphas)
        a.xy_split = xy_split
        a.v_splats = _ptr(v_splats)
        det = _det_begin(a, ctx.binning, dev)
        _lib.run("dnsplat_raster_bwd", _lib.lib().dnsplat_raster_bwd, ctypes.byref(a), _stream())
        _det_finish(det, ctx.binning, v_splats)
        if absgrad:
            # gsplat contract (dn_model.py:512, consumed by nerfstudio after_train via self.xys.absgrad)
            means2d.absgrad = v_splats[:, 14:16].reshape(means2d.shape)
        return (_hand_over_means2d_grad(means2d, v_splats), v_splats) + (None,) * 12


def rasterize(means2d, splats, depths, radii, tiles, *, background=None, width, height, tile_size=16, D,
              ed_channel=-1, xy_split=None, absgrad=False, holder=None, tight=False, tile_boxes=None):
    """-> render [C,H,W,D], alphas [C,H,W] for the C cameras of ``means2d`` [C,N,2] / ``splats`` [C*N,16].
    ``tiles`` / ``tight``: ``tiles_bin`` and ``tight_tiles`` of the project() result, always taken together (the binning walks the
    boxes the counts were taken over)."""
    if tight or tile_boxes is not None:
        holder = {} if holder is None else holder
        holder["tight"] = bool(tight)
        holder["tile_boxes"] = tile_boxes          # (first tile, width) per entry, of the same projection as ``tiles``
    if tile_size != 16:
        raise NotImplementedError("libdnsplat composites 16x16 tiles (dn_model.py:470-472 uses BLOCK_WIDTH = 16)")
    if xy_split is None:
        xy_split = D
    render, alphas = _RasterFn.apply(means2d, splats, depths, radii, tiles, background, width, height, tile_size, D,
                                     ed_channel, xy_split, absgrad, holder)
    return render, alphas


# The fused pass hands the forward's per-half-tile rectangle-test results to the backward (dnsplat_raster_args.keep_masks);
# DNSPLAT_KEEP_MASKS=0 makes the backward re-derive them (kernel A/B runs).
KEEP_MASKS = os.environ.get("DNSPLAT_KEEP_MASKS", "1") != "0"
# The fused get_outputs path keeps its tile lists to itself, so it bins over the tight tile boxes (dnsplat_camera.tight_tiles:
# ~1/3 fewer intersections on the benchmark scenes, same images and gradients).  DNSPLAT_TIGHT_TILES=0 (or setting this to
# False) makes it use gsplat's boxes, as the drop-in calls always do.
TIGHT_TILES = os.environ.get("DNSPLAT_TIGHT_TILES", "1") != "0"
# dnsplat_raster_args.saturation_flag for the fused path's backward (DNSPLAT_SATURATION_FLAG=0: always the clamping loop)
SATURATION_FLAG = os.environ.get("DNSPLAT_SATURATION_FLAG", "1") != "0"
# dnsplat_raster_args.zero_fill: the fused forward clears the gradient records its backward accumulates into
FORWARD_ZERO_FILL = os.environ.get("DNSPLAT_FORWARD_ZERO_FILL", "1") != "0"
# dnsplat_proj_out.skip_culled_records for the fused path (its records never leave the library; DNSPLAT_SKIP_CULLED_RECORDS=0: zero-filled)
SKIP_CULLED_RECORDS = os.environ.get("DNSPLAT_SKIP_CULLED_RECORDS", "1") != "0"
# the fused path's projection as two launches, the SH colour half on a side stream beside the binning kernels (ProjCfg.split_colours)
SPLIT_COLOURS = os.environ.get("DNSPLAT_SPLIT_COLOURS", "0") != "0"

# Measurement hook (bench.py's VALU roofline): a uint64 [8] device tensor makes the fused pass run the COUNTING instantiation of
# both compositing kernels, which tally list entries / splats walked / pairs evaluated / pairs blended / slots issued.
PAIR_COUNTERS: Optional[Tensor] = None

_BG7: Dict[torch.device, Tensor] = {}


def _bg7(device) -> Tensor:
    """Kernel-level background of the fused pass: none for rgb | depth, ones for the normal channels (the
    legacy gsplat.rasterize_gaussians default, dn_model.py:564-575 passes no background)."""
    t = _BG7.get(device)
    if t is None:
        t = torch.tensor([0.0, 0.0, 0.0, 0.0, 1.0, 1.0, 1.0], device=device)
        _BG7[device] = t
    return t


class _RasterDnFn(torch.autograd.Function):
    """Bins, composites the 7 fused channels AND applies dn-splatter's per-pixel post-ops in the same
    kernels (``dnsplat_dn_post``): returns the images ``get_outputs`` hands out — rgb, depth (filled),
    normal, accumulation, surface_normal — instead of the raw composite."""

    @staticmethod
    def forward(ctx, means2d, splats, depths, radii, tiles, bg_rgb, width, height, intr, absgrad, holder):
        dev = splats.device
        tight = bool(holder is not None and holder.get("tight"))
        ctx.saturation_flag = holder.get("saturation_flag") if holder is not None else None
        C = means2d.shape[0]                     # cameras of the batch; intr = [(fx, fy, cx, cy)] * C
        f32 = dict(dtype=torch.float32, device=dev)
        render = torch.empty(C, height, width, 7, **f32)
        alphas = torch.empty(C, height, width, **f32)
        last_ids = torch.empty(C, height, width, dtype=torch.int32, device=dev)
        rgb = torch.empty(C, height, width, 3, **f32)
        depth_raw = torch.empty(C, height, width, **f32)
        normal = torch.empty(C, height, width, 3, **f32)
        # per-camera maximum of the expected depth: zeroed by the frame's camera_prepare launch when it came with the saturation
        # flag, otherwise (and again if the capacity guess forces a re-run) by a fill in composite()
        sat = holder.get("saturation_flag") if holder is not None else None
        depth_max = DEPTH_MAX_OF.pop(sat.data_ptr(), None) if sat is not None else None
        fresh = {"depth_max": depth_max is not None and depth_max.numel() == C}
        if not fresh["depth_max"]:
            depth_max = torch.empty(C, **f32)
        depth_out = torch.empty(C, height, width, 1, **f32)
        surface_normal = torch.empty(C, height, width, 3, **f32)
        bg_rgb = _f32c(bg_rgb, "background")
        bg7 = _bg7(dev)
        dn = DnPost()
        dn.background_rgb, dn.rgb, dn.depth, dn.normal = _ptr(bg_rgb), _ptr(rgb), _ptr(depth_raw), _ptr(normal)
        dn.depth_max = _ptr(depth_max)
        counters = holder.get("pair_counters") if holder is not None else None
        if counters is None:
            counters = PAIR_COUNTERS
        keep = {}
        # gradient records of the backward (atomically accumulated, so they must start at zero): allocated here and cleared by the
        # forward kernel itself when a backward can follow (FORWARD_ZERO_FILL = False: torch.zeros in the backward)
        fill = {"v_splats": torch.empty(splats.shape[0], RECORD_FLOATS, **f32)
                if (FORWARD_ZERO_FILL and ctx.needs_input_grad[1] and counters is None) else None}

        def composite(b: Binning):
            if not fresh["depth_max"]:
                depth_max.zero_()
            fresh["depth_max"] = False
            ready = holder.get("colours_ready") if holder is not None else None
            if ready is not None:          # the projection's colour phase runs on a side stream (ProjCfg.split_colours)
                torch.cuda.current_stream(dev).wait_event(ready)
            a = RasterArgs()
            a.n_cameras = C
            if KEEP_MASKS:
                # the forward's rectangle-test ballots, one 64-bit word per (half tile, 64 list entries): the backward takes
                # them instead of re-testing (and re-gathering) every list entry
                stride = (b.flatten_ids.numel() >> 6) + C * b.tile_width * b.tile_height + 1
                keep["masks"], keep["stride"] = torch.empty(2 * stride, dtype=torch.int64, device=dev), stride
                a.keep_masks, a.keep_mask_stride = _ptr(keep["masks"]), stride
            a.width, a.height, a.tile_size, a.D = width, height, 16, 7
            a.splats, a.flatten_ids, a.tile_offsets = _ptr(splats), _ptr(b.flatten_ids), _ptr(b.tile_offsets)
            a.tile_ends = _ptr(b.tile_ends)
            a.background = _ptr(bg7)
            a.ed_channel = 3
            a.render, a.alphas, a.last_ids = _ptr(render), _ptr(alphas), _ptr(last_ids)
            a.dn = ctypes.pointer(dn)
            a.pair_counters = _ptr(counters)
            if fill["v_splats"] is not None:      # the backward's accumulation buffer, cleared by this launch on the way
                a.zero_fill, a.zero_fill_bytes = _ptr(fill["v_splats"]), fill["v_splats"].numel() * 4
            _lib.run("dnsplat_raster_fwd", _lib.lib().dnsplat_raster_fwd, ctypes.byref(a), _stream())

        b = bin_tiles(means2d.detach().reshape(-1, 2), radii.reshape(-1), depths.detach().reshape(-1), tiles.reshape(-1), width,
                      height, 16, after_emit=composite, n_cameras=C, tight_splats=splats.detach() if tight else None,
                      defer_ok=True, want_ends=True, tile_boxes=holder.get("tile_boxes") if holder is not None else None)
        for c in range(C):
            fx, fy, cx, cy = intr[c]
            _lib.run("dnsplat_dn_depth_normals", _lib.lib().dnsplat_dn_depth_normals, width, height, fx, fy, cx, cy,
                     _ptr(depth_raw[c]), _ptr(alphas[c]), _ptr(depth_max[c:]), _ptr(depth_out[c]), _ptr(surface_normal[c]), _stream())
        if holder is not None:
            holder["binning"] = b
        ctx.save_for_backward(means2d, splats, b.flatten_ids, b.tile_offsets, render, alphas, last_ids, bg_rgb)
        ctx.keep = keep
        ctx.binning = b
        ctx.v_splats = fill["v_splats"]
        ctx.cfg = (width, height, absgrad, C, counters)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(surface_normal)
        return rgb, depth_out, normal, alphas.unsqueeze(-1), surface_normal

    @staticmethod
    def backward(ctx, v_rgb, v_depth, v_normal, v_acc, _v_sn):
        means2d, splats, flatten_ids, tile_offsets, render, alphas, last_ids, bg_rgb = ctx.saved_tensors
        width, height, absgrad, C, counters = ctx.cfg
        N = splats.shape[0]
        dev = splats.device
        ctx.binning.verify_if_ready()     # "deferred" bin policy: the frame's count has normally arrived by now (never waits)
        # cleared by the forward launch of this frame; a second backward through the same graph gets a fresh one
        v_splats, ctx.v_splats = ctx.v_splats, None
        if v_splats is None:
            v_splats = torch.zeros(N, RECORD_FLOATS, dtype=torch.float32, device=dev)
        none = (None,) * 9
        if v_rgb is None and v_depth is None and v_normal is None and v_acc is None:
            return (None, v_splats) + none
        z = lambda t, c: torch.zeros(C, height, width, c, dtype=torch.float32, device=dev) if t is None else t.contiguous()  # noqa: E731
        v_rgb, v_depth, v_normal = z(v_rgb, 3), z(v_depth, 1), z(v_normal, 3)
        v_acc = v_acc.contiguous() if v_acc is not None else None
        dn = DnPost()
        dn.background_rgb = _ptr(bg_rgb)
        dn.v_rgb, dn.v_depth, dn.v_normal, dn.v_accumulation = _ptr(v_rgb), _ptr(v_depth), _ptr(v_normal), _ptr(v_acc)
        a = RasterArgs()
        a.n_cameras = C
        a.width, a.height, a.tile_size, a.D = width, height, 16, 7
        a.splats, a.flatten_ids, a.tile_offsets = _ptr(splats), _ptr(flatten_ids), _ptr(tile_offsets)
        a.tile_ends = _ptr(ctx.binning.tile_ends)
        a.background = _ptr(_bg7(dev))
        a.ed_channel = 3
        a.render, a.alphas, a.last_ids = _ptr(render), _ptr(alphas), _ptr(last_ids)
        a.xy_split = 4
        a.v_splats = _ptr(v_splats)
        a.dn = ctypes.pointer(dn)
        a.pair_counters = _ptr(counters)
        if ctx.keep:
            a.keep_masks, a.keep_mask_stride = _ptr(ctx.keep["masks"]), ctx.keep["stride"]
        # "no visible opacity above the alpha cap in this frame" (written by the projection): lets the kernel drop the clamp handling
        a.saturation_flag = _ptr(ctx.saturation_flag) if SATURATION_FLAG else None
        det = _det_begin(a, ctx.binning, dev) if counters is None else None
        _lib.run("dnsplat_raster_bwd", _lib.lib().dnsplat_raster_bwd, ctypes.byref(a), _stream())
        _det_finish(det, ctx.binning, v_splats)
        if absgrad:
            means2d.absgrad = v_splats[:, 14:16].reshape(means2d.shape)
        return (_hand_over_means2d_grad(means2d, v_splats), v_splats) + none


def rasterize_dn(means2d, splats, depths, radii, tiles, *, background_rgb, width, height, intrinsics, absgrad=True,
                 holder=None, tight=False, tile_boxes=None):
    """``intrinsics``: one (fx, fy, cx, cy) per camera of ``means2d`` [C,N,2].
    -> rgb[C,H,W,3], depth[C,H,W,1], normal[C,H,W,3], accumulation[C,H,W,1], surface_normal[C,H,W,3].
    ``holder["pair_counters"]`` (optional uint64 [8] device tensor) switches both compositing kernels to their measurement
    instantiation (bench.py's VALU roofline)."""
    if isinstance(intrinsics[0], (int, float)):
        intrinsics = [tuple(intrinsics)]
    if tight or tile_boxes is not None:    # ``tiles`` = tiles_bin (+ flag, + boxes): always taken together from the project() result
        holder = {} if holder is None else holder
        holder["tight"] = bool(tight)
        holder["tile_boxes"] = tile_boxes
    return _RasterDnFn.apply(means2d, splats, depths, radii, tiles, background_rgb, width, height, list(intrinsics), absgrad,
                             holder)


def camera_prepare(c2w: Tensor, fx: float, fy: float, cx: float, cy: float, with_normal_frame: bool = True, with_flag: bool = False,
                   n_depth_max: int = 0):
    """One-launch replacement of get_viewmat + intrinsics + normal frame (dn_model.py:475-479, 550-560).
    ``c2w`` [3,4] (or [1,3,4]) on the GPU -> viewmat[4,4], K[3,3], normal_frame[12] | None
    (+ with_flag: a zeroed int32 [1] device word, the frame's opacity-saturation flag for project(..., saturation_flag=);
    n_depth_max > 0: the flag is followed by that many zeroed floats, the per-camera depth maxima of the fused epilogue —
    ``flag.depth_max`` — so that the frame needs no fill launch for either)."""
    c2w = _f32c(c2w.reshape(-1)[:12], "camera_to_worlds")
    dev = c2w.device
    out = torch.empty(16 + 9 + 12, dtype=torch.float32, device=dev)
    viewmat, K, nf = out[:16], out[16:25], out[25:37]
    # the words the launch zeroes live in their own buffer: autograd saves viewmat / K, and a later in-place fill of depth_max
    # (a re-run after a capacity overflow) must not touch their version counter
    zeros = torch.empty(1 + n_depth_max, dtype=torch.float32, device=dev) if with_flag else None
    flag = zeros[0:1].view(torch.int32) if with_flag else None
    _lib.run("dnsplat_camera_prepare", _lib.lib().dnsplat_camera_prepare, _ptr(c2w), fx, fy, cx, cy, _ptr(viewmat), _ptr(K),
             _ptr(nf) if with_normal_frame else None, _ptr(flag), 1 + n_depth_max, _stream())
    if with_flag and n_depth_max:
        DEPTH_MAX_OF[flag.data_ptr()] = zeros[1:1 + n_depth_max]
        while len(DEPTH_MAX_OF) > 16:          # prepared cameras nobody rendered (an exception in between): oldest first
            DEPTH_MAX_OF.pop(next(iter(DEPTH_MAX_OF)))
    res = (viewmat.view(4, 4), K.view(3, 3), (nf if with_normal_frame else None))
    return res + (flag,) if with_flag else res


# zeroed depth_max words that came with a saturation flag (keyed by the flag's address; taken once by _RasterDnFn.forward)
DEPTH_MAX_OF: Dict[int, Tensor] = {}


def camera_prepare_batch(cameras, with_normal_frame: bool = True, with_flag: bool = False):
    """camera_prepare for a list of camera records (``camera_to_worlds`` [1,3,4], fx, fy, cx, cy) ->
    viewmats[C,4,4], Ks[C,3,3], normal_frames[C,12] | None (+ with_flag: ONE zeroed flag word for the whole batch)."""
    parts = [camera_prepare(c.camera_to_worlds, float(c.fx), float(c.fy), float(c.cx), float(c.cy), with_normal_frame,
                            with_flag=(with_flag and i == 0), n_depth_max=(len(cameras) if (with_flag and i == 0) else 0))
             for i, c in enumerate(cameras)]
    vm = torch.stack([p[0] for p in parts])
    K = torch.stack([p[1] for p in parts])
    nf = torch.stack([p[2] for p in parts]) if with_normal_frame else None
    return (vm, K, nf, parts[0][3]) if with_flag else (vm, K, nf)


class _PackFn(torch.autograd.Function):
    """(xys, conics, opacity, colors) -> records; front end of the legacy rasterize_gaussians call."""

    @staticmethod
    def forward(ctx, xys, conics, opacity, colors):
        xys = _f32c(xys, "xys"); conics = _f32c(conics, "conics")
        opacity = _f32c(opacity, "opacity"); colors = _f32c(colors, "colors")
        N, C = colors.shape
        splats = torch.empty(N, RECORD_FLOATS, dtype=torch.float32, device=xys.device)
        _lib.run("dnsplat_pack_splats", _lib.lib().dnsplat_pack_splats, N, _ptr(xys), _ptr(conics), _ptr(opacity.reshape(-1)), _ptr(colors), C,
                                                  _ptr(splats), _stream())
        ctx.C = C
        ctx.oshape = opacity.shape
        return splats

    @staticmethod
    def backward(ctx, v_splats):
        C = ctx.C
        return (v_splats[:, 0:2], v_splats[:, 2:5], v_splats[:, 5].reshape(ctx.oshape), v_splats[:, 6:6 + C])
