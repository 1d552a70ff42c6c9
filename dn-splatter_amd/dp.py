"""Multi-view data parallelism: one camera per GPU, one RCCL all-reduce of the splat gradients per step.

The reference renders a single camera per step (``dn_model.py:421``) and never exercises its inherited DDP
wrap (``dn_pipeline.py:122-128``).  BASELINE.json's multi-GPU configs (C4/C5) render 8 cameras per step, one
per GPU of a node: every rank holds the full set of Gaussians, renders its own camera, and the gradients of
the six optimised tensors named by ``get_gaussian_param_groups`` (``dn_model.py:388-402``; ``normals`` gets
no gradient) are averaged across ranks — DDP semantics, 236 B per Gaussian at SH degree 3.

MI355X mapping.  xGMI is point-to-point (7 links per GPU), so per-message latency and ring steps are paid
per collective: instead of six collectives (12/12/16/12/180/4 B per Gaussian) the backward kernel
(``dnsplat_project_bwd``) writes all six gradients straight into ONE flat fp32 bucket (`GradArena`) whose
slices are the ``.grad`` tensors, and a single in-place ``all_reduce`` runs over it — no flatten/unflatten
copies (that would be 2 x 236 MB of extra HBM traffic per step at 1 M Gaussians).  ``torch.distributed``
backend ``"nccl"`` is RCCL on ROCm; CPU tests use ``gloo``.
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, Optional, Tuple

import torch
import torch.distributed as dist
from torch import Tensor

# order = layout of the bucket (the parameter groups of dn_model.py:388-402 minus "normals"); the four geometry tensors
# come first so that they form one contiguous prefix when the SH gradients travel as factors (ShFactorExchange)
GRAD_KEYS = ("means", "scales", "quats", "opacities", "features_dc", "features_rest")
GEOMETRY_KEYS = GRAD_KEYS[:4]

# "mean": the collectives average (DDP semantics; RCCL's AVG).  "sum": they add, and the CALLER has scaled its loss (its image
# cotangents) by 1 / world_size — the same numbers up to rounding, but nothing in the exchange step multiplies by 1 / W any more: RCCL
# implements AVG as pre-multiply + sum, which on ONE rank still runs a kernel over the whole 44 B / Gaussian prefix (oneRankReduce:
# 0.32 ms at 5 M Gaussians, profiles/r06_frame_timeline_single_rank_rccl_c5_*.txt), where a one-rank in-place SUM is nothing at all.
REDUCTION = {"mode": "mean"}


def set_reduction(mode: str) -> None:
    if mode not in ("mean", "sum"):
        raise ValueError(mode)
    REDUCTION["mode"] = mode


def reduction_scale(w: int) -> float:
    """What a rank's own contribution is multiplied by inside the exchange step (own-camera SH rows, rebuilt rows)."""
    return 1.0 / w if REDUCTION["mode"] == "mean" else 1.0


class GradArena:
    """One flat fp32 buffer holding the gradients of all optimised tensors of a Gaussian set.

    ``_ops._ProjectFn.backward`` asks ``take(name, like)`` for its output tensors, so the kernels write
    directly into the bucket and autograd installs those views as ``param.grad``.

    Contract: ``param.grad`` should be ``None`` when a backward starts (``zero_grad(set_to_none=True)``).  If a ``.grad`` that
    already lives in the bucket is still set — gradient accumulation over several cameras, ``zero_grad(set_to_none=False)`` —
    the kernels write into a fresh tensor instead (``_ops._grad_like``) and autograd's in-place ``+=`` lands in the bucket:
    still correct, one extra pass over the gradients.  (Writing into the bucket in that case would alias old and new
    gradient and the sum would come out as 2 x new.)
    """

    def __init__(self, params: Dict[str, Tensor]):
        self.slices: Dict[str, Tuple[int, int, torch.Size]] = {}
        off = 0
        dev = None
        for k in GRAD_KEYS:
            p = params[k]
            assert p.dtype == torch.float32 and p.is_contiguous(), k
            self.slices[k] = (off, p.numel(), p.shape)
            off += p.numel()
            dev = p.device
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self._by_ptr = {params[k].data_ptr(): k for k in GRAD_KEYS}
        self._leaf = {k: params[k] for k in GRAD_KEYS}      # the leaves whose .grad the slices become
        # dnsplat_proj_grads.sh_zero_state: one bit per Gaussian, "the SH coefficient-gradient rows of this Gaussian are zero in the
        # bucket" — the projection backward then does not write the zero rows of a Gaussian that is culled again (28 % of the rows on
        # the benchmark scenes).  The bucket starts zero-filled, so every bit starts set.  CONTRACT: whoever else writes non-zero
        # values into the features_dc / features_rest slices (a collective over the bucket, an accumulation `+=` by autograd, user
        # code) calls invalidate_sh_state(); scaling or zeroing them in place keeps the bits valid.
        # OFF by default (DNSPLAT_SH_ZERO_STATE=1 turns it on): measured on MI355X (profiles/r06_ab_per_gaussian.txt) the skipped rows
        # turn the backward's one streaming store per workgroup into 16-byte pieces with holes — partial lines — and the kernel gets
        # 5-10 % SLOWER although it writes 10 % fewer bytes.
        n_gauss = params["features_dc"].shape[0]
        on = os.environ.get("DNSPLAT_SH_ZERO_STATE", "0") == "1"
        self.sh_state = (torch.full(((n_gauss + 63) // 64,), -1, dtype=torch.int64, device=dev)
                         if on and dev is not None and dev.type == "cuda" else None)

    def view(self, name: str) -> Tensor:
        off, n, shape = self.slices[name]
        return self.flat[off:off + n].view(shape)

    def take(self, like: Tensor) -> Optional[Tensor]:
        """Bucket slice for the gradient of parameter ``like`` (matched by storage address), or None."""
        name = self._by_ptr.get(like.data_ptr())
        if name is None or self.slices[name][1] != like.numel():
            return None
        return self.view(name)

    def in_use(self, like: Tensor) -> bool:
        """True when the bucket slice of parameter ``like`` (a leaf, or a view of one such as opacities.reshape(N)) is
        currently installed as that leaf's ``.grad`` — the backward must then not write into it (see the class docstring)."""
        name = self._by_ptr.get(like.data_ptr())
        if name is None:
            return False
        g = self._leaf[name].grad
        return g is not None and self.holds(g)

    def holds(self, t: Optional[Tensor]) -> bool:
        if t is None:
            return False
        a = self.flat.data_ptr()
        return a <= t.data_ptr() < a + self.flat.numel() * 4

    def bytes(self) -> int:
        return self.flat.numel() * 4

    def invalidate_sh_state(self) -> None:
        """Nothing is known about the SH gradient rows any more (see ``sh_state``): the next projection backward writes them all."""
        if self.sh_state is not None:
            self.sh_state.zero_()


class ShFactorExchange:
    """Compact exchange of the SH-coefficient gradients between the cameras of a data-parallel step.

    For one camera the gradient of a Gaussian's 16 x 3 SH coefficients is the outer product of the SH basis of its view
    direction with the 3 colour gradients, and the view direction is something every rank can work out itself for every
    camera (replicated means, 12-byte camera position).  Summed over the ranks' cameras the gradient is no longer rank
    one, but every rank can rebuild the sum from the other ranks' colour gradients.  So instead of all-reducing 192 B per
    Gaussian (~2 x 7/8 x 192 = 336 B over xGMI per GPU at 8 ranks) the ranks ALL-GATHER 12 B per Gaussian (7 x 12 = 84 B
    received) and run ``dnsplat_sh_grads_from_factors``; only the 44 B of geometry gradients are all-reduced.  Per-GPU
    xGMI traffic per Gaussian: 413 B -> 161 B at 8 GPUs, 236 B -> 56 B at 2.  The sums are the same up to fp32 summation
    order.  ``dnsplat_project_bwd`` also stops writing the 192 B rows itself.
    One rank's slab: ``[N,3]`` colour gradients, then its camera position (3 floats) and a pad word: 3 N + 4 floats.
    """

    def __init__(self, own_rows: Optional[bool] = None, packed: bool = False, capacity: Optional[int] = None):
        # own_rows: the projection backward writes THIS rank's camera's coefficient rows itself (pre-scaled by 1 / world) and the
        # rebuild only ADDS the other cameras' (dnsplat_sh_grads_add_factors / _from_packed with skip_view = rank): with one rank there
        # is nothing to add and no pass over the 192 B / Gaussian at all.  At world >= 2 the read-modify-write moves more bytes than
        # rebuilding every row from the slabs, hence None = "only when the world is one rank".
        self.own_rows = own_rows
        # packed: slabs of the VISIBLE Gaussians' rows only (include/dnsplat.h, dnsplat_visible_index): mask + block offsets + rows,
        # ``capacity`` rows per slab — the same on every rank (all-gather of equal pieces); None = N (never overflows, saves nothing on
        # the wire: call calibrate() once the scene has been rendered).  Whole scenes only (not SlicedShExchange).
        self.packed = bool(packed)
        self.capacity = capacity
        self.scratch: Optional[Tensor] = None
        self.scale_override: Optional[float] = None   # tests: the own-rows pre-scale of a world this process is not part of
        self.mine: Optional[Tensor] = None
        self.gathered: Optional[Tensor] = None
        self.meta = None
        self.means: Optional[Tensor] = None
        self.work = None      # the all-gather in flight (launch())
        self.group = None     # the process group collectives issued from INSIDE the backward use (launch / eager slice gathers)
        # True: launch() does nothing and finish() runs the all-gather itself — for a backward that is being captured into a HIP
        # graph (graph.GraphedDpStep): no collective is issued from inside the capture, the exchange follows the replay eagerly
        self.deferred = False

    def drop(self) -> None:
        """Forgets factors that were produced but never rebuilt (warm-up frames of a graph capture)."""
        self.meta, self.work = None, None

    def slab_floats(self, N: int) -> int:
        if self.packed:
            cap = N if self.capacity is None else min(int(self.capacity), N)
            nb = (N + 63) // 64
            rows0 = (8 + 3 * nb + 3) & ~3                    # = dnsplat_packed_slab_floats(N, cap): header | masks | offsets | rows
            return (rows0 + 3 * cap + 3) & ~3
        return 3 * N + 4

    def packed_capacity(self, N: int) -> int:
        return N if self.capacity is None else min(int(self.capacity), N)

    def use_own_rows(self, group=None) -> bool:
        return (world_size(self.group if group is None else group) == 1) if self.own_rows is None else bool(self.own_rows)

    def calibrate(self, radii: Tensor, group=None, slack: float = 1.1) -> int:
        """Packed slabs: sets ``capacity`` to ``slack`` x the largest visible count any rank reports for ``radii`` (its last frame),
        rounded up to 1024 rows — one host sync and one tiny collective, outside the step.  Call it again after densification."""
        n_vis = int((radii.reshape(-1) > 0).sum().item())
        n_vis = int(max_over_ranks(float(n_vis), radii.device, self.group if group is None else group))
        self.capacity = min(radii.numel(), (int(n_vis * slack) + 1023) // 1024 * 1024)
        self.mine = self.gathered = None
        return self.capacity

    def overflowed(self) -> bool:
        """Packed slabs: True if a gathered slab of the last step reported more visible Gaussians than its capacity (rows were dropped:
        the SH gradients of that step are incomplete).  Synchronises; call it now and then, like GraphedStep.check()."""
        if not self.packed or self.gathered is None:
            return False
        hdr = self.gathered.view(torch.int32).reshape(self.gathered.shape[0], -1)[:, :8].cpu()
        return bool((hdr[:, 0] > hdr[:, 4]).any())

    def begin(self, N, device, sh_degree, sh_K, means: Optional[Tensor] = None) -> Tensor:
        """Called by the projection backward: returns the slab (3 N + 4 floats) dnsplat_sh_factors fills and remembers the
        shape of the gradients to rebuild and the means the view directions are re-derived from."""
        if self.meta is not None:
            raise RuntimeError("ShFactorExchange: the factors of the previous backward were never rebuilt — with set_sh_exchange() "
                               "active every backward must be followed by dp.allreduce_gradients(..., exchange=...) (or "
                               "exchange.finish()); until then features_dc.grad / features_rest.grad are unwritten")
        n = self.slab_floats(N)
        if self.mine is None or self.mine.shape[0] != n or self.mine.device != device:
            self.mine = torch.empty(n, dtype=torch.float32, device=device)
        # no gradient tensors are kept here: autograd only adopts the returned gradient tensors as .grad (instead of
        # cloning them) while nobody else holds them
        self.meta = (N, sh_degree, sh_K)
        if means is not None:
            self.means = means.detach()
        self.work = None
        return self.mine

    def _gather_buffer(self, w: int) -> Tensor:
        n = self.slab_floats(self.meta[0])
        if self.gathered is None or self.gathered.shape != (w, n) or self.gathered.device != self.mine.device:
            self.gathered = torch.empty(w, n, dtype=torch.float32, device=self.mine.device)
        return self.gathered

    def launch(self, group=None) -> None:
        """Start the all-gather of the slabs (enqueued after whatever filled ``mine`` on the current stream) without
        waiting for it: the projection backward calls this right after ``dnsplat_sh_factors`` and BEFORE
        ``dnsplat_project_bwd``, so the 12 B/Gaussian travel while the geometry gradients are computed."""
        group = self.group if group is None else group
        if self.meta is None or self.deferred or not _collectives_on(group):
            return
        buf = self._gather_buffer(world_size(group))
        self.work = dist.all_gather_into_tensor(buf.view(-1), self.mine.view(-1), group=group, async_op=True)

    def finish(self, group=None, v_coeffs: Optional[Tensor] = None, v_sh0: Optional[Tensor] = None,
               v_shN: Optional[Tensor] = None) -> int:
        """Complete the all-gather (or run it, if launch() was not called) and rebuild the averaged coefficient gradients
        into the given ``.grad`` tensors (``v_coeffs`` [N,16,3], or ``v_sh0`` [N,3] + ``v_shN`` [N,15,3]).  Returns the
        bytes received."""
        if self.meta is None:
            return 0
        N, sh_degree, sh_K = self.meta
        w = world_size(group)
        own = self.use_own_rows(group)
        if own and w == 1 and not _collectives_on(group):
            self.meta = None              # one rank, no process group: the projection backward has written the complete rows
            return 0
        buf = self._gather_buffer(w)
        if self.work is not None:
            self.work.wait()
            self.work = None
        elif w > 1 or _collectives_on(group):
            dist.all_gather_into_tensor(buf.view(-1), self.mine.view(-1), group=group)
        else:
            buf.copy_(self.mine[None])
        self.meta = None
        extra = {} if REDUCTION["mode"] == "mean" else {"scale": 1.0}
        if own or self.packed:
            if not (own and w == 1):      # a single view whose rows are already in place: nothing to add
                self._rebuild(buf, self.means, N, w, sh_degree, sh_K, v_coeffs, v_sh0, v_shN,
                              skip_view=(rank_of(group) if own else -1), packed_capacity=(self.packed_capacity(N) if self.packed else None),
                              **extra)
        else:
            self._rebuild(buf, self.means, N, w, sh_degree, sh_K, v_coeffs, v_sh0, v_shN, **extra)
        return (w - 1) * self.slab_floats(N) * 4


class SlicedShExchange(ShFactorExchange):
    """The exchange step cut into K slices of Gaussians, so that xGMI starts carrying gradients a quarter of the way into the
    projection backward instead of behind it ("bucketed ... overlapped with backward").

    ``dnsplat_project_bwd`` needs no change for this: a slice [g0, g1) is the same entry point called on a smaller scene — every
    pointer of ``dnsplat_scene`` / ``dnsplat_proj_grads`` advanced by g0 rows, N = g1 - g0 — and each slice writes its own mini slab
    ``[3 n_k colour gradients | camera position | pad]``.  As soon as slice k has been launched its slab's all-gather is queued on
    the communication stream (it waits for that launch only) and travels while slices k+1 ... compute; the geometry gradients
    (44 B per Gaussian, one contiguous prefix of the bucket) are all-reduced in ONE collective behind the last slice — the links are
    busy with the slabs until then anyway (K x 7 x 12 n_k bytes take longer than the K launches at every BASELINE size) — and the
    coefficient rows of slice k are rebuilt (``dnsplat_sh_grads_from_factors`` on the slice) as soon as its slab has arrived.

    Two ways of running it:
      * eager: the projection backward launches the K slices itself; ``dp.allreduce_gradients`` runs the collectives;
      * recorded (``record_only``, used by ``graph.GraphedDpStep``): the projection backward that is being CAPTURED launches
        nothing and leaves the argument structs of the K launches here; ``run_recorded`` issues launch k + gather k for every
        slice right behind each graph replay.  No collective is ever part of a HIP graph (see GraphedDpStep).
    Slice bounds are multiples of 256 Gaussians (16-byte alignment of every row pointer, whole staging workgroups)."""

    ALIGN = 256

    def __init__(self, slices: int = 4):
        super().__init__(own_rows=False)          # every slice's rows are rebuilt from the gathered mini slabs
        self.slices = max(1, int(slices))
        self.record_only = False
        self.records = None            # [(entry point args..., keep-alive tensors)] of the captured backward
        self.bounds = None
        self.total = 0
        self.works = None

    def plan(self, N: int):
        K = max(1, min(self.slices, N // self.ALIGN))
        step = -(-N // K)
        step = -(-step // self.ALIGN) * self.ALIGN
        return [(g0, min(N, g0 + step)) for g0 in range(0, max(N, 1), step)] if N > 0 else [(0, 0)]

    def _slab_offset(self, k: int) -> int:
        return 3 * self.bounds[k][0] + 4 * k

    def begin(self, N, device, sh_degree, sh_K, means: Optional[Tensor] = None):
        """-> the K mini slabs (views of one buffer of 3 N + 4 K floats), slab k = 3 n_k + 4 floats."""
        if self.meta is not None:
            raise RuntimeError("SlicedShExchange: the factors of the previous backward were never rebuilt — every backward must be "
                               "followed by dp.allreduce_gradients(..., exchange=...) (or run_recorded / finish)")
        self.bounds = self.plan(N)
        self.total = 3 * N + 4 * len(self.bounds)
        if self.mine is None or self.mine.shape[0] != self.total or self.mine.device != device:
            self.mine = torch.empty(self.total, dtype=torch.float32, device=device)
        self.meta = (N, sh_degree, sh_K)
        if means is not None:
            self.means = means.detach()
        self.works = None
        return [self.slab(k) for k in range(len(self.bounds))]

    def slab(self, k: int) -> Tensor:
        g0, g1 = self.bounds[k]
        o = self._slab_offset(k)
        return self.mine[o:o + 3 * (g1 - g0) + 4]

    def _gathered(self, k: int, w: int) -> Tensor:
        """[w, 3 n_k + 4] view of slice k's receive buffer (one buffer of w x (3 N + 4 K) floats, slice-major)."""
        if self.gathered is None or self.gathered.numel() != w * self.total or self.gathered.device != self.mine.device:
            self.gathered = torch.empty(w * self.total, dtype=torch.float32, device=self.mine.device)
        g0, g1 = self.bounds[k]
        n = 3 * (g1 - g0) + 4
        o = w * self._slab_offset(k)
        return self.gathered[o:o + w * n].view(w, n)

    def launch(self, group=None) -> None:      # the unsliced early all-gather does not apply
        return

    def gather_slice(self, k: int, group=None):
        """Queues the all-gather of slab k behind whatever filled it on the current stream; returns the handle (or None)."""
        group = self.group if group is None else group
        w = world_size(group)
        buf = self._gathered(k, w)
        if w > 1 or _collectives_on(group):
            return dist.all_gather_into_tensor(buf.view(-1), self.slab(k), group=group, async_op=True)
        buf.copy_(self.slab(k)[None])
        return None

    def rebuild_slice(self, k: int, w: int, v_sh0: Tensor, v_shN: Tensor) -> None:
        N, sh_degree, sh_K = self.meta
        g0, g1 = self.bounds[k]
        if g1 > g0:
            extra = {} if REDUCTION["mode"] == "mean" else {"scale": 1.0}
            self._rebuild(self._gathered(k, w), self.means[g0:g1], g1 - g0, w, sh_degree, sh_K, None, v_sh0[g0:g1], v_shN[g0:g1], **extra)

    def finish(self, group=None, v_coeffs: Optional[Tensor] = None, v_sh0: Optional[Tensor] = None,
               v_shN: Optional[Tensor] = None) -> int:
        """All slices: gathers not yet in flight are queued now (all of them before the first wait), then each slice is rebuilt as
        its slab arrives.  Returns the bytes received."""
        if self.meta is None:
            return 0
        if v_coeffs is not None:
            raise RuntimeError("SlicedShExchange rebuilds into the split features_dc / features_rest gradients only")
        N = self.meta[0]
        w = world_size(group)
        K = len(self.bounds)
        works = self.works if self.works is not None else [self.gather_slice(k, group) for k in range(K)]
        self.works = None
        for k in range(K):
            if works[k] is not None:
                works[k].wait()
            self.rebuild_slice(k, w, v_sh0, v_shN)
        self.meta = None
        return (w - 1) * self.total * 4

    # ---- recorded mode (graph.GraphedDpStep) -----------------------------------------------------------------------------
    def record(self, launches, keep) -> None:
        self.records = (launches, keep)

    def run_recorded(self, params: Dict[str, Tensor], arena: "GradArena", group=None, collectives: bool = True,
                     direct: Optional[Dict[str, Tensor]] = None) -> int:
        """Behind a replay of the captured step: launch k (dnsplat_project_bwd on slice k) + all-gather k for every slice, ONE
        all-reduce of the geometry prefix, the rebuilds.  ``direct``: gradients that reach a geometry tensor NOT through the
        renderer (a scale regulariser): the captured step leaves them in tensors of its own, they are added to the bucket slice
        before it is reduced.  ``collectives=False``: the launches only (what the step costs when nothing travels; the coefficient
        rows stay unwritten).  Returns the bytes exchanged."""
        launches, _keep = self.records
        works = []
        for k, args in enumerate(launches):
            self._launch(args)
            if collectives:
                works.append(self.gather_slice(k, group))
        if direct:
            for name, t in direct.items():
                arena.view(name).add_(t.view(arena.view(name).shape))
        if not collectives:
            return 0
        n_geo = sum(arena.slices[k][1] for k in GEOMETRY_KEYS)
        geo = allreduce_mean_(arena.flat[:n_geo], group, async_op=True)
        self.works = works
        got = self.finish(group, v_sh0=arena.view("features_dc"), v_shN=arena.view("features_rest"))
        geo.wait()
        return n_geo * 4 + got


def _rebuild_hip(gathered: Tensor, means: Tensor, N: int, w: int, sh_degree: int, sh_K: int, v_coeffs, v_sh0, v_shN, skip_view: int = -1,
                 packed_capacity: Optional[int] = None, scale: Optional[float] = None) -> None:
    from . import _lib
    from ._ops import _ptr, _stream

    scale = 1.0 / w if scale is None else float(scale)

    if v_coeffs is not None:      # gsplat layout [N,16,3]
        p0, s0, pN, sN = v_coeffs, 3 * sh_K, v_coeffs.view(-1)[3:], 3 * sh_K
    else:
        p0, s0, pN, sN = v_sh0, 3, v_shN, 3 * (sh_K - 1)
    if packed_capacity is not None:
        _lib.run("dnsplat_sh_grads_from_packed", _lib.lib().dnsplat_sh_grads_from_packed, N, int(packed_capacity), w, int(skip_view),
                 _ptr(gathered), _ptr(means.contiguous()), sh_degree, sh_K, scale, _ptr(p0), s0, _ptr(pN), sN, _stream())
        return
    if skip_view >= 0:
        _lib.run("dnsplat_sh_grads_add_factors", _lib.lib().dnsplat_sh_grads_add_factors, N, w, int(skip_view), _ptr(gathered),
                 _ptr(means.contiguous()), sh_degree, sh_K, scale, _ptr(p0), s0, _ptr(pN), sN, _stream())
        return
    _lib.run("dnsplat_sh_grads_from_factors", _lib.lib().dnsplat_sh_grads_from_factors, N, w, _ptr(gathered), _ptr(means.contiguous()),
             sh_degree, sh_K, scale, _ptr(p0), s0, _ptr(pN), sN, _stream())


ShFactorExchange._rebuild = staticmethod(_rebuild_hip)   # the CPU gloo test swaps in a torch reference
SlicedShExchange._rebuild = staticmethod(_rebuild_hip)


def _launch_project_bwd(args) -> None:
    from . import _lib
    from ._ops import _stream

    _lib.run("dnsplat_project_bwd", _lib.lib().dnsplat_project_bwd, *args, _stream())


SlicedShExchange._launch = staticmethod(_launch_project_bwd)   # the CPU gloo test swaps in a stand-in that fills slab + bucket slice


def init_from_env(device_type: Optional[str] = None):
    """(rank, world, local_rank, device).  Initialises the default process group from the RANK / WORLD_SIZE /
    MASTER_* variables ``python -m torch.distributed.run`` exports; single process when they are absent."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if device_type is None:
        device_type = "cuda" if torch.cuda.is_available() else "cpu"
    if device_type == "cuda":
        # DNSPLAT_SHARE_GPU=1: several ranks on one GPU (functional tests on a single-GPU box; RCCL refuses that, so the
        # backend must be gloo, which stages device tensors through the host)
        dev_index = local % torch.cuda.device_count() if os.environ.get("DNSPLAT_SHARE_GPU", "0") == "1" else local
        torch.cuda.set_device(dev_index)
        device = torch.device("cuda", dev_index)
    else:
        device = torch.device("cpu")
    force = os.environ.get("DNSPLAT_FORCE_DIST", "0") == "1"   # exercise the collective path with a single rank
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL needs it)
        backend = os.environ.get("DNSPLAT_DIST_BACKEND") or ("nccl" if device_type == "cuda" else "gloo")
        kw = {"device_id": device} if backend == "nccl" else {}
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local, device


def world_size(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def rank_of(group=None) -> int:
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


def _collectives_on(group=None) -> bool:
    """True when a process group exists and either has > 1 rank or was forced on for testing."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or os.environ.get("DNSPLAT_FORCE_DIST", "0") == "1"


class _MeanWork:
    """Handle of an in-place mean over ranks: ``wait()`` completes it (RCCL averages natively; gloo sums and is scaled
    here)."""

    def __init__(self, work, t: Optional[Tensor], scale: float):
        self.work, self.t, self.scale = work, t, scale

    def wait(self) -> None:
        if self.work is not None:
            self.work.wait()
        if self.t is not None:
            self.t.mul_(self.scale)


def allreduce_mean_(t: Tensor, group=None, async_op: bool = False):
    """In-place mean over ranks (REDUCTION "sum": in-place sum — the caller scaled its loss by 1 / world).  With ``async_op`` the
    collective is only enqueued; call ``wait()`` on the result."""
    w = world_size(group)
    if not _collectives_on(group):
        return _MeanWork(None, None, 1.0) if async_op else None
    if REDUCTION["mode"] == "sum":
        work = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        return _MeanWork(work, None, 1.0) if async_op else work
    if t.is_cuda:
        work = dist.all_reduce(t, op=dist.ReduceOp.AVG, group=group, async_op=async_op)
        return _MeanWork(work, None, 1.0) if async_op else work
    work = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    if async_op:
        return _MeanWork(work, t, 1.0 / w)
    t.mul_(1.0 / w)
    return work


def allreduce_gradients(params: Dict[str, Tensor], arena: Optional[GradArena] = None, group=None,
                        exchange: Optional[ShFactorExchange] = None) -> int:
    """Average ``params[k].grad`` (k in GRAD_KEYS) across ranks; returns the bytes exchanged per rank.

    Fast path: every grad is a slice of ``arena`` -> one in-place collective over the bucket (with ``exchange``: over its
    geometry prefix only, the SH part travels as factors).  Otherwise the grads are packed into a temporary bucket,
    reduced and copied back."""
    if exchange is not None and exchange.meta is not None:
        if arena is None or not all(arena.holds(params[k].grad) for k in GEOMETRY_KEYS):
            raise RuntimeError("the SH factor exchange needs the gradients in a GradArena (dp.GradArena + set_grad_arena)")
        n_geo = sum(arena.slices[k][1] for k in GEOMETRY_KEYS)
        # the slabs first: their all-gather is what the rebuild waits for, and the rebuild (on the compute stream) then runs
        # while the geometry all-reduce, queued behind the gather on the communication stream, is still travelling
        if isinstance(exchange, SlicedShExchange):
            if exchange.works is None:
                exchange.works = [exchange.gather_slice(k, group) for k in range(len(exchange.bounds))]
        elif exchange.work is None and _collectives_on(group):
            buf = exchange._gather_buffer(world_size(group))
            exchange.work = dist.all_gather_into_tensor(buf.view(-1), exchange.mine.view(-1), group=group, async_op=True)
        # the geometry all-reduce is queued behind the factor all-gather on the communication stream and runs while the
        # SH rows are rebuilt from the gathered factors on the compute stream
        geo = allreduce_mean_(arena.flat[:n_geo], group, async_op=True)
        got = exchange.finish(group, v_sh0=params["features_dc"].grad, v_shN=params["features_rest"].grad)
        geo.wait()
        return n_geo * 4 + got
    if not _collectives_on(group):
        return 0
    grads = [params[k].grad for k in GRAD_KEYS]
    if arena is not None and all(arena.holds(g) for g in grads):
        arena.invalidate_sh_state()          # rows culled on this rank receive the other ranks' gradients
        allreduce_mean_(arena.flat, group)
        return arena.bytes()
    present = [g for g in grads if g is not None]
    if not present:
        return 0
    flat = torch.cat([g.reshape(-1) for g in present])
    allreduce_mean_(flat, group)
    off = 0
    for g in present:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()
    return flat.numel() * 4


def barrier(group=None) -> None:
    if _collectives_on(group):
        dist.barrier(group)


def max_over_ranks(value: float, device, group=None) -> float:
    if not _collectives_on(group):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def sum_over_ranks(values: Iterable[float], device, group=None):
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    if _collectives_on(group):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.tolist()


def gather_over_ranks(value: float, device, group=None):
    """[value of rank 0, value of rank 1, ...] on every rank."""
    if not _collectives_on(group):
        return [value]
    w = world_size(group)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    out = torch.empty(w, dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(out, t, group=group)
    return out.tolist()
