"""A whole training step — ``get_outputs`` + loss + backward — captured once into a HIP graph and replayed.

One frame of the fused path is ~35 kernel launches issued from Python through ctypes and autograd: about 2 ms of host
time against ~2.5 ms of GPU time at the 1 M-Gaussian / 1080p workload, i.e. the host is barely ahead and every hiccup
of the interpreter stalls the GPU (a tenth of the steps took 0.3-0.8 ms longer).  The launch sequence of a frame does
not depend on the data — same kernels, same buffers, the element counts are read from device words — so it is captured
once (``torch.cuda.graph`` = hipStreamBeginCapture on ROCm) and replayed with ONE host call per step.

What makes the frame capturable:
  * bin policy "static" (``_ops.BIN_POLICY``): nothing on the host waits for, or even looks at, the frame's
    intersection count; the capacity is the one the eager warm-up frames established (x1.25) and the device keeps a
    running maximum (``dnsplat_bin_args.n_isects_max``) that ``check()`` compares with it;
  * every buffer a kernel touches is allocated during capture from the graph's private pool (or before it), so
    addresses are stable across replays; gradients land in the same tensors each time (``dp.GradArena`` slices or the
    tensors autograd installed during capture);
  * the camera pose is read on the device from ``camera.camera_to_worlds`` — copy a new pose INTO that tensor and the
    next replay renders it.  Intrinsics and the image size are kernel arguments: one graph per (W, H, fx, fy, cx, cy).

Usage::

    step = GraphedStep(lambda: loss_fn(renderer.get_outputs(cam)).backward(), params=gauss_params)
    for it in range(n):
        cam.camera_to_worlds.copy_(next_pose)      # optional
        step()                                      # replays; outputs / .grad tensors are the ones of the capture
        optimizer.step()
    step.check()                                    # raises if any replayed frame overflowed its intersection buffers

``copies=2`` captures the step twice (two graphs, two sets of buffers; gradients still land in the same ``GradArena``)
and alternates between them: while one copy runs, the host can read what the other one left behind — bench.py uses it
to read the HIP-event brackets of every replay without ever making the GPU wait for the host.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import torch
from torch import Tensor

from . import _lib, _ops


class GraphedStep:
    def __init__(self, fn: Callable[[], object], params: Optional[Dict[str, Tensor]] = None, warmup: int = 2,
                 copies: int = 1, before_capture: Optional[Callable[[int], None]] = None):
        self.fn = fn
        self.params = params
        self._warmup, self._before_capture = warmup, before_capture      # recapture() repeats the capture as it was asked for
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.stream = torch.cuda.Stream(self.device)
        self.graphs: List[torch.cuda.CUDAGraph] = []
        self.results: List[object] = []
        self.done = [torch.cuda.Event() for _ in range(copies)]     # recorded behind every replay of a copy
        self.launched = [False] * copies
        self.replays = 0
        self.overflowed = None
        prev = _ops.BIN_POLICY["mode"]
        try:
            self._capture(warmup, copies, before_capture)
        finally:
            _ops.BIN_POLICY["mode"] = prev

    def _zero_grads(self):
        if self.params is not None:
            for p in self.params.values():
                if isinstance(p, Tensor):
                    p.grad = None

    def _capture(self, warmup: int, copies: int, before_capture) -> None:
        s = self.stream
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            # eager frames on the capture stream: the first one sizes the intersection buffers with a host round trip
            # ("static" falls back to an exact count while no capacity is known), the rest run as the capture will
            _ops.set_bin_policy("static")
            for _ in range(max(warmup, 2)):
                self._zero_grads()
                self.fn()
            s.synchronize()
            over = self._overflow()
            if over is not None:
                raise _lib.DnsplatError(f"GraphedStep: warm-up frame with {over} intersections exceeded its capacity")
        # the device words the captured bin kernels keep their running maxima in: held here as well, so that nothing the host
        # forgets meanwhile can free memory a replay still writes (check() refuses to vouch for a step whose records are gone)
        skey = (self.device, s.cuda_stream)
        self._n_max = {k: t for k, t in _ops.BUFFERS.n_max.items() if k[:2] == skey}
        for c in range(copies):
            self._zero_grads()
            if before_capture is not None:
                before_capture(c)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                self.results.append(self.fn())
            self.graphs.append(g)
        torch.cuda.current_stream(self.device).wait_stream(s)

    def _overflow(self):
        return _ops.static_overflow(self.device, self.stream)

    def __call__(self, copy: Optional[int] = None):
        """Replays the captured step (copy ``replays % copies`` unless given) on the CURRENT stream.  With several copies the
        caller must not replay a copy whose previous replay it still wants to read (``wait(copy)`` first)."""
        c = self.replays % len(self.graphs) if copy is None else copy
        self.graphs[c].replay()
        self.done[c].record()
        self.launched[c] = True
        self.replays += 1
        return self.results[c]

    def wait(self, copy: int) -> bool:
        """Blocks until the last replay of ``copy`` has finished; False if it was never replayed."""
        if not self.launched[copy]:
            return False
        self.done[copy].synchronize()
        return True

    def check(self) -> None:
        """Raises if any frame replayed so far produced more intersections than the captured buffers hold (its lists were
        truncated).  Synchronises with the capture stream.  After the error this step stays invalid (``overflowed``; every later
        ``check()`` raises again, replaying it keeps truncating): build a new GraphedStep — the capacity guess has been raised to
        1.25 x the count that did not fit, so the new capture gets buffers that hold it."""
        self.stream.synchronize()
        gone = [k for k in getattr(self, "_n_max", {}) if _ops.BUFFERS.n_max.get(k) is not self._n_max[k]]
        if gone and self.graphs:
            raise _lib.DnsplatError(
                "GraphedStep.check(): the bin policy's running-maximum records of this step were dropped while it was alive "
                f"({len(gone)} of {len(self._n_max)}): overflow of the replayed frames can no longer be ruled out — close() this "
                "step and capture a new one")
        over = self._overflow()
        if over is not None:
            self.overflowed = max(self.overflowed or 0, over)
        if self.overflowed:
            raise _lib.DnsplatError(
                f"GraphedStep: a replayed frame produced {self.overflowed} intersections, more than the captured buffers hold; "
                "the results of that frame (and of every frame with as many) are invalid. The capacity guess has been raised: "
                "discard this step and capture a new GraphedStep")

    def close(self) -> None:
        """Releases the graphs and the per-stream bookkeeping of the bin policy (call before capturing a replacement)."""
        self.graphs, self.results = [], []
        _ops.forget_static(self.device, self.stream)

    def recapture(self, **kw) -> "GraphedStep":
        """close() + a new capture of the same function over the same parameters — what follows a check() that reported an overflow
        (the capacity guess has been raised by then), a refinement that kept the tensors, or a change of image size.  Warm-up count,
        number of copies and the ``before_capture`` hook are the original ones unless overridden.  Returns the new step; this one is
        finished.  (The inner step of a GraphedDpStep is re-captured through GraphedDpStep.recapture(), which also restores the
        exchange's recorded state.)"""
        n = max(len(self.done), 1)
        self.close()
        self._zero_grads()
        kw.setdefault("warmup", self._warmup)
        kw.setdefault("before_capture", self._before_capture)
        return GraphedStep(self.fn, self.params, copies=kw.pop("copies", n), **kw)


class GraphedDpStep:
    """The data-parallel step with the host out of the way: ``fn`` (get_outputs + loss + backward of THIS rank's camera) is
    captured into one HIP graph, and the exchange step — the all-gather of the SH colour-gradient slabs, the all-reduce of
    the geometry gradients, the rebuild kernel (``dp.allreduce_gradients``) — is issued eagerly right behind each replay.

    Why not the collectives inside the graph: RCCL kernels captured into a HIP graph cannot be exercised on the one-GPU boxes
    this code is developed on, and a hang on an 8-GPU node is not an acceptable way to find out.  What the eager step loses
    to the host are the ~35 kernel launches of the frame (about 2 ms of Python per 2.3 ms of GPU at the 1 M / 1080p
    workload); behind a replay the host issues two collectives and one kernel.  The price: the all-gather no longer starts
    before ``dnsplat_project_bwd`` (it used to travel behind that kernel: 0.12 ms at 1 M Gaussians, 0.6 ms at 5 M).

    Gradients land in the ``dp.GradArena`` slices autograd installed during the capture; ``wire`` holds the bytes the last
    step exchanged."""

    def __init__(self, fn: Callable[[], object], params: Dict[str, Tensor], arena, exchange=None, group=None, **kw):
        from . import dp

        self._dp = dp
        self.params, self.arena, self.exchange, self.group = params, arena, exchange, group
        self.wire = 0
        # dp.SlicedShExchange: the captured step ends in front of dnsplat_project_bwd; its K slice launches are issued behind each
        # replay, each followed by the all-gather of its slab (the exchange overlaps the projection backward again)
        self.sliced = exchange is not None and getattr(exchange, "slices", 1) > 1
        if self.sliced and int(kw.get("copies", 1)) != 1:
            # the exchange keeps ONE set of recorded launches / direct-gradient tensors: with two captures the replays of copy 0
            # would run copy 1's slice launches on copy 1's pool tensors
            raise _lib.DnsplatError("GraphedDpStep: the sliced (recorded) exchange supports copies=1 only")
        self._fn, self._kw = fn, dict(kw)
        self.direct: Dict[str, Tensor] = {}
        if exchange is not None:
            exchange.deferred = True
            if self.sliced:
                exchange.record_only = True

        def compute():
            if exchange is not None:
                exchange.drop()          # an eager warm-up frame leaves factors nobody rebuilds
            return fn()

        try:
            self.step = GraphedStep(compute, params={k: params[k] for k in dp.GRAD_KEYS}, **kw)
            self._adopt_recorded(dp)
        except BaseException:
            # warm-up or capture failed (overflow, capture error): the caller falls back to the eager step with the SAME exchange
            # object, which must start its all-gather from the backward again and must not find the warm-up's factors pending
            if exchange is not None:
                exchange.deferred = False
                if self.sliced:
                    exchange.record_only = False
                    exchange.records = None
                exchange.drop()
                if getattr(self, "step", None) is not None:
                    self.step.close()
            raise
        # what ShFactorExchange.begin() recorded while the backward was captured: restored before every exchange, because a replay
        # runs no Python
        self._meta = exchange.meta if exchange is not None else None

    def _adopt_recorded(self, dp) -> None:
        exchange, params, arena = self.exchange, self.params, self.arena
        if self.sliced:
            if exchange.records is None:
                raise _lib.DnsplatError("GraphedDpStep: the captured step never reached the projection backward (no slices recorded)")
            # What autograd left in .grad of a geometry tensor during the capture did not come from the renderer (the recorded
            # launches hand autograd nothing): a loss term that feeds the parameter directly, e.g. the scale regulariser.  The
            # graph rewrites that tensor on every replay; it is added to the bucket slice before the all-reduce, and the bucket
            # slice — which the recorded launches write — becomes .grad.
            for k in dp.GEOMETRY_KEYS:
                g = params[k].grad
                if g is not None and not arena.holds(g):
                    self.direct[k] = g
                params[k].grad = arena.view(k)
            for k in ("features_dc", "features_rest"):
                if not arena.holds(params[k].grad):
                    raise _lib.DnsplatError(f"GraphedDpStep: {k}.grad is not a slice of the gradient bucket — a loss term that feeds "
                                            "the SH coefficients directly cannot be combined with the factor exchange")

    def compute_only(self):
        """Replay without the exchange (bench.py: what the step costs when nothing travels)."""
        out = self.step()
        if self.sliced:
            self.exchange.meta = self._meta
            self.exchange.run_recorded(self.params, self.arena, self.group, collectives=False, direct=self.direct)
            self.exchange.meta = None
        return out

    def exchange_only(self) -> int:
        if self.exchange is not None:
            self.exchange.meta = self._meta
        if self.sliced:
            self.wire = self.exchange.run_recorded(self.params, self.arena, self.group, direct=self.direct)
        else:
            self.wire = self._dp.allreduce_gradients(self.params, self.arena, self.group, exchange=self.exchange)
        return self.wire

    def __call__(self):
        out = self.step()
        self.exchange_only()
        return out

    def check(self) -> None:
        self.step.check()

    def recapture(self, **kw) -> "GraphedDpStep":
        """close() + a new capture of the same step with the same exchange (record_only / deferred state rebuilt by the constructor).
        Do not call ``self.step.recapture()``: that would bypass the exchange's recorded launches."""
        self.close()
        for p in self.params.values():
            if isinstance(p, Tensor):
                p.grad = None
        return GraphedDpStep(self._fn, self.params, self.arena, self.exchange, self.group, **{**self._kw, **kw})

    def close(self) -> None:
        self.step.close()
        if self.exchange is not None:
            self.exchange.deferred = False
            if self.sliced:
                self.exchange.record_only = False
                self.exchange.records = None
            self.exchange.drop()
