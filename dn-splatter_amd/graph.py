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
