"""Densification statistics on device.

dn-splatter inherits ``after_train`` from nerfstudio's ``SplatfactoModel`` (registered at
``dn_splatter/dn_model.py:938-942``); it reads ``self.xys.absgrad`` and ``self.radii`` — both produced by our
renderer — and accumulates the three per-Gaussian statistics ``refinement_after`` consumes
(``dn_model.py:286-296``: ``xys_grad_norm / vis_counts * 0.5 * max(size)`` against ``densify_grad_thresh``, and
``max_2Dsize`` for the screen-size split/cull rules).  nerfstudio 1.1.3 is not vendored in the reference, so the body is
restated from its published source:

    visible = radii > 0
    xys_grad_norm[visible] += xys.absgrad[0][visible].norm(dim=-1)      # .grad if use_absgrad is off
    vis_counts[visible]    += 1                                         # vis_counts starts at ones
    max_2Dsize[visible]     = max(max_2Dsize[visible], radii[visible] / max(W, H))

Here it is one kernel (``dnsplat_densify_stats``) instead of ~10 boolean-mask gathers/scatters, and — new with
multi-view data parallelism — the per-rank statistics are combined across ranks so that every replica takes the same
split/cull decisions.  The split/duplicate/cull surgery itself (optimizer state included) stays with nerfstudio.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch
import torch.distributed as dist
from torch import Tensor

from . import _lib, dp
from ._ops import _ptr, _stream


class DensifyStats:
    def __init__(self, num_points: int, device):
        self.xys_grad_norm = torch.zeros(num_points, dtype=torch.float32, device=device)
        self.vis_counts = torch.ones(num_points, dtype=torch.float32, device=device)
        self.max_2Dsize = torch.zeros(num_points, dtype=torch.float32, device=device)

    def after_train(self, renderer, width: int, height: int, use_absgrad: bool = True) -> None:
        """Accumulate from the renderer's last backward (``renderer.xys`` must have .grad / .absgrad)."""
        xys = renderer.xys
        grads = xys.absgrad if use_absgrad else xys.grad
        if grads is None:
            raise RuntimeError("after_train before backward: means2d has no gradient yet (dn_model.py:517-519)")
        grads = grads.reshape(-1, grads.shape[-1])
        N = grads.shape[0]
        stride = grads.stride(0)
        assert grads.stride(1) == 1 and grads.dtype == torch.float32
        _lib.run("dnsplat_densify_stats", _lib.lib().dnsplat_densify_stats, N, _ptr(renderer.radii.contiguous()),
                 _ptr(grads), stride, 1.0 / float(max(width, height)), _ptr(self.xys_grad_norm), _ptr(self.vis_counts),
                 _ptr(self.max_2Dsize), _stream())

    def allreduce(self, prev: Optional["DensifyStats"] = None, group=None) -> None:
        """Data parallel: combine THIS step's increments of all ranks.  ``prev`` holds the values before the step
        (sums are reduced on the increment so that the common history is not multiplied by the world size)."""
        if not dp._collectives_on(group):
            return
        if prev is None:
            raise ValueError("pass the pre-step statistics so that only the increment is summed")
        for cur, old in ((self.xys_grad_norm, prev.xys_grad_norm), (self.vis_counts, prev.vis_counts)):
            inc = cur - old
            dist.all_reduce(inc, op=dist.ReduceOp.SUM, group=group)
            cur.copy_(old + inc)
        dist.all_reduce(self.max_2Dsize, op=dist.ReduceOp.MAX, group=group)

    def clone(self) -> "DensifyStats":
        c = DensifyStats.__new__(DensifyStats)
        c.xys_grad_norm, c.vis_counts, c.max_2Dsize = (self.xys_grad_norm.clone(), self.vis_counts.clone(),
                                                       self.max_2Dsize.clone())
        return c
