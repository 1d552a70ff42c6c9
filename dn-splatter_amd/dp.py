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

# order = layout of the bucket; matches dn_model.py:388-402 minus "normals"
GRAD_KEYS = ("means", "scales", "quats", "features_dc", "features_rest", "opacities")


class GradArena:
    """One flat fp32 buffer holding the gradients of all optimised tensors of a Gaussian set.

    ``_ops._ProjectFn.backward`` asks ``take(name, like)`` for its output tensors, so the kernels write
    directly into the bucket and autograd installs those views as ``param.grad``.
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

    def view(self, name: str) -> Tensor:
        off, n, shape = self.slices[name]
        return self.flat[off:off + n].view(shape)

    def take(self, like: Tensor) -> Optional[Tensor]:
        """Bucket slice for the gradient of parameter ``like`` (matched by storage address), or None."""
        name = self._by_ptr.get(like.data_ptr())
        if name is None or self.slices[name][1] != like.numel():
            return None
        return self.view(name)

    def holds(self, t: Optional[Tensor]) -> bool:
        if t is None:
            return False
        a = self.flat.data_ptr()
        return a <= t.data_ptr() < a + self.flat.numel() * 4

    def bytes(self) -> int:
        return self.flat.numel() * 4


def init_from_env(device_type: Optional[str] = None):
    """(rank, world, local_rank, device).  Initialises the default process group from the RANK / WORLD_SIZE /
    MASTER_* variables ``python -m torch.distributed.run`` exports; single process when they are absent."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if device_type is None:
        device_type = "cuda" if torch.cuda.is_available() else "cpu"
    if device_type == "cuda":
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
    else:
        device = torch.device("cpu")
    force = os.environ.get("DNSPLAT_FORCE_DIST", "0") == "1"   # exercise the collective path with a single rank
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = "nccl" if device_type == "cuda" else "gloo"
        kw = {"device_id": device} if device_type == "cuda" else {}
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local, device


def world_size(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def _collectives_on(group=None) -> bool:
    """True when a process group exists and either has > 1 rank or was forced on for testing."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or os.environ.get("DNSPLAT_FORCE_DIST", "0") == "1"


def allreduce_mean_(t: Tensor, group=None, async_op: bool = False):
    """In-place mean over ranks.  RCCL has a native AVG; gloo sums and we scale."""
    w = world_size(group)
    if not _collectives_on(group):
        return None
    if t.is_cuda:
        return dist.all_reduce(t, op=dist.ReduceOp.AVG, group=group, async_op=async_op)
    work = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=False)
    t.mul_(1.0 / w)
    return work


def allreduce_gradients(params: Dict[str, Tensor], arena: Optional[GradArena] = None, group=None) -> int:
    """Average ``params[k].grad`` (k in GRAD_KEYS) across ranks; returns the bytes put on the wire per rank.

    Fast path: every grad is a slice of ``arena`` -> one in-place collective over the bucket.
    Otherwise the grads are packed into a temporary bucket, reduced and copied back."""
    if not _collectives_on(group):
        return 0
    grads = [params[k].grad for k in GRAD_KEYS]
    if arena is not None and all(arena.holds(g) for g in grads):
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
