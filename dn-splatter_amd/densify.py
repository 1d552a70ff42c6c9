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
split/cull decisions.

``refinement_after`` below is the surgery itself (``dn_model.py:271-386`` plus the helpers it inherits from nerfstudio's
``SplatfactoModel``: ``split_gaussians``, ``dup_gaussians``, ``cull_gaussians``, ``dup_in_all_optim``,
``remove_from_all_optim`` — restated from the published nerfstudio 1.1.3 source, which the reference does not vendor): the
decisions come from ONE kernel (``dnsplat_densify_classify``) evaluated on the all-reduced statistics, the children's means
and scales from a second (``dnsplat_densify_split``) with noise drawn from a generator every rank seeds identically, and the
parameter / Adam-state rows are moved by index gathers.  All ranks therefore end a refinement step with identical Gaussian
sets — the precondition for the next data-parallel step.  ``oracle/densify_ref.py`` restates the reference sequence in
plain torch; the tests compare against it.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Callable, Dict, Optional, Tuple

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

    def reset(self, num_points: Optional[int] = None) -> None:
        """Start over (dn_model.py:359-363 sets the three statistics to None after a refinement), for ``num_points`` Gaussians if
        the set changed size."""
        n = self.xys_grad_norm.shape[0] if num_points is None else num_points
        dev = self.xys_grad_norm.device
        self.__init__(n, dev)

    def clone(self) -> "DensifyStats":
        c = DensifyStats.__new__(DensifyStats)
        c.xys_grad_norm, c.vis_counts, c.max_2Dsize = (self.xys_grad_norm.clone(), self.vis_counts.clone(),
                                                       self.max_2Dsize.clone())
        return c


@dataclass
class RefineConfig:
    """The fields of DNSplatterModelConfig / SplatfactoModelConfig that refinement_after reads.  dn_model.py:102,112 set
    warmup_length and stop_split_at; the rest are nerfstudio 1.1.3 defaults (dn-splatter-big overrides cull_alpha_thresh =
    0.005 and continue_cull_post_densification = False, dn_config.py:150-153)."""
    warmup_length: int = 500
    refine_every: int = 100
    reset_alpha_every: int = 30
    stop_split_at: int = 15000
    stop_screen_size_at: int = 4000
    densify_grad_thresh: float = 0.0008
    densify_size_thresh: float = 0.01
    split_screen_size: float = 0.05
    n_split_samples: int = 2
    cull_alpha_thresh: float = 0.1
    cull_scale_thresh: float = 0.5
    cull_screen_size: float = 0.15
    continue_cull_post_densification: bool = True


SPLIT, DUP, CULL, CULL_CHILD, CULL_DUP = 1, 2, 4, 8, 16      # include/dnsplat.h DNSPLAT_DENSIFY_*


def classify(gauss_params: Dict[str, Tensor], stats: Optional[DensifyStats], cfg: RefineConfig, step: int, last_size,
             do_densify: bool) -> Tensor:
    """uint8 [N] flag byte per Gaussian (dnsplat_densify_classify)."""
    scales = gauss_params["scales"].detach().contiguous()
    opac = gauss_params["opacities"].detach().reshape(-1).contiguous()
    N = scales.shape[0]
    flags = torch.empty(N, dtype=torch.uint8, device=scales.device)
    a = _lib.DensifyArgs()
    a.N = N
    a.scales, a.opacities = _ptr(scales), _ptr(opac)
    if stats is not None:
        a.xys_grad_norm, a.vis_counts, a.max_2Dsize = _ptr(stats.xys_grad_norm), _ptr(stats.vis_counts), _ptr(stats.max_2Dsize)
    a.do_densify = int(do_densify)
    a.screen_rules = int(step < cfg.stop_screen_size_at)
    a.cull_big = int(step > cfg.refine_every * cfg.reset_alpha_every)
    a.max_image_side = float(max(last_size[0], last_size[1]))
    a.densify_grad_thresh, a.densify_size_thresh, a.split_screen_size = cfg.densify_grad_thresh, cfg.densify_size_thresh, cfg.split_screen_size
    a.cull_alpha_thresh, a.cull_scale_thresh, a.cull_screen_size = cfg.cull_alpha_thresh, cfg.cull_scale_thresh, cfg.cull_screen_size
    a.flags = _ptr(flags)
    _lib.run("dnsplat_densify_classify", _lib.lib().dnsplat_densify_classify, ctypes.byref(a), _stream())
    return flags


def split_children(gauss_params: Dict[str, Tensor], parents: Tensor, noise: Tensor) -> Tuple[Tensor, Tensor]:
    """(means, log-scales) of the split children, sample-major (dnsplat_densify_split)."""
    n_children, n_parents = noise.shape[0], parents.shape[0]
    dev = noise.device
    new_means = torch.empty(n_children, 3, dtype=torch.float32, device=dev)
    new_scales = torch.empty(n_children, 3, dtype=torch.float32, device=dev)
    _lib.run("dnsplat_densify_split", _lib.lib().dnsplat_densify_split, n_children, n_parents, _ptr(parents.to(torch.int32).contiguous()),
             _ptr(noise.contiguous()), _ptr(gauss_params["means"].detach().contiguous()),
             _ptr(gauss_params["scales"].detach().contiguous()), _ptr(gauss_params["quats"].detach().contiguous()),
             _ptr(new_means), _ptr(new_scales), _stream())
    return new_means, new_scales


def spatial_order(means: Tensor, bits: int = 10) -> Tensor:
    """int64 [N] permutation that lays the Gaussians out along a 3-D Morton (Z-order) curve over the bounding box of their means
    (``bits`` per axis).  The order of the rows of the parameter tensors means nothing to the reference (densification appends and
    culls, dn_model.py:309-357), but it decides what the per-Gaussian kernels move: a camera culls ~28 % of the benchmark scenes'
    Gaussians, and in a random order every 64-Gaussian workgroup of ``dnsplat_project_fwd`` / ``_bwd`` holds a few of them, whose SH
    rows share their cache lines with visible neighbours (1.3 x the algorithmic bytes).  Along the curve any
    frustum cuts the rows into a few contiguous runs: workgroups are all visible or all culled.  Pure torch (one sort); identical on
    every rank for identical parameters (stable sort on integer keys)."""
    m = torch.nan_to_num(means.detach().float())
    lo, hi = m.amin(0), m.amax(0)
    top = (1 << bits) - 1
    q = ((m - lo) / (hi - lo).clamp_min(1e-30) * top).to(torch.int64).clamp_(0, top)

    def spread(v):                      # bit i -> bit 3 i (bits <= 21)
        out = torch.zeros_like(v)
        for i in range(bits):
            out |= ((v >> i) & 1) << (3 * i)
        return out

    code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    return torch.sort(code, stable=True)[1]


def reorder(gauss_params: Dict[str, Tensor], perm: Tensor, adam_state: Optional[Dict[str, Dict[str, Tensor]]] = None):
    """Rows of every per-Gaussian tensor (parameters, ``normals``, Adam moments) gathered in the order ``perm``.  Returns
    ``(new_gauss_params, new_adam_state)`` as plain tensors, like ``refinement_after``; per-Gaussian statistics
    (``DensifyStats``) belong to the old order and start over (``after_refinement``)."""
    n = int(perm.shape[0])
    with torch.no_grad():
        new_params = {k: (v.detach().index_select(0, perm) if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == n else v)
                      for k, v in gauss_params.items()}
        new_adam = adam_state
        if adam_state is not None:
            new_adam = {name: {key: (t.index_select(0, perm) if torch.is_tensor(t) and t.dim() > 0 and t.shape[0] == n else t)
                               for key, t in st.items()} for name, st in adam_state.items()}
    return new_params, new_adam


def refinement_after(gauss_params: Dict[str, Tensor], stats: Optional[DensifyStats], cfg: RefineConfig, step: int,
                     num_train_data: int, last_size, adam_state: Optional[Dict[str, Dict[str, Tensor]]] = None,
                     seed: int = 0, classify_fn: Optional[Callable] = None, split_fn: Optional[Callable] = None,
                     spatial_reorder: bool = False):
    """One refinement step (dn_model.py:271-386) on device.  Returns ``(new_gauss_params, new_adam_state, report)`` —
    plain tensors (the caller re-wraps them as Parameters / re-seats the optimizer state, as nerfstudio's
    dup_in_all_optim / remove_from_all_optim do); ``report`` counts what happened.  ``stats`` must already be combined
    across ranks (``DensifyStats.allreduce``); ``seed`` (plus the step) seeds the split noise identically on every rank.
    ``classify_fn`` / ``split_fn`` exist so that tests can run this very function over another implementation of the two
    kernels; the package only ever passes its HIP ones.
    ``spatial_reorder``: whenever the set changed (rows were appended / removed anyway), lay the new set out along the Morton curve
    (``spatial_order``; the reference appends children and duplicates at the end — the result is then a permutation of the
    reference's, ``report["perm"]``)."""
    classify_fn = classify_fn or classify
    split_fn = split_fn or split_children
    report = dict(step=step, n_before=int(gauss_params["means"].shape[0]), n_split=0, n_dup=0, n_culled=0, opacity_reset=False)
    new_adam = adam_state
    if step <= cfg.warmup_length:
        report["n_after"] = report["n_before"]
        return gauss_params, adam_state, report
    reset_interval = cfg.reset_alpha_every * cfg.refine_every
    do_densify = step < cfg.stop_split_at and step % reset_interval > num_train_data + cfg.refine_every
    cull_only = (not do_densify) and step >= cfg.stop_split_at and cfg.continue_cull_post_densification
    with torch.no_grad():
        params = {k: v.detach() for k, v in gauss_params.items()}
        if do_densify or cull_only:
            if do_densify and stats is None:
                raise RuntimeError("densification needs the accumulated statistics (DensifyStats)")
            flags = classify_fn(params, stats, cfg, step, last_size, do_densify)
            dev = flags.device
            split_par = torch.nonzero((flags & SPLIT) != 0).reshape(-1)            # parents, ascending
            dup_src = torch.nonzero(((flags & DUP) != 0) & ((flags & CULL_DUP) == 0)).reshape(-1)
            keep = torch.nonzero((flags & CULL) == 0).reshape(-1)
            ns = cfg.n_split_samples
            # children are laid out sample-major over ALL split parents (`repeat(samps, 1)`), then filtered by their own cull
            child_parent_all = split_par.repeat(ns)
            child_keep = (flags[child_parent_all] & CULL_CHILD) == 0
            gen = torch.Generator(device=dev).manual_seed((int(seed) * 1_000_003 + int(step)) & 0x7FFFFFFF)
            noise = torch.randn(child_parent_all.shape[0], 3, device=dev, generator=gen)     # split_gaussians: centered_samples
            if child_parent_all.numel():
                ch_means, ch_scales = split_fn(params, split_par, noise)
            else:
                ch_means = ch_scales = torch.empty(0, 3, device=dev)
            child_sel = torch.nonzero(child_keep).reshape(-1)
            child_parent = child_parent_all[child_sel]
            src = torch.cat([keep, child_parent, dup_src])            # row of the old arrays each new row is copied from
            n_keep, n_child = keep.numel(), child_parent.numel()
            new_params = {}
            for name, p in params.items():
                rows = p.index_select(0, src)
                if name == "means":
                    rows[n_keep:n_keep + n_child] = ch_means[child_sel]
                elif name == "scales":
                    rows[n_keep:n_keep + n_child] = ch_scales[child_sel]
                    # split_gaussians shrinks its parents in place BEFORE the duplicates are copied (dn_model.py:309-316): the
                    # duplicate of a Gaussian that was also split carries the shrunk scale
                    both = (flags[dup_src] & SPLIT) != 0
                    if bool(both.any()):
                        d = rows[n_keep + n_child:]
                        d[both] = torch.log(torch.exp(d[both]) / 1.6)
                new_params[name] = rows
            if adam_state is not None:
                new_adam = {}
                for name, st in adam_state.items():
                    new_adam[name] = {}
                    for key, t in st.items():
                        if torch.is_tensor(t) and t.dim() > 0 and t.shape[0] == report["n_before"]:
                            rows = t.index_select(0, src)
                            rows[n_keep:] = 0                         # dup_in_optim appends zeros for the new entries
                            new_adam[name][key] = rows
                        else:
                            new_adam[name][key] = t
            # n_dup counts every Gaussian the reference duplicates (incl. duplicates its cull removes straight away, which also
            # count in n_culled); n_dup_kept is what actually stays
            report.update(n_split=int(split_par.numel()), n_dup=int(((flags & DUP) != 0).sum()), n_dup_kept=int(dup_src.numel()),
                          n_children_kept=int(n_child),
                          n_culled=report["n_before"] + ns * int(split_par.numel()) + int(((flags & DUP) != 0).sum()) - int(src.numel()))
            params = new_params
        if step < cfg.stop_split_at and step % reset_interval == cfg.refine_every:
            # opacity reset (dn_model.py:365-383): clamp to logit(2 x cull_alpha_thresh), zero the Adam moments of the opacities
            reset_value = cfg.cull_alpha_thresh * 2.0
            params = dict(params)
            params["opacities"] = torch.clamp(params["opacities"], max=torch.logit(torch.tensor(reset_value)).item())
            if new_adam is not None and "opacities" in new_adam:
                new_adam = dict(new_adam)
                new_adam["opacities"] = {k: (torch.zeros_like(v) if k in ("exp_avg", "exp_avg_sq") else v)
                                         for k, v in new_adam["opacities"].items()}
            report["opacity_reset"] = True
        if spatial_reorder and (do_densify or cull_only):
            perm = spatial_order(params["means"])
            params, new_adam = reorder(params, perm, new_adam)
            report["perm"] = perm
    report["n_after"] = int(params["means"].shape[0])
    return params, new_adam, report


def after_refinement(new_params: Dict[str, Tensor], stats=None, report: Optional[dict] = None):
    """What the caller of ``refinement_after`` has to renew when the Gaussian set changed size, in one place: the statistics start
    over (dn_model.py:359-363 sets xys_grad_norm / vis_counts / max_2Dsize to None), the flat gradient bucket and the SH factor
    exchange are rebuilt for the new tensors (their slices and data pointers belong to the old ones), and the binning capacity
    guesses of the old size are dropped.  Returns the new ``dp.GradArena`` (or None if none was installed)."""
    from . import _ops, dp

    if stats is not None:
        stats.reset(int(new_params["means"].shape[0]))
    # the capacity guesses / static capacities / running maxima are keyed by the number of Gaussians: those of the old size would
    # only accumulate (the scratch buffers themselves are grow-only and stay)
    dev = new_params["means"].device
    n_new, n_old = int(new_params["means"].shape[0]), (report or {}).get("n_before")
    if n_old and n_old != n_new:
        # ... carried over to the new size in proportion (a capture right after the refinement then needs no eager frames over the
        # poses to size its buffers again: _ops.carry_capacity_guesses); without the report they are dropped
        _ops.carry_capacity_guesses(dev if dev.type == "cuda" else None, int(n_old), n_new)
    elif not n_old:
        _ops.forget_capacity_guesses(dev if dev.type == "cuda" else None)
    arena = None
    if _ops.GRAD_ARENA is not None:
        arena = dp.GradArena({k: new_params[k] for k in dp.GRAD_KEYS})
        _ops.set_grad_arena(arena)
    if _ops.SH_EXCHANGE is not None:
        old = _ops.SH_EXCHANGE       # the same kind of exchange (dp.SlicedShExchange keeps its slice count)
        _ops.set_sh_exchange(dp.SlicedShExchange(old.slices) if isinstance(old, dp.SlicedShExchange) else dp.ShFactorExchange())
    return arena
