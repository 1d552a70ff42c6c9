"""CPU oracle for the dn-splatter rendering hot path (TEST INFRASTRUCTURE ONLY).

Python face of ``oracle/dnsplat_oracle.c``: the four symbols dn-splatter imports
from gsplat (``dn_splatter/dn_model.py:29-35``) restated on the CPU with torch
autograd wiring, so a parity test can call them exactly the way
``DNSplatterModel.get_outputs`` (``dn_splatter/dn_model.py:495-516``, ``:564-575``)
calls gsplat.

PARITY UNPINNED: gsplat==1.0.0 is neither vendored in the reference nor
installable here; see the header of ``dnsplat_oracle.c``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product package never does.
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess
from pathlib import Path
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "_build" / "libdnsplat_oracle.so"
_lib = None

ALPHA_MIN = 1.0 / 255.0
ALPHA_MAX = 0.999
T_MIN = 1e-4


def build(force: bool = False) -> Path:
    """Compile the oracle with gcc (``make -C oracle``)."""
    if force or not _LIB_PATH.exists() or any(
        (_HERE / f).stat().st_mtime > _LIB_PATH.stat().st_mtime
        for f in ("dnsplat_oracle.c", "oracle_impl.inc")
    ):
        subprocess.run(["make", "-C", str(_HERE)] + (["-B"] if force else []), check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(str(_LIB_PATH))
        _lib.orc_isect_tiles_f32.restype = ctypes.c_int64
        _lib.orc_isect_tiles_f64.restype = ctypes.c_int64
    return _lib


_KEEP: list = []  # tensors whose storage must outlive the current C call (``x.contiguous()`` temporaries)


def set_exact_accumulation(on: bool) -> bool:
    """Compositing backward: accumulate the gradient scatter in double instead of fp32 omp atomics (dnsplat_oracle.c
    ``orc_exact_accum``) — an order-independent result for the deterministic-mode checks.  Returns the previous setting."""
    L = lib()
    prev = bool(L.orc_get_exact_accum())
    L.orc_set_exact_accum(ctypes.c_int(1 if on else 0))
    return prev


def _p(t: Optional[Tensor]):
    """Pointer to a contiguous CPU tensor.  The tensor is parked in _KEEP so a temporary made by
    ``.contiguous()`` cannot be freed (and its storage recycled) before the C call runs."""
    if t is None:
        return ctypes.c_void_p(0)
    assert t.is_contiguous() and t.device.type == "cpu", "oracle wants contiguous CPU tensors"
    _KEEP.append(t)
    if len(_KEEP) > 256:
        del _KEEP[:128]
    return ctypes.c_void_p(t.data_ptr())


def _suffix(t: Tensor) -> str:
    if t.dtype == torch.float32:
        return "f32"
    if t.dtype == torch.float64:
        return "f64"
    raise TypeError(f"oracle supports float32/float64, got {t.dtype}")


def _real(t: Tensor, v: float):
    return ctypes.c_float(v) if t.dtype == torch.float32 else ctypes.c_double(v)


# --------------------------------------------------------------------------- helpers


def num_sh_bases(degree: int) -> int:
    """gsplat.cuda_legacy._wrapper.num_sh_bases (dn_model.py:35,139)."""
    table = {0: 1, 1: 4, 2: 9, 3: 16, 4: 25}
    if degree not in table:
        raise AssertionError("We don't support degree greater than 4.")
    return table[degree]


def quat_to_rotmat(quat: Tensor) -> Tensor:
    """gsplat.cuda_legacy._torch_impl.quat_to_rotmat (dn_model.py:34,547): wxyz, normalises."""
    assert quat.shape[-1] == 4, quat.shape
    w, x, y, z = torch.unbind(torch.nn.functional.normalize(quat, dim=-1), dim=-1)
    mat = torch.stack(
        [
            1 - 2 * (y**2 + z**2), 2 * (x * y - w * z), 2 * (x * z + w * y),
            2 * (x * y + w * z), 1 - 2 * (x**2 + z**2), 2 * (y * z - w * x),
            2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x**2 + y**2),
        ],
        dim=-1,
    )
    return mat.reshape(quat.shape[:-1] + (3, 3))


# --------------------------------------------------------------------------- raw kernels


def project_fwd(means, quats, scales, viewmat, K, width, height, eps2d=0.3, near_plane=0.01,
                far_plane=1e10, radius_clip=0.0, tile_size=16, calc_compensations=False):
    N = means.shape[0]
    dt = means.dtype
    radii = torch.zeros(N, dtype=torch.int32)
    means2d = torch.zeros(N, 2, dtype=dt)
    depths = torch.zeros(N, dtype=dt)
    conics = torch.zeros(N, 3, dtype=dt)
    comp = torch.zeros(N, dtype=dt) if calc_compensations else None
    tiles = torch.zeros(N, dtype=torch.int32)
    fn = getattr(lib(), "orc_project_fwd_" + _suffix(means))
    fn(ctypes.c_int(N), _p(means.contiguous()), _p(quats.contiguous()), _p(scales.contiguous()),
       _p(viewmat.contiguous()), _p(K.contiguous()), ctypes.c_int(width), ctypes.c_int(height),
       _real(means, eps2d), _real(means, near_plane), _real(means, far_plane), _real(means, radius_clip),
       ctypes.c_int(tile_size), _p(radii), _p(means2d), _p(depths), _p(conics), _p(comp), _p(tiles))
    return radii, means2d, depths, conics, comp, tiles


def project_edge(means, quats, scales, viewmat, K, width, height, eps2d=0.3, near_plane=0.01, far_plane=1e10,
                 radius_clip=0.0, tile_size=16):
    """bool [N]: Gaussians whose integer outputs hinge on a comparison inside the fp32 rounding envelope of the activated
    inputs (orc_project_edge).  Only needed by parity tests that hand both sides RAW parameters."""
    N = means.shape[0]
    edge = torch.zeros(N, dtype=torch.uint8)
    fn = getattr(lib(), "orc_project_edge_" + _suffix(means))
    fn(ctypes.c_int(N), _p(means.contiguous()), _p(quats.contiguous()), _p(scales.contiguous()),
       _p(viewmat.contiguous()), _p(K.contiguous()), ctypes.c_int(width), ctypes.c_int(height),
       _real(means, eps2d), _real(means, near_plane), _real(means, far_plane), _real(means, radius_clip),
       ctypes.c_int(tile_size), _p(edge))
    return edge.bool()


def project_bwd(means, quats, scales, viewmat, K, width, height, eps2d, near_plane, far_plane,
                radius_clip, radii, v_means2d, v_depths, v_conics, v_compensations=None):
    N = means.shape[0]
    dt = means.dtype
    v_means = torch.zeros(N, 3, dtype=dt)
    v_quats = torch.zeros(N, 4, dtype=dt)
    v_scales = torch.zeros(N, 3, dtype=dt)
    fn = getattr(lib(), "orc_project_bwd_" + _suffix(means))
    fn(ctypes.c_int(N), _p(means.contiguous()), _p(quats.contiguous()), _p(scales.contiguous()),
       _p(viewmat.contiguous()), _p(K.contiguous()), ctypes.c_int(width), ctypes.c_int(height),
       _real(means, eps2d), _real(means, near_plane), _real(means, far_plane), _real(means, radius_clip),
       _p(radii), _p(v_means2d.contiguous()), _p(v_depths.contiguous()), _p(v_conics.contiguous()),
       _p(v_compensations.contiguous() if v_compensations is not None else None),
       _p(v_means), _p(v_quats), _p(v_scales))
    return v_means, v_quats, v_scales


def sh_fwd(degree, dirs, coeffs, radii=None):
    N, Ktot = coeffs.shape[0], coeffs.shape[1]
    colors = torch.zeros(N, 3, dtype=coeffs.dtype)
    fn = getattr(lib(), "orc_sh_fwd_" + _suffix(coeffs))
    fn(ctypes.c_int(N), ctypes.c_int(degree), ctypes.c_int(Ktot), _p(dirs.contiguous()),
       _p(coeffs.contiguous()), _p(radii), _p(colors))
    return colors


def sh_bwd(degree, dirs, coeffs, radii, v_colors, need_dirs=True):
    N, Ktot = coeffs.shape[0], coeffs.shape[1]
    v_coeffs = torch.zeros_like(coeffs)
    v_dirs = torch.zeros(N, 3, dtype=coeffs.dtype) if need_dirs else None
    fn = getattr(lib(), "orc_sh_bwd_" + _suffix(coeffs))
    fn(ctypes.c_int(N), ctypes.c_int(degree), ctypes.c_int(Ktot), _p(dirs.contiguous()),
       _p(coeffs.contiguous()), _p(radii), _p(v_colors.contiguous()), _p(v_coeffs), _p(v_dirs))
    return v_coeffs, v_dirs


def isect_tiles(means2d, radii, depths, tile_size, tile_width, tile_height, sort=True):
    """A.3: (tiles_per_gauss, isect_ids[int64], flatten_ids[int32]) for one camera."""
    N = means2d.shape[0]
    sfx = _suffix(means2d)
    fn = getattr(lib(), "orc_isect_tiles_" + sfx)
    tiles = torch.zeros(N, dtype=torch.int32)
    m2 = means2d.contiguous()
    d = depths.contiguous()
    r = radii.contiguous()
    n = fn(ctypes.c_int(N), _p(m2), _p(r), _p(d), ctypes.c_int(tile_size), ctypes.c_int(tile_width),
           ctypes.c_int(tile_height), _p(tiles), _p(None), _p(None))
    isect_ids = torch.zeros(n, dtype=torch.int64)
    flatten_ids = torch.zeros(n, dtype=torch.int32)
    fn(ctypes.c_int(N), _p(m2), _p(r), _p(d), ctypes.c_int(tile_size), ctypes.c_int(tile_width),
       ctypes.c_int(tile_height), _p(tiles), _p(isect_ids), _p(flatten_ids))
    if sort:
        lib().orc_sort_isects(ctypes.c_int64(n), _p(isect_ids), _p(flatten_ids))
    return tiles, isect_ids, flatten_ids


def isect_offset_encode(isect_ids, tile_width, tile_height):
    T = tile_width * tile_height
    offsets = torch.zeros(T, dtype=torch.int32)
    lib().orc_isect_offsets(ctypes.c_int64(isect_ids.shape[0]), _p(isect_ids.contiguous()),
                            ctypes.c_int(T), _p(offsets))
    return offsets.reshape(tile_height, tile_width)


def rasterize_fwd(means2d, conics, colors, opacities, background, width, height, tile_size,
                  isect_offsets, flatten_ids, borderline=None, flip_weight=None):
    """``borderline``: optional uint8 [H,W] output, 1 where a hard decision of the rule set fell inside the fp32
    rounding envelope (see orc_rasterize_fwd); ``flip_weight``: optional float32 [H,W] output (needs ``borderline``), the
    compositing weight those decisions can move at the pixel."""
    N, D = colors.shape
    dt = colors.dtype
    render = torch.zeros(height, width, D, dtype=dt)
    alphas = torch.zeros(height, width, dtype=dt)
    last_ids = torch.zeros(height, width, dtype=torch.int32)
    fn = getattr(lib(), "orc_rasterize_fwd_" + _suffix(colors))
    fn(ctypes.c_int(N), ctypes.c_int(D), _p(means2d.contiguous()), _p(conics.contiguous()),
       _p(colors.contiguous()), _p(opacities.contiguous()),
       _p(background.contiguous() if background is not None else None),
       ctypes.c_int(width), ctypes.c_int(height), ctypes.c_int(tile_size),
       _p(isect_offsets.contiguous()), _p(flatten_ids.contiguous()), ctypes.c_int64(flatten_ids.shape[0]),
       _p(render), _p(alphas), _p(last_ids), _p(borderline), _p(flip_weight))
    return render, alphas, last_ids


def rasterize_bwd(means2d, conics, colors, opacities, background, width, height, tile_size,
                  isect_offsets, flatten_ids, alphas, last_ids, v_render, v_alphas, absgrad=False):
    N, D = colors.shape
    dt = colors.dtype
    v_means2d = torch.zeros(N, 2, dtype=dt)
    v_abs = torch.zeros(N, 2, dtype=dt) if absgrad else None
    v_conics = torch.zeros(N, 3, dtype=dt)
    v_colors = torch.zeros(N, D, dtype=dt)
    v_opac = torch.zeros(N, dtype=dt)
    fn = getattr(lib(), "orc_rasterize_bwd_" + _suffix(colors))
    fn(ctypes.c_int(N), ctypes.c_int(D), _p(means2d.contiguous()), _p(conics.contiguous()),
       _p(colors.contiguous()), _p(opacities.contiguous()),
       _p(background.contiguous() if background is not None else None),
       ctypes.c_int(width), ctypes.c_int(height), ctypes.c_int(tile_size),
       _p(isect_offsets.contiguous()), _p(flatten_ids.contiguous()), ctypes.c_int64(flatten_ids.shape[0]),
       _p(alphas.contiguous()), _p(last_ids.contiguous()), _p(v_render.contiguous()), _p(v_alphas.contiguous()),
       _p(v_means2d), _p(v_abs), _p(v_conics), _p(v_colors), _p(v_opac))
    return v_means2d, v_abs, v_conics, v_colors, v_opac


def rasterize_bwd_hull(means2d, conics, colors, opacities, background, width, height, tile_size, isect_offsets, flatten_ids,
                       mask, v_render, v_alphas, max_flags=10, combo=-1):
    """Backward of the pixels selected by ``mask`` (bool [H,W]) under every combination of outcomes of their flagged
    (borderline) decisions — orc_rasterize_bwd_hull: per gradient entry the smallest / largest value any admissible combination
    gives, as float64 tensors.  Returns (lo, hi, status) with lo / hi dicts of means2d [N,2], absgrad [N,2], conics [N,3],
    colors [N,D], opacities [N] and status = dict(max_flags_seen, pixels_over_cap, pixels_incomplete, pixels).  ``combo >= 0``:
    that single combination only (bit j = outcome of the pixel's j-th flagged decision), lo == hi."""
    N, D = colors.shape
    n_tot = (8 + D) * N
    lo = torch.zeros(n_tot, dtype=torch.float64)
    hi = torch.zeros(n_tot, dtype=torch.float64)
    status = torch.zeros(4, dtype=torch.int64)
    fn = getattr(lib(), "orc_rasterize_bwd_hull_" + _suffix(colors))
    fn(ctypes.c_int(N), ctypes.c_int(D), _p(means2d.contiguous()), _p(conics.contiguous()), _p(colors.contiguous()),
       _p(opacities.contiguous()), _p(background.contiguous() if background is not None else None),
       ctypes.c_int(width), ctypes.c_int(height), ctypes.c_int(tile_size),
       _p(isect_offsets.contiguous()), _p(flatten_ids.contiguous()), ctypes.c_int64(flatten_ids.shape[0]),
       _p(mask.to(torch.uint8).contiguous()), _p(v_render.contiguous()), _p(v_alphas.contiguous()),
       ctypes.c_int(max_flags), ctypes.c_int(combo), _p(lo), _p(hi), _p(status))

    def split(t):
        return {"means2d": t[:2 * N].view(N, 2), "absgrad": t[2 * N:4 * N].view(N, 2), "conics": t[4 * N:7 * N].view(N, 3),
                "colors": t[7 * N:(7 + D) * N].view(N, D), "opacities": t[(7 + D) * N:]}

    st = dict(max_flags_seen=int(status[0]), pixels_over_cap=int(status[1]), pixels_incomplete=int(status[2]), pixels=int(status[3]))
    return split(lo), split(hi), st


def input_perturbation(means2d, conics, width, height):
    """[N,3] float64 for ``rasterize_bwd_cond(pert=)``: how many roundings (2^-24) the projection's fp32 outputs are away from exact —
    the conic relatively: 8 x kappa_det, kappa_det = (a c + b^2) / (a c - b^2) the cancellation of the 2x2 determinant behind the
    inverse (1 for a round splat, 10-100 for an elongated diagonal one); the centre absolutely, in pixels:
    4 x (|x| + W/2), 4 x (|y| + H/2) (f x / z + c: products, a quotient and the sum with the principal point)."""
    m = means2d.detach().double().reshape(-1, 2)
    c = conics.detach().double().reshape(-1, 3)
    ac, b2 = c[:, 0] * c[:, 2], c[:, 1] * c[:, 1]
    kdet = torch.where(ac - b2 > 0, (ac + b2) / (ac - b2).clamp_min(1e-300), torch.ones_like(ac))
    return torch.stack([8.0 * kdet, 4.0 * (m[:, 0].abs() + width / 2), 4.0 * (m[:, 1].abs() + height / 2)], dim=1).contiguous()


def rasterize_bwd_cond(means2d, conics, colors, opacities, background, width, height, tile_size, isect_offsets, flatten_ids,
                       alphas, last_ids, vabs_render, vabs_alphas, pert=None):
    """orc_rasterize_bwd_cond: per (Gaussian, raster-level gradient entry) the kappa-weighted sum of term magnitudes of A.7, float64
    [N, 8 + D] with columns (vx vy |vx| |vy| conic_a conic_b conic_c opacity colours[D]) — the running error bound, in units of one
    rounding, of that entry under any evaluation in the precision of the inputs (see oracle_impl.inc)."""
    N, D = colors.shape
    A = torch.zeros(N, 8 + D, dtype=torch.float64)
    B = torch.zeros(N, 8 + D, dtype=torch.float64)
    fn = getattr(lib(), "orc_rasterize_bwd_cond_" + _suffix(colors))
    fn(ctypes.c_int(N), ctypes.c_int(D), _p(means2d.contiguous()), _p(conics.contiguous()), _p(colors.contiguous()),
       _p(opacities.contiguous()), _p(background.contiguous() if background is not None else None),
       ctypes.c_int(width), ctypes.c_int(height), ctypes.c_int(tile_size),
       _p(isect_offsets.contiguous()), _p(flatten_ids.contiguous()), ctypes.c_int64(flatten_ids.shape[0]),
       _p(alphas.contiguous()), _p(last_ids.contiguous()), _p(vabs_render.contiguous()), _p(vabs_alphas.contiguous()), _p(A), _p(B),
       _p(pert.contiguous() if pert is not None else None))
    return A, B


class ConditionTrace:
    """Context manager: while active, every compositing call of this module (``_RasterizeToPixels``) remembers the autograd tensors
    it was fed (means2d, conics, colours, opacities) and, when its backward runs, the running error bound ``A`` [N, 8 + D] of its
    raster-level gradients for the cotangents it received (``rasterize_bwd_cond``).  Afterwards ``param_condition(leaves)`` chains
    those bounds to the leaf parameters through the entry-wise absolute value of the per-Gaussian Jacobian of everything between the
    leaves and the compositing inputs (activations, projection, SH, normals — Gaussians are independent there, so J column i is one
    backward pass with a unit cotangent in column i):
        cond[leaf][g] = sum over calls, over raster inputs i of |d input_i[g] / d leaf[g]| x A_call[g][i].
    The main backward must keep the graph (``retain_graph=True``) for the Jacobian passes.  Test infrastructure for
    tests/test_gpu_determinism.py: ||hip_g - oracle_g|| <= c x 2^-24 x ||cond_g|| for EVERY visible Gaussian."""

    def __init__(self, input_rounding: bool = True):
        self.calls = []
        # count the roundings of the projection's outputs (conic, centre) as a coherent shift of each splat's alphas (input_perturbation)
        self.input_rounding = input_rounding

    def __enter__(self):
        global _TRACE
        self._prev, _TRACE = _TRACE, self
        return self

    def __exit__(self, *exc):
        global _TRACE
        _TRACE = self._prev
        return False

    def register(self, means2d, conics, colors, opacities):
        slot = {"inputs": (means2d, conics, colors, opacities), "A": None, "B": None}
        self.calls.append(slot)
        return slot

    # columns of A belonging to each compositing input
    @staticmethod
    def _columns(D):
        return {0: [0, 1], 1: [4, 5, 6], 2: list(range(8, 8 + D)), 3: [7]}

    def raster_condition(self, call: int = 0, which: str = "A"):
        """dict(means2d [N,2], absgrad [N,2], conics [N,3], colors [N,D], opacities [N]) of compositing call ``call``: the worst-case
        bound A, or with ``which="B"`` the standard deviation sqrt(B) of the independent-roundings model (both in units of u)."""
        A = self.calls[call][which]
        if which == "B":
            A = A.sqrt()
        D = A.shape[1] - 8
        return {"means2d": A[:, 0:2], "absgrad": A[:, 2:4], "conics": A[:, 4:7], "opacities": A[:, 7], "colors": A[:, 8:8 + D]}

    def param_condition(self, leaves: Dict[str, Tensor]):
        """-> (A, S): dicts leaf name -> float64 tensor of the leaf's shape.  A = sum_i |J_i| A_i (worst case),
        S = sqrt(sum_i J_i^2 B_i) (independent roundings), both in units of one rounding.  One Jacobian sweep serves both."""
        names = [k for k, v in leaves.items() if v.requires_grad]
        out_a = {k: torch.zeros(leaves[k].shape, dtype=torch.float64) for k in names}
        out_b = {k: torch.zeros(leaves[k].shape, dtype=torch.float64) for k in names}
        # tensors the caller called retain_grad() on (info["means2d"], dn_model.py:517-518) would collect the unit cotangents of the
        # sweeps in their .grad / keep the last sweep's .absgrad: put both back afterwards
        held = [(t, t.grad, getattr(t, "absgrad", None)) for slot in self.calls for t in slot["inputs"]
                if torch.is_tensor(t) and t.requires_grad and t.retains_grad]
        lib().orc_set_cond_quat_abs(ctypes.c_int(1))      # the quaternion stage of orc_project_bwd in magnitudes (see there)
        try:
            self._sweep(leaves, names, out_a, out_b)
        finally:
            lib().orc_set_cond_quat_abs(ctypes.c_int(0))
            for t, g, ag in held:
                t.grad = g
                if ag is not None:
                    t.absgrad = ag
        return out_a, {k: v.sqrt() for k, v in out_b.items()}

    def _sweep(self, leaves, names, out_a, out_b):
        for slot in self.calls:
            A, B = slot["A"], slot["B"]
            if A is None:
                continue
            N, D = A.shape[0], A.shape[1] - 8
            cols = self._columns(D)
            for slot_input, t in enumerate(slot["inputs"]):
                if not (torch.is_tensor(t) and t.requires_grad):
                    continue
                t2 = t.reshape(N, -1)
                for j, col in enumerate(cols[slot_input]):
                    if j >= t2.shape[1]:
                        break
                    if float(A[:, col].max()) == 0.0:
                        continue
                    unit = torch.zeros_like(t2)
                    unit[:, j] = 1.0
                    grads = torch.autograd.grad([t], [leaves[k] for k in names], grad_outputs=[unit.reshape(t.shape)],
                                                retain_graph=True, allow_unused=True)
                    for k, gk in zip(names, grads):
                        if gk is not None:
                            j_abs = gk.detach().double().abs()
                            shape = (N,) + (1,) * (gk.dim() - 1)
                            out_a[k] += j_abs * A[:, col].reshape(shape)
                            out_b[k] += j_abs * j_abs * B[:, col].reshape(shape)


_TRACE: Optional["ConditionTrace"] = None


# --------------------------------------------------------------------------- autograd wiring


class _Projection(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means, quats, scales, viewmat, K, width, height, eps2d, near, far, radius_clip,
                tile_size, calc_comp):
        radii, means2d, depths, conics, comp, tiles = project_fwd(
            means, quats, scales, viewmat, K, width, height, eps2d, near, far, radius_clip, tile_size, calc_comp)
        ctx.save_for_backward(means, quats, scales, viewmat, K, radii)
        ctx.cfg = (width, height, eps2d, near, far, radius_clip)
        ctx.calc_comp = calc_comp
        ctx.mark_non_differentiable(radii, tiles)
        if comp is None:
            comp = torch.zeros(0, dtype=means.dtype)
        return radii, means2d, depths, conics, comp, tiles

    @staticmethod
    def backward(ctx, _vr, v_means2d, v_depths, v_conics, v_comp, _vt):
        means, quats, scales, viewmat, K, radii = ctx.saved_tensors
        width, height, eps2d, near, far, radius_clip = ctx.cfg
        v_means, v_quats, v_scales = project_bwd(
            means, quats, scales, viewmat, K, width, height, eps2d, near, far, radius_clip, radii,
            v_means2d, v_depths, v_conics, v_comp if ctx.calc_comp else None)
        return (v_means, v_quats, v_scales) + (None,) * 10


class _SphericalHarmonics(torch.autograd.Function):
    @staticmethod
    def forward(ctx, degree, dirs, coeffs, radii):
        colors = sh_fwd(degree, dirs, coeffs, radii)
        ctx.save_for_backward(dirs, coeffs, radii)
        ctx.degree = degree
        return colors

    @staticmethod
    def backward(ctx, v_colors):
        dirs, coeffs, radii = ctx.saved_tensors
        v_coeffs, v_dirs = sh_bwd(ctx.degree, dirs, coeffs, radii, v_colors, ctx.needs_input_grad[1])
        return None, v_dirs, v_coeffs, None


class _RasterizeToPixels(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means2d, conics, colors, opacities, background, width, height, tile_size,
                isect_offsets, flatten_ids, absgrad, borderline=None, flip_weight=None):
        # means2d may arrive as [1,N,2] (the tensor dn_model.py:517-519 retains grad on) or [N,2]
        render, alphas, last_ids = rasterize_fwd(means2d.reshape(-1, 2), conics, colors, opacities, background,
                                                 width, height, tile_size, isect_offsets, flatten_ids, borderline, flip_weight)
        ctx.save_for_backward(means2d, conics, colors, opacities, isect_offsets, flatten_ids, alphas, last_ids)
        ctx.background = background
        ctx.cfg = (width, height, tile_size, absgrad)
        ctx.trace_slot = _TRACE.register(means2d, conics, colors, opacities) if _TRACE is not None else None
        return render, alphas

    @staticmethod
    def backward(ctx, v_render, v_alphas):
        means2d, conics, colors, opacities, isect_offsets, flatten_ids, alphas, last_ids = ctx.saved_tensors
        width, height, tile_size, absgrad = ctx.cfg
        v_means2d, v_abs, v_conics, v_colors, v_opac = rasterize_bwd(
            means2d.reshape(-1, 2), conics, colors, opacities, ctx.background, width, height, tile_size, isect_offsets,
            flatten_ids, alphas, last_ids, v_render, v_alphas, absgrad)
        if absgrad:
            means2d.absgrad = v_abs.reshape(means2d.shape)
        if ctx.trace_slot is not None:
            # the compositing inputs are the fp32 projection's outputs when they come out of this module's graph (requires_grad / grad_fn),
            # exact when the caller hands them in as data (the legacy call's detached xys: still rounded values -> keep the term)
            pert = input_perturbation(means2d, conics, width, height) if _TRACE is None or _TRACE.input_rounding else None
            A, B = rasterize_bwd_cond(means2d.reshape(-1, 2), conics, colors, opacities, ctx.background, width, height, tile_size,
                                      isect_offsets, flatten_ids, alphas, last_ids, v_render.abs(), v_alphas.abs(), pert)
            ctx.trace_slot["A"] = A if ctx.trace_slot["A"] is None else ctx.trace_slot["A"] + A
            ctx.trace_slot["B"] = B if ctx.trace_slot["B"] is None else ctx.trace_slot["B"] + B
        return (v_means2d.reshape(means2d.shape), v_conics, v_colors, v_opac) + (None,) * 9


# --------------------------------------------------------------------------- gsplat-shaped API


def rasterization(
    means: Tensor, quats: Tensor, scales: Tensor, opacities: Tensor, colors: Tensor,
    viewmats: Tensor, Ks: Tensor, width: int, height: int,
    near_plane: float = 0.01, far_plane: float = 1e10, radius_clip: float = 0.0, eps2d: float = 0.3,
    sh_degree: Optional[int] = None, packed: bool = False, tile_size: int = 16,
    backgrounds: Optional[Tensor] = None, render_mode: str = "RGB", sparse_grad: bool = False,
    absgrad: bool = False, rasterize_mode: str = "classic",
) -> Tuple[Tensor, Tensor, Dict]:
    """CPU restatement of ``gsplat.rendering.rasterization`` as called at
    ``dn_splatter/dn_model.py:495-516`` (single camera, packed=False)."""
    assert render_mode in ("RGB", "D", "ED", "RGB+D", "RGB+ED"), render_mode
    if viewmats.shape[0] > 1:
        return _rasterization_batch(means, quats, scales, opacities, colors, viewmats, Ks, width, height, near_plane=near_plane,
                                    far_plane=far_plane, radius_clip=radius_clip, eps2d=eps2d, sh_degree=sh_degree, packed=packed,
                                    tile_size=tile_size, backgrounds=backgrounds, render_mode=render_mode, sparse_grad=sparse_grad,
                                    absgrad=absgrad, rasterize_mode=rasterize_mode)
    assert not packed and not sparse_grad
    N = means.shape[0]
    viewmat, K = viewmats[0], Ks[0]
    tw, th = math.ceil(width / tile_size), math.ceil(height / tile_size)

    radii, means2d, depths, conics, comp, tiles = _Projection.apply(
        means, quats, scales, viewmat, K, width, height, eps2d, near_plane, far_plane, radius_clip,
        tile_size, rasterize_mode == "antialiased")
    if rasterize_mode == "antialiased":
        opacities = opacities * comp

    with torch.no_grad():
        _t, isect_ids, flatten_ids = isect_tiles(means2d.detach(), radii, depths.detach(), tile_size, tw, th)
        isect_offsets = isect_offset_encode(isect_ids, tw, th)

    if sh_degree is None:
        cols = colors.reshape(N, -1)
    else:
        camtoworld = torch.inverse(viewmat)
        dirs = means - camtoworld[:3, 3][None, :]
        cols = _SphericalHarmonics.apply(sh_degree, dirs, colors, radii)
        cols = torch.clamp_min(cols + 0.5, 0.0)

    if render_mode in ("RGB+D", "RGB+ED"):
        cols = torch.cat([cols, depths[:, None]], dim=-1)
    elif render_mode in ("D", "ED"):
        cols = depths[:, None]

    bg = backgrounds[0] if backgrounds is not None else None
    if bg is not None and render_mode in ("D", "ED"):
        bg = bg.new_zeros(1)                        # gsplat replaces the background by zeros in the depth-only modes
    elif bg is not None and bg.shape[0] == cols.shape[-1] - 1 and render_mode != "RGB":
        bg = torch.cat([bg, bg.new_zeros(1)])      # gsplat appends a zero background for the depth channel
    assert bg is None or bg.shape[0] == cols.shape[-1], (bg.shape, cols.shape)
    means2d_c = means2d[None]  # [1,N,2]: the object the caller retains grad / reads .absgrad on
    borderline = torch.zeros(height, width, dtype=torch.uint8)
    flip_weight = torch.zeros(height, width, dtype=torch.float32)
    render, alphas = _RasterizeToPixels.apply(means2d_c, conics, cols, opacities, bg, width, height,
                                              tile_size, isect_offsets, flatten_ids, absgrad, borderline, flip_weight)
    # what one flipped decision can move, per channel: the largest |colour| any visible splat (or the background) contributes
    vis_rows = cols.detach()[radii > 0]
    channel_absmax = vis_rows.abs().amax(0) if vis_rows.numel() else cols.new_zeros(cols.shape[-1])
    if bg is not None:
        channel_absmax = torch.maximum(channel_absmax, bg.detach().abs().to(channel_absmax.dtype))
    raw_last = render.detach()[..., -1].clone()     # the depth channel before the ED division
    if render_mode in ("ED", "RGB+ED"):
        render = torch.cat([render[..., :-1], render[..., -1:] / alphas[..., None].clamp(min=1e-10)], dim=-1)

    meta = {
        "camera_ids": None, "gaussian_ids": None,
        "radii": radii[None], "means2d": means2d_c, "depths": depths[None], "conics": conics[None],
        "opacities": opacities[None], "tile_width": tw, "tile_height": th,
        "tiles_per_gauss": tiles[None], "isect_ids": isect_ids, "flatten_ids": flatten_ids,
        "isect_offsets": isect_offsets[None], "width": width, "height": height, "tile_size": tile_size,
        "n_cameras": 1,
        # not gsplat keys: pixels whose skip / stop / clamp decisions fell inside the fp32 rounding envelope, and
        # Gaussians whose integer outputs hinge on the last place of the activated inputs
        "borderline": borderline.bool(),
        # per pixel: the compositing weight the flagged decisions can move (orc_rasterize_fwd), and the per-channel |colour| bound
        # that turns it into a bound on the image (tests/_scenes.py flip_bound_*)
        "flip_weight": flip_weight, "channel_absmax": channel_absmax.float(), "raw_depth_channel": raw_last,
        "edge_gaussians": project_edge(means.detach(), quats.detach(), scales.detach(), viewmat, K, width, height, eps2d,
                                       near_plane, far_plane, radius_clip, tile_size),
    }
    return render[None], alphas[None, ..., None], meta


def _rasterization_batch(means, quats, scales, opacities, colors, viewmats, Ks, width, height, **kw):
    """C > 1 cameras (gsplat batches them in one call; SURVEY.md A.3: keys camera << (32 + tb) | tile << 32 | depth bits,
    values camera * N + g): the cameras are independent, so the oracle renders them one after the other and stacks."""
    C, N = viewmats.shape[0], means.shape[0]
    tile_size = kw.get("tile_size", 16)
    tw, th = math.ceil(width / tile_size), math.ceil(height / tile_size)
    tb = int(math.floor(math.log2(tw * th))) + 1
    bgs = kw.pop("backgrounds", None)
    outs = [rasterization(means, quats, scales, opacities, colors, viewmats[c:c + 1], Ks[c:c + 1], width, height,
                          backgrounds=None if bgs is None else bgs[c:c + 1], **kw) for c in range(C)]
    infos = [o[2] for o in outs]
    base, flat, ids, offs = 0, [], [], []
    for c, i in enumerate(infos):
        flat.append(i["flatten_ids"] + c * N)
        ids.append(i["isect_ids"] | (c << (32 + tb)))
        offs.append(i["isect_offsets"] + base)
        base += i["flatten_ids"].shape[0]
    meta = dict(infos[0])
    for k in ("radii", "means2d", "depths", "conics", "opacities", "tiles_per_gauss"):      # each [1, N, ...]
        meta[k] = torch.cat([i[k] for i in infos], 0)
    for k in ("borderline", "edge_gaussians", "flip_weight", "raw_depth_channel"):          # [H, W] / [N]
        meta[k] = torch.stack([i[k] for i in infos])
    meta["channel_absmax"] = torch.stack([i["channel_absmax"] for i in infos]).amax(0)
    meta["per_camera_means2d"] = [i["means2d"] for i in infos]     # the leaves that receive .grad / .absgrad
    meta.update(flatten_ids=torch.cat(flat), isect_ids=torch.cat(ids), isect_offsets=torch.cat(offs, 0), n_cameras=C)
    return torch.cat([o[0] for o in outs], 0), torch.cat([o[1] for o in outs], 0), meta


def rasterize_gaussians(xys, depths, radii, conics, num_tiles_hit, colors, opacity, img_height,
                        img_width, block_width, background=None, return_alpha=False):
    """CPU restatement of the legacy ``gsplat.rasterize_gaussians`` as called at
    ``dn_splatter/dn_model.py:564-575``: re-bin, re-sort, composite with background
    defaulting to ones."""
    assert 1 < block_width <= 16
    if background is None:
        background = torch.ones(colors.shape[-1], dtype=colors.dtype)
    tw, th = math.ceil(img_width / block_width), math.ceil(img_height / block_width)
    with torch.no_grad():
        _t, isect_ids, flatten_ids = isect_tiles(xys.detach(), radii, depths.detach(), block_width, tw, th)
        offsets = isect_offset_encode(isect_ids, tw, th)
    if flatten_ids.shape[0] < 1:
        out = torch.ones(img_height, img_width, colors.shape[-1], dtype=colors.dtype) * background
        return (out, torch.zeros(img_height, img_width, dtype=colors.dtype)) if return_alpha else out
    opac = opacity.reshape(-1)
    global last_borderline, last_flip_weight
    border = torch.zeros(img_height, img_width, dtype=torch.uint8)
    flip = torch.zeros(img_height, img_width, dtype=torch.float32)
    render, alphas = _RasterizeToPixels.apply(xys, conics, colors, opac, background, img_width, img_height,
                                              block_width, offsets, flatten_ids, False, border, flip)
    last_borderline = border.bool()     # the legacy call returns no info dict: parity tests read the mask (and the weight) here
    last_flip_weight = flip
    return (render, alphas) if return_alpha else render


last_borderline: Optional[Tensor] = None
last_flip_weight: Optional[Tensor] = None


def tight_tile_boxes(means2d: Tensor, conics: Tensor, opacities: Tensor, radii: Tensor, tile_size: int, tile_width: int,
                     tile_height: int):
    """CPU restatement of ``dns_snug_tile_bbox`` (dn-splatter_amd/csrc/splat_common.h; dnsplat_camera.tight_tiles in
    include/dnsplat.h): per Gaussian the tile box [x0, x1) x [y0, y1) of the ellipse sigma <= ln(255 opacity), widened by the
    kernel's rounding margin and clipped to gsplat's 3-sigma box.  Not a gsplat rule — gsplat 1.0.0 always uses
    the 3-sigma box — but the rule the product's fused path bins with; tests check on the CPU that every tile it leaves out is
    unreachable (alpha < 1/255 at all pixel centres).  float32 like the kernel.  Returns int64 tensors (x0, y0, x1, y1)."""
    f = torch.float32
    mx, my = means2d[..., 0].to(f), means2d[..., 1].to(f)
    ca, cb, cc = conics[..., 0].to(f), conics[..., 1].to(f), conics[..., 2].to(f)
    o, r = opacities.to(f), radii.to(f)
    ts = float(tile_size)
    lx0 = torch.floor(mx / ts - r / ts).clamp(0, tile_width); lx1 = torch.ceil(mx / ts + r / ts).clamp(0, tile_width)
    ly0 = torch.floor(my / ts - r / ts).clamp(0, tile_height); ly1 = torch.ceil(my / ts + r / ts).clamp(0, tile_height)
    tau = torch.log(255.0 * o)
    det = ca * cc - cb * cb
    ok = (tau >= 2e-3) & (det > 0)                      # opacity within 0.2 % of 1/255: gsplat's box (rounding of the alpha test)
    s = 2.0 * tau / det
    rel = 1e-4 + 2.4e-7 * ((ca * cc + cb * cb) / det)
    hx = torch.sqrt(s * cc) * (1.0 + rel) + 0.01
    hy = torch.sqrt(s * ca) * (1.0 + rel) + 0.01
    sx0 = torch.ceil((mx - hx - (ts - 0.5)) / ts).clamp(0, tile_width); sx1 = (torch.floor((mx + hx - 0.5) / ts) + 1).clamp(0, tile_width)
    sy0 = torch.ceil((my - hy - (ts - 0.5)) / ts).clamp(0, tile_height); sy1 = (torch.floor((my + hy - 0.5) / ts) + 1).clamp(0, tile_height)
    x0 = torch.where(ok, torch.maximum(lx0, sx0), lx0); y0 = torch.where(ok, torch.maximum(ly0, sy0), ly0)
    x1 = torch.where(ok, torch.maximum(torch.minimum(lx1, sx1), x0), lx1); y1 = torch.where(ok, torch.maximum(torch.minimum(ly1, sy1), y0), ly1)
    dead = ~(tau > 0)                                   # opacity <= 1/255: never composited
    x1 = torch.where(dead, x0, x1); y1 = torch.where(dead, y0, y1)
    vis = r > 0
    z = torch.zeros_like(x0)
    return tuple(torch.where(vis, t, z).long() for t in (x0, y0, x1, y1))
