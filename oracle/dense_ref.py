"""Dense pure-torch restatement of the same rendering math, differentiable by autograd.

TEST INFRASTRUCTURE ONLY.  Purpose: pin the hand-derived backward passes of
``dnsplat_oracle.c`` (and hence of the HIP kernels) against torch autograd in
fp64 on tiny problems (SURVEY.md §4 tier T1).  It is written independently of the
C oracle — vectorised over pixels, sequential over depth-sorted Gaussians — so it
also cross-checks the forward.  O(N * H * W) memory: keep N <= ~256, images <= 64x64.

Follows SURVEY.md Appendix A.1-A.6 (restating gsplat==1.0.0 as called from
dn_splatter/dn_model.py:495-516).
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import Tensor


def quat_to_rotmat(q: Tensor) -> Tensor:
    q = q / q.norm(dim=-1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    R = torch.stack(
        [
            1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
            2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
            2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y),
        ],
        dim=-1,
    )
    return R.reshape(q.shape[:-1] + (3, 3))


def project(means, quats, scales, viewmat, K, W, H, eps2d=0.3, near=0.01, far=1e10, radius_clip=0.0):
    """A.2.  Returns dict with radii (int), means2d, depths, conics, compensations, valid mask."""
    Rv, t = viewmat[:3, :3], viewmat[:3, 3]
    mean_c = means @ Rv.T + t
    Rq = quat_to_rotmat(quats)
    M = Rq * scales[:, None, :]
    covar = M @ M.transpose(1, 2)
    covar_c = Rv @ covar @ Rv.T
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    x, y, z = mean_c.unbind(-1)
    lim_x = 1.3 * 0.5 * W / fx
    lim_y = 1.3 * 0.5 * H / fy
    rz = 1.0 / z
    tx = z * torch.minimum(lim_x, torch.maximum(-lim_x, x * rz))
    ty = z * torch.minimum(lim_y, torch.maximum(-lim_y, y * rz))
    zero = torch.zeros_like(z)
    J = torch.stack(
        [
            torch.stack([fx * rz, zero, -fx * tx * rz * rz], -1),
            torch.stack([zero, fy * rz, -fy * ty * rz * rz], -1),
        ],
        dim=-2,
    )
    cov2d = J @ covar_c @ J.transpose(1, 2)
    means2d = torch.stack([fx * x * rz + cx, fy * y * rz + cy], -1)
    det_orig = cov2d[:, 0, 0] * cov2d[:, 1, 1] - cov2d[:, 0, 1] * cov2d[:, 1, 0]
    c00 = cov2d[:, 0, 0] + eps2d
    c11 = cov2d[:, 1, 1] + eps2d
    c01 = 0.5 * (cov2d[:, 0, 1] + cov2d[:, 1, 0])
    det = c00 * c11 - c01 * c01
    comp = torch.sqrt(torch.clamp(det_orig / det, min=0.0))
    conics = torch.stack([c11 / det, -c01 / det, c00 / det], -1)
    with torch.no_grad():
        b = 0.5 * (c00 + c11)
        v1 = b + torch.sqrt(torch.clamp(b * b - det, min=0.01))
        radius = torch.ceil(3.0 * torch.sqrt(v1))
        ok = (z >= near) & (z <= far) & (det > 0) & (radius > radius_clip)
        ok &= ~((means2d[:, 0] + radius <= 0) | (means2d[:, 0] - radius >= W)
                | (means2d[:, 1] + radius <= 0) | (means2d[:, 1] - radius >= H))
        radii = torch.where(ok, radius, torch.zeros_like(radius)).to(torch.int32)
    return dict(radii=radii, means2d=means2d, depths=z, conics=conics, compensations=comp, ok=ok)


def sh_colors(degree: int, dirs: Tensor, coeffs: Tensor) -> Tensor:
    """A.5 (raw SH sum, before +0.5/clamp)."""
    d = dirs / dirs.norm(dim=-1, keepdim=True)
    x, y, z = d.unbind(-1)
    res = 0.2820947917738781 * coeffs[:, 0]
    if degree >= 1:
        res = res + 0.48860251190292 * (-y[:, None] * coeffs[:, 1] + z[:, None] * coeffs[:, 2] - x[:, None] * coeffs[:, 3])
    if degree >= 2:
        z2 = z * z
        fTmp0B = -1.092548430592079 * z
        fC1 = x * x - y * y
        fS1 = 2 * x * y
        res = res + (0.5462742152960395 * fS1)[:, None] * coeffs[:, 4] + (fTmp0B * y)[:, None] * coeffs[:, 5] \
            + (0.9461746957575601 * z2 - 0.3153915652525201)[:, None] * coeffs[:, 6] \
            + (fTmp0B * x)[:, None] * coeffs[:, 7] + (0.5462742152960395 * fC1)[:, None] * coeffs[:, 8]
    if degree >= 3:
        fTmp0C = -2.285228997322329 * z2 + 0.4570457994644658
        fTmp1B = 1.445305721320277 * z
        fC2 = x * fC1 - y * fS1
        fS2 = x * fS1 + y * fC1
        res = res + (-0.5900435899266435 * fS2)[:, None] * coeffs[:, 9] + (fTmp1B * fS1)[:, None] * coeffs[:, 10] \
            + (fTmp0C * y)[:, None] * coeffs[:, 11] \
            + (z * (1.865881662950577 * z2 - 1.119528997770346))[:, None] * coeffs[:, 12] \
            + (fTmp0C * x)[:, None] * coeffs[:, 13] + (fTmp1B * fC1)[:, None] * coeffs[:, 14] \
            + (-0.5900435899266435 * fC2)[:, None] * coeffs[:, 15]
    return res


def tile_membership(means2d, radii, W, H, tile_size):
    """A.3 bbox rule, as a dense [N, H*W] bool mask (pixel belongs to a tile the Gaussian is binned to)."""
    tw, th = math.ceil(W / tile_size), math.ceil(H / tile_size)
    with torch.no_grad():
        r = radii.to(means2d.dtype)
        x0 = torch.clamp(torch.floor((means2d[:, 0] - r) / tile_size), 0, tw)
        x1 = torch.clamp(torch.ceil((means2d[:, 0] + r) / tile_size), 0, tw)
        y0 = torch.clamp(torch.floor((means2d[:, 1] - r) / tile_size), 0, th)
        y1 = torch.clamp(torch.ceil((means2d[:, 1] + r) / tile_size), 0, th)
        jj = torch.arange(W) // tile_size
        ii = torch.arange(H) // tile_size
        mx = (jj[None, :] >= x0[:, None]) & (jj[None, :] < x1[:, None])  # [N,W]
        my = (ii[None, :] >= y0[:, None]) & (ii[None, :] < y1[:, None])  # [N,H]
        m = my[:, :, None] & mx[:, None, :] & (radii > 0)[:, None, None]
    return m.reshape(means2d.shape[0], H * W)


def composite(means2d, conics, colors, opacities, depths, radii, W, H, tile_size=16,
              background: Optional[Tensor] = None):
    """A.6 over all pixels at once; Gaussians visited in (depth bits, index) order."""
    N, D = colors.shape
    member = tile_membership(means2d, radii, W, H, tile_size)
    with torch.no_grad():
        dbits = depths.detach().to(torch.float32).view(torch.int32).to(torch.int64)
        order = torch.argsort(dbits * (N + 1) + torch.arange(N), stable=True)
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    px = (xs.reshape(-1) + 0.5).to(means2d.dtype)
    py = (ys.reshape(-1) + 0.5).to(means2d.dtype)
    P = H * W
    T = torch.ones(P, dtype=means2d.dtype)
    out = torch.zeros(P, D, dtype=means2d.dtype)
    done = torch.zeros(P, dtype=torch.bool)
    for g in order.tolist():
        if radii[g] <= 0:
            continue
        dx = means2d[g, 0] - px
        dy = means2d[g, 1] - py
        a, b, c = conics[g]
        sigma = 0.5 * (a * dx * dx + c * dy * dy) + b * dx * dy
        alpha = torch.clamp(opacities[g] * torch.exp(-sigma), max=0.999)
        with torch.no_grad():
            valid = member[g] & ~done & (sigma >= 0) & (alpha >= 1.0 / 255.0)
            stop = valid & (T * (1 - alpha) <= 1e-4)
            done = done | stop
            apply = valid & ~stop
        w = torch.where(apply, alpha * T, torch.zeros_like(T))
        out = out + w[:, None] * colors[g][None, :]
        T = torch.where(apply, T * (1 - alpha), T)
    alphas = 1 - T
    if background is not None:
        out = out + T[:, None] * background[None, :]
    return out.reshape(H, W, D), alphas.reshape(H, W)


def render(means, quats, scales, opacities, colors, viewmat, K, W, H, sh_degree=None,
           render_mode="RGB+ED", tile_size=16, eps2d=0.3, near=0.01, far=1e10,
           rasterize_mode="classic", background=None):
    """The dn_model.py:495-516 call, dense."""
    pr = project(means, quats, scales, viewmat, K, W, H, eps2d, near, far)
    if rasterize_mode == "antialiased":
        opacities = opacities * pr["compensations"]
    if sh_degree is None:
        cols = colors.reshape(means.shape[0], -1)
    else:
        cam = torch.inverse(viewmat)[:3, 3]
        cols = sh_colors(sh_degree, means - cam[None], colors)
        cols = torch.where((pr["radii"] > 0)[:, None], cols, torch.zeros_like(cols))
        cols = torch.clamp_min(cols + 0.5, 0.0)
    if render_mode in ("RGB+D", "RGB+ED"):
        cols = torch.cat([cols, pr["depths"][:, None]], -1)
    out, alphas = composite(pr["means2d"], pr["conics"], cols, opacities, pr["depths"], pr["radii"], W, H,
                            tile_size, background)
    if render_mode == "RGB+ED":
        out = torch.cat([out[..., :-1], out[..., -1:] / alphas[..., None].clamp(min=1e-10)], -1)
    return out, alphas, pr
