"""Synthetic inputs for tests and the benchmark: the reference's own random-init distribution.

BASELINE.md §4 / SURVEY.md §8(d): ``means = (rand(N,3)-0.5)*10`` (dn_model.py:135), isotropic
``scales = log(mean 3-NN distance)`` (dn_model.py:186-189,217), ``quats = random_quat_tensor(N)``
(dn_model.py:218,1497-1509), ``opacities = logit(0.1)`` (dn_model.py:156), ``features_dc = rand(N,3)``
(dn_model.py:153), ``features_rest = 0`` (dn_model.py:154) or N(0, 0.1) to exercise SH degree 3.

Scale initialisation: the reference runs sklearn's kd-tree kNN on the CPU (inherited
``k_nearest_sklearn``).  ``scale_init="knn"`` does exactly that; ``scale_init="closed_form"``
(default above 100k points) uses the expectation of the same quantity for a uniform Poisson
process of the same density, E[(d1+d2+d3)/3] = 1.15747 * (3 / (4 pi rho))^(1/3), identical for every
point — documented in DESIGN.md, and what bench.py reports in its config.

Cameras (builder-chosen, the reference hard-codes none): pinhole, principal point at the image
centre, radius-8 orbit around the origin in the x-z plane looking at the origin, +y up,
nerfstudio/OpenGL camera-to-world convention.
"""
from __future__ import annotations

import math
from typing import Dict

import torch
from torch import Tensor

from .model import Camera

_MEAN_3NN_FACTOR = (0.8929795115692493 + 1.1906393487589990 + 1.3890792402188323) / 3.0  # Gamma(k+1/3)/Gamma(k), k=1..3


def random_quat_tensor(N: int, generator=None) -> Tensor:
    """Uniformly distributed unit quaternions from three uniform variates per quaternion (Shoemake's subgroup algorithm) — the
    distribution, and the order in which the variates are drawn and combined, of dn_model.py:1497-1509, so that a seed gives
    the reference's initial rotations."""
    u, v, w = (torch.rand(N, generator=generator) for _ in range(3))
    r1, r2 = torch.sqrt(1.0 - u), torch.sqrt(u)
    a, b = 2.0 * math.pi * v, 2.0 * math.pi * w
    return torch.stack((r1 * torch.sin(a), r1 * torch.cos(a), r2 * torch.sin(b), r2 * torch.cos(b)), dim=-1)


def mean_3nn_distance_closed_form(N: int, extent: float = 10.0) -> float:
    rho = N / extent ** 3
    return _MEAN_3NN_FACTOR * (3.0 / (4.0 * math.pi * rho)) ** (1.0 / 3.0)


def make_gauss_params(N: int, sh_degree: int = 3, seed: int = 0, sh_rest_std: float = 0.0,
                      scale_init: str = "auto", device="cpu") -> Dict[str, Tensor]:
    """The 7-entry ParameterDict of dn_model.py:227-237 (as plain tensors with requires_grad)."""
    g = torch.Generator().manual_seed(seed)
    means = (torch.rand(N, 3, generator=g) - 0.5) * 10
    if scale_init == "auto":
        scale_init = "knn" if N <= 100_000 else "closed_form"
    if scale_init == "knn":
        from sklearn.neighbors import NearestNeighbors

        nn_model = NearestNeighbors(n_neighbors=4, algorithm="auto", metric="euclidean").fit(means.numpy())
        distances, _ = nn_model.kneighbors(means.numpy())
        avg_dist = torch.from_numpy(distances[:, 1:]).float().mean(dim=-1, keepdim=True)
    elif scale_init == "closed_form":
        avg_dist = torch.full((N, 1), mean_3nn_distance_closed_form(N))
    else:
        raise ValueError(scale_init)
    scales = torch.log(avg_dist.repeat(1, 3))
    quats = random_quat_tensor(N, generator=g)
    dim_sh = (sh_degree + 1) ** 2
    features_dc = torch.rand(N, 3, generator=g)
    features_rest = torch.zeros(N, dim_sh - 1, 3)
    if sh_rest_std > 0:
        features_rest = torch.randn(N, dim_sh - 1, 3, generator=g) * sh_rest_std
    opacities = torch.logit(0.1 * torch.ones(N, 1))
    params = {
        "means": means, "scales": scales, "quats": quats, "features_dc": features_dc,
        "features_rest": features_rest, "opacities": opacities,
    }
    out = {k: v.to(device).contiguous().requires_grad_(True) for k, v in params.items()}
    out["normals"] = torch.zeros(N, 3, device=device)
    return out


def orbit_camera(index: int, n_views: int = 8, width: int = 1920, height: int = 1080, focal: float = 1200.0,
                 radius: float = 8.0, device="cpu") -> Camera:
    theta = 2.0 * math.pi * index / n_views
    pos = torch.tensor([radius * math.cos(theta), 0.0, radius * math.sin(theta)])
    forward = -pos / pos.norm()
    up = torch.tensor([0.0, 1.0, 0.0])
    right = torch.linalg.cross(forward, up)
    right = right / right.norm()
    true_up = torch.linalg.cross(right, forward)
    c2w = torch.stack([right, true_up, -forward, pos], dim=1)  # [3,4], OpenGL: camera looks down -z
    return Camera(camera_to_worlds=c2w[None].to(device), fx=focal, fy=focal, cx=width / 2.0, cy=height / 2.0,
                  width=width, height=height)
