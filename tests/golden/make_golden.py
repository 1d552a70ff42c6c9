"""Generates tests/golden/*.npz from the CPU oracle (builder-authored vectors).

PARITY UNPINNED: the reference ships no golden vectors for this path and its renderer (gsplat==1.0.0)
cannot be imported or built in this container, so these vectors freeze the ORACLE's outputs — they
protect against regressions of the oracle and of the HIP path, they are not reference-derived.

    python tests/golden/make_golden.py          # rewrites c1_small.npz, dn_outputs_small.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from _scenes import cotangents, gsplat_inputs, keep_mask, to_leaf, zero_borderline  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def rasterization_case(path, N=3000, W=128, H=96, focal=90, seed=17, cot_seed=23):
    inp, viewmat, K, _ = gsplat_inputs(N, W, H, focal=float(focal), seed=seed, anisotropic=True)
    ci = to_leaf(inp, "cpu")
    r, a, info = orc.rasterization(**ci, viewmats=viewmat, Ks=K, width=W, height=H, packed=False, sh_degree=3,
                                   render_mode="RGB+ED", absgrad=True)
    v_r, v_a = cotangents([r.shape, a.shape], cot_seed)
    keep = keep_mask(info["borderline"], "golden c1_small")     # borderline pixels carry no cotangent (tests/_scenes.py)
    v_r, v_a = zero_borderline(v_r, keep), zero_borderline(v_a[..., 0], keep)[..., None]
    info["means2d"].retain_grad()
    ((r * v_r).sum() + (a * v_a).sum()).backward()
    out = dict(N=N, W=W, H=H, focal=focal, seed=seed, cot_seed=cot_seed,
               render=r.detach().numpy(), alpha=a.detach().numpy(), borderline=info["borderline"].numpy(),
               # what the flagged decisions can move at each pixel (oracle/oracle_impl.inc): the GPU test holds borderline pixels to it
               flip_weight=info["flip_weight"].numpy(), channel_absmax=info["channel_absmax"].numpy(),
               radii=info["radii"].numpy(), tiles_per_gauss=info["tiles_per_gauss"].numpy(),
               flatten_ids=info["flatten_ids"].numpy(), isect_offsets=info["isect_offsets"].numpy(),
               isect_ids=info["isect_ids"].numpy(),
               means2d=info["means2d"].detach().numpy(), depths=info["depths"].detach().numpy(),
               conics=info["conics"].detach().numpy(),
               means2d_grad=info["means2d"].grad.numpy(), means2d_absgrad=info["means2d"].absgrad.numpy())
    for k in ci:
        out["grad_" + k] = ci[k].grad.numpy()
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KiB", "n_isects", info["flatten_ids"].shape[0])


def get_outputs_case(path, N=2000, W=96, H=64, focal=60.0, seed=29):
    """The six-key dict of DNSplatterModel.get_outputs (dn_model.py:605-612) through the host mirror with the
    oracle plugged in for the two gsplat calls."""
    import dn_splatter_amd as dns
    from dn_splatter_amd import synthetic

    gp = synthetic.make_gauss_params(N, sh_rest_std=0.2, seed=seed)
    cam = synthetic.orbit_camera(5, width=W, height=H, focal=focal)
    params = {k: v.detach().clone().requires_grad_(k != "normals") for k, v in gp.items()}
    m = dns.DNSplatterRenderer(params, fused=False, rasterization_fn=orc.rasterization,
                               rasterize_gaussians_fn=orc.rasterize_gaussians)
    out = m.get_outputs(cam)
    border = m.last_info["borderline"] | orc.last_borderline
    keep = keep_mask(border, "golden dn_outputs_small")
    gen = torch.Generator().manual_seed(31)
    loss = 0
    for k in ("rgb", "depth", "normal", "accumulation"):
        loss = loss + (out[k] * zero_borderline(torch.rand(out[k].shape, generator=gen) * 2 - 1, keep)).sum()
    loss.backward()
    save = dict(N=N, W=W, H=H, focal=focal, seed=seed, borderline=border.numpy())
    for k, v in out.items():
        save["out_" + k] = v.detach().numpy()
    for k in ("means", "scales", "quats", "features_dc", "features_rest", "opacities"):
        save["grad_" + k] = params[k].grad.numpy()
    save["normals_world"] = params["normals"].detach().numpy()
    np.savez_compressed(path, **save)
    print(path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    rasterization_case(os.path.join(HERE, "c1_small.npz"))
    get_outputs_case(os.path.join(HERE, "dn_outputs_small.npz"))
