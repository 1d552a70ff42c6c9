"""Multi-view data parallelism on CPU: world_size 2 over gloo (SURVEY.md §4 tier T5, §8e).

Each rank renders its own camera of the same Gaussians (host mirror of get_outputs with the oracle plugged in for
the two gsplat calls — tests may use the oracle), the six gradient tensors are averaged with
``dp.allreduce_gradients`` and must equal the single-process mean over both cameras.  Also covers the flat
``GradArena`` bucket and the rank helpers.  The N > 1 GPU path differs only in the backend ("nccl" = RCCL).
"""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("means", "scales", "quats", "features_dc", "features_rest", "opacities")
N, W, H = 600, 64, 48


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _camera_grads(view):
    import dn_splatter_amd as dns
    from dn_splatter_amd import synthetic
    from oracle import oracle as orc

    gp = synthetic.make_gauss_params(N, sh_rest_std=0.2, seed=11)
    g = torch.Generator().manual_seed(12)
    gp["scales"] = (gp["scales"].detach() + torch.randn(N, 3, generator=g) * 0.4).requires_grad_(True)
    params = {k: v.detach().clone().requires_grad_(k != "normals") for k, v in gp.items()}
    cam = synthetic.orbit_camera(view, n_views=8, width=W, height=H, focal=40.0)
    m = dns.DNSplatterRenderer(params, fused=False, rasterization_fn=orc.rasterization,
                               rasterize_gaussians_fn=orc.rasterize_gaussians)
    out = m.get_outputs(cam)
    gen = torch.Generator().manual_seed(100 + view)
    cots = [torch.rand(out[k].shape, generator=gen) * 2 - 1 for k in ("rgb", "depth", "normal", "accumulation")]
    torch.autograd.backward([out[k] for k in ("rgb", "depth", "normal", "accumulation")], cots)
    return params


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from dn_splatter_amd import dp

    r, w, local, dev = dp.init_from_env("cpu")
    assert (r, w) == (rank, world) and dev.type == "cpu" and dp.world_size() == world
    params = _camera_grads(rank)
    # path 1: grads are ordinary tensors -> packed, reduced, copied back
    wire = dp.allreduce_gradients(params)
    assert wire == sum(params[k].numel() for k in KEYS) * 4
    # path 2: grads living in a GradArena -> one in-place collective
    arena = dp.GradArena(params)
    for k in KEYS:
        arena.view(k).copy_(torch.full_like(params[k], float(rank + 1)))
        params[k].grad2 = arena.view(k)
    assert arena.holds(arena.view("quats")) and not arena.holds(torch.zeros(3))
    fake = {k: torch.nn.Parameter(params[k].detach().clone()) for k in KEYS}
    arena2 = dp.GradArena(fake)
    for k in KEYS:
        fake[k].grad = arena2.take(fake[k])
        fake[k].grad.fill_(float(rank + 1))
    assert dp.allreduce_gradients(fake, arena2) == arena2.bytes()
    assert all(torch.allclose(fake[k].grad, torch.full_like(fake[k], (1 + world) / 2)) for k in KEYS)
    assert dp.max_over_ranks(float(rank), dev) == world - 1
    assert dp.sum_over_ranks([1.0, float(rank)], dev) == [float(world), float(sum(range(world)))]
    # densification statistics: sums over the step's increments, max over ranks (densify.py)
    from dn_splatter_amd.densify import DensifyStats
    prev = DensifyStats(8, "cpu")
    prev.xys_grad_norm += 5.0           # common history
    cur = prev.clone()
    cur.xys_grad_norm[rank] += 1.0 + rank
    cur.vis_counts[rank] += 1
    cur.max_2Dsize[rank] = 0.1 * (rank + 1)
    cur.allreduce(prev)
    assert cur.xys_grad_norm.tolist()[:2] == [6.0, 7.0] and cur.xys_grad_norm[2] == 5.0
    assert cur.vis_counts.tolist()[:3] == [2.0, 2.0, 1.0]
    assert abs(float(cur.max_2Dsize[1]) - 0.2) < 1e-7 and abs(float(cur.max_2Dsize[0]) - 0.1) < 1e-7
    dp.barrier()
    if rank == 0:
        ret.update({k: params[k].grad.clone() for k in KEYS})
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_equals_single_process_mean():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
            assert p.exitcode == 0, f"rank exited with {p.exitcode}"
        got = {k: ret[k] for k in KEYS}
    sys.path.insert(0, ROOT)
    ref = [_camera_grads(v) for v in range(world)]
    for k in KEYS:
        mean = sum(r[k].grad for r in ref) / world
        scale = mean.abs().max().item() + 1e-30
        err = (got[k] - mean).abs().max().item()
        assert err <= 1e-5 * scale + 1e-7, f"{k}: {err} vs scale {scale}"     # fp32 reduction order only


def test_single_process_helpers_are_no_ops():
    sys.path.insert(0, ROOT)
    from dn_splatter_amd import dp

    assert dp.world_size() == 1
    p = {k: torch.nn.Parameter(torch.ones(4, 3)) for k in KEYS}
    for v in p.values():
        v.grad = torch.ones_like(v)
    assert dp.allreduce_gradients(p) == 0
    assert dp.max_over_ranks(3.5, torch.device("cpu")) == 3.5
    dp.barrier()
