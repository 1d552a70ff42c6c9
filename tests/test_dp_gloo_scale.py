"""Data parallelism beyond two ranks, on CPU over gloo: world sizes 4 and 8 — the node size BASELINE's C4 / C5
configs name — for the flat gradient bucket, the SH factor exchange and the data-parallel refinement step (N3).

  * bucket / exchange offsets: every rank fills its gradient bucket with a rank- and element-dependent pattern; after
    ``dp.allreduce_gradients(..., exchange=...)`` the 44 B geometry prefix must hold the mean over ranks element by element
    and the SH rows the mean of the ranks' outer products (all-gathered 12-byte colour gradients, rebuilt per rank);
  * refinement: ranks accumulate DIFFERENT per-camera densification statistics, combine them (``DensifyStats.allreduce``) and
    run ``densify.refinement_after``: all ranks must end with bit-identical Gaussian sets and Adam moments, equal to a single
    process that saw all cameras.  (The two HIP kernels are swapped for their torch restatements, as the 2-rank test does for
    the SH rebuild kernel: the GPU tests compare the kernels with those restatements.)
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
SHAPES = {"means": 3, "scales": 3, "quats": 4, "opacities": 1, "features_dc": 3}
N_G = 37          # odd on purpose: no slice of the bucket is aligned to anything


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _scene(n=3000):
    from dn_splatter_amd import synthetic

    gp = synthetic.make_gauss_params(n, sh_rest_std=0.1, seed=1)
    gp = {k: v.detach() for k, v in gp.items()}
    g = torch.Generator().manual_seed(2)
    gp["scales"] = gp["scales"] + torch.randn(n, 3, generator=g) * 1.5 - 3.5
    gp["opacities"] = gp["opacities"] + torch.randn(n, 1, generator=g) * 2
    gp["normals"] = torch.randn(n, 3, generator=g)
    adam = {k: {"exp_avg": torch.randn(v.shape, generator=g), "exp_avg_sq": torch.rand(v.shape, generator=g)}
            for k, v in gp.items() if k != "normals"}
    return gp, adam


def _camera_increment(n, view):
    """What one camera's backward adds to the densification statistics (stand-in numbers, different per camera)."""
    g = torch.Generator().manual_seed(50 + view)
    vis = torch.rand(n, generator=g) < 0.6
    grad = torch.rand(n, generator=g) * 0.01 * vis
    size = torch.rand(n, generator=g) * 0.1 * vis
    return grad, vis.float(), size


def _refine(gp, adam, stats, step=3500):
    from dn_splatter_amd import densify
    from oracle import densify_ref as ref

    return densify.refinement_after(gp, stats, densify.RefineConfig(), step, 100, (480, 640), adam_state=adam, seed=9,
                                    classify_fn=ref.classify_torch, split_fn=ref.split_children_torch)


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from dn_splatter_amd import dp
    from dn_splatter_amd.densify import DensifyStats
    from oracle import dense_ref

    r, w, _local, dev = dp.init_from_env("cpu")
    assert (r, w) == (rank, world)

    # ---- bucket + factor exchange at this world size ------------------------------------------------------------------
    def rebuild_ref(gathered, means_, n, w_, deg, K, v_coeffs, v_sh0, v_shN):
        tot = torch.zeros(n, K, 3)
        for v in range(w_):
            co = torch.zeros(n, K, 3, requires_grad=True)
            cols_v, pos_v = gathered[v, :3 * n].reshape(n, 3), gathered[v, 3 * n:3 * n + 3]
            (dense_ref.sh_colors(deg, torch.nn.functional.normalize(means_ - pos_v, dim=-1), co) * cols_v).sum().backward()
            tot += co.grad
        tot /= w_
        v_sh0.copy_(tot[:, 0])
        v_shN.copy_(tot[:, 1:])

    ex = dp.ShFactorExchange()
    ex._rebuild = rebuild_ref
    fpar = {k: torch.nn.Parameter(torch.zeros((N_G, SHAPES[k]) if k in SHAPES else (N_G, 15, 3))) for k in KEYS}
    arena = dp.GradArena(fpar)
    pattern = {}
    for i, k in enumerate(dp.GRAD_KEYS):
        fpar[k].grad = arena.take(fpar[k])
        assert arena.holds(fpar[k].grad)
        pattern[k] = torch.arange(fpar[k].numel(), dtype=torch.float32).reshape(fpar[k].shape) * 0.001 + 10 * i
        fpar[k].grad.copy_(pattern[k] + rank)                     # element- and rank-dependent
    gen = torch.Generator().manual_seed(7 + rank)
    means_g = torch.randn(N_G, 3, generator=torch.Generator().manual_seed(99)) * 2      # replicated on every rank
    campos = torch.randn(3, generator=gen) * 6                                          # this rank's camera
    dirs = torch.nn.functional.normalize(means_g - campos, dim=-1)
    cols = torch.randn(N_G, 3, generator=gen)
    ex.begin(N_G, torch.device("cpu"), 3, 16, means=means_g).copy_(torch.cat([cols.reshape(-1), campos, torch.zeros(1)]))
    got = dp.allreduce_gradients(fpar, arena, exchange=ex)
    assert got == 11 * N_G * 4 + (world - 1) * (3 * N_G + 4) * 4, got
    mean_rank = (world - 1) / 2
    for k in dp.GEOMETRY_KEYS:                                     # the contiguous 44-byte prefix: element-wise mean
        assert torch.allclose(fpar[k].grad, pattern[k] + mean_rank, atol=1e-5), k
    own = torch.zeros(N_G, 16, 3, requires_grad=True)
    (dense_ref.sh_colors(3, dirs, own) * cols).sum().backward()
    allc = own.grad.clone()
    dist.all_reduce(allc, op=dist.ReduceOp.SUM)
    assert torch.allclose(allc[:, 1:] / world, fpar["features_rest"].grad, atol=1e-5)
    assert torch.allclose(allc[:, 0] / world, fpar["features_dc"].grad, atol=1e-5)
    # dense fallback: one all-reduce of the whole bucket
    for k in dp.GRAD_KEYS:
        fpar[k].grad.copy_(pattern[k] + rank)
    assert dp.allreduce_gradients(fpar, arena) == arena.bytes() == 59 * N_G * 4
    for k in dp.GRAD_KEYS:
        assert torch.allclose(fpar[k].grad, pattern[k] + mean_rank, atol=1e-5), k

    # ---- data-parallel refinement: identical Gaussian sets on every rank ---------------------------------------------
    gp, adam = _scene()
    n = gp["means"].shape[0]
    stats = DensifyStats(n, "cpu")
    stats.xys_grad_norm += 0.002                                   # history shared by all ranks (earlier, already combined steps)
    prev = stats.clone()
    grad, vis, size = _camera_increment(n, rank)                   # this rank's camera
    stats.xys_grad_norm += grad
    stats.vis_counts += vis
    stats.max_2Dsize = torch.maximum(stats.max_2Dsize, size)
    stats.allreduce(prev)
    new, new_adam, report = _refine(gp, adam, stats)
    assert report["n_split"] > 100 and report["n_dup"] > 10 and report["n_culled"] > 100, report
    sig = torch.cat([torch.tensor([float(new["means"].shape[0])])] + [new[k].double().reshape(-1)[::7].float() for k in sorted(new)]
                    + [new_adam[k]["exp_avg"].reshape(-1)[::11] for k in sorted(new_adam)])
    ref_sig = sig.clone()
    dist.broadcast(ref_sig, src=0)
    assert torch.equal(sig, ref_sig), "replicas diverged in the refinement step"
    if rank == 0:
        ret["n_after"] = int(new["means"].shape[0])
        ret["means"] = new["means"].clone()
        ret["scales"] = new["scales"].clone()
        ret["exp_avg_quats"] = new_adam["quats"]["exp_avg"].clone()
    dp.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
def test_bucket_exchange_and_refinement_at_node_scale(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(600)
            assert p.exitcode == 0, f"a rank exited with {p.exitcode}"
        got = dict(ret)
    # single process that saw all cameras
    sys.path.insert(0, ROOT)
    from dn_splatter_amd.densify import DensifyStats

    gp, adam = _scene()
    n = gp["means"].shape[0]
    stats = DensifyStats(n, "cpu")
    stats.xys_grad_norm += 0.002
    for v in range(world):
        grad, vis, size = _camera_increment(n, v)
        stats.xys_grad_norm += grad
        stats.vis_counts += vis
        stats.max_2Dsize = torch.maximum(stats.max_2Dsize, size)
    new, new_adam, report = _refine(gp, adam, stats)
    assert got["n_after"] == new["means"].shape[0]
    # the rank-summed statistics differ from the sequential sum only by fp32 order; decisions are threshold tests on them
    assert torch.allclose(got["means"], new["means"], atol=1e-6) and torch.allclose(got["scales"], new["scales"], atol=1e-6)
    assert torch.equal(got["exp_avg_quats"], new_adam["quats"]["exp_avg"])
