"""The data-parallel step with REAL ranks on the GPU: two processes share the one GPU of the test box (gloo carries the
collectives, RCCL refuses two ranks on one device), each renders its own camera with the HIP kernels, gradients land in the
flat ``GradArena`` bucket, the SH gradients travel as 12-byte colour gradients (``ShFactorExchange``: factor kernel,
all-gather, HIP rebuild) and the geometry gradients through the in-place all-reduce.  Rank 0's averaged gradients must equal
the single-process mean over both cameras computed without any of that machinery (the multi-GPU run proper,
RCCL over xGMI, is the driver's).
"""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("means", "scales", "quats", "features_dc", "features_rest", "opacities")
OUT = ("rgb", "depth", "normal", "accumulation")
N, W, H = 20_000, 320, 256

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _render_and_backward(dns, synthetic, gp, view, dev):
    cam = synthetic.orbit_camera(view, n_views=8, width=W, height=H, focal=220.0).to(dev)
    out = dns.DNSplatterRenderer(gp, fused=True).get_outputs(cam)
    gen = torch.Generator(device=dev).manual_seed(100 + view)
    cots = [torch.rand(out[k].shape, device=dev, generator=gen) * 2 - 1 for k in OUT]
    torch.autograd.backward([out[k] for k in OUT], cots)


def _worker(rank, world, port, path):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), DNSPLAT_DIST_BACKEND="gloo", DNSPLAT_SHARE_GPU="1")
    import dn_splatter_amd as dns
    from dn_splatter_amd import dp, synthetic

    r, w, _, dev = dp.init_from_env()
    assert (r, w) == (rank, world) and dev.type == "cuda" and dp.world_size() == world
    gp = synthetic.make_gauss_params(N, sh_rest_std=0.2, seed=5, device=dev)
    arena = dp.GradArena(gp)
    dns.set_grad_arena(arena)
    exchange = dp.ShFactorExchange()
    dns.set_sh_exchange(exchange)
    for it in range(2):                                   # twice: the second step reuses buckets, slabs and capacity guesses
        for k in KEYS:
            gp[k].grad = None
        _render_and_backward(dns, synthetic, gp, rank, dev)
        wire = dp.allreduce_gradients(gp, arena, exchange=exchange)
        assert all(arena.holds(gp[k].grad) for k in KEYS)
    torch.cuda.synchronize()
    eager = {k: gp[k].grad.detach().clone() for k in KEYS}

    # the same step with the compute captured into a HIP graph and the exchange issued eagerly behind each replay
    # (graph.GraphedDpStep — what bench.py --gpus N times): same averaged gradients
    from dn_splatter_amd.graph import GraphedDpStep

    cam = synthetic.orbit_camera(rank, n_views=8, width=W, height=H, focal=220.0).to(dev)
    renderer = dns.DNSplatterRenderer(gp, fused=True)
    gen = torch.Generator(device=dev).manual_seed(100 + rank)
    shapes = {"rgb": (H, W, 3), "depth": (H, W, 1), "normal": (H, W, 3), "accumulation": (H, W, 1)}
    cots = [torch.rand(shapes[k], device=dev, generator=gen) * 2 - 1 for k in OUT]

    def compute():
        out = renderer.get_outputs(cam)
        torch.autograd.backward([out[k] for k in OUT], cots)

    for k in KEYS:
        gp[k].grad = None
    gdp = GraphedDpStep(compute, gp, arena, exchange=exchange)
    worst = 0.0
    for it in range(3):
        gdp()
        torch.cuda.synchronize()
        assert all(arena.holds(gp[k].grad) for k in KEYS) and gdp.wire == wire
        for k in KEYS:
            scale = float(eager[k].abs().max()) + 1e-30
            worst = max(worst, float((gp[k].grad - eager[k]).abs().max()) / scale)
    gdp.check()
    gdp.close()

    # ---- the SLICED exchange (dp.SlicedShExchange): the projection backward as 4 slice launches, slab k all-gathered behind launch
    # k.  Eager form first (the backward launches the slices itself), then recorded under graph.GraphedDpStep — there with a loss
    # term that feeds `scales` directly (a scale regulariser), which the captured step leaves in a tensor of its own and
    # run_recorded adds to the bucket slice before the all-reduce.  Both must give the unsliced step's averaged gradients.
    sliced = dp.SlicedShExchange(4)
    dns.set_sh_exchange(sliced)
    for k in KEYS:
        gp[k].grad = None
    _render_and_backward(dns, synthetic, gp, rank, dev)
    assert len(sliced.bounds) == 4 and sliced.bounds[0] == (0, 5120) and sliced.bounds[-1][1] == N
    wire_s = dp.allreduce_gradients(gp, arena, exchange=sliced)
    torch.cuda.synchronize()
    assert wire_s == wire + (world - 1) * 4 * 3 * 4 and all(arena.holds(gp[k].grad) for k in KEYS)
    worst_s = max(float((gp[k].grad - eager[k]).abs().max()) / (float(eager[k].abs().max()) + 1e-30) for k in KEYS)

    reg_w = 0.37 * (rank + 1)

    def compute_reg():
        out = renderer.get_outputs(cam)
        torch.autograd.backward([out[k] for k in OUT] + [(gp["scales"] * reg_w).sum()], cots + [None])

    for k in KEYS:
        gp[k].grad = None
    renderer.forget()
    gdp = GraphedDpStep(compute_reg, gp, arena, exchange=sliced)
    assert gdp.sliced and list(gdp.direct) == ["scales"]
    worst_r = 0.0
    for it in range(3):
        gdp()
        torch.cuda.synchronize()
        assert all(arena.holds(gp[k].grad) for k in KEYS) and gdp.wire == wire_s
        for k in KEYS:
            want = eager[k] + (0.37 * (1 + world) / 2 if k == "scales" else 0.0)     # the mean over ranks of the direct term
            worst_r = max(worst_r, float((gp[k].grad - want).abs().max()) / (float(want.abs().max()) + 1e-30))
    # compute_only: the launches without the exchange leave this rank's own geometry gradients (+ its direct term) in the bucket
    gdp.compute_only()
    torch.cuda.synchronize()
    gdp.check()
    gdp.close()
    if rank == 0:
        torch.save({"grads": {k: eager[k].cpu() for k in KEYS}, "wire": int(wire), "graph_vs_eager": worst, "sliced_vs_eager": worst_s,
                    "sliced_graph_with_direct_term_vs_eager": worst_r}, path)
    dns.set_grad_arena(None)
    dns.set_sh_exchange(None)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_ranks_on_one_gpu_average_like_one_process(tmp_path):
    import dn_splatter_amd as dns
    from dn_splatter_amd import synthetic
    from _scenes import assert_close

    world = 2
    path = str(tmp_path / "rank0.pt")
    mp.spawn(_worker, args=(world, _free_port(), path), nprocs=world, join=True)
    got = torch.load(path)
    dev = "cuda:0"
    ref = None
    for view in range(world):
        gp = synthetic.make_gauss_params(N, sh_rest_std=0.2, seed=5, device=dev)
        _render_and_backward(dns, synthetic, gp, view, dev)
        g = {k: gp[k].grad.detach().cpu().double() for k in KEYS}
        ref = g if ref is None else {k: ref[k] + g[k] for k in KEYS}
    for k in KEYS:
        # same kernels on both sides; the sums only differ in the order of the atomics and of the mean over the ranks
        assert_close(got["grads"][k].reshape(ref[k].shape), (ref[k] / world).float(), f"two ranks: grad {k}", tol=2e-5)
    # what travelled: 11 geometry floats per Gaussian in the all-reduce + the (3 N + 4)-float slab of every rank
    assert got["wire"] > 0
    # graph replay + eager exchange gave the gradients of the fully eager step (same kernels; the atomics' order differs)
    assert got["graph_vs_eager"] <= 2e-5, got["graph_vs_eager"]
    # the sliced exchange (4 slice launches + 4 slab all-gathers), eager and recorded behind a graph replay with a direct term
    assert got["sliced_vs_eager"] <= 2e-5, got["sliced_vs_eager"]
    assert got["sliced_graph_with_direct_term_vs_eager"] <= 2e-5, got["sliced_graph_with_direct_term_vs_eager"]
