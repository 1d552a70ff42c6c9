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
    # SH factor exchange: all-gather of the colour gradients (+ each rank's camera position) + rebuild == mean of the outer
    # products basis(normalize(mean - camera)) (x) colour gradient.
    # The HIP rebuild kernel is swapped for a torch reference here (autograd through the oracle's dense SH evaluation).
    from oracle import dense_ref

    def rebuild_ref(gathered, means_, n, w, deg, K, v_coeffs, v_sh0, v_shN):
        tot = torch.zeros(n, K, 3)
        for v in range(w):
            co = torch.zeros(n, K, 3, requires_grad=True)
            # copies: the sliced exchange keeps all slices' receive buffers in ONE tensor, and an all-gather of another slice that
            # is still in flight bumps the version counter autograd checks on views of it
            cols_v, pos_v = gathered[v, :3 * n].reshape(n, 3).clone(), gathered[v, 3 * n:3 * n + 3].clone()
            dirs_v = torch.nn.functional.normalize(means_ - pos_v, dim=-1)
            (dense_ref.sh_colors(deg, dirs_v, co) * cols_v).sum().backward()
            tot += co.grad
        tot /= w
        v_sh0.copy_(tot[:, 0])
        v_shN.copy_(tot[:, 1:])

    ex = dp.ShFactorExchange()
    ex._rebuild = rebuild_ref
    n_g = 40
    gen = torch.Generator().manual_seed(7 + rank)
    fpar = {k: torch.nn.Parameter(torch.zeros(s)) for k, s in (("means", (n_g, 3)), ("scales", (n_g, 3)), ("quats", (n_g, 4)),
                                                              ("opacities", (n_g, 1)), ("features_dc", (n_g, 3)),
                                                              ("features_rest", (n_g, 15, 3)))}
    far = dp.GradArena(fpar)
    for k in KEYS:
        fpar[k].grad = far.take(fpar[k])
        fpar[k].grad.fill_(float(rank))
    means_g = torch.randn(n_g, 3, generator=torch.Generator().manual_seed(99)) * 2          # replicated on every rank
    campos = torch.randn(3, generator=gen) * 6                                              # this rank's camera
    dirs = torch.nn.functional.normalize(means_g - campos, dim=-1)
    cols = torch.randn(n_g, 3, generator=gen)
    slab = torch.cat([cols.reshape(-1), campos, torch.zeros(1)])
    mine = ex.begin(n_g, torch.device("cpu"), 3, 16, means=means_g)
    mine.copy_(slab)
    got_bytes = dp.allreduce_gradients(fpar, far, exchange=ex)
    assert got_bytes == 11 * n_g * 4 + (world - 1) * (3 * n_g + 4) * 4
    assert torch.allclose(fpar["quats"].grad, torch.full((n_g, 4), (world - 1) / 2))      # geometry part: plain mean
    # every rank must hold the same rebuilt SH gradient = mean over ranks of basis (x) colour
    chk = fpar["features_rest"].grad.clone()
    dist.all_reduce(chk, op=dist.ReduceOp.SUM)
    assert torch.allclose(chk / world, fpar["features_rest"].grad, atol=1e-6)
    own = torch.zeros(n_g, 16, 3, requires_grad=True)
    (dense_ref.sh_colors(3, dirs, own) * cols).sum().backward()
    allc = own.grad.clone()
    dist.all_reduce(allc, op=dist.ReduceOp.SUM)
    assert torch.allclose(allc[:, 1:] / world, fpar["features_rest"].grad, atol=1e-5)
    assert torch.allclose(allc[:, 0] / world, fpar["features_dc"].grad, atol=1e-5)
    # the same through the early launch: the all-gather is started before the geometry gradients exist (what the projection
    # backward does right after dnsplat_sh_factors) and completed inside allreduce_gradients
    want_dc, want_rest = fpar["features_dc"].grad.clone(), fpar["features_rest"].grad.clone()
    class _LaunchInBackward(torch.autograd.Function):
        """The product starts the all-gather from inside the projection backward, i.e. on autograd's worker thread."""

        @staticmethod
        def forward(ctx, x):
            return x.clone()

        @staticmethod
        def backward(ctx, g):
            mine = ex.begin(n_g, torch.device("cpu"), 3, 16, means=means_g)
            mine.copy_(slab)
            ex.launch()
            return g

    _LaunchInBackward.apply(torch.zeros(1, requires_grad=True)).sum().backward()
    assert ex.work is not None
    for k in dp.GEOMETRY_KEYS:
        fpar[k].grad.fill_(float(rank))
    fpar["features_dc"].grad.zero_(); fpar["features_rest"].grad.zero_()
    assert dp.allreduce_gradients(fpar, far, exchange=ex) == got_bytes and ex.work is None
    assert torch.allclose(fpar["means"].grad, torch.full((n_g, 3), (world - 1) / 2))
    assert torch.equal(fpar["features_dc"].grad, want_dc) and torch.equal(fpar["features_rest"].grad, want_rest)

    # the deferred form graph.GraphedDpStep relies on: the backward runs (here: is replayed) WITHOUT any collective — launch() is a
    # no-op, nothing is in flight — and the whole exchange follows it; begin()'s bookkeeping is restored by hand as a replay runs
    # no Python.  Same gradients as the two forms above.
    ex.deferred = True
    _LaunchInBackward.apply(torch.zeros(1, requires_grad=True)).sum().backward()
    assert ex.work is None and ex.meta is not None
    meta = ex.meta
    ex.drop()                                   # "warm-up frame": factors produced, never rebuilt
    _LaunchInBackward.apply(torch.zeros(1, requires_grad=True)).sum().backward()      # begin() must not complain
    for rep in range(2):                        # two "replays"
        ex.meta = meta
        for k in dp.GEOMETRY_KEYS:
            fpar[k].grad.fill_(float(rank))
        fpar["features_dc"].grad.zero_(); fpar["features_rest"].grad.zero_()
        assert dp.allreduce_gradients(fpar, far, exchange=ex) == got_bytes and ex.meta is None
        assert torch.allclose(fpar["scales"].grad, torch.full((n_g, 3), (world - 1) / 2))
        assert torch.equal(fpar["features_dc"].grad, want_dc) and torch.equal(fpar["features_rest"].grad, want_rest)
    ex.deferred = False

    # ---- the SLICED exchange (dp.SlicedShExchange): K slices of Gaussians, one mini slab + one all-gather +
    # one rebuild per slice, one all-reduce of the geometry prefix — must give what the unsliced exchange gives on the same data.
    n2 = 1300
    fp2 = {k: torch.nn.Parameter(torch.zeros(s)) for k, s in (("means", (n2, 3)), ("scales", (n2, 3)), ("quats", (n2, 4)),
                                                             ("opacities", (n2, 1)), ("features_dc", (n2, 3)),
                                                             ("features_rest", (n2, 15, 3)))}
    far2 = dp.GradArena(fp2)
    for k in KEYS:
        fp2[k].grad = far2.take(fp2[k])
    means2 = torch.randn(n2, 3, generator=torch.Generator().manual_seed(98)) * 2
    gen2 = torch.Generator().manual_seed(17 + rank)
    cols2, geo2 = torch.randn(n2, 3, generator=gen2), torch.randn(11 * n2, generator=gen2)
    n_geo2 = 11 * n2

    def fill_geometry():
        far2.flat[:n_geo2].copy_(geo2)
        fp2["features_dc"].grad.zero_(); fp2["features_rest"].grad.zero_()

    ex1 = dp.ShFactorExchange()
    ex1._rebuild = rebuild_ref
    fill_geometry()
    ex1.begin(n2, torch.device("cpu"), 3, 16, means=means2).copy_(torch.cat([cols2.reshape(-1), campos, torch.zeros(1)]))
    bytes1 = dp.allreduce_gradients(fp2, far2, exchange=ex1)
    want = far2.flat.clone()
    exs = dp.SlicedShExchange(4)
    exs._rebuild = rebuild_ref
    assert exs.plan(n2) == [(0, 512), (512, 1024), (1024, 1300)] and exs.plan(100) == [(0, 100)] and exs.plan(0) == [(0, 0)]
    assert exs.plan(5_000_000)[-1][1] == 5_000_000 and all(a % 256 == 0 for a, _ in exs.plan(5_000_000)) and len(exs.plan(5_000_000)) == 4

    def fill_slab(k):          # what dnsplat_project_bwd on slice k leaves behind: its colour gradients, the camera position, a pad word
        g0, g1 = exs.bounds[k]
        exs.slab(k).copy_(torch.cat([cols2[g0:g1].reshape(-1), campos, torch.zeros(1)]))

    # (a) eager form: the backward launched the K slices itself, allreduce_gradients runs the collectives
    fill_geometry()
    slabs = exs.begin(n2, torch.device("cpu"), 3, 16, means=means2)
    assert len(slabs) == 3 and [t.numel() for t in slabs] == [3 * 512 + 4, 3 * 512 + 4, 3 * 276 + 4]
    for k in range(3):
        fill_slab(k)
    bytes_s = dp.allreduce_gradients(fp2, far2, exchange=exs)
    assert bytes_s == n_geo2 * 4 + (world - 1) * (3 * n2 + 4 * 3) * 4 and bytes1 == n_geo2 * 4 + (world - 1) * (3 * n2 + 4) * 4
    assert torch.allclose(far2.flat, want, atol=1e-6), float((far2.flat - want).abs().max())
    # (b) recorded form (graph.GraphedDpStep): "replays" leave the inputs of the K launches behind; run_recorded issues launch k +
    # all-gather k, adds the gradient a loss term fed to a geometry tensor directly, reduces the geometry prefix, rebuilds
    exs.begin(n2, torch.device("cpu"), 3, 16, means=means2)
    meta = exs.meta
    exs.drop()
    launched = []

    def launch_stand_in(args):
        (k,) = args
        g0, g1 = exs.bounds[k]
        for name in dp.GEOMETRY_KEYS:          # the slice's rows of the four geometry gradients (a launch OVERWRITES them)
            off, n, shape = far2.slices[name]
            wd = n // n2
            far2.flat[off + g0 * wd:off + g1 * wd].copy_(geo2[off + g0 * wd:off + g1 * wd])
        fill_slab(k)
        launched.append(k)

    exs._launch = launch_stand_in
    exs.record([(k,) for k in range(3)], [])
    direct = {"scales": torch.full((n2, 3), 0.25 * (rank + 1))}
    for rep in range(2):
        far2.flat.fill_(123.0)                  # whatever the previous step left
        exs.meta = meta
        assert exs.run_recorded(fp2, far2, direct=direct) == bytes_s and exs.meta is None
        got2 = far2.flat.clone()
        off, n, _shape = far2.slices["scales"]
        got2[off:off + n] -= 0.25 * (1 + world) / 2       # the mean of the direct term
        assert torch.allclose(got2, want, atol=1e-6), float((got2 - want).abs().max())
    assert launched == [0, 1, 2, 0, 1, 2]
    far2.flat.fill_(5.0)
    exs.meta = meta
    assert exs.run_recorded(fp2, far2, collectives=False) == 0 and float(far2.flat[n_geo2:].min()) == 5.0     # launches only
    exs.meta = None

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
