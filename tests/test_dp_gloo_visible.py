"""The exchange step with own-camera rows and slabs of visible rows only (reference gradient set
dn_splatter/dn_model.py:388-402), on CPU over gloo at world sizes 2 and 8: every rank writes its OWN camera's SH coefficient rows
(pre-scaled by 1 / world, what dnsplat_project_bwd does with dnsplat_proj_grads.sh_grad_scale), all-gathers a PACKED slab — mask, block
offsets and the colour gradients of its visible Gaussians only — and adds the other cameras' shares; the result must equal the dense
mean over the cameras' outer products on every rank, and the geometry prefix the plain mean.  The HIP kernels are swapped for the
torch restatements of tests/_dp_ref.py (the GPU suite compares the kernels with the same arithmetic)."""
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
N_G = 203          # not a multiple of 64: the last mask word is partial


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, mode):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    import _dp_ref
    from dn_splatter_amd import dp
    from oracle import dense_ref

    r, w, _local, _dev = dp.init_from_env("cpu")
    assert (r, w) == (rank, world)
    own_rows, packed = mode in ("own", "own+packed"), mode in ("packed", "own+packed")
    gen = torch.Generator().manual_seed(7 + rank)
    means_g = torch.randn(N_G, 3, generator=torch.Generator().manual_seed(99)) * 2      # replicated on every rank
    campos = torch.randn(3, generator=gen) * 6                                          # this rank's camera
    visible = torch.rand(N_G, generator=gen) < 0.7                                      # ... and what it sees
    cols = torch.randn(N_G, 3, generator=gen) * visible[:, None]                        # culled Gaussians: zero colour gradient
    dirs = torch.nn.functional.normalize(means_g - campos, dim=-1)
    # all ranks agree on the capacity: the largest visible count, rounded up (dp.ShFactorExchange.calibrate)
    radii = visible.int()
    ex = dp.ShFactorExchange(own_rows=own_rows, packed=packed)
    ex._rebuild = _dp_ref.make_rebuild_ref(dense_ref)
    if packed:
        cap = ex.calibrate(radii, slack=1.0)
        assert cap >= int(visible.sum()) and cap <= N_G
        counts = torch.tensor([float(visible.sum())])
        dist.all_reduce(counts, op=dist.ReduceOp.MAX)
        assert cap == min(N_G, (int(counts.item()) + 1023) // 1024 * 1024)
        ex.capacity = cap = int(counts.item())           # tight for the test: the slab then is smaller than a dense one
    fpar = {k: torch.nn.Parameter(torch.zeros((N_G, SHAPES[k]) if k in SHAPES else (N_G, 15, 3))) for k in KEYS}
    arena = dp.GradArena(fpar)
    for i, k in enumerate(dp.GRAD_KEYS):
        fpar[k].grad = arena.take(fpar[k])
        fpar[k].grad.fill_(float(rank + i))
    own = torch.zeros(N_G, 16, 3, requires_grad=True)
    (dense_ref.sh_colors(3, dirs, own) * cols).sum().backward()
    if own_rows:
        # what dnsplat_project_bwd leaves with sh_grad_scale = 1 / world: this camera's rows, pre-scaled
        assert ex.use_own_rows()
        fpar["features_dc"].grad.copy_(own.grad[:, 0] / world)
        fpar["features_rest"].grad.copy_(own.grad[:, 1:] / world)
    mine = ex.begin(N_G, torch.device("cpu"), 3, 16, means=means_g)
    if packed:
        mine.copy_(_dp_ref.pack_slab_ref(cols, visible, campos, cap))
        assert mine.numel() < 3 * N_G + 4 or world == 1
    else:
        mine.copy_(torch.cat([cols.reshape(-1), campos, torch.zeros(1)]))
    got = dp.allreduce_gradients(fpar, arena, exchange=ex)
    assert got == 11 * N_G * 4 + (world - 1) * ex.slab_floats(N_G) * 4, got
    for i, k in enumerate(dp.GEOMETRY_KEYS):
        assert torch.allclose(fpar[k].grad, torch.full_like(fpar[k], i + (world - 1) / 2)), k
    allc = own.grad.clone()
    dist.all_reduce(allc, op=dist.ReduceOp.SUM)
    assert torch.allclose(allc[:, 1:] / world, fpar["features_rest"].grad, atol=1e-5), mode
    assert torch.allclose(allc[:, 0] / world, fpar["features_dc"].grad, atol=1e-5), mode
    assert not ex.overflowed()
    dp.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,mode", [(2, "own+packed"), (2, "own"), (2, "packed"), (8, "own+packed"), (8, "packed")])
def test_own_rows_and_packed_slabs_equal_the_dense_mean(world, mode):
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0, f"a rank exited with {p.exitcode}"


def test_packed_slab_reference_round_trip():
    """pack -> unpack of the slab layout of include/dnsplat.h (mask words, exclusive block offsets, rows of the visible ones)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _dp_ref

    g = torch.Generator().manual_seed(3)
    for n in (1, 63, 64, 65, 1000):
        vis = torch.rand(n, generator=g) < 0.6
        cols = torch.randn(n, 3, generator=g) * vis[:, None]
        pos = torch.randn(3, generator=g)
        cap = int(vis.sum())
        slab = _dp_ref.pack_slab_ref(cols, vis, pos, cap)
        back, pos_b = _dp_ref.unpack_slab_ref(slab, n, cap)
        assert torch.equal(back, cols) and torch.equal(pos_b, pos)
        assert slab.numel() == _dp_ref.packed_layout(n, cap)[2]
