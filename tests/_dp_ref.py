"""Torch restatements of the exchange step's HIP kernels for the CPU (gloo) tests — test infrastructure, as the oracle is:
the packed (visible rows only) colour-gradient slab of include/dnsplat.h (dnsplat_visible_index + dnsplat_proj_grads.sh_packed) and the
rebuild / add kernels (dnsplat_sh_grads_from_factors, _add_factors, _from_packed).  tests/test_gpu_parity.py compares the kernels
with the same arithmetic on the GPU."""
import torch


def packed_layout(n: int, capacity: int):
    nb = (n + 63) // 64
    rows0 = (8 + 3 * nb + 3) & ~3
    return nb, rows0, (rows0 + 3 * capacity + 3) & ~3


def pack_slab_ref(cols: torch.Tensor, visible: torch.Tensor, campos: torch.Tensor, capacity: int) -> torch.Tensor:
    """[N,3] colour gradients + bool [N] -> one packed slab (float32 words; integers through a view), as the kernels write it."""
    n = cols.shape[0]
    nb, rows0, total = packed_layout(n, capacity)
    slab = torch.zeros(total, dtype=torch.float32)
    ints = slab.view(torch.int32)
    vis = visible.bool()
    ints[0], ints[4], ints[5] = int(vis.sum()), capacity, n
    slab[1:4] = campos.float()
    bits = torch.zeros(nb * 64, dtype=torch.int64)
    bits[:n] = vis.long()
    words = (bits.view(nb, 64) << torch.arange(64)).sum(1)          # bit b of word w = Gaussian 64 w + b (wraps into the sign bit)
    ints[8:8 + 2 * nb] = words.view(torch.int32)
    counts = bits.view(nb, 64).sum(1)
    ints[8 + 2 * nb:8 + 3 * nb] = (torch.cumsum(counts, 0) - counts).int()
    rows = cols[vis][:capacity].float()
    slab[rows0:rows0 + 3 * rows.shape[0]] = rows.reshape(-1)
    return slab


def unpack_slab_ref(slab: torch.Tensor, n: int, capacity: int):
    """-> ([N,3] colour gradients with zeros for the Gaussians the slab does not hold, camera position [3])."""
    nb, rows0, _ = packed_layout(n, capacity)
    ints = slab.view(torch.int32)
    words = ints[8:8 + 2 * nb].contiguous().view(torch.int64)
    bits = ((words[:, None] >> torch.arange(64)) & 1).reshape(-1)[:n].bool()
    offs = ints[8 + 2 * nb:8 + 3 * nb].long()
    k = torch.cumsum(bits.long(), 0) - bits.long()                   # = offs[block] + popcount below: a global exclusive count
    assert torch.equal(k[::64][:nb], offs), "block offsets are not the exclusive prefix sum of the mask popcounts"
    cols = torch.zeros(n, 3)
    ok = bits & (k < int(ints[4]))
    cols[ok] = slab[rows0:].reshape(-1)[: 3 * int(ints[4])].reshape(-1, 3)[k[ok]]
    return cols, slab[1:4].clone()


def make_rebuild_ref(dense_ref):
    """The stand-in the gloo tests install as ``dp.ShFactorExchange._rebuild``: autograd through the oracle's dense SH evaluation,
    summed over the views; ``skip_view >= 0`` ADDS the other views to rows that hold that view's pre-scaled share."""

    def rebuild_ref(gathered, means_, n, w, deg, K, v_coeffs, v_sh0, v_shN, skip_view=-1, packed_capacity=None, scale=None):
        tot = torch.zeros(n, K, 3)
        for v in range(w):
            if v == skip_view:
                continue
            if packed_capacity is not None:
                cols_v, pos_v = unpack_slab_ref(gathered[v].clone(), n, packed_capacity)
            else:
                cols_v, pos_v = gathered[v, :3 * n].reshape(n, 3).clone(), gathered[v, 3 * n:3 * n + 3].clone()
            co = torch.zeros(n, K, 3, requires_grad=True)
            (dense_ref.sh_colors(deg, torch.nn.functional.normalize(means_ - pos_v, dim=-1), co) * cols_v).sum().backward()
            tot += co.grad
        tot *= (1.0 / w) if scale is None else scale
        if skip_view >= 0:
            v_sh0.add_(tot[:, 0])
            v_shN.add_(tot[:, 1:])
        else:
            v_sh0.copy_(tot[:, 0])
            v_shN.copy_(tot[:, 1:])

    return rebuild_ref
