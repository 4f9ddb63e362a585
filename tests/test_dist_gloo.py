"""The N>1 path on CPU: world_size-2 gloo processes exercise the sharding and the one optional
exchange step of the path (raw knots, 640 B per image).  The pixel kernels themselves are not run here."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, n_images, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from curl_amd import shard
        g = torch.Generator().manual_seed(0)
        all_knots = torch.randn(n_images, 160, generator=g)  # what one shared encoder would produce
        # (1) shared encoder on rank 0, broadcast of the raw knots
        buf = all_knots.clone() if rank == 0 else torch.zeros(n_images, 160)
        shard.broadcast_knots(buf, src=0)
        ok_bcast = torch.equal(buf, all_knots)
        # (2) every rank encodes its image shard, all-gather (ragged shards)
        a, b = shard.image_shard(n_images, rank, world)
        gathered = shard.allgather_knots(all_knots[a:b].clone())
        ok_gather = torch.equal(gathered, all_knots)
        # (3) bench-style timing reduction: MAX over ranks
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        # (4) split-pixels: row slabs tile the image and a pointwise layer commutes with the split
        img = torch.rand(n_images, 3, 10, 6, generator=g)

        def fake_layer(x, m, L, R, H):  # pointwise stand-in for the HIP layer
            return x * L[:, :1].reshape(-1, 1, 1, 1) + 1.0, L.sum(1)
        out, reg, (r0, r1) = shard.apply_row_slab(fake_layer, img, None, buf, buf, buf, rank, world)
        full, _ = fake_layer(img, None, buf, buf, buf)
        # the result comes back as rows [r0, r1) of a FULL-size tensor (what the in-place HIP entry point writes)
        ok_slab = out.shape == img.shape and torch.equal(out[:, :, r0:r1], full[:, :, r0:r1])
        # ... so that the slabs of all ranks assemble the whole output with one sum (zero elsewhere)
        mine = torch.zeros_like(img)
        mine[:, :, r0:r1] = out[:, :, r0:r1]
        dist.all_reduce(mine)
        ok_slab = ok_slab and torch.equal(mine, full)
        rows = torch.tensor([r1 - r0])
        dist.all_reduce(rows)
        ret[rank] = (ok_bcast, ok_gather, float(t), ok_slab, int(rows))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_images", [8, 5])
def test_world_size_2_gloo(n_images):
    world = 2
    port = 29500 + (os.getpid() % 2000) + n_images
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, n_images, ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        ok_bcast, ok_gather, tmax, ok_slab, rows = ret[r]
        assert ok_bcast and ok_gather and ok_slab and tmax == 2.0 and rows == 10


def test_apply_row_slab_keeps_gradients_and_the_regulariser():
    """the slab route must stay differentiable through `layer` (the in-place HIP slab entry has no autograd
    node and is taken only for a CURLLayer with nothing requiring grad), any callable must actually be called, and a
    rank with an EMPTY slab (more ranks than rows) still returns the knots-only regulariser every other rank returns."""
    from curl_amd import shard
    torch.manual_seed(0)
    img = torch.rand(2, 3, 3, 5)
    L = torch.randn(2, 48, requires_grad=True)
    calls = []

    def layer(x, m, L, R, H):
        calls.append(tuple(x.shape))
        return x * L[:, :1].reshape(-1, 1, 1, 1), (L ** 2).sum(1)
    out, reg, (r0, r1) = shard.apply_row_slab(layer, img, None, L, L, L, rank=1, world=3)
    assert (r0, r1) == (1, 2) and calls == [(2, 3, 1, 5)]
    (out[:, :, r0:r1].sum() + reg.sum()).backward()
    want = img[:, :, 1:2].sum((1, 2, 3))
    assert torch.allclose(L.grad[:, 0], want + 2 * L.detach()[:, 0]) and torch.allclose(L.grad[:, 1:], 2 * L.detach()[:, 1:])
    # five ranks, three rows: ranks 3 and 4 own no row but owe the same regulariser
    _, reg_full, _ = shard.apply_row_slab(layer, img, None, L, L, L, rank=0, world=5)
    _, reg_empty, (a, b) = shard.apply_row_slab(layer, img, None, L, L, L, rank=4, world=5)
    assert a == b and torch.equal(reg_empty, reg_full)
