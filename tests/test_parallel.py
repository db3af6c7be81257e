"""CPU (gloo, world_size 2): the data-parallel plumbing -- flat bucket views, tile sharding and
the single all-reduce -- gives rank-averaged gradients equal to the single-process
global-batch gradients."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sequitr_amd.parallel import FlatBucket, allreduce_sum_, shard_range, epoch_schedule


def test_flat_bucket_views_share_storage_and_stay_aligned():
    b = FlatBucket([("a/kernel", (3, 3, 1, 16)), ("a/bias", (16,)), ("h/kernel", (1, 1, 16, 2)), ("h/bias", (2,))], "cpu")
    assert b.numel % 4 == 0 and b.numel >= 144 + 16 + 32 + 2
    b.view("h/bias").fill_(7.0)
    o, n = b.offsets["h/bias"]
    assert o % 4 == 0 and torch.all(b.flat[o:o + n] == 7.0) and b.flat.sum() == 14.0
    assert set(b.views()) == {"a/kernel", "a/bias", "h/kernel", "h/bias"} and b.view("a/kernel").shape == (3, 3, 1, 16)


def test_shard_range_partitions_tiles():
    for n, world in ((32, 8), (33, 8), (5, 8), (128, 3)):
        r = [shard_range(n, k, world) for k in range(world)]
        assert r[0][0] == 0 and r[-1][1] == n
        assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
        sizes = [e - b for b, e in r]
        assert max(sizes) - min(sizes) <= 1


def test_epoch_schedule_same_step_count_on_every_rank_full_batches_only():
    """63 tiles, world 2, batch 16 used to give rank 0 two steps and rank 1 one (a dead-locked all-reduce)."""
    for n, batch, world in ((63, 16, 2), (64, 16, 2), (65, 16, 3), (129, 16, 8)):
        order, steps = epoch_schedule(n, batch, world)
        assert steps == (n // world) // batch
        seen = [order(0, r) for r in range(world)]
        assert all(len(o) == steps * batch for o in seen)                    # every rank: same count, full batches
        allidx = np.concatenate(seen)
        assert len(set(allidx.tolist())) == len(allidx) and allidx.max() < n  # disjoint shards of one permutation
        assert not np.array_equal(order(0, 0), order(1, 0))                   # reshuffled every epoch
    with pytest.raises(ValueError):
        epoch_schedule(31, 16, 2)                                             # no rank may step with a short batch
    # a single rank with a small stack trains on one short batch (no collective, gradient scale 1) ...
    order, steps = epoch_schedule(5, 16, 1)
    assert steps == 1 and order.batch == 5 and order.dropped == 0 and sorted(order(0, 0).tolist()) == list(range(5))
    # ... and the tiles an epoch leaves out are reported for the caller to log
    order, steps = epoch_schedule(63, 16, 2)
    assert order.batch == 16 and order.dropped == 63 - 2 * 16
    with pytest.raises(ValueError):
        epoch_schedule(0, 16, 1)


def _sched_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        order, steps = epoch_schedule(63, 16, world)                          # odd tile count
        n = 0
        for epoch in range(2):
            idx = order(epoch, rank)
            for s in range(steps):
                t = torch.tensor([float(idx[s * 16:(s + 1) * 16].sum())])
                allreduce_sum_(t)                                             # one collective per step, as trainer.step
                n += 1
        ret[rank] = n
    finally:
        dist.destroy_process_group()


def test_odd_tile_count_does_not_desynchronise_the_allreduce():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_sched_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: 2, 1: 2}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model_grads(w, x, y):
    """tiny conv net on torch CPU: returns loss grad w.r.t. w with MEAN-over-batch loss."""
    w = w.clone().requires_grad_(True)
    out = torch.nn.functional.conv2d(x, w, padding=1)
    loss = ((out - y) ** 2).mean()
    loss.backward()
    return w.grad


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)
        w = torch.randn(4, 1, 3, 3, generator=g, dtype=torch.float64)
        x = torch.randn(8, 1, 16, 16, generator=g, dtype=torch.float64)
        y = torch.randn(8, 4, 16, 16, generator=g, dtype=torch.float64)
        b, e = shard_range(8, rank, world)
        bucket = FlatBucket([("w", (4, 1, 3, 3))], "cpu", dtype=torch.float64)
        bucket.view("w").copy_(_model_grads(w, x[b:e], y[b:e]))
        n = allreduce_sum_(bucket.flat)
        assert n == world
        avg = bucket.view("w") / n
        ref = _model_grads(w, x, y)                      # equal shard sizes: mean of means = global mean
        ret[rank] = float((avg - ref).abs().max() / ref.abs().max())
    finally:
        dist.destroy_process_group()


def test_allreduce_equals_global_batch_gradient():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert len(ret) == world and all(v <= 1e-12 for v in ret.values()), dict(ret)


def test_allreduce_is_noop_without_process_group():
    t = torch.ones(8)
    assert allreduce_sum_(t) == 1 and torch.all(t == 1)


def _gan_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from types import SimpleNamespace
        from sequitr_amd.networks.gan import GenerativeAdverserialNetwork as GAN
        g = torch.Generator().manual_seed(rank)
        grads = [torch.randn(3, 3, 4, 8, generator=g), None, torch.randn(8, generator=g), torch.randn(1, 1, 8, 2, generator=g)]
        mine = [None if t is None else t.clone() for t in grads]
        scale = GAN._allreduce(SimpleNamespace(group=None), grads)          # the solver's ONE collective per step
        others = []
        for r in range(world):
            gr = torch.Generator().manual_seed(r)
            others.append([torch.randn(3, 3, 4, 8, generator=gr), None, torch.randn(8, generator=gr), torch.randn(1, 1, 8, 2, generator=gr)])
        ok = scale == 1.0 / world and grads[1] is None
        for i in (0, 2, 3):
            want = sum(o[i] for o in others)
            ok = ok and torch.allclose(grads[i], want, rtol=0, atol=1e-6) and grads[i].shape == mine[i].shape
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_gan_solver_allreduce_sums_every_live_gradient_in_one_collective():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_gan_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}
