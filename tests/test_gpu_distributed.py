"""GPU, two processes on ONE card (gloo carries the CUDA tensors; RCCL needs distinct GPUs): the data-parallel
training path end to end on the HIP kernels -- SURVEY 8(d) config 4: rank-averaged gradients equal the
single-process global-batch gradients (<= 1e-5 rel, fp32), and replicas stay identical after a step."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batch(n, size=32, seed=7):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, size, size, 1)).astype(np.float32)
    lab = rng.integers(0, 2, (n, size, size))
    onehot = (lab[..., None] == np.arange(2)).astype(np.uint8)
    w = (1 + rng.random((n, size, size, 1))).astype(np.float32)
    return x, onehot, w


PARAMS = {"shape": (32, 32), "filters": (16, 32), "dropout": 0.0, "device": "cuda:0", "seed": 5}


def _unet_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sequitr_amd.parallel import shard_range
        from sequitr_amd.train import UNetTrainer
        x, onehot, w = _batch(4)
        b, e = shard_range(4, rank, world)
        d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
        t = UNetTrainer(dict(PARAMS), learning_rate=0.01)
        t.forward_backward(d(x[b:e]), d(onehot[b:e]), d(w[b:e]))
        from sequitr_amd.parallel import allreduce_sum_
        n = allreduce_sum_(t.gbucket.flat)
        avg = {k: v / n for k, v in t.grads().items()}
        # a full optimiser step on a second, identical trainer (the step does its own all-reduce)
        t2 = UNetTrainer(dict(PARAMS), learning_rate=0.01)
        t2.step(d(x[b:e]), d(onehot[b:e]), d(w[b:e]))
        ret[rank] = (n, {k: v.copy() for k, v in avg.items()}, t2.state_dict())
    finally:
        dist.destroy_process_group()


def test_unet_rank_averaged_gradients_equal_global_batch_gradients():
    from sequitr_amd.train import UNetTrainer
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_unet_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert sorted(ret.keys()) == [0, 1] and ret[0][0] == 2
    x, onehot, w = _batch(4)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    t = UNetTrainer(dict(PARAMS), learning_rate=0.01)
    t.forward_backward(d(x), d(onehot), d(w))
    ref = t.grads()
    for k, g in ref.items():
        scale = max(float(np.abs(g).max()), 1e-30)
        for r in (0, 1):
            assert float(np.abs(ret[r][1][k] - g).max()) / scale <= 1e-5, (k, r)
    # replicas are identical after the step, and equal the single-process global-batch step
    w0, w1 = ret[0][2], ret[1][2]
    assert all(np.array_equal(w0[k], w1[k]) for k in w0)
    one = UNetTrainer(dict(PARAMS), learning_rate=0.01)
    one.step(d(x), d(onehot), d(w))
    ws = one.state_dict()
    for k in ws:                                               # Adam's first step is +-lr per weight: only sign ties may differ
        assert float(np.mean(np.abs(ws[k] - w0[k]) > 1e-6)) <= 0.01, k


def _gan_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sequitr_amd.networks import gan
        g = gan.GenerativeAdverserialNetwork({"num_levels": 3, "batch_size": 4, "repeat_batch": 1, "learning_rate": 1e-3,
                                              "device": "cuda:0", "seed": 3, "graph": True, "dtype": "bf16"}, mode=None)
        g.build()
        g.set_level(1)
        rng = np.random.default_rng(100 + rank)                # every rank its own minibatch
        for _ in range(3):                                     # eager warm step, capture, replay
            z = torch.from_numpy(rng.standard_normal((4, 1, 1, 512)).astype(np.float32)).cuda()
            x = torch.from_numpy(rng.standard_normal((4, 8, 8, 2)).astype(np.float32)).cuda()
            g.d_solver(x, z, 1.0)
            g.g_solver(x, z, 1.0)
        ret[rank] = g.store.state_dict()
    finally:
        dist.destroy_process_group()


def test_gan_replicas_stay_identical_with_graph_replay_and_allreduce():
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_gan_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    w0, w1 = ret[0], ret[1]
    assert all(np.array_equal(w0[k], w1[k]) for k in w0) and all(np.isfinite(v).all() for v in w0.values())
