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


def _gan_worker(rank, world, port, ret, shared=False):
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
            if shared:
                g.iteration(x, z, 1.0)                         # one generator pass, four graphs, an all-reduce per solver between them
            else:
                g.d_solver(x, z, 1.0)
                g.g_solver(x, z, 1.0)
        ret[rank] = g.store.state_dict()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("shared", [False, True])
def test_gan_replicas_stay_identical_with_graph_replay_and_allreduce(shared):
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_gan_worker, args=(world, _free_port(), ret, shared), nprocs=world, join=True)
    w0, w1 = ret[0], ret[1]
    assert all(np.array_equal(w0[k], w1[k]) for k in w0) and all(np.isfinite(v).all() for v in w0.values())


def test_gan_iteration_equals_the_two_solver_calls_under_data_parallelism():
    """the shared-generator iteration and the two solver calls: the same replicas after three all-reduced iterations"""
    world, out = 2, []
    for shared in (False, True):
        ret = mp.Manager().dict()
        mp.spawn(_gan_worker, args=(world, _free_port(), ret, shared), nprocs=world, join=True)
        out.append(dict(ret[0]))
    assert all(np.array_equal(out[0][k], out[1][k]) for k in out[0])


CFG4 = {"shape": (512, 512), "dropout": 0.0, "device": "cuda:0", "seed": 0, "dtype": "bf16"}
CFG4_LR = 1e-3


def _config4_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        from sequitr_amd.train import UNetTrainer
        d = torch.device("cuda:0")
        mbs = [bench.config3_inputs(d, seed=2 + 2 * rank + k, nb=16) for k in range(2)]
        t = UNetTrainer(dict(CFG4), learning_rate=CFG4_LR, warmup_steps=0)
        salt0 = int(t.drop_salt.item())
        t.capture(*mbs[0], warmup=1)                       # step 1: the eager warm-up step (all-reduce), then the capture
        l0 = float(t.last_loss.item())
        l1 = float(t.step(*mbs[1]).item())                 # step 2: replay, gradient all-reduce BETWEEN the two graphs
        l2 = float(t.step_accumulate(mbs).item())          # step 3: two replayed micro-batches, ONE all-reduce, ONE Adam
        torch.cuda.synchronize()
        ret[rank] = (l0, l1, l2, t.state_dict(), salt0, int(t.drop_salt.item()), t.step_count,
                     int(t.step_state[0].item()))
    finally:
        dist.destroy_process_group()


def test_config4_per_rank_workload_captured_step_allreduce_between_graphs_and_accumulation():
    """BASELINE configs[3] at its PER-RANK size: two ranks (one card, gloo), each with config 3's
    16 x 512x512 bf16 batch (seeds 2 + 2 rank + k): an eager data-parallel step, the captured step replayed with the
    all-reduce between the (forward/backward) and (Adam) graphs, and one step_accumulate over two micro-batches (a
    global batch of 64).  Replicas must be bit-identical after the three steps; the rank-averaged loss of every step
    must equal the single-process step_accumulate over the same micro-batches (exactly for step 1 -- the same passes
    from the same weights --, within 2 % afterwards: Adam's first step is +-lr per weight and the summation order
    of all-reduce vs. accumulation flips the sign of gradients that are zero to rounding)."""
    import bench
    from sequitr_amd.train import UNetTrainer
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_config4_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert sorted(ret.keys()) == [0, 1]
    r0, r1 = ret[0], ret[1]
    assert all(np.array_equal(r0[3][k], r1[3][k]) for k in r0[3]), "replicas differ after the captured DP steps"
    assert all(np.isfinite(v).all() for v in r0[3].values())
    for r in (r0, r1):
        assert r[5] - r[4] == 1 + 1 + 2 and r[6] == 3 == r[7]                  # 4 passes, 3 optimiser steps
    assert r0[4] != r1[4]                                                       # rank-dependent dropout salt
    d = torch.device("cuda:0")
    mb = [bench.config3_inputs(d, seed=2 + k, nb=16) for k in range(4)]         # rank 0: 0, 1; rank 1: 2, 3
    one = UNetTrainer(dict(CFG4), learning_rate=CFG4_LR, warmup_steps=0)
    L0 = float(one.step_accumulate([mb[0], mb[2]]).item())
    L1 = float(one.step_accumulate([mb[1], mb[3]]).item())
    L2 = float(one.step_accumulate(mb).item())
    m0, m1, m2 = [(r0[i] + r1[i]) / 2 for i in range(3)]
    assert abs(m0 - L0) <= 1e-6 * abs(L0), (m0, L0)
    assert abs(m1 - L1) <= 2e-2 * abs(L1), (m1, L1)
    assert abs(m2 - L2) <= 2e-2 * abs(L2), (m2, L2)
    ws = one.state_dict()
    for k in ws:
        assert float(np.mean(np.abs(ws[k] - r0[3][k]) > 3.5 * CFG4_LR)) <= 0.01, k


def _train_job_worker(rank, world, port, job_params, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", SQ_DIST_BACKEND="gloo")
    from sequitr_amd import core, jobs
    core.TensorflowConfiguration.MODELDIR = job_params["_modeldir"]
    p = {k: v for k, v in job_params.items() if not k.startswith("_")}
    p["output"] = job_params["_out"]
    try:
        ret[rank] = jobs.SERVER_train(p, {"gpu": 0})
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_config4_through_the_job_entry_point(tmp_path):
    """configs[3] through SERVER_train itself (what worker.py:195-215 calls): two ranks, 64 tiles of 512x512 bf16,
    batch 16 per rank, dropout 0.4 -- the job captures the step on every rank, gathers batches on the device, all-reduces
    between the graphs; both ranks take the same number of steps, finish with bit-identical parameters (the job's own
    replica check) and rank 0 alone saves the model."""
    import bench
    os.mkdir(str(tmp_path / "models")), os.mkdir(str(tmp_path / "out"))
    d = torch.device("cuda:0")
    parts = [bench.disk_image_inputs(d, seed=20 + k, nb=16) for k in range(4)]
    np.save(str(tmp_path / "im.npy"), np.concatenate([p[0].cpu().numpy() for p in parts])[..., 0])
    np.save(str(tmp_path / "lab.npy"), np.concatenate([p[3] for p in parts]).astype(np.uint8))
    del parts
    params = {"images": str(tmp_path / "im.npy"), "labels": str(tmp_path / "lab.npy"), "shape": (512, 512),
              "num_outputs": 2, "learning_rate": 0.001, "num_epochs": 3, "batch_size": 16, "dropout": 0.4, "seed": 0,
              "dtype": "bf16", "_modeldir": str(tmp_path / "models"), "_out": str(tmp_path / "out")}
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_train_job_worker, args=(world, _free_port(), params, ret), nprocs=world, join=True)
    a, b = ret[0], ret[1]
    assert a["steps"] == b["steps"] == 6 and a["world"] == 2 and a["graph"] and a["resident"]
    assert a["replicas_identical"] and b["replicas_identical"] and a["param_checksum"] == b["param_checksum"]
    assert "model_dir" in a and "model_dir" not in b and os.path.exists(os.path.join(a["model_dir"], "weights.npz"))
    assert np.isfinite(a["last_loss"]) and a["ms_per_step"] > 0
