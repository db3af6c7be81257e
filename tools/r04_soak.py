#!/usr/bin/env python3
"""Soak of the shipped fast paths on the GPU box: N captured bf16 training steps at config-3 size (losses finite, falling on
the disk-label data, replicas of the step deterministic) and M level-6 GAN iterations with bf16 storage through the graphed
iteration() (one shared generator pass; losses and every parameter finite; device memory must not grow), then 256 x 8 tiles
through the streamed inference path.  python tools/r04_soak.py [train_steps] [gan_iterations]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from sequitr_amd.train import UNetTrainer  # noqa: E402
from sequitr_amd.networks import gan  # noqa: E402

n_train = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
n_gan = int(sys.argv[2]) if len(sys.argv) > 2 else 400
dev = "cuda:0"
params = {"shape": (512, 512), "num_inputs": 1, "num_outputs": 2, "device": dev, "dtype": "bf16", "dropout": 0.2, "seed": 1}
x, onehot, wmap, _ = bench.disk_image_inputs(dev, seed=3)        # config 3 with tiles that carry their labels
tr = UNetTrainer(params)
tr.capture(x, onehot, wmap)
losses = []
t0 = time.time()
for i in range(n_train):
    tr.step(x, onehot, wmap)
    if i % 50 == 0 or i == n_train - 1:
        losses.append(float(tr.last_loss.item()))
torch.cuda.synchronize()
print("train: %d captured steps in %.1f s, losses every 50 steps: %s" % (n_train, time.time() - t0, " ".join("%.4f" % v for v in losses)))
assert all(np.isfinite(losses)), "training loss went non-finite"
assert losses[-1] < losses[0], "training loss did not fall"
assert all(np.isfinite(v).all() for v in tr.state_dict().values()), "a parameter went non-finite"

g = gan.GenerativeAdverserialNetwork({"num_levels": 7, "batch_size": 32, "repeat_batch": 1, "learning_rate": 1e-3, "device": dev,
                                      "seed": 0, "dtype": "bf16", "graph": True}, mode=None)
g.build()
g.set_level(6)
rng = np.random.default_rng(0)
X = torch.from_numpy(rng.standard_normal((32, 256, 256, 2)).astype(np.float32)).to(dev)
gl, dl = [], []
t0 = time.time()
mem = []
for i in range(n_gan):
    Z = g.build_latent()
    dloss, gloss = g.iteration(X, Z, 1.0)
    if i in (10, n_gan - 1):
        torch.cuda.synchronize()
        mem.append(torch.cuda.memory_allocated())
    if i % 25 == 0 or i == n_gan - 1:
        dl.append(float(dloss.item())); gl.append(float(gloss.item()))
torch.cuda.synchronize()
print("gan: %d iterations in %.1f s\n  d_loss %s\n  g_loss %s" % (n_gan, time.time() - t0, " ".join("%.3g" % v for v in dl), " ".join("%.3g" % v for v in gl)))
assert all(np.isfinite(dl)) and all(np.isfinite(gl)), "a GAN loss went non-finite"
assert all(bool(torch.isfinite(v).all()) for v in g.store.vars.values()), "a GAN parameter went non-finite"
print("gan: device memory allocated after iteration 10 / the last: %.1f / %.1f MiB" % (mem[0] / 2 ** 20, mem[-1] / 2 ** 20))
assert mem[-1] <= mem[0] + (8 << 20), "device memory grows with the iterations"

# the same without graphs for a few iterations (eager tape: reference cycles would show as growth here)
ge = gan.GenerativeAdverserialNetwork({"num_levels": 7, "batch_size": 8, "repeat_batch": 1, "learning_rate": 1e-3, "device": dev,
                                       "seed": 0, "dtype": "bf16", "graph": False}, mode=None)
ge.build()
ge.set_level(6)
Xe = X[:8]
em = []
for i in range(12):
    ge.iteration(Xe, ge.build_latent(), 1.0)
    ge.d_solver(Xe, ge.build_latent(), 1.0)                 # a discriminator step alone: its generator pass is never differentiated
    torch.cuda.synchronize()
    em.append(torch.cuda.memory_allocated())
print("gan eager: memory allocated after iterations 3 / 12: %.1f / %.1f MiB" % (em[2] / 2 ** 20, em[-1] / 2 ** 20))
assert em[-1] <= em[2] + (8 << 20), "eager GAN steps leak device memory"

from sequitr_amd.networks.unet import UNet2D, init_unet_weights
from sequitr_amd.frontend import TileStreamer
p = {"shape": (512, 512), "num_inputs": 1, "num_outputs": 2, "device": dev}
net = UNet2D(p, "infer"); net.load_state_dict(init_unet_weights(p, seed=0))
tiles = np.random.default_rng(5).standard_normal((256, 512, 512, 1)).astype(np.float32)
st = TileStreamer(net, batch=32); st.warm_up((512, 512, 1))
first, _ = st.run(tiles)
t0 = time.time()
for _ in range(8):
    m, _ = st.run(tiles)
dt = time.time() - t0
assert np.array_equal(m, first)
print("stream: 8 x 256 tiles in %.2f s = %.0f Mpix/s, masks identical run to run" % (dt, 8 * 256 * 512 * 512 / dt / 1e6))
print("soak ok")
