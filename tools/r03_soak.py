#!/usr/bin/env python3
"""Soak of the shipped fast paths on the GPU box: N captured bf16 training steps at config-3 size (losses finite, falling on
the disk-label data, replicas of the step deterministic) and M level-6 GAN iterations with bf16 storage through the graphed
solvers (losses and every parameter finite).  python tools/r03_soak.py [train_steps] [gan_iterations]"""
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
for i in range(n_gan):
    Z = g.build_latent()
    dloss = g.d_solver(X, Z, 1.0)
    gloss = g.g_solver(X, Z, 1.0)
    if i % 25 == 0 or i == n_gan - 1:
        dl.append(float(dloss.item())); gl.append(float(gloss.item()))
torch.cuda.synchronize()
print("gan: %d iterations in %.1f s\n  d_loss %s\n  g_loss %s" % (n_gan, time.time() - t0, " ".join("%.3g" % v for v in dl), " ".join("%.3g" % v for v in gl)))
assert all(np.isfinite(dl)) and all(np.isfinite(gl)), "a GAN loss went non-finite"
assert all(bool(torch.isfinite(v).all()) for v in g.store.vars.values()), "a GAN parameter went non-finite"
print("soak ok")
