#!/usr/bin/env python3
"""Host-side cost of one GAN iteration (d_solver + g_solver at level 6, batch 32): cProfile top functions and
the wall time with / without a device sync per step.  python tools/gan_host_profile.py [f32|bf16]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from sequitr_amd.networks.gan import GenerativeAdverserialNetwork

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
g = GenerativeAdverserialNetwork({"num_levels": 7, "batch_size": 32, "repeat_batch": 1, "learning_rate": 1e-3,
                                  "device": "cuda:0", "seed": 0, "dtype": dtype}, mode=None)
g.build()
g.set_level(6)
rng = np.random.default_rng(3)
X = torch.from_numpy(rng.standard_normal((32, 256, 256, 2)).astype(np.float32)).cuda()
Z = torch.from_numpy(rng.standard_normal((32, 1, 1, 512)).astype(np.float32)).cuda()
for _ in range(2):
    g.d_solver(X, Z, 1.0)
    g.g_solver(X, Z, 1.0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    g.d_solver(X, Z, 1.0)
    g.g_solver(X, Z, 1.0)
torch.cuda.synchronize()
print("wall ms/iter", (time.perf_counter() - t0) / 5 * 1e3)
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    g.d_solver(X, Z, 1.0)
    g.g_solver(X, Z, 1.0)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
