"""GPU probe: the discriminator step on one lane vs two (SQ_GAN_TWO_LANES): bit-identity of a step's gradients and the
iteration time at level 6"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from sequitr_amd.networks import gan
dev = "cuda:0"
rng = np.random.default_rng(0)
X = torch.from_numpy(rng.standard_normal((32, 256, 256, 2)).astype(np.float32)).to(dev)
Z = torch.from_numpy(rng.standard_normal((32, 1, 1, 512)).astype(np.float32)).to(dev)
res = {}
for lanes in (False, True):
    g = gan.GenerativeAdverserialNetwork({"num_levels": 7, "batch_size": 32, "repeat_batch": 1, "learning_rate": 1e-3, "device": dev,
                                          "seed": 0, "dtype": "bf16", "graph": True, "two_lanes": lanes}, mode=None)
    g.build(); g.set_level(6)
    for _ in range(4):
        g.iteration(X, Z, 1.0)
    torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(20):
            g.iteration(X, Z, 1.0)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 20 * 1e3)
    res[lanes] = {k: v.detach().clone() for k, v in g.store.vars.items()}
    print("two_lanes", lanes, "ms per iteration", [round(t, 3) for t in ts], "losses", g.last_losses)
diff = max(float((res[True][k] - res[False][k]).abs().max()) for k in res[True])
print("max |weight difference| after 64 iterations:", diff)
