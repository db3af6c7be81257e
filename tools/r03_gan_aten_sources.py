"""debug: which framework (ATen) kernels one eager D + G step of the level-6 GAN launches, and from where"""
import sys, collections
import numpy as np
import torch
sys.path.insert(0, '.')
from sequitr_amd.networks import gan
from torch.profiler import profile, ProfilerActivity

g = gan.GenerativeAdverserialNetwork({"num_levels": 7, "batch_size": 8, "repeat_batch": 1, "learning_rate": 1e-3,
                                      "device": "cuda:0", "seed": 0, "dtype": "bf16"}, mode=None)
g.build()
g.set_level(6)
rng = np.random.default_rng(0)
z = torch.from_numpy(rng.standard_normal((8, 1, 1, 512)).astype(np.float32)).cuda()
x = torch.from_numpy(rng.standard_normal((8, 256, 256, 2)).astype(np.float32)).cuda()
it = getattr(g, '_iteration', None)
(it(x, z, 1.0) if it else (g.d_solver(x, z, 1.0), g.g_solver(x, z, 1.0)))
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    (it(x, z, 1.0) if it else (g.d_solver(x, z, 1.0), g.g_solver(x, z, 1.0)))
    torch.cuda.synchronize()
ev = prof.events()
cnt = collections.Counter()
LEAF = ("aten::fill_", "aten::zero_", "aten::copy_", "aten::add", "aten::add_", "aten::cat", "aten::sum", "aten::mul", "aten::neg",
        "aten::uniform_", "aten::_to_copy")
for e in ev:
    if e.name in LEAF:
        st = [f for f in (e.stack or []) if "sequitr_amd" in f]
        where = st[0] if st else "<autograd engine>"
        cnt[(e.name, str(getattr(e, "input_shapes", ""))[:60], where[-80:])] += 1
for (name, shp, where), n in sorted(cnt.items(), key=lambda kv: -kv[1])[:70]:
    print("%3d %-14s %-60s %s" % (n, name, shp, where))
