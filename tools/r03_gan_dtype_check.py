"""debug: every tape entry of sequitr_amd.functional must hand autograd gradients of its inputs' dtypes (the engine casts a
mismatch silently with a framework kernel).  Runs one eager D + G step of the bf16-storage GAN with checks patched in."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from sequitr_amd import functional as F
from sequitr_amd.networks import gan

bad = []
for name, cls in list(vars(F).items()):
    if isinstance(cls, type) and issubclass(cls, torch.autograd.Function) and cls is not torch.autograd.Function:
        def patch(cls, name):
            fwd, bwd = cls.forward, cls.backward

            def forward(ctx, *a):
                ctx._in = [(t.dtype, tuple(t.shape)) if isinstance(t, torch.Tensor) else None for t in a]
                return fwd(ctx, *a)

            def backward(ctx, *g):
                out = bwd(ctx, *g)
                outs = out if isinstance(out, tuple) else (out,)
                for i, (o, meta) in enumerate(zip(outs, ctx._in)):
                    if isinstance(o, torch.Tensor) and meta is not None and o.dtype != meta[0]:
                        bad.append((name, i, str(o.dtype), str(meta[0]), meta[1]))
                return out
            cls.forward, cls.backward = staticmethod(forward), staticmethod(backward)
        patch(cls, name)

g = gan.GenerativeAdverserialNetwork({"num_levels": 3, "batch_size": 4, "repeat_batch": 1, "learning_rate": 1e-3,
                                      "device": "cuda:0", "seed": 3, "dtype": "bf16"}, mode=None)
g.build()
g.set_level(2)
rng = np.random.default_rng(0)
z = torch.from_numpy(rng.standard_normal((4, 1, 1, 512)).astype(np.float32)).cuda()
x = torch.from_numpy(rng.standard_normal((4, 16, 16, 2)).astype(np.float32)).cuda()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    g.d_solver(x, z, 0.5)
    g.g_solver(x, z, 0.5)
    torch.cuda.synchronize()
print("dtype mismatches:", bad)
for e in prof.key_averages():
    if e.key.startswith("aten::") and ("to" in e.key or "copy" in e.key or "cast" in e.key):
        print(e.key, e.count)
# stacks of aten::to / _to_copy calls
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    g.d_solver(x, z, 0.5)
    g.g_solver(x, z, 0.5)
seen = set()
for ev in prof.events():
    if ev.name in ("aten::_to_copy",) and ev.stack:
        key = tuple(ev.stack[:6])
        if key not in seen:
            seen.add(key)
            print("---- aten::_to_copy", ev.input_shapes if hasattr(ev, "input_shapes") else "")
            for fr in ev.stack[:8]:
                print("   ", fr)
