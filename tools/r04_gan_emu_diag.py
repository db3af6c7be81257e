"""GPU diagnostic: where does the HIP bf16 GAN forward leave the rounding-point emulation (oracle/gan_bf16_ref.py)?
Records the output of every weighted_conv2d / pixel_norm-free stage in call order on both sides, level 6, batch 4."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from sequitr_amd.networks import gan
from sequitr_amd import ops
from oracle import gan_bf16_ref as emu

level, nb, levels = int(sys.argv[1]) if len(sys.argv) > 1 else 6, 4, 7
g = gan.GenerativeAdverserialNetwork({"num_levels": levels, "batch_size": nb, "device": "cuda:0", "seed": 3, "dtype": "bf16"}, mode=None)
g.build()
g.set_level(level)
rng = np.random.default_rng(2)
z = rng.standard_normal((nb, 1, 1, 512)).astype(np.float32)
x = rng.standard_normal((nb,) + g.get_size(level) + (2,)).astype(np.float32)
dev = lambda a: torch.from_numpy(a).cuda()

hip, names = [], []
orig = gan.weighted_conv2d
def rec(**kw):
    y = orig(**kw)
    hip.append(y.detach().float().cpu().double().numpy())
    names.append("%s %s->%d %s" % (kw.get('name'), tuple(kw['inputs'].shape), kw['filters'], 'pool' if kw.get('pool') else ''))
    return y
gan.weighted_conv2d = rec
em = []
orig_w = emu.wconv
def rec_e(x_, W, name, sx, so, act=True, norm=True):
    y = orig_w(x_, W, name, sx, so, act, norm)
    em.append(y.detach().numpy())
    return y
emu.wconv = rec_e
orig_pool = emu.Pool.apply

f = g.filters[:level + 1]
with torch.no_grad(), g.precision():
    outs, last = g.generator(dev(z), f, levels=(-2, -1))
    n_g = len(hip)
    layers, logits = g.discriminator(dev(x), f[::-1])
W = emu.to_torch(g.store.state_dict(), requires_grad=False)
t64 = lambda a: torch.as_tensor(a, dtype=torch.float64)
with torch.no_grad():
    imgs = emu.generator(t64(z), W, f)
    elog = emu.discriminator(t64(x), W, f[::-1])
def rel(a, b):
    return np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-30)
print("generator: %d stages hip, discriminator %d; emulation %d" % (n_g, len(hip) - n_g, len(em)))
for i, (a, n) in enumerate(zip(hip, names)):
    b = em[i]
    if a.shape != b.shape:                                     # hip records the POOLED tensor of a fused conv + pool
        bb = torch.as_tensor(b)
        b = emu.Pool.apply(bb, 0.25, 'b').numpy()
    print("%2d %-60s rel diff %.3e   exact-equal fraction %.4f" % (i, n, rel(a, b), float((a == b).mean())))
print("logits hip", logits.cpu().numpy(), "\nlogits emu", elog.numpy())
