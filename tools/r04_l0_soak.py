#!/usr/bin/env python3
"""Run-to-run identity soak of the round-4 f32 kernels at BASELINE config 2's size: the whole fused network 300 times, every
level-0 form and the transpose convs 100 times each -- every repeat must equal the first bit for bit (the store hazard of
profiles/r04_l0_issue_model.txt section 4 showed as ~1 differing pixel in 3000, a different one each run)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sequitr_amd import ops
from sequitr_amd.networks.unet import UNet2D

D, N, h = "cuda:0", 32, 512
g = torch.Generator(device=D); g.manual_seed(0)
r = lambda *s: torch.randn(*s, device=D, generator=g)
bad = 0
net = UNet2D({"shape": (h, h), "device": D}, "infer").initialize()
x = r(N, h, h, 1)
m0 = net.predict(x).clone(); l0 = net.logits().clone()
for i in range(300):
    m = net.predict(x)
    if not (torch.equal(m, m0) and torch.equal(net.logits(), l0)):
        bad += 1
        print("network repeat", i, "differs", flush=True)
print("network: 300 repeats,", bad, "differing", flush=True)
x1, xx, xl = r(N, h, h, 1), r(N, h, h, 16), r(N, h // 2, h // 2, 32)
w1, b1 = r(3, 3, 1, 16) * 0.3, r(16) * 0.1
w, b = r(3, 3, 16, 16) * 0.08, r(16) * 0.1
w32, b32 = r(3, 3, 16, 32) * 0.08, r(32) * 0.1
wt, bt = r(2, 2, 16, 32) * 0.1, r(16) * 0.1
wh, bh = r(1, 1, 16, 2), r(2) * 0.1
xt = r(N, 128, 128, 64); wtt, btt = r(2, 2, 32, 64) * 0.1, r(32) * 0.1; skt = r(N, 256, 256, 32)
forms = {
    "plain": lambda: (ops.conv2d(xx, w, b, act="relu"),),
    "plain32": lambda: (ops.conv2d(xx, w32, b32, act="relu"),),
    "pool": lambda: ops.conv3x3_pool(xx, w, b),
    "head": lambda: ops.conv3x3_head(xx, w, b, wh, bh, act="relu"),
    "first": lambda: ops.conv3x3_first_block(x1, w1, b1, w, b, want_pool=True),
    "up": lambda: (ops.convT_conv3x3(xl, wt, bt, xx, "eltwise_mul", w, b, act="relu"),),
    "convT": lambda: (ops.convT2x2s2(xt, wtt, btt, skip=skt, bridge="eltwise_mul"),),
}
for name, fn in forms.items():
    ref = [t.clone() for t in fn()]
    nb = 0
    for i in range(100):
        out = fn()
        if not all(torch.equal(a, bb) for a, bb in zip(out, ref)):
            nb += 1
    bad += nb
    print("%-8s 100 repeats, %d differing" % (name, nb), flush=True)
print("total differing:", bad)
sys.exit(1 if bad else 0)
