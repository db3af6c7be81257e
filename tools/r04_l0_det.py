#!/usr/bin/env python3
"""Level-0 f32 kernel family at BASELINE config 2's size (32 x 512^2): every form run three times and against the
generic kernel (SQ_CONV_L0=0), with the location of any mismatch (tile, row, column, channel)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sequitr_amd import ops

D, N, h = "cuda:0", int(os.environ.get("N", 32)), 512
g = torch.Generator(device=D); g.manual_seed(0)
r = lambda *s: torch.randn(*s, device=D, generator=g)
x1, x, xl = r(N, h, h, 1), r(N, h, h, 16), r(N, h // 2, h // 2, 32)
w1, b1 = r(3, 3, 1, 16) * 0.3, r(16) * 0.1
w, b = r(3, 3, 16, 16) * 0.08, r(16) * 0.1
wt, bt = r(2, 2, 16, 32) * 0.1, r(16) * 0.1
wh, bh = r(1, 1, 16, 2), r(2) * 0.1
forms = {
    "plain": lambda: (ops.conv2d(x, w, b, act="relu"),),
    "pool": lambda: ops.conv3x3_pool(x, w, b),
    "head": lambda: ops.conv3x3_head(x, w, b, wh, bh, act="relu"),
    "first": lambda: ops.conv3x3_first_block(x1, w1, b1, w, b, want_pool=True),
    "up": lambda: (ops.convT_conv3x3(xl, wt, bt, x, "eltwise_mul", w, b, act="relu"),),
}


def where(a, bb):
    d = (a != bb).nonzero()
    k = d[:, 0].unique().tolist()
    return "%d mismatches; images %s; first %s; rows %s cols %s" % (
        d.shape[0], k[:8], d[0].tolist(), sorted(set((d[:200, 1] % 16).tolist())), sorted(set((d[:200, 2] % 16).tolist())))


for name, fn in forms.items():
    os.environ["SQ_CONV_L0"] = "0"
    ref = [t.clone() for t in fn()]
    os.environ["SQ_CONV_L0"] = "1"
    for rep in range(3):
        out = fn()
        torch.cuda.synchronize()
        for i, (a, bb) in enumerate(zip(out, ref)):
            print("%-6s run %d output %d: %s" % (name, rep, i, "identical" if torch.equal(a, bb) else where(a, bb)), flush=True)
