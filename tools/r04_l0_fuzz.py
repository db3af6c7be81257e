#!/usr/bin/env python3
"""Fuzz of the level-0 f32 kernel family and the persistent transpose conv against the generic kernels (in-process A/B
switches SQ_CONV_L0 / SQ_CONVT_V2; the generic kernels are the ones pinned bit-exact against the C oracle): random
shapes with one / two / many tiles per image row, more tiles than blocks, every form and bridge."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from sequitr_amd import ops

D = "cuda:0"
rng = np.random.default_rng(int(os.environ.get("SEED", 0)))
g = torch.Generator(device=D); g.manual_seed(1)
r = lambda *s: torch.randn(*s, device=D, generator=g)
bad = 0


def ab(name, fn, var):
    global bad
    os.environ[var] = "0"
    ref = [t.clone() if t is not None else None for t in fn()]
    os.environ[var] = "1"
    out = fn()
    for i, (a, b) in enumerate(zip(out, ref)):
        if a is None and b is None:
            continue
        if not torch.equal(a, b):
            bad += 1
            print("MISMATCH", name, "output", i, int((a != b).sum()), flush=True)


shapes = [(1, 16, 16), (1, 16, 32), (2, 32, 16), (1, 48, 48), (3, 64, 80), (1, 528, 512), (5, 512, 496), (2, 1040, 528)]
for _ in range(10):
    shapes.append((int(rng.integers(1, 5)), 16 * int(rng.integers(1, 14)), 16 * int(rng.integers(1, 14))))
for (N, H, W) in shapes:
    x1, x, xl = r(N, H, W, 1), r(N, H, W, 16), r(N, H // 2, W // 2, 32)
    w1, b1 = r(3, 3, 1, 16) * 0.3, r(16) * 0.1
    w, b = r(3, 3, 16, 16) * 0.08, r(16) * 0.1
    w32, b32 = r(3, 3, 16, 32) * 0.08, r(32) * 0.1
    wt, bt = r(2, 2, 16, 32) * 0.1, r(16) * 0.1
    wh, bh = r(1, 1, 16, 2), r(2) * 0.1
    ab("plain", lambda: (ops.conv2d(x, w, b, act="relu"),), "SQ_CONV_L0")
    ab("plain32", lambda: (ops.conv2d(x, w32, b32, act="relu"),), "SQ_CONV_L0")
    ab("pool", lambda: ops.conv3x3_pool(x, w, b), "SQ_CONV_L0")
    ab("head", lambda: ops.conv3x3_head(x, w, b, wh, bh, act="relu"), "SQ_CONV_L0")
    ab("first", lambda: ops.conv3x3_first_block(x1, w1, b1, w, b, want_pool=True), "SQ_CONV_L0")
    ab("first-nopool", lambda: ops.conv3x3_first_block(x1, w1, b1, w, b, want_pool=False), "SQ_CONV_L0")
    for br in ("eltwise_mul", "eltwise_add", "eltwise_sub", None):
        ab("up-%s" % br, lambda: (ops.convT_conv3x3(xl, wt, bt, x, br, w, b, act="relu"),), "SQ_CONV_L0")
    print("shape", (N, H, W), "done", flush=True)
for (N, H, W, ci, co) in [(2, 9, 13, 64, 32), (1, 200, 190, 64, 32), (3, 64, 64, 128, 64), (1, 100, 72, 256, 128), (2, 128, 128, 32, 32)]:
    x = r(N, H, W, ci)
    wt, bt = r(2, 2, co, ci) * 0.1, r(co) * 0.1
    sk = r(N, 2 * H, 2 * W, co)
    for br in ("eltwise_mul", "eltwise_add", "eltwise_sub", None):
        ab("convT %d->%d %s" % (ci, co, br), lambda: (ops.convT2x2s2(x, wt, bt, skip=sk if br else None, bridge=br),), "SQ_CONVT_V2")
    print("convT", (N, H, W, ci, co), "done", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
