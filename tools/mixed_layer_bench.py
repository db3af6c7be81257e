#!/usr/bin/env python3
"""Per-launch time of the GAN's deep mixed-precision conv layers (SQ_CONV_BF16_PF=0|1, SQ_CONV_BF16_NARROW=0|1 A/B)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sequitr_amd import ops

out = []
for (n, h, ci, co) in [(32, 4, 512, 512), (32, 8, 512, 256), (32, 8, 256, 256), (32, 16, 256, 128), (32, 16, 128, 128),
                       (32, 32, 128, 64), (32, 32, 64, 64), (16, 32, 256, 256)]:
    x = torch.randn(n, h, h, ci, device="cuda:0")
    w = (torch.randn(3, 3, ci, co, device="cuda:0") * 0.05).requires_grad_(True)
    with ops.mixed_precision():
        for _ in range(50):
            ops.conv2d(x, w, None, act="leaky", wscale=0.1)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(200):
            ops.conv2d(x, w, None, act="leaky", wscale=0.1)
        e.record()
        torch.cuda.synchronize()
    out.append("%dx%d %d->%d %.1f us" % (h, h, ci, co, s.elapsed_time(e) / 200 * 1e3))
print(" | ".join(out))
