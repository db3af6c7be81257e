#!/usr/bin/env python3
"""Warm per-operator timing of the inference step's launches (round 3): every op of UNet2D's fused inference graph at
BASELINE config 2's shapes, timed alone with HIP events after the clocks have settled (300 launches of warm-up
traffic first, tools/clock_probe.py explains why).  Prints us per launch, algorithmic TFLOP/s and GB/s.
   python tools/r03_op_bench.py [convT|level0|all]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sequitr_amd import ops

D = "cuda:0"
N = 32


def timeit(fn, reps=60, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def warm_clocks():
    x = torch.randn(N, 128, 128, 64, device=D)
    w = torch.randn(3, 3, 64, 64, device=D) * 0.05
    for _ in range(300):
        ops.conv2d(x, w, None, act="relu")
    torch.cuda.synchronize()


def convT_cases():
    for h, cin, cout in ((128, 64, 32), (64, 128, 64), (32, 256, 128)):
        x = torch.randn(N, h, h, cin, device=D)
        w = torch.randn(2, 2, cout, cin, device=D) * 0.05
        b = torch.zeros(cout, device=D)
        skip = torch.randn(N, 2 * h, 2 * h, cout, device=D)
        out = torch.empty_like(skip)
        us = timeit(lambda: ops.convT2x2s2(x, w, b, skip=skip, bridge="eltwise_mul", out=out))
        byt = (x.numel() + 2 * skip.numel()) * 4
        fl = 2.0 * N * h * h * 4 * cin * cout
        print("convT %3d->%3d @%3d^2 -> %3d^2: %7.1f us  %6.1f TF  %6.0f GB/s" % (cin, cout, h, 2 * h, us, fl / us / 1e6, byt / us / 1e3),
              flush=True)


def level0_cases():
    h = 512
    x1 = torch.randn(N, h, h, 1, device=D)
    x = torch.randn(N, h, h, 16, device=D)
    xl = torch.randn(N, h // 2, h // 2, 32, device=D)
    w1, b1 = torch.randn(3, 3, 1, 16, device=D) * 0.3, torch.zeros(16, device=D)
    w, b = torch.randn(3, 3, 16, 16, device=D) * 0.08, torch.zeros(16, device=D)
    wt, bt = torch.randn(2, 2, 16, 32, device=D) * 0.1, torch.zeros(16, device=D)
    wh, bh = torch.randn(1, 1, 16, 2, device=D), torch.zeros(2, device=D)
    fl = 2.0 * N * h * h * 9 * 16 * 16
    for name, fn, extra in (
            ("plain 16->16", lambda: ops.conv2d(x, w, b, act="relu"), 0.0),
            ("first_block", lambda: ops.conv3x3_first_block(x1, w1, b1, w, b, want_pool=True), 2.0 * N * h * h * 9 * 16),
            ("convT_conv3x3", lambda: ops.convT_conv3x3(xl, wt, bt, x, "eltwise_mul", w, b, act="relu"), 2.0 * N * h * h * 32 * 16),
            ("conv3x3_head", lambda: ops.conv3x3_head(x, w, b, wh, bh, act="relu"), 2.0 * N * h * h * 16 * 2)):
        us = timeit(fn)
        print("%-14s @512^2: %7.1f us  %6.1f TF (%.3f of 157.3)" % (name, us, (fl + extra) / us / 1e6, (fl + extra) / us / 1e6 / 157.3), flush=True)


def other_cases():
    for h, ci, co in ((256, 16, 32), (256, 32, 32), (128, 32, 64), (128, 64, 64), (64, 128, 128), (32, 256, 256)):
        x = torch.randn(N, h, h, ci, device=D)
        w = torch.randn(3, 3, ci, co, device=D) * 0.05
        b = torch.zeros(co, device=D)
        us = timeit(lambda: ops.conv2d(x, w, b, act="relu"))
        fl = 2.0 * N * h * h * 9 * ci * co
        print("conv %3d->%3d @%3d^2: %7.1f us  %6.1f TF (%.3f)" % (ci, co, h, us, fl / us / 1e6, fl / us / 1e6 / 157.3), flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    warm_clocks()
    if what in ("convT", "all"):
        convT_cases()
    if what in ("level0", "all"):
        level0_cases()
    if what in ("other", "all"):
        other_cases()
