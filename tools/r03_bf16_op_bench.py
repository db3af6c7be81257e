#!/usr/bin/env python3
"""Warm per-launch timing of the bf16 training step's deep convolutions and weight gradients (round 3):
   python tools/r03_bf16_op_bench.py [conv|wgrad|all]      N = 16 tiles, BASELINE config 3's level shapes"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sequitr_amd import _lib
if os.environ.get("SQ_LIB_PATH"):                               # an experimental build of the library (tools/_exp/...)
    _lib.LIB_PATH = os.environ["SQ_LIB_PATH"]
from sequitr_amd import ops_bf16 as ob

D = "cuda:0"
N = int(os.environ.get("BENCH_N", 16))
BF = torch.bfloat16


def timeit(fn, reps=100, warm=30):
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
    x = torch.randn(N, 128, 128, 64, device=D).to(BF)
    wp = ob.pack_weights(torch.randn(3, 3, 64, 64, device=D) * 0.05)
    for _ in range(400):
        ob.conv2d(x, wp, None, 3, 64, act="relu")
    torch.cuda.synchronize()


SHAPES = ((256, 16, 32), (256, 32, 32), (128, 32, 64), (128, 64, 64), (64, 64, 128), (64, 128, 128), (32, 128, 256), (32, 256, 256))


def conv_cases():
    for h, ci, co in SHAPES:
        x = torch.randn(N, h, h, ci, device=D).to(BF)
        wp = ob.pack_weights(torch.randn(3, 3, ci, co, device=D) * 0.05)
        b = torch.zeros(co, device=D)
        us = timeit(lambda: ob.conv2d(x, wp, b, 3, co, act="relu"))
        fl = 2.0 * N * h * h * 9 * ci * co
        byt = N * h * h * (ci + co) * 2
        print("conv  %3d->%3d @%3d^2: %6.1f us  %6.0f TF  %5.0f GB/s" % (ci, co, h, us, fl / us / 1e6, byt / us / 1e3), flush=True)


def wgrad_cases():
    for h, ci, co in SHAPES:
        x = torch.randn(N, h, h, ci, device=D).to(BF)
        dy = torch.randn(N, h, h, co, device=D).to(BF)
        us = timeit(lambda: ob.conv2d_wgrad(x, dy, 3))
        fl = 2.0 * N * h * h * 9 * ci * co
        print("wgrad %3d->%3d @%3d^2: %6.1f us  %6.0f TF" % (ci, co, h, us, fl / us / 1e6), flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    warm_clocks()
    if what in ("conv", "all"):
        conv_cases()
    if what in ("wgrad", "all"):
        wgrad_cases()
