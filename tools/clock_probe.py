#!/usr/bin/env python3
"""Shader clock and power while one conv layer runs back to back (is a layer MFMA-bound at a reduced clock?).
  python tools/clock_probe.py        # level 0 / 1 / 2 / 4 layers of the BASELINE U-Net, 3 s each, rocm-smi polled"""
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sequitr_amd import ops


def poll(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], stdout=subprocess.PIPE,
                               stderr=subprocess.DEVNULL, text=True, timeout=5).stdout
            out.append(r)
        except Exception as e:                                  # noqa
            out.append("ERR %r" % (e,))
        time.sleep(0.4)


for (n, h, ci, co) in [(32, 512, 16, 16), (32, 256, 32, 32), (32, 128, 64, 64), (32, 32, 256, 256)]:
    x = torch.randn(n, h, h, ci, device="cuda:0")
    w = torch.randn(3, 3, ci, co, device="cuda:0") * 0.05
    b = torch.zeros(co, device="cuda:0")
    stop, out = threading.Event(), []
    th = threading.Thread(target=poll, args=(stop, out))
    th.start()
    t0 = time.time()
    k = 0
    while time.time() - t0 < 3.0:
        for _ in range(50):
            ops.conv2d(x, w, b, act="relu")
        torch.cuda.synchronize()
        k += 50
    dt = time.time() - t0
    stop.set()
    th.join()
    us = dt / k * 1e6
    print("C=%d %dx%d: %.1f us/launch %.1f TF" % (ci, h, h, us, 2.0 * n * h * h * 9 * ci * co / us / 1e6), flush=True)
    for o in out[2:5]:
        print("   ", " | ".join(l for l in o.strip().splitlines()[:3]), flush=True)
