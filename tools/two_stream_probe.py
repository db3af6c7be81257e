#!/usr/bin/env python3
"""Does splitting the 32-tile batch over two streams (two half-batches in flight) beat one stream?
  python tools/two_stream_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from sequitr_amd.networks.unet import UNet2D, init_unet_weights

params = {"shape": (512, 512), "device": "cuda:0"}
w = init_unet_weights(params, 0)
x = torch.from_numpy(np.random.default_rng(1).standard_normal((32, 512, 512, 1)).astype(np.float32)).cuda()


def make():
    n = UNet2D(params, "infer")
    n.load_state_dict(w)
    return n


def timed(fn, reps=40, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


one = make()
print("one stream, 32 tiles: %.3f ms" % timed(lambda: one.predict(x)), flush=True)
for parts in (2, 4):
    nets = [make() for _ in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    chunks = [c.contiguous() for c in torch.chunk(x, parts)]
    ref = one.predict(x).clone()

    def run():
        main = torch.cuda.current_stream()
        outs = []
        for n, s, c in zip(nets, streams, chunks):
            s.wait_stream(main)
            with torch.cuda.stream(s):
                outs.append(n.predict(c))
        for s in streams:
            main.wait_stream(s)
        return outs
    outs = run()
    torch.cuda.synchronize()
    assert torch.equal(torch.cat(outs), ref)
    print("%d streams, %d tiles each: %.3f ms" % (parts, 32 // parts, timed(run)), flush=True)
    print("%d sequential parts on one stream: %.3f ms" % (parts, timed(lambda: [n.predict(c) for n, c in zip(nets, chunks)])), flush=True)

# free-running: each stream loops over its own part without a join per step (as two processes on one GPU do)
for parts in (2, 4):
    nets = [make() for _ in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    chunks = [c.contiguous() for c in torch.chunk(x, parts)]

    def run_free(reps):
        main = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(main)
        for _ in range(reps):
            for n, s, c in zip(nets, streams, chunks):
                with torch.cuda.stream(s):
                    n.predict(c)
        for s in streams:
            main.wait_stream(s)
    run_free(10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_free(40)
    torch.cuda.synchronize()
    print("%d free-running streams, %d tiles each: %.3f ms per 32 tiles" % (parts, 32 // parts, (time.perf_counter() - t0) / 40 * 1e3), flush=True)
# two full batches in flight (the same batch size per launch as the headline, twice the tiles in flight)
nets = [make() for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]
main = torch.cuda.current_stream()
for k in (10, 40):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in streams:
        s.wait_stream(main)
    for _ in range(k):
        for n, s in zip(nets, streams):
            with torch.cuda.stream(s):
                n.predict(x)
    for s in streams:
        main.wait_stream(s)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (2 * k) * 1e3
print("2 free-running streams, 32 tiles each: %.3f ms per 32 tiles" % dt, flush=True)
