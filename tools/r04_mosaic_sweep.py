#!/usr/bin/env python3
"""GPU: per-launch time of the GAN's small-image (mosaic) convolutions for forced block widths / splits.
   python tools/r04_mosaic_sweep.py  -- runs itself once per (SQ_MOS_BN, SQ_MOS_S) in child processes (the switches are read once)"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CASES = ((32, 4, 512, 512), (64, 4, 512, 512), (32, 8, 512, 256), (32, 8, 256, 256), (64, 8, 256, 512), (64, 8, 512, 512),
         (32, 8, 512, 512), (32, 8, 256, 512))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from sequitr_amd import ops
    from sequitr_amd import ops_gan_bf16 as gb
    D = "cuda:0"
    def timeit(fn, reps=200, warm=50):
        for _ in range(warm): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps): fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / reps * 1e3
    xw = torch.randn(16, 128, 128, 64, device=D).to(torch.bfloat16); ww = torch.randn(3, 3, 64, 64, device=D)
    with ops.mixed_precision(True, store_bf16=True):
        for _ in range(300): gb.conv2d(xw, ww, None, "leaky", 0.1)
        out = []
        for n, h, ci, co in CASES:
            x = torch.randn(n, h, h, ci, device=D).to(torch.bfloat16)
            w = (torch.randn(3, 3, ci, co, device=D)).requires_grad_(True)      # a leaf: its pack is cached
            b = torch.zeros(co, device=D)
            for _ in range(3): gb.conv2d(x, w, b, "leaky", 0.02)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()                          # 20 launches per replay: GPU time, not the host's issue rate
            with torch.cuda.graph(g):
                for _ in range(20): y = gb.conv2d(x, w, b, "leaky", 0.02)
            us = timeit(g.replay, reps=20, warm=5) / 20
            out.append("%6.1f" % us)
    print(" ".join(out), flush=True)
    sys.exit(0)
print("case (N,h,Cin,Cout):      " + " ".join("%d,%d,%d,%d" % c for c in CASES))
for bn in ("", "16", "32", "64"):
    for S in ("", "1", "2", "4", "8"):
        env = dict(os.environ)
        if bn: env["SQ_MOS_BN"] = bn
        if S: env["SQ_MOS_S"] = S
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        print("BN %-4s S %-4s us: %s" % (bn or "auto", S or "auto", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "FAILED " + r.stderr[-200:]), flush=True)
