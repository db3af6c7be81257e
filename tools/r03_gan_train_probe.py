#!/usr/bin/env python3
"""Round 3: does the bf16-STORAGE GAN train like the f32 graph over a few hundred solver steps?  Four levels (4x4 .. 32x32),
fade + stabilisation phases of STEPS iterations each, batch 16, synthetic "blob" images, the same seeds in every dtype;
prints per phase the mean / last Wasserstein estimate (-d_loss without penalty is not available separately, so d_loss and
g_loss themselves), whether anything went non-finite, and how far the generator's weights moved.  GPU box only."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sequitr_amd.networks import gan  # noqa: E402


def blobs(rng, n, size):
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32) / size
    out = np.zeros((n, size, size, 2), np.float32)
    for i in range(n):
        cy, cx, r = rng.random(3) * np.array([1, 1, 0.3]) + np.array([0, 0, 0.1])
        d = np.sqrt((yy - cy) ** 2 + (xx - cx) ** 2)
        out[i, ..., 0] = np.exp(-(d / r) ** 2) * 2 - 1
        out[i, ..., 1] = (d < r) * 2.0 - 1
    return out


def run(dtype, steps, levels, graph):
    g = gan.GenerativeAdverserialNetwork({"num_levels": levels, "batch_size": 16, "repeat_batch": 1, "learning_rate": 1e-3,
                                          "device": "cuda:0", "seed": 5, "dtype": dtype, "graph": graph}, mode=None)
    g.build()
    w0 = {k: v.copy() for k, v in g.store.state_dict().items() if k.startswith("GAN/generator")}
    rng = np.random.default_rng(9)
    report = []
    for level in range(levels):
        g.set_level(level)
        size = g.get_size(level)[0]
        for phase in ("fade", "stab"):
            if level == 0 and phase == "fade":
                continue
            dl, gl = [], []
            for it in range(steps):
                alpha = (it + 1) / steps if phase == "fade" else 1.0
                x = torch.from_numpy(blobs(rng, 16, size)).cuda()
                z = torch.from_numpy(rng.standard_normal((16, 1, 1, 512)).astype(np.float32)).cuda()
                g.d_solver(x, z, alpha)
                g.g_solver(x, z, alpha)
                d, gg = g.last_losses
                dl.append(d), gl.append(gg)
            report.append({"level": level, "phase": phase, "d_mean": round(float(np.mean(dl)), 4), "d_last10": round(float(np.mean(dl[-10:])), 4),
                           "g_mean": round(float(np.mean(gl)), 4), "g_last10": round(float(np.mean(gl[-10:])), 4),
                           "finite": bool(np.isfinite(dl).all() and np.isfinite(gl).all()), "d_max_abs": round(float(np.max(np.abs(dl))), 3)})
    w1 = g.store.state_dict()
    moved = float(np.sqrt(sum(((w1[k] - w0[k]) ** 2).sum() for k in w0)))
    img = g.predict(latent=rng.standard_normal((8, 1, 1, 512)).astype(np.float32)).cpu().numpy()
    return report, moved, float(img.std()), bool(np.isfinite(img).all())


if __name__ == "__main__":
    steps = int(os.environ.get("STEPS", 60))
    levels = int(os.environ.get("LEVELS", 4))
    for dtype in os.environ.get("DTYPES", "f32,mixed,bf16").split(","):
        rep, moved, istd, ifin = run(dtype, steps, levels, graph=True)
        print(json.dumps({"dtype": dtype, "steps_per_phase": steps, "generator_weights_moved": round(moved, 3),
                          "image_std": round(istd, 4), "images_finite": ifin}))
        for r in rep:
            print("   ", json.dumps(r))
