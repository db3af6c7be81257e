#!/usr/bin/env python3
"""Round 3, VERDICT item 6: which default learning rate / warm-up trains the BASELINE config-3 net (5 levels, 512x512,
16 tiles) without the step-2 blow-up of lr 0.01 (gpurun_out/dbg/eager.log: 1.44 -> 1.8e15 -> 4968 -> 1.37)?
Runs <= 60 captured steps per setting on disk-label tiles (bench.disk_image_inputs) and prints the loss curve and the
foreground IoU of the trained net's masks vs the labels.  GPU box only; output feeds DESIGN.md section 8."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from sequitr_amd.train import UNetTrainer  # noqa: E402
from sequitr_amd.networks.unet import UNet2D, UNet2DBf16  # noqa: E402


def iou_fg(mask, lab):
    m = mask.astype(bool)
    return float(np.logical_and(m, lab).sum() / max(1, np.logical_or(m, lab).sum()))


def run(dtype, lr, warm, steps, data, dropout=0.4, extra=None, every=0):
    x, onehot, wmap, lab = data
    params = dict({"shape": (512, 512), "dropout": dropout, "device": "cuda:0", "seed": 0, "dtype": dtype}, **(extra or {}))
    t = UNetTrainer(params, learning_rate=lr, warmup_steps=warm)
    t.capture(x, onehot, wmap, warmup=1)
    losses = [float(t.last_loss.item())]
    log = torch.zeros(steps, device="cuda:0")
    cls = UNet2DBf16 if dtype == "bf16" else UNet2D
    net = cls(dict(params, dropout=0.0), "infer")
    trace = []

    def evaluate():
        net.load_state_dict(t.state_dict())
        return iou_fg(net.predict(x).cpu().numpy(), lab)
    for k in range(1, steps):
        log[k].copy_(t.step(x, onehot, wmap))
        if every and (k + 1) % every == 0:
            trace.append((k + 1, round(float(log[k].item()), 4), round(evaluate(), 4)))
    losses += [float(v) for v in log[1:].cpu().numpy()]
    iou = evaluate()
    del t
    return losses, iou, trace


def main():
    steps = int(os.environ.get("STEPS", 60))
    data = bench.disk_image_inputs("cuda:0", seed=int(os.environ.get("DATA_SEED", 2)), nb=16)
    grid = [("bf16", 0.01, 0), ("bf16", 0.01, 10), ("bf16", 0.01, 30), ("bf16", 0.003, 0), ("bf16", 0.003, 10),
            ("bf16", 0.001, 0), ("bf16", 0.001, 10), ("f32", 0.01, 10), ("f32", 0.003, 10), ("f32", 0.001, 10),
            ("f32", 0.001, 0)]
    if os.environ.get("GRID"):
        grid = [tuple(json.loads(g)) for g in os.environ["GRID"].split(";")]
    every = int(os.environ.get("EVERY", 0))
    dropout = float(os.environ.get("DROPOUT", 0.4))
    extra = json.loads(os.environ.get("EXTRA", "{}"))
    for dtype, lr, warm in grid:
        losses, iou, trace = run(dtype, lr, warm, steps, data, dropout=dropout, extra=extra, every=every)
        if trace:
            print(json.dumps({"dtype": dtype, "lr": lr, "warmup": warm, "dropout": dropout, "extra": extra,
                              "trace(step, loss, iou)": trace}), flush=True)
        mono = all(losses[i + 1] <= losses[i] * 1.0001 for i in range(3, len(losses) - 1))
        print(json.dumps({"dtype": dtype, "lr": lr, "warmup": warm, "iou_fg": round(iou, 4), "max_loss": max(losses),
                          "monotone_after_3": mono, "last": round(losses[-1], 5),
                          "argmax_loss": int(np.argmax(losses)),
                          "losses": [round(v, 4) for v in losses[:int(os.environ.get("HEAD", 8))]] + ["..."] + [round(v, 5) for v in losses[-4:]]}),
              flush=True)


if __name__ == "__main__":
    main()
