#!/usr/bin/env python3
"""one grouped weight-gradient launch per deep layer shape of config 3 (the group kernel with a single item, SQ_WGRAD_GROUP_SHRINK
as in the step), a few launches each: for per-layer PMC traffic (tools/r03_pmc_wgrad_layers.sh)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sequitr_amd import ops_bf16 as ob
dev = 'cuda:0'
SHAPES = [(256, 16, 32), (256, 32, 32), (128, 32, 64), (128, 64, 64), (64, 64, 128), (64, 128, 128), (32, 128, 256), (32, 256, 256)]
only = int(sys.argv[1]) if len(sys.argv) > 1 else -1
for i, (h, ci, co) in enumerate(SHAPES):
    if only >= 0 and i != only:
        continue
    x = torch.randn(16, h, h, ci, device=dev).to(torch.bfloat16)
    dy = torch.randn(16, h, h, co, device=dev).to(torch.bfloat16)
    dw = torch.zeros(3, 3, ci, co, device=dev)
    db = torch.zeros(co, device=dev)
    for _ in range(4):
        with ob.deferred_wgrads():
            ob.conv2d_wgrad(x, dy, 3, dw_out=dw, db_out=db)
    torch.cuda.synchronize()
