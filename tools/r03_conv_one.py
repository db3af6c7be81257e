#!/usr/bin/env python3
"""one deep bf16 conv shape, a few launches (PMC passes): python tools/r03_conv_one.py H Cin Cout [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sequitr_amd import ops_bf16 as ob
h, ci, co = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 16
x = torch.randn(n, h, h, ci, device="cuda:0").to(torch.bfloat16)
wp = ob.pack_weights(torch.randn(3, 3, ci, co, device="cuda:0") * 0.05)
b = torch.zeros(co, device="cuda:0")
for _ in range(12):
    ob.conv2d(x, wp, b, 3, co, act="relu")
torch.cuda.synchronize()
