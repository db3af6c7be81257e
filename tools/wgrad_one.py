import sys, torch
sys.path.insert(0, '/root/repo')
from sequitr_amd import ops_bf16 as ob
dev = 'cuda:0'
for (n, h, ci, co, k) in [(16, 256, 32, 32, 3), (16, 128, 64, 64, 3), (16, 32, 256, 256, 3)]:
    x = torch.randn(n, h, h, ci, device=dev).to(torch.bfloat16); dy = torch.randn(n, h, h, co, device=dev).to(torch.bfloat16)
    for _ in range(3): ob.conv2d_wgrad(x, dy, k)
    torch.cuda.synchronize()
