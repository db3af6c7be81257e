import sys

import torch
sys.path.insert(0,'/root/repo')
from sequitr_amd import ops_bf16 as ob
dev='cuda:0'
for (n,h,ci,co,k) in [(16,512,16,16,3),(16,256,32,32,3),(16,128,64,64,3),(16,64,128,128,3),(16,32,256,256,3),(16,256,16,32,3),(16,128,32,64,3),(16,64,64,128,3),(16,32,128,256,3),(16,256,32,64,1),(16,128,64,128,1),(16,64,128,256,1),(16,32,256,512,1)]:
    x=torch.randn(n,h,h,ci,device=dev).to(torch.bfloat16); dy=torch.randn(n,h,h,co,device=dev).to(torch.bfloat16)
    for _ in range(3): ob.conv2d_wgrad(x,dy,k)
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): ob.conv2d_wgrad(x,dy,k)
    e.record(); torch.cuda.synchronize()
    ms=s.elapsed_time(e)/20
    gb=(x.numel()+dy.numel())*2/1e9
    print((n,h,ci,co,k), round(ms*1e3,1),'us', round(gb/ms*1e3,1),'GB/s algorithmic')
