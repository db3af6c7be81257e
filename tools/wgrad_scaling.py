import sys, torch
sys.path.insert(0, '/root/repo')
from sequitr_amd import ops_bf16 as ob
dev = 'cuda:0'
def t(n, h, ci, co, k, reps=10):
    x = torch.randn(n, h, h, ci, device=dev).to(torch.bfloat16); dy = torch.randn(n, h, h, co, device=dev).to(torch.bfloat16)
    for _ in range(3): ob.conv2d_wgrad(x, dy, k)
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): ob.conv2d_wgrad(x, dy, k)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
for (h, c) in [(32, 256), (64, 128), (128, 64), (256, 32)]:
    print((h, c), ["N=%d: %.1f us" % (n, t(n, h, c, c, 3)) for n in (4, 8, 16, 32, 64)])
