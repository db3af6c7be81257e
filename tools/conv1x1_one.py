import sys, torch
sys.path.insert(0, '/root/repo')
from sequitr_amd import ops_bf16 as ob
dev = 'cuda:0'
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
for (n, h, k, co) in [(16, 256, 64, 32), (16, 128, 128, 64), (16, 64, 256, 128), (16, 32, 512, 256)]:
    g = torch.randn(n, h, h, k, device=dev).to(torch.bfloat16)
    gate = torch.randn(n, h, h, co, device=dev).to(torch.bfloat16)
    w = torch.randn(1, 1, k, co, device=dev) * 0.1
    wp = ob.pack_weights(w)
    a = t(lambda: ob.conv2d_dgrad_relu(g, wp, gate, 1, scale=1.6667))
    b = t(lambda: ob.conv2d(g, wp, None, 1, co))
    mb = (g.numel() + 2 * gate.numel()) * 2 / 1e6
    print((n, h, k, co), "gated %.1f us (%.0f GB/s)  plain %.1f us" % (a, mb / a * 1e3 / 1e3, b))
