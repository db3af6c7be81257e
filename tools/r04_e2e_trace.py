"""GPU-side spans of net.predict inside TileStreamer.run (events recorded round the call): network time per batch and the
gaps between batches, from a pageable and from a pinned source"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import bench
from sequitr_amd.networks.unet import UNet2D, init_unet_weights
from sequitr_amd.frontend import TileStreamer
dev = torch.device('cuda', 0)
params = {"shape": (512, 512), "num_inputs": 1, "num_outputs": 2, "filters": bench.FILTERS, "bridge": "eltwise_mul", "device": str(dev)}
net = UNet2D(params, "infer"); net.load_state_dict(init_unet_weights(params, seed=0))
NT = int(os.environ.get('NT', 1024))
x = np.random.default_rng(1).standard_normal((NT, 512, 512, 1)).astype(np.float32)
for _ in range(40): net.predict(torch.from_numpy(x[:32]).to(dev))
torch.cuda.synchronize()
spans = []
real = net.predict
def traced(t):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); m = real(t); b.record(); spans.append((a, b)); return m
st = TileStreamer(net, batch=32); st.warm_up((512, 512, 1))
print("probe", st.probe_ms)
for name, src in (("pageable", x), ("pageable", x), ("pageable", x), ("pinned", torch.from_numpy(x).pin_memory())):
    net.predict = traced; spans.clear()
    t0 = time.perf_counter(); st.run(src); dt = time.perf_counter() - t0
    net.predict = real
    torch.cuda.synchronize()
    dur = [a.elapsed_time(b) for a, b in spans]
    gap = [spans[i][1].elapsed_time(spans[i + 1][0]) for i in range(len(spans) - 1)]
    print("%s: %.0f Mpix/s; net ms %s" % (name, NT * 512 * 512 / dt / 1e6, " ".join("%.2f" % v for v in dur)))
    print("   gaps ms %s" % " ".join("%.2f" % v for v in gap))
    print("   wall %.1f ms, first net start -> last net end %.1f ms, sum net %.1f, sum gaps %.1f" % (dt * 1e3, spans[0][0].elapsed_time(spans[-1][1]), sum(dur), sum(gap)))
