"""GPU-side timeline of the three streams of a TileStreamer-like loop: when each batch's upload, network pass and download
start and end (HIP events on their own streams), to see what really overlaps"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import bench
from sequitr_amd.networks.unet import UNet2D, init_unet_weights
from sequitr_amd.frontend import TileStreamer
dev = torch.device('cuda', 0)
params = {"shape": (512, 512), "num_inputs": 1, "num_outputs": 2, "filters": bench.FILTERS, "bridge": "eltwise_mul", "device": str(dev)}
net = UNet2D(params, "infer"); net.load_state_dict(init_unet_weights(params, seed=0))
B, NB = 32, 12
xh = torch.from_numpy(np.random.default_rng(1).standard_normal((B, 512, 512, 1)).astype(np.float32)).pin_memory()
xd = [torch.empty((B, 512, 512, 1), device=dev) for _ in range(2)]
mh = [torch.empty((B, 512, 512), dtype=torch.uint8).pin_memory() for _ in range(2)]
for _ in range(40): net.predict(xd[0])
torch.cuda.synchronize()
mode = sys.argv[1] if len(sys.argv) > 1 else "picked"
if mode == "picked":
    st = TileStreamer(net, batch=B); st.warm_up((512, 512, 1))
    s_in, s_out = st.s_in, st.s_out
    print("overlap_found", st.overlap_found)
elif mode == "high":
    s_in, s_out = torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=-1)
else:
    s_in, s_out = torch.cuda.Stream(), torch.cuda.Stream()
print("mode", mode)
main = torch.cuda.current_stream()
E = lambda: torch.cuda.Event(enable_timing=True)
ev = {k: [(E(), E()) for _ in range(NB)] for k in ("up", "net", "down")}
up = [torch.cuda.Event() for _ in range(2)]; used = [torch.cuda.Event() for _ in range(2)]; done = [None] * NB
for e in up + used: e.record(main)
t0 = E(); t0.record(main)
held = [None, None]
for b in range(NB):
    k = b & 1
    with torch.cuda.stream(s_in):
        s_in.wait_event(used[k])
        ev["up"][b][0].record(s_in)
        xd[k].copy_(xh, non_blocking=True)
        ev["up"][b][1].record(s_in)
        up[k].record(s_in)
    main.wait_event(up[k])
    ev["net"][b][0].record(main)
    m = net.predict(xd[k])
    ev["net"][b][1].record(main)
    used[k].record(main)
    d = torch.cuda.Event(); d.record(main)
    held[k] = m
    with torch.cuda.stream(s_out):
        s_out.wait_event(d)
        ev["down"][b][0].record(s_out)
        mh[k].copy_(m, non_blocking=True)
        ev["down"][b][1].record(s_out)
torch.cuda.synchronize()
for b in range(NB if "-v" in sys.argv else 0):
    row = []
    for k in ("up", "net", "down"):
        row.append("%s %7.2f-%7.2f" % (k, t0.elapsed_time(ev[k][b][0]), t0.elapsed_time(ev[k][b][1])))
    print("batch %2d  " % b + "   ".join(row))
print("per batch (net start to net start): %.3f ms" % ((t0.elapsed_time(ev["net"][NB - 1][0]) - t0.elapsed_time(ev["net"][2][0])) / (NB - 3)))
