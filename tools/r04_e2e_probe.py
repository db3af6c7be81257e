"""GPU probe: TileStreamer rate vs the plain three-stream loop (bench r03) on the same box, with host timelines"""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import bench
from sequitr_amd.networks.unet import UNet2D, init_unet_weights
from sequitr_amd.frontend import TileStreamer
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
params = {"shape": (512, 512), "num_inputs": 1, "num_outputs": 2, "filters": bench.FILTERS, "bridge": "eltwise_mul", "device": str(dev)}
net = UNet2D(params, "infer"); net.load_state_dict(init_unet_weights(params, seed=0))
x_dev = torch.from_numpy(np.random.default_rng(1).standard_normal((32, 512, 512, 1)).astype(np.float32)).to(dev)
for _ in range(3): net.predict(x_dev)
torch.cuda.synchronize()
TILE = 512

def old_loop(iters=5, prio=0):
    n = x_dev.shape[0]
    xh = x_dev.cpu().pin_memory()
    mh = [torch.empty(x_dev.shape[:3], dtype=torch.uint8).pin_memory() for _ in range(2)]
    xd = [torch.empty_like(x_dev) for _ in range(2)]
    main = torch.cuda.current_stream()
    s_in, s_out = torch.cuda.Stream(priority=prio), torch.cuda.Stream(priority=prio)
    up = [torch.cuda.Event() for _ in range(2)]; used = [torch.cuda.Event() for _ in range(2)]; down = [torch.cuda.Event() for _ in range(2)]
    def run(k):
        for i in range(k):
            b = i & 1
            with torch.cuda.stream(s_in):
                if i >= 2: s_in.wait_event(used[b])
                xd[b].copy_(xh, non_blocking=True); up[b].record(s_in)
            main.wait_event(up[b])
            mask = net.predict(xd[b]); used[b].record(main)
            done = torch.cuda.Event(); done.record(main)
            with torch.cuda.stream(s_out):
                s_out.wait_event(done)
                if i >= 2: s_out.wait_event(down[b])
                mh[b].copy_(mask, non_blocking=True); mask.record_stream(s_out); down[b].record(s_out)
        torch.cuda.synchronize()
    run(4)
    k = 10
    t0 = time.perf_counter(); run(k); return (time.perf_counter() - t0) / k * 1e3

print("old r03 loop, ms per batch:", [round(old_loop(), 3) for _ in range(6)])
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else None)
n = 32
xh = x_dev.cpu().repeat(20, 1, 1, 1).pin_memory()
for rep in range(6):
    st = TileStreamer(net, batch=n)
    st.warm_up((512, 512, 1))
    t0 = time.perf_counter(); st.run(xh, out_masks=np.empty((640, 512, 512), np.uint8)); torch.cuda.synchronize()
    print("fresh TileStreamer", rep, "overlapping candidates found", st.overlap_found, "ms per batch", round((time.perf_counter() - t0) / 20 * 1e3, 3))
masks = np.empty((640, 512, 512), np.uint8)
for src, name in ((xh, "pinned"), (xh.numpy().copy(), "pageable")):
    st.run(src[:64], out_masks=masks[:64])
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter(); st.run(src, out_masks=masks); ts.append((time.perf_counter() - t0) / 20 * 1e3)
    print("TileStreamer", name, "ms per batch:", [round(t, 3) for t in ts])
# compute alone
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): net.predict(x_dev)
torch.cuda.synchronize(); print("predict alone ms:", (time.perf_counter() - t0) / 10 * 1e3)
# H2D / D2H alone
xd = torch.empty_like(x_dev); mh = torch.empty((32, 512, 512), dtype=torch.uint8).pin_memory(); m = net.predict(x_dev)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(10): xd.copy_(xh[i * 32:(i + 1) * 32], non_blocking=True)
torch.cuda.synchronize(); print("H2D 33.5 MB from a slice of the big pinned tensor ms:", (time.perf_counter() - t0) / 10 * 1e3)
xs = x_dev.cpu().pin_memory()
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(10): xd.copy_(xs, non_blocking=True)
torch.cuda.synchronize(); print("H2D 33.5 MB from its own pinned tensor ms:", (time.perf_counter() - t0) / 10 * 1e3)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(10): mh.copy_(m, non_blocking=True)
torch.cuda.synchronize(); print("D2H 8 MB ms:", (time.perf_counter() - t0) / 10 * 1e3)
