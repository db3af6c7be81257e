import os, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import bench
from sequitr_amd.networks.unet import UNet2D, init_unet_weights
from sequitr_amd.frontend import TileStreamer
dev = torch.device('cuda', 0)
params = {"shape": (512, 512), "num_inputs": 1, "num_outputs": 2, "filters": bench.FILTERS, "bridge": "eltwise_mul", "device": str(dev)}
net = UNet2D(params, "infer"); net.load_state_dict(init_unet_weights(params, seed=0))
x = np.random.default_rng(1).standard_normal((1024, 512, 512, 1)).astype(np.float32)
for _ in range(40): net.predict(torch.from_numpy(x[:32]).to(dev))
torch.cuda.synchronize()
for poll in ("1e-4",) * 6:
    os.environ["SQ_STREAM_POLL"] = poll
    st = TileStreamer(net, batch=32); st.warm_up((512, 512, 1))
    r = []
    for _ in range(3):
        t0 = time.perf_counter(); st.run(x); r.append(1024 * 512 * 512 / (time.perf_counter() - t0) / 1e6)
    print("probe", getattr(st, "probe_ms", None), "rates", " ".join("%.0f" % v for v in r), flush=True)
    st.close()
