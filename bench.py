#!/usr/bin/env python3
"""bench.py -- segmented Mpixels/s of the U-Net inference hot path on MI355X.

Workload (BASELINE.json configs[1]): 2-class U-Net, filters (16,32,64,128,256), batch of
32 synthetic 512x512x1 tiles per GPU, fp32, inputs resident in HBM before the timed
region.  One "step" = one pass of the hot path over the batch: UNet2D.predict() =
logits + uint8 argmax mask for every tile.  Multi-GPU: one process per GPU, each rank
segments its own 32-tile batch (tiles are independent units -> weak scaling, no
data-path collective); value = pixels all ranks segmented / max-over-ranks time.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel = the f32 MFMA implicit-GEMM
convolution; achieved = its algorithmic FLOPs / its HIP-event time inside the timed region)
and `cpu_baseline` (the torch-CPU oneDNN restatement of the same net on the host cores,
standing in for the reference's TF-CPU path -- see BASELINE.md section 3).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
TILE = 512
BATCH = 32
FILTERS = (16, 32, 64, 128, 256)


def init_ranks(dist, local_rank):
    """one process per GPU over RCCL (torch's "nccl" backend).  SQ_BENCH_BACKEND=gloo is the rehearsal mode for a
    one-GPU box: every rank uses GPU 0 and gloo carries the (tiny) collectives of the harness -- it exercises the
    N > 1 control flow (barriers, max-over-ranks time, rank-0 line), not the interconnect."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if os.environ.get("SQ_BENCH_BACKEND", "nccl") == "gloo":
        torch.cuda.set_device(0)
        dist.init_process_group("gloo")
    else:
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))


def rank_setup():
    """(world, rank, dist-or-None, device) from the RANK / LOCAL_RANK / WORLD_SIZE the launcher (torch.distributed.run or
    launch_ranks below) put in the environment; initialises the process group when world > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        init_ranks(dist, local_rank)
    else:
        torch.cuda.set_device(0)
    return world, rank, dist, torch.device("cuda", torch.cuda.current_device())


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh child processes of this script, one per GPU, with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, and relay rank 0's single JSON line.  The parent never touches the
    GPU (no torch.cuda call at all: `import torch` does not initialise HIP) and never re-execs itself; any failed
    child makes the exit code non-zero and ends the others (by PID)."""
    import socket
    import subprocess
    gloo = os.environ.get("SQ_BENCH_BACKEND", "nccl") == "gloo"
    # The launcher must never initialise HIP (it fork+execs the ranks): the GPU count comes from the render nodes / KFD
    # topology / *_VISIBLE_DEVICES (sequitr_amd/hwinfo.py), not from torch.cuda -- whose amdsmi route falls back to
    # hipGetDeviceCount.  Unknown (None) or the gloo rehearsal: no check, a rank that cannot set its device exits
    # non-zero and that exit code is propagated below.
    from sequitr_amd.hwinfo import count_gpus
    have = None if gloo else count_gpus()
    if have is not None and have < n:
        sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) visible (SQ_BENCH_BACKEND=gloo rehearses the N-rank "
                         "control flow on one card)\n" % (n, have))
        return 2
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=(r == 0)))
    import threading
    buf = []
    reader = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    reader.start()                                      # drain rank 0's pipe while polling every child
    rc = 0
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for q in procs:                         # one rank died: the others would wait in a barrier for ever
                    if q.poll() is None:
                        q.terminate()
        time.sleep(0.05)
    reader.join(10)
    out0 = buf[0] if buf else ""
    lines = [l for l in (out0 or "").splitlines() if l.strip()]
    if rc == 0 and lines:
        print(lines[-1])
        sys.stdout.flush()
    elif rc == 0:
        sys.stderr.write("bench.py: rank 0 printed no line\n")
        rc = 1
    return rc


def mfma_conv_flops(n, h, w, cin, cout, k):
    return 2.0 * n * h * w * cin * cout * k * k


class ConvTimer(object):
    """HIP-event timing of the MFMA implicit-GEMM convolution launches (conv2d and its fused forms
    conv3x3_pool / conv3x3_head / conv3x3_first_block / convT_conv3x3) on the stream they are launched on
    (torch's current stream IS the launch stream).  Consecutive conv launches share one event pair: a group is
    opened by the first conv after any other kernel and closed right before the next other kernel (the
    transpose convs), so a step costs 10 event records instead of 34 and the instrumentation perturbs the
    measured step by < 1 %.  achieved = sum of algorithmic conv FLOPs / sum of group times; the average launch
    duration is group time / launches in the group."""

    NAMES = ("conv2d", "conv3x3_pool", "conv3x3_head", "conv3x3_first_block", "convT_conv3x3")
    BREAKERS = ("convT2x2s2", "conv1x1_argmax", "maxpool2x2", "argmax_u8", "bridge")

    def __init__(self, ops_mod, expected_groups=0):
        self.ops = ops_mod
        self.orig = {n: getattr(ops_mod, n) for n in self.NAMES + self.BREAKERS}
        self.records = []        # (start_event, end_event, flops, launches)
        self.open = None
        # events are created BEFORE the timed region: hipEventCreate in the launch path costs host time
        self.pool = [torch.cuda.Event(enable_timing=True) for _ in range(2 * expected_groups)]

    def _event(self):
        return self.pool.pop() if self.pool else torch.cuda.Event(enable_timing=True)

    @staticmethod
    def _flops(name, a):
        x = a[0]
        n, h, w = int(x.shape[0]), int(x.shape[1]), int(x.shape[2])
        if name == "conv3x3_first_block":                     # conv1 (1->16) + conv2 (16->16), one launch
            return mfma_conv_flops(n, h, w, 1, 16, 3) + mfma_conv_flops(n, h, w, 16, 16, 3)
        if name == "convT_conv3x3":                           # transpose conv (32->16 at the skip's size) + 3x3 conv
            sk = a[3]
            n, h, w = int(sk.shape[0]), int(sk.shape[1]), int(sk.shape[2])
            return 2.0 * n * h * w * 32 * 16 + mfma_conv_flops(n, h, w, 16, 16, 3)
        wt = a[1]
        cin, cout, k = int(wt.shape[2]), int(wt.shape[3]), int(wt.shape[0])
        if name == "conv2d" and not ((cin % 16 == 0 or cin == 8) and not (k == 1 and cout <= 4)):
            return None                                       # VALU kernels (first layer, small head)
        f = mfma_conv_flops(n, h, w, cin, cout, k)
        if name == "conv3x3_head":
            f += mfma_conv_flops(n, h, w, 16, int(a[3].shape[3]), 1)
        return f

    def close(self):
        """end the open group (call before the timed region's final barrier)"""
        if self.open is not None:
            e = self._event()
            e.record()
            s, fl, nl = self.open
            self.records.append((s, e, fl, nl))
            self.open = None

    def __enter__(self):
        def wrap(name):
            orig = self.orig[name]

            def timed(*a, **kw):
                fl = self._flops(name, a)
                if fl is None:
                    self.close()
                    return orig(*a, **kw)
                if self.open is None:
                    s = self._event()
                    s.record()
                    self.open = [s, 0.0, 0]
                y = orig(*a, **kw)
                self.open[1] += fl
                self.open[2] += 1
                return y
            return timed

        def breaker(name):
            orig = self.orig[name]

            def other(*a, **kw):
                self.close()
                return orig(*a, **kw)
            return other
        for n in self.NAMES:
            setattr(self.ops, n, wrap(n))
        for n in self.BREAKERS:
            setattr(self.ops, n, breaker(n))
        return self

    def __exit__(self, *a):
        self.close()
        for n, f in self.orig.items():
            setattr(self.ops, n, f)

    def summary(self):
        ms = sum(s.elapsed_time(e) for s, e, _, _ in self.records)
        fl = sum(f for _, _, f, _ in self.records)
        return ms, fl, sum(n for _, _, _, n in self.records)


def pmc_conv_traffic():
    """HBM bytes of the conv_mfma launches from the rocprofv3 PMC passes of THIS command, collected and corrected as
    MI355X_MICROARCH.md prescribes (separate --pmc FETCH_SIZE / WRITE_SIZE runs; FETCH_SIZE x2 on gfx950): a dict
    {per_step, per_launch, launches_per_step, algorithmic_per_step} or None.  PMC counters cannot be read from inside
    the timed run, so the committed summary is reported."""
    import glob
    try:
        fn = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.json")))[-1]   # newest round
        with open(fn) as f:
            d = json.load(f)["_summary"]
        pmc_conv_traffic.source = os.path.relpath(fn, ROOT)
        return {"per_step": round(d["conv_mfma_hbm_bytes_per_step"], 1),
                "per_launch": round(d["conv_mfma_hbm_bytes_per_launch_avg"], 1),
                "launches_per_step": d.get("conv_mfma_launches_per_step"),
                "algorithmic_per_step": d.get("conv_mfma_algorithmic_bytes_per_step")}
    except Exception:
        return None


pmc_conv_traffic.source = None


def pmc_mfma_busy():
    """MFMA-pipe busy fraction per conv kernel from the committed SQ counter passes (tools/pmc_sq.sh +
    tools/pmc_sq_summary.py: SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)); None if absent.  It counts
    every MFMA issued, including the recomputed halos of the fused forms, so it sits above `frac`."""
    import glob
    try:
        fn = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_sq_summary.json")))[-1]
        pmc_mfma_busy.source = os.path.relpath(fn, ROOT)
        with open(fn) as f:
            d = json.load(f)
        return {k.replace("conv_mfma_f32_v2_kernel", "v2").replace("conv_l0_kernel", "l0"): v["mfma_busy_frac"]
                for k, v in d.items() if "mfma_busy_frac" in v and k.startswith(("conv_mfma_f32_v2", "conv_l0_kernel"))}
    except Exception:
        return None


pmc_mfma_busy.source = None


def o1_weights(params, seed=0, gain=1.35, bias_std=0.05):
    """Non-degenerate parity fixture: the seeded initial weights with every 3x3 kernel scaled by
    `gain` and seeded N(0, bias_std) biases.  With variance_scaling kernels, zero biases and eltwise_mul bridges the
    benchmark net's logits are ~1e-5 (positively homogeneous of degree ~39 in the kernel scale); gain 1.35 puts them at
    std ~ 1, max ~ 10, and the biases break the homogeneity, so tolerance statements (north_star: logits within 1e-3)
    and IoU figures are made on logits of order one.  A deterministic function of (params, seed): no file."""
    from sequitr_amd.networks.unet import init_unet_weights
    w = init_unet_weights(params, seed)
    rng = np.random.default_rng(seed + 7)
    for k in sorted(w):
        if k.endswith("kernel") and "upscale" not in k and "to_image" not in k:
            w[k] = (w[k] * np.float32(gain)).astype(np.float32)
        elif k.endswith("bias"):
            w[k] = (rng.standard_normal(w[k].shape) * bias_std).astype(np.float32)
    return w


def parity_on_o1_fixture(params, device, with_cpu=True):
    """IoU / logits parity of the GPU path on the O(1)-logit fixture: (a) one 512x512 tile against the C oracle
    (oracle/sq_oracle.c: bit-exact logits and mask expected), (b) 8 tiles against the torch-oneDNN CPU restatement
    (another summation order: logits within 1e-3, masks equal except near-ties), with the near-tie pixel counts SURVEY 7
    asks for (|z1 - z0| below 1e-3 / 1e-6)."""
    from oracle import unet_oracle
    from oracle.torch_ref import TorchCpuUNet
    from sequitr_amd.networks.unet import UNet2D
    w = o1_weights(params)
    net = UNet2D(dict(params, device=str(device)), "infer")
    net.load_state_dict(w)
    xb = np.random.default_rng(11).standard_normal((8, TILE, TILE, 1)).astype(np.float32)
    gmask = net.predict(torch.from_numpy(xb).to(device)).cpu().numpy()
    glog = net.logits().cpu().numpy()
    gap = np.abs(glog[..., 1] - glog[..., 0])
    out = {"weights": "bench.o1_weights(seed 0, gain 1.35, bias std 0.05)", "logits_std": float("%.3g" % glog.std()),
           "logits_max_abs": float("%.3g" % np.abs(glog).max()), "foreground_fraction": round(float(gmask.mean()), 4),
           "near_tie_pixels": {"below_1e-3": int((gap < 1e-3).sum()), "below_1e-6": int((gap < 1e-6).sum()),
                               "of": int(gap.size)}}
    ref = unet_oracle.unet_forward(xb[:1], w, params)                         # ~5 s of scalar C on one tile
    out["vs_c_oracle_one_tile"] = {"logits_bit_exact": bool(np.array_equal(glog[:1], ref)),
                                   "mask_bit_exact": bool(np.array_equal(gmask[:1], unet_oracle.predict_mask(ref))),
                                   "logits_max_abs_diff": float(np.abs(glog[:1] - ref).max())}
    if with_cpu:
        threads = int(os.environ.get("SQ_CPU_THREADS", min(os.cpu_count() or 1, 16)))    # a parity leg, not a timing
        cl = TorchCpuUNet(w, params, threads=threads)(xb)
        cm = np.argmax(cl, -1).astype(np.uint8)
        diff = gmask != cm
        out["vs_cpu_onednn_8_tiles"] = {"iou_per_class": [round(v, 6) for v in iou_per_class(gmask, cm)],
                                        "pixels_differing": int(diff.sum()),
                                        "largest_gap_at_a_differing_pixel": float(gap[diff].max()) if diff.any() else 0.0,
                                        "logits_max_abs_diff": float("%.3g" % np.abs(glog - cl).max())}
    return out


def iou_per_class(a, b, nclass=2):
    out = []
    for c in range(nclass):
        pa, pb = (a == c), (b == c)
        union = np.logical_or(pa, pb).sum()
        out.append(float(np.logical_and(pa, pb).sum()) / float(union) if union else 1.0)
    return out


def end_to_end_rate(net, x_dev, iters=5, dist=None, world=1):
    """PCIe-inclusive rates (reported beside `value`, never as it), measured through the PRODUCT's streamed data path
    (sequitr_amd.frontend.TileStreamer -- what jobs.SERVER_segment runs): float32 tiles in host memory -> H2D ->
    predict -> uint8 masks -> D2H -> host array.  `value`: the tiles already sit in PINNED host memory (uploaded in
    place); `from_pageable`: they sit in an ordinary numpy array and pass through the streamer's pinned staging
    threads first (the job's case: np.load / memmap); `serial`: H2D + predict + D2H of one batch on ONE stream.
    With world > 1 every rank streams its own tiles at the same time (barrier before, MAX of the rank times), so the
    number shows whether the host can feed N cards at once."""
    from sequitr_amd.frontend import TileStreamer
    n = x_dev.shape[0]
    k = max(2 * iters, 32)                                     # a 32-batch stream: pipeline fill, drain and the clock ramp of
                                                               # its first batches are ~4 ms, 2 % of it (a 10-batch stream: 7 %)
    xh1 = x_dev.cpu()
    xh = xh1.repeat(k, 1, 1, 1).pin_memory()                   # k batches, pinned (allocated after set_device)
    st = TileStreamer(net, batch=n, want_logits=False)
    st.warm_up(tuple(x_dev.shape[1:]))
    masks = np.zeros((k * n,) + tuple(x_dev.shape[1:3]), np.uint8)   # zeros: the pages are touched before the timed pass

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(src):
        st.run(src[:2 * n], out_masks=masks[:2 * n])
        sync_all()
        t0 = time.perf_counter()
        st.run(src, out_masks=masks)
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=x_dev.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt / k

    piped = timed(xh)
    ref_mask = net.predict(x_dev).cpu().numpy()
    same = bool(np.array_equal(masks[:n], ref_mask) and np.array_equal(masks[-n:], ref_mask))
    pageable = timed(xh.numpy().copy())

    mh = torch.empty(x_dev.shape[:3], dtype=torch.uint8).pin_memory()
    xd = torch.empty_like(x_dev)
    for _ in range(2):
        xd.copy_(xh[:n], non_blocking=True), mh.copy_(net.predict(xd), non_blocking=True)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(iters):
        xd.copy_(xh[:n], non_blocking=True)
        mh.copy_(net.predict(xd), non_blocking=True)
    torch.cuda.synchronize()
    serial = (time.perf_counter() - t0) / iters
    tiles = float(n)
    if dist is not None:                                       # shards may differ by a tile: count what really streamed
        tn = torch.tensor([tiles], dtype=torch.float64, device=x_dev.device)
        dist.all_reduce(tn, op=dist.ReduceOp.SUM)
        tiles = float(tn.item())
    px = tiles * TILE * TILE

    def rate(t):
        return {"value": round(px / t / 1e6, 3), "ms_per_step": round(t * 1e3, 4)}
    out = dict(rate(piped), unit="Mpixels/s", ranks_streaming=world, masks_equal_predict=same,
               from_pageable=rate(pageable),
               serial={"value": round(n * TILE * TILE / serial / 1e6, 3), "ms_per_step": round(serial * 1e3, 4)},
               what="sequitr_amd.frontend.TileStreamer over %d batches of %d tiles per rank: host f32 tiles -> H2D -> "
                    "predict -> uint8 masks -> D2H -> host array, three streams, double-buffered; value: pinned source; "
                    "from_pageable: numpy source through the pinned staging threads; serial: one stream, one batch "
                    "at a time (this rank alone)" % (k, n))
    return out


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def timed_passes(fn, warm=3, timed=10, budget_s=12.0):
    """>= `warm` warm-up and `timed` timed passes; the timed count shrinks only when one pass
    is so slow that `timed` of them would exceed the budget.  Returns (times, n_warm)."""
    for _ in range(warm):
        t0 = time.perf_counter()
        fn()
        one = time.perf_counter() - t0
    n = int(max(3, min(timed, budget_s / max(one, 1e-4))))
    times = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        times.append(time.perf_counter() - t0)
    return times, warm


def usable_cpus():
    """CPUs this process may really use: scheduler affinity clipped by the cgroup CPU quota (a 1-GPU box owns a
    16-core share of a 256-thread host; os.cpu_count() still says 256)."""
    ncpu = os.cpu_count() or 1
    try:
        ncpu = min(ncpu, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for fn in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(fn) as f:
                tok = f.read().split()
            if fn.endswith("cpu.max"):
                quota, period = tok[0], float(tok[1])
            else:
                quota = tok[0]
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                    period = float(f2.read().split()[0])
            if quota not in ("max", "-1") and period > 0:
                ncpu = min(ncpu, max(1, int(-(-float(quota) // period))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return ncpu


def cpu_thread_candidates():
    """thread counts the CPU legs try (BASELINE.md section 3 / VERDICT r2 item 9): 16 (the 1-GPU box's CPU share), 64
    and os.cpu_count(), each kept only while it is at most 4 x the CPUs this process may really use (usable_cpus():
    256 threads on a 16-core cgroup share run 250 x slower than 16 and would cost the default run minutes);
    SQ_CPU_THREADS pins one."""
    if os.environ.get("SQ_CPU_THREADS"):
        return [int(os.environ["SQ_CPU_THREADS"])]
    ncpu, use = os.cpu_count() or 1, usable_cpus()
    cands = sorted({min(t, ncpu) for t in (16, 64, ncpu, use) if t >= 1})
    return [t for t in cands if t <= 4 * use] or [use]


def cpu_baseline(weights, params, gpu_net=None):
    """Bounded CPU sample as BASELINE.md section 3 prescribes: the torch-CPU (oneDNN, fp32, channels_last) restatement
    of the same net on tiles of the same workload, threads in {16, 64, os.cpu_count()} x batch in {1, 8}, 3 warm-up +
    <= 10 timed passes each (about 4 s per cell), median / min / max; `value` is the BEST cell's median with its thread
    count and batch, CPU model stated.  Never the thing shipped, only the reported baseline."""
    from oracle.torch_ref import TorchCpuUNet
    xb = np.random.default_rng(1).standard_normal((8, TILE, TILE, 1)).astype(np.float32)
    rows, best = {}, None
    net = None
    probe_best = None
    for threads in cpu_thread_candidates():
        net = TorchCpuUNet(weights, params, threads=threads)
        net(xb[:1])                                             # primitive creation for this thread count
        t0 = time.perf_counter()
        net(xb[:1])
        probe = time.perf_counter() - t0
        if probe_best is not None and probe > 2.5 * probe_best:
            # oversubscribed (the cgroup quota is not always visible): one probe pass says enough, no timed cells
            rows["threads_%d" % net.threads] = {"skipped": "probe pass %.2f s vs %.2f s at fewer threads" % (probe, probe_best)}
            break
        probe_best = probe if probe_best is None else min(probe_best, probe)
        for nb in (1, 8):
            xin = xb[:nb]
            times, warm = timed_passes(lambda: net(xin), budget_s=4.0)
            rate = [nb * TILE * TILE / t / 1e6 for t in times]
            cell = {"median": round(float(np.median(rate)), 3), "min": round(min(rate), 3), "max": round(max(rate), 3),
                    "warmup": warm, "timed": len(times), "threads": net.threads, "batch": nb}
            rows["threads_%d_batch_%d" % (net.threads, nb)] = cell
            if best is None or cell["median"] > best["median"]:
                best = cell
    if net.threads != best["threads"]:
        net = TorchCpuUNet(weights, params, threads=best["threads"])
    iou = near = None
    if gpu_net is not None:                       # matched-IoU check of the timed GPU net against this CPU run
        cpu_logits = net(xb)
        cpu_mask = np.argmax(cpu_logits, axis=-1).astype(np.uint8)
        gpu_mask = gpu_net.predict(torch.from_numpy(xb).to(gpu_net.device)).cpu().numpy()
        iou = [round(v, 6) for v in iou_per_class(gpu_mask, cpu_mask, gpu_net.n_outputs)]
        near = int((gpu_mask != cpu_mask).sum())
    return {"value": best["median"], "unit": "Mpixels/s", "cores": best["threads"], "batch": best["batch"],
            "kind": "port", "cpu_model": cpu_model_name(), "os_cpu_count": os.cpu_count(), "usable_cpus": usable_cpus(),
            "passes": rows,
            "iou_gpu_vs_cpu_per_class": iou, "pixels_differing_gpu_vs_cpu": near,
            "sample": "CPU restatement (torch-oneDNN fp32 channels_last, oracle/torch_ref.py) of the same U-Net, standing "
                      "in for the reference TF-CPU path (TensorFlow not installable); threads x batch sweep, value = the "
                      "best cell: median over %d timed passes of %d x 512x512 tiles after 3 warm-up passes, %d threads"
                      % (best["timed"], best["batch"], best["threads"])}


def best_cpu_threads(fn_for_threads, budget_s=6.0):
    """pick the thread count for a slow CPU leg (training / GAN): one pass per candidate, fastest wins"""
    cands = cpu_thread_candidates()
    if len(cands) == 1:
        return cands[0]
    best, best_t = cands[0], None
    for th in cands:
        torch.set_num_threads(th)
        t0 = time.perf_counter()
        fn_for_threads()
        dt = time.perf_counter() - t0
        if best_t is None or dt < best_t:
            best, best_t = th, dt
        elif dt > 1.5 * best_t:                        # more threads are already slower: larger counts will not help
            break
        if dt > budget_s:
            break
    return best


PEAK_BF16_MFMA_TFLOPS = 2500.0    # dense bf16 MFMA, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
PEAK_HBM_GBS = 8000.0             # HBM3E spec, MI355X_MICROARCH.md "HBM3E peak BW"


def unet_work_per_tile(filters=FILTERS, tile=TILE, nout=2):
    """SURVEY A.6 per 512x512 tile: (forward FLOPs, forward layer-by-layer activation elements read + written,
    first-layer MACs).  Training FLOPs = 3 x forward - the first layer's dgrad (its input needs no gradient)."""
    flops, elems, cin = 0.0, 0, 1
    first = None
    for i, c in enumerate(filters):                                         # encoder
        h = tile >> i
        if i > 0:
            elems += (2 * h) * (2 * h) * cin + h * h * cin                  # pool: read + write
        for ci in (cin, c):
            f = mfma_conv_flops(1, h, h, ci, c, 3)
            first = f if first is None else first
            flops += f
            elems += h * h * (ci + c)
        cin = c
    for i in reversed(range(len(filters) - 1)):                             # decoder
        h, c, c2 = tile >> i, filters[i], filters[i + 1]
        flops += 2.0 * (h // 2) * (h // 2) * 4 * c2 * c
        elems += (h // 2) * (h // 2) * c2 + h * h * c                       # transpose conv
        elems += 3 * h * h * c                                              # bridge: two reads + write
        flops += 2 * mfma_conv_flops(1, h, h, c, c, 3)
        elems += 2 * h * h * 2 * c
    flops += mfma_conv_flops(1, tile, tile, filters[0], nout, 1)
    elems += tile * tile * (filters[0] + nout)
    return flops, elems, first


def disk_labels(rng, n, tile=TILE, disks=60):
    """config 3's label definition: the union of 60 random disks of radius 6-15 px per tile"""
    yy, xx = np.mgrid[0:tile, 0:tile]
    lab = np.zeros((n, tile, tile), np.bool_)
    for i in range(n):
        for _ in range(disks):
            cy, cx, r = rng.integers(0, tile), rng.integers(0, tile), rng.integers(6, 16)
            lab[i] |= (yy - cy) ** 2 + (xx - cx) ** 2 <= r * r
    return lab


def config3_inputs(dev, seed=2, nb=16, tile=TILE):
    """BASELINE config 3's tensors on the device: tiles default_rng(seed).standard_normal, labels = 60 random disks
    per tile as one-hot uint8 (N,H,W,2), weights = ImageWeightMap(w0=10, sigma=5) of the label (sequitr/pipeline.py:
    455-479) computed on the GPU by sq_weightmap_edt_f32 and left there, float32 (N,H,W,1), range [1, ~11]."""
    from sequitr_amd.weightmap import device_weightmaps
    rng = np.random.default_rng(seed)
    x = torch.from_numpy(rng.standard_normal((nb, tile, tile, 1)).astype(np.float32)).to(dev)
    lab = disk_labels(rng, nb, tile)
    onehot = torch.from_numpy(np.stack([~lab, lab], -1).astype(np.uint8)).to(dev)
    wmap = device_weightmaps(lab.astype(np.float32), 10., 5., device=dev)
    return x, onehot, wmap


def disk_image_inputs(dev, seed=2, nb=16, tile=TILE, noise=0.5):
    """config 3's tensors with tiles that CARRY their labels (the matched-IoU training test / job): image = the disk
    label plus N(0, noise) pixel noise, per-tile ImageNorm'd (sequitr/pipeline.py:350-356: (x - mean) / std), labels
    and ImageWeightMap(10, 5) weights as config3_inputs.  Returns (x, onehot, wmap, lab) -- lab a host bool array."""
    from sequitr_amd.weightmap import device_weightmaps
    rng = np.random.default_rng(seed)
    lab = disk_labels(rng, nb, tile)
    img = lab.astype(np.float32) + noise * rng.standard_normal((nb, tile, tile)).astype(np.float32)
    img = (img - img.mean(axis=(1, 2), keepdims=True)) / (1e-99 + img.std(axis=(1, 2), keepdims=True))
    x = torch.from_numpy(img[..., None].astype(np.float32)).to(dev)
    onehot = torch.from_numpy(np.stack([~lab, lab], -1).astype(np.uint8)).to(dev)
    wmap = device_weightmaps(lab.astype(np.float32), 10., 5., device=dev)
    return x, onehot, wmap, lab


def pmc_step_traffic(tag):
    """HBM bytes per step from the committed rocprofv3 PMC passes of this mode (profiles/r*_pmc_<tag>_traffic.json:
    FETCH_SIZE x 2 + WRITE_SIZE summed over every kernel of the step; separate passes), with its source file; or None."""
    import glob
    try:
        fn = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_%s_traffic.json" % tag)))[-1]
        with open(fn) as f:
            return round(json.load(f)["_summary"]["hbm_bytes_per_step"], 1), os.path.relpath(fn, ROOT)
    except Exception:
        return None, None


def cpu_baseline_train(params, x, onehot, wmap):
    """CPU leg of the training line: fp32 torch-CPU autograd of the same graph (oracle/torch_ref.py
    unet_loss_and_grads: forward + weighted softmax-CE + backward) + a numpy Adam update, on 2 of the 16 tiles."""
    from oracle import torch_ref
    from sequitr_amd.networks.unet import init_unet_weights
    w = init_unet_weights(params, seed=0)
    m = {k: np.zeros_like(v) for k, v in w.items()}
    v2 = {k: np.zeros_like(v) for k, v in w.items()}
    nt = 2
    xs, os_, ws = x[:nt].cpu().numpy(), onehot[:nt].cpu().numpy(), wmap[:nt].cpu().numpy()
    state = {"t": 0}

    def one():
        _, g, _ = torch_ref.unet_loss_and_grads(xs, os_, ws, w, params, dtype=torch.float32)
        state["t"] += 1
        t = state["t"]
        lr_t = 0.01 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        for k in w:
            m[k] = 0.9 * m[k] + 0.1 * g[k]
            v2[k] = 0.999 * v2[k] + 0.001 * g[k] * g[k]
            w[k] = (w[k] - lr_t * m[k] / (np.sqrt(v2[k]) + 1e-8)).astype(np.float32)
    threads = best_cpu_threads(one)                  # one pass per candidate thread count; the fastest is timed
    torch.set_num_threads(threads)
    times, warm = timed_passes(one, warm=1, timed=5, budget_s=15.0)
    rate = [nt * TILE * TILE / t / 1e6 for t in times]
    return {"value": round(float(np.median(rate)), 3), "unit": "Mpixels/s", "cores": threads, "kind": "port",
            "cpu_model": cpu_model_name(), "min": round(min(rate), 3), "max": round(max(rate), 3),
            "sample": "fp32 torch-CPU autograd restatement of the same training step (forward + weighted softmax-CE + "
                      "backward + Adam, oracle/torch_ref.py) on 2 of the 16 tiles, %d warm-up + %d timed passes, %d "
                      "threads; stands in for the reference's TF-CPU training (TensorFlow not installable)"
                      % (warm, len(times), threads)}


def main_train(args):
    """BASELINE configs[2] / [3]: U-Net training step (fwd + weighted CE + bwd + flat all-reduce + Adam).
    weak: 16 tiles per GPU per step (config 3 per rank, seeds 2 + rank).  strong: config 4's global batch of 128 tiles
    sharded over the ranks; a rank with more than 16 tiles runs them as micro-batches of 16 with gradient accumulation
    (UNetTrainer.step_accumulate) -- still one all-reduce and one Adam launch per step."""
    world, rank, dist, dev = rank_setup()
    out = train_line(args, world, rank, dist, dev)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def train_line(args, world, rank, dist, dev):
    """the training line as a dict (rank 0; None elsewhere) -- printed by main_train, or attached to the headline as
    `train_bf16` by the default run"""
    from sequitr_amd.train import UNetTrainer
    from sequitr_amd.parallel import shard_range
    nb = 16
    params = {"shape": (TILE, TILE), "num_inputs": 1, "num_outputs": 2, "filters": FILTERS,
              "bridge": "eltwise_mul", "dropout": 0.4, "device": str(dev), "seed": 0, "dtype": args.dtype}
    tr = UNetTrainer(params)                                         # the trainer's default learning rate + warm-up
    if args.scaling == "strong":
        g_tiles = args.global_tiles or 128
        lo, hi = shard_range(g_tiles // nb, rank, world)             # whole micro-batches of 16 per rank
        micro = [config3_inputs(dev, seed=2 + mb, nb=nb) for mb in range(lo, hi)]
        total_tiles = (g_tiles // nb) * nb
        if not micro:
            raise SystemExit("bench.py --mode train --scaling strong: %d ranks but only %d micro-batches of 16" % (world, g_tiles // nb))
    else:
        micro = [config3_inputs(dev, seed=2 + rank, nb=nb)]
        total_tiles = world * nb
    x, onehot, wmap = micro[0]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.graph:                     # hipGraph replay; step() still copies the batch into the static buffers
        tr.capture(x, onehot, wmap, warmup=2)

    def step():
        return tr.step_accumulate(micro)

    for _ in range(args.warmup):
        step()
    barrier()
    tr.allreduce_events = [] if dist is not None else None     # HIP events round the gradient all-reduce alone
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()              # HIP events on the launch stream round the whole step (graph replays + all-reduce + Adam)
        step()
        ev[i][1].record()
    barrier()
    dt = time.perf_counter() - t0
    ar_ms = None
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # the all-reduce as this rank's stream saw it (it includes waiting for the slowest rank to arrive): mean per
        # step, then MAX and MIN over the ranks -- MIN is the nearest thing to the collective's own cost
        mine = float(np.mean([a.elapsed_time(b) for a, b in tr.allreduce_events])) if tr.allreduce_events else 0.0
        hi = torch.tensor([mine], dtype=torch.float64, device=dev)
        lo = hi.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        ar_ms = {"max_over_ranks": round(float(hi.item()), 4), "min_over_ranks": round(float(lo.item()), 4),
                 "bytes": int(tr.gbucket.flat.numel()) * 4, "backend": dist.get_backend(),
                 "what": "HIP events on the launch stream round dist.all_reduce(flat gradient bucket), mean per step; "
                         "max includes waiting for the slowest rank"}
    if rank == 0:
        pix = float(total_tiles) * TILE * TILE * args.steps
        dev_ms = float(np.mean([s.elapsed_time(e) for s, e in ev]))
        f_fwd, e_fwd, f_first = unet_work_per_tile()
        esz = 2 if args.dtype == "bf16" else 4
        tiles_rank = len(micro) * nb
        flops = (3.0 * f_fwd - f_first) * tiles_rank                 # fwd + dgrad + wgrad, no dgrad into the input image
        alg_bytes = 3.0 * e_fwd * esz * tiles_rank                   # layer-by-layer traffic, x3 with backward
        peak_tf = PEAK_BF16_MFMA_TFLOPS if args.dtype == "bf16" else PEAK_F32_MFMA_TFLOPS
        traffic, src = pmc_step_traffic("train_" + args.dtype)
        # SURVEY 8d's COMPULSORY bytes of a step: what must cross HBM whatever the schedule -- the batch's inputs (image
        # f32 + one-hot u8 x 2 + weight map f32 per pixel) and the optimiser state (read p, g, m, v; write p, m, v).  At
        # these bytes the step is compute-bound (AI ~ 10^4 FLOP/B): the schedule bytes above are a property of running
        # layer by layer, not a floor -- both fractions are reported so that `frac` is not read as "at a bound".
        n_par = int(tr.pbucket.numel)
        comp_bytes = float(tiles_rank) * TILE * TILE * (4 + 2 + 4) + 7.0 * 4 * n_par
        out = {"metric": "trained Mpixels/sec on 512x512 tiles (fwd+loss+bwd+allreduce+Adam)",
               "value": round(pix / dt / 1e6, 3), "unit": "Mpixels/s", "n_gpus": world,
               "ranks_seen": dist.get_world_size() if dist is not None else 1, "allreduce_ms": ar_ms,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
               "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": args.dtype,
               "data": "synthetic",
               "config": {"workload": "U-Net training, weightmap-weighted softmax-CE (disk labels, ImageWeightMap(10,5) "
                                      "weights: BASELINE configs[2]), launches of 16 512x512x1 tiles, dropout 0.4, Adam "
                                      "(fp32 master weights); activations " + args.dtype
                                      + ("; hipGraph replay" if args.graph else "; eager launches")
                                      + ("; strong scaling: global batch %d tiles per optimiser step (configs[3]), "
                                         "micro-batches of 16 accumulated per rank" % total_tiles
                                         if args.scaling == "strong" else "; one 16-tile batch per GPU per step"),
                          "tiles_per_step": int(total_tiles), "tiles_this_rank": int(tiles_rank),
                          "loss": float(tr.last_loss.item())},
               # the network is compute-bound at its compulsory bytes (AI ~ 10^4 FLOP/B), so the line's
               # roofline is FLOPs / dense MFMA peak; the layer-by-layer schedule's HBM fraction sits under `hbm_schedule`
               "roofline": {"bound": "mfma",
                            "kernel": "the whole captured step (forward + loss + backward + Adam); achieved = (3 x forward "
                                      "FLOPs - the first layer's dgrad) x tiles / HIP-event time of the step",
                            "achieved": round(flops / (dev_ms * 1e-3) / 1e12, 2), "peak": peak_tf, "unit": "TFLOP/s",
                            "frac": round(flops / (dev_ms * 1e-3) / 1e12 / peak_tf, 4),
                            "traffic": traffic, "traffic_source": src,
                            "flops_per_step": flops, "device_ms_per_step": round(dev_ms, 4),
                            "hbm_schedule": {
                                "achieved": round(alg_bytes / (dev_ms * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                "frac": round(alg_bytes / (dev_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                                "algorithmic_bytes_per_step": alg_bytes,
                                "algorithmic_bytes_are": "the layer-by-layer schedule's activation traffic (SURVEY A.6: "
                                                         "elements read + written by every layer, x 3 for forward + dgrad + "
                                                         "wgrad), NOT a floor"},
                            "compulsory_bytes_per_step": comp_bytes,
                            "compulsory": {"achieved": round(comp_bytes / (dev_ms * 1e-3) / 1e9, 2), "unit": "GB/s",
                                           "frac": round(comp_bytes / (dev_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 5)}}}
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline_train(params, x, onehot, wmap)
            except Exception as e:                              # noqa: BLE001
                out["cpu_baseline"] = {"value": None, "unit": "Mpixels/s", "cores": 0, "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        return out
    return None


def main_centroids(args):
    """mask -> connected components -> centroids (CentroidWriter.write, utils.py:531-578)
    on 32 resident 512x512 uint8 masks (60 random disks of radius 6-15 per tile); CPU leg = the reference's
    scipy loop (oracle/centroids_ref.py) on a few of the same frames."""
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from sequitr_amd import centroids, _lib
    rng = np.random.default_rng(2)
    yy, xx = np.mgrid[0:TILE, 0:TILE]
    mask = np.zeros((BATCH, TILE, TILE), np.uint8)
    for i in range(BATCH):
        for _ in range(60):
            cy, cx, r = rng.integers(0, TILE), rng.integers(0, TILE), rng.integers(6, 16)
            mask[i][(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = 1
    md = torch.from_numpy(mask).to(dev)
    for _ in range(args.warmup):
        frames = centroids.mask_centroids(md)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        frames = centroids.mask_centroids(md)          # includes the D2H of the rows and the host ordering
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    # kernel-only time of the same call (HIP events around the C-ABI launch sequence)
    lib = _lib.load()
    ws = torch.empty(lib.sq_mask_centroids_workspace(BATCH, TILE, TILE) // 4 + 4, dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    out = torch.empty((1 << 18, 5), dtype=torch.float32, device=dev)
    keys = torch.empty((1 << 18,), dtype=torch.int32, device=dev)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(args.steps):
        lib.sq_mask_centroids_u8(md.data_ptr(), BATCH, TILE, TILE, ws.data_ptr(), cnt.data_ptr(), out.data_ptr(),
                                 keys.data_ptr(), 1 << 18, torch.cuda.current_stream().cuda_stream)
    e.record()
    torch.cuda.synchronize()
    kms = s.elapsed_time(e) / args.steps
    npx = BATCH * TILE * TILE
    alg_bytes = npx * (3 * 1 + 3 * 4)                  # mask read by 3 passes, parent written once + read twice
    res = {"metric": "mask -> centroids Mpixels/sec on 512x512 uint8 masks", "value": round(npx / dt / 1e6, 2),
           "unit": "Mpixels/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(dt * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "u8/int32", "data": "synthetic",
           "config": {"workload": "CentroidWriter.write on 32 x 512x512 masks, 60 disks per tile",
                      "components": int(sum(len(f) for f in frames))},
           "roofline": {"bound": "hbm", "achieved": round(alg_bytes / (kms * 1e-3) / 1e9, 1), "peak": 8000.0,
                        "unit": "GB/s", "frac": round(alg_bytes / (kms * 1e-3) / 8e12, 4), "traffic": pmc_step_traffic("centroids")[0], "traffic_source": pmc_step_traffic("centroids")[1],
                        "kernel_ms_per_step": round(kms, 4),
                        "algorithmic_bytes_per_pixel": 15}}
    if not args.no_cpu_baseline:
        from oracle import centroids_ref
        t0 = time.perf_counter()
        ref = centroids_ref.mask_centroids(mask[:4])
        ct = time.perf_counter() - t0
        same = all(np.array_equal(a, b) for a, b in zip(frames[:4], ref))
        res["cpu_baseline"] = {"value": round(4 * TILE * TILE / ct / 1e6, 3), "unit": "Mpixels/s", "cores": 1,
                               "kind": "port", "sample": "the reference's scipy label + center_of_mass loop "
                               "(oracle/centroids_ref.py) on 4 of the 32 frames", "rows_identical": bool(same)}
    print(json.dumps(res))


def main_weightmap(args):
    """ImageWeightMap (pipeline.py:475-479) of 16 resident 512x512 label tiles (config 3's
    label definition: 60 disks of radius 6-15); CPU leg = the reference's scipy expression on 4 tiles."""
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from sequitr_amd import ops as sq_ops
    nb = 16
    rng = np.random.default_rng(2)
    yy, xx = np.mgrid[0:TILE, 0:TILE]
    lab = np.zeros((nb, TILE, TILE), np.float32)
    for i in range(nb):
        for _ in range(60):
            cy, cx, r = rng.integers(0, TILE), rng.integers(0, TILE), rng.integers(6, 16)
            lab[i][(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = 1
    ld = torch.from_numpy(lab).to(dev)
    for _ in range(args.warmup):
        w = sq_ops.weightmap_edt(ld, 10., 5.)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    s.record()
    for _ in range(args.steps):
        w = sq_ops.weightmap_edt(ld, 10., 5.)
    e.record()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    kms = s.elapsed_time(e) / args.steps
    npx = nb * TILE * TILE
    alg = npx * (4 * 3 + 2 + 2 + 4)                    # image read by 3 passes, g written + read, f32 map written
    res = {"metric": "EDT weight maps Mpixels/sec on 512x512 label tiles", "value": round(npx / dt / 1e6, 2),
           "unit": "Mpixels/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(dt * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "int32/f64", "data": "synthetic",
           "config": {"workload": "ImageWeightMap(w0=10, sigma=5) on 16 x 512x512 binary label tiles, f32 maps left in HBM"},
           "roofline": {"bound": "hbm", "achieved": round(alg / (kms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                        "frac": round(alg / (kms * 1e-3) / 8e12, 4), "traffic": pmc_step_traffic("weightmap")[0], "traffic_source": pmc_step_traffic("weightmap")[1],
                        "kernel_ms_per_step": round(kms, 4), "algorithmic_bytes_per_pixel": 20}}
    if not args.no_cpu_baseline:
        from oracle import weightmap_ref
        t0 = time.perf_counter()
        ref = [weightmap_ref.image_weight_map(lab[i]) for i in range(4)]
        ct = time.perf_counter() - t0
        wn = w[:4].cpu().numpy()
        same = all(np.array_equal(wn[i], ref[i][..., 0].astype(np.float32)) for i in range(4))
        res["cpu_baseline"] = {"value": round(4 * TILE * TILE / ct / 1e6, 3), "unit": "Mpixels/s", "cores": 1,
                               "kind": "port", "sample": "the reference's scipy EDT expression "
                               "(oracle/weightmap_ref.py) on 4 of the 16 tiles", "maps_identical_f32": bool(same)}
    print(json.dumps(res))


def main_weightmap2(args):
    """SURVEY 8f rank 2, second half: ImageWeightMap2 (pipeline.py:482-571) of 16 x 512x512 label tiles (config 3's
    labels), labels resident in HBM.  One step = the whole map with NO scipy in it (round 3): boundary points on the
    device (sq_wm2_boundary_points_u8) -> compaction -> D2H (~50 KB per tile) -> the library's exact integer Delaunay on
    a pool of host threads (sq_delaunay2d_batch_i32) -> H2D -> point location by rasterisation, Gaussian, weight
    expression (sq_weightmap2_delaunay_f32).  value = end to end; the roofline object is the device part.  CPU leg:
    the vectorised numpy/scipy restatement (oracle/weightmap_ref.py image_weight_map2: the reference's own per-pixel
    Python loop takes 2.7 s per tile)."""
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from sequitr_amd import ops as sq_ops
    from sequitr_amd.weightmap import device_weightmaps2
    nb = 16
    lab = disk_labels(np.random.default_rng(2), nb)
    img = torch.from_numpy(lab.astype(np.float32)).to(dev)
    tri = os.environ.get("SQ_WM2_TRIANGULATION", "native")

    def step():
        return device_weightmaps2(img, 10., 5., device=dev, triangulation=tri)
    for _ in range(max(args.warmup, 1)):
        w = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        w = step()
    torch.cuda.synchronize()
    step_s = (time.perf_counter() - t0) / args.steps
    # the stages of one step, timed apart (host clock round synchronised stages; the device part also with events)
    def clock(fn, reps=5):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            r = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps, r
    t_pts, idx = clock(lambda: torch.nonzero(sq_ops.wm2_boundary_points(img)))
    counts = torch.bincount(idx[:, 0], minlength=nb).cpu()
    t_d2h, xy = clock(lambda: idx[:, 1:].to(torch.int32).contiguous().cpu())   # nonzero's (P,3) result is column-major
    offsets = torch.zeros(nb + 1, dtype=torch.int64)
    offsets[1:] = torch.cumsum(counts, 0)
    t_tri, (simp, lng) = clock(lambda: sq_ops.delaunay2d_batch(xy, offsets))
    t_h2d, (simp_d, lng_d) = clock(lambda: (simp.to(dev), lng.to(dev)))
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    sq_ops.weightmap_delaunay(img, simp_d, lng_d, 10., 5.)
    s.record()
    for _ in range(10):
        sq_ops.weightmap_delaunay(img, simp_d, lng_d, 10., 5.)
    e.record()
    torch.cuda.synchronize()
    kms = s.elapsed_time(e) / 10
    npx = nb * TILE * TILE
    alg = npx * (8 + 4 + 8 + 8 + 8 * 2 + 4 + 4)                  # cover zero + image + cover read + tmp write/read(+halo) + image + f32 map
    res = {"metric": "Delaunay weight maps (ImageWeightMap2) Mpixels/sec on 512x512 label tiles",
           "value": round(npx / step_s / 1e6, 2), "unit": "Mpixels/s", "n_gpus": 1, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(step_s * 1e3, 3), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "int64/f64", "data": "synthetic",
           "config": {"workload": "ImageWeightMap2(w0=10, sigma=5) on 16 x 512x512 binary label tiles resident in HBM: device "
                                  "boundary points, native exact Delaunay on host threads, device point location + Gaussian "
                                  "+ weights (triangulation = %s)" % tri,
                      "parity": ("PARITY RELAXED (Delaunay tie-breaking): the native triangulation takes another valid diagonal "
                                 "than Qhull in co-circular lattice polygons -- mean |dw| 0.035, 2.3 % of background pixels "
                                 "off by > 0.25 vs the reference vector; SQ_WM2_TRIANGULATION=scipy is the reference-equal "
                                 "(default-in-product) path, ~12 Mpix/s" if tri == "native" else
                                 "reference-equal: scipy morphology + Qhull on the host, exact vs the reference vectors"),
                      "simplices": int((simp[:, 0] >= 0).sum()), "boundary_points": int(xy.shape[0]),
                      "stage_ms": {"boundary_points_and_compaction": round(t_pts * 1e3, 3), "points_d2h": round(t_d2h * 1e3, 3),
                                   "host_triangulation": round(t_tri * 1e3, 3), "simplices_h2d": round(t_h2d * 1e3, 3),
                                   "device_raster_gauss_weights": round(kms, 4)},
                      "host_threads": int(os.environ.get("SQ_HOST_THREADS", 16))},
           "roofline": {"bound": "hbm", "achieved": round(alg / (kms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                        "frac": round(alg / (kms * 1e-3) / 8e12, 4), "traffic": None, "kernel_ms_per_step": round(kms, 4),
                        "algorithmic_bytes_per_pixel": 52}}
    if not args.no_cpu_baseline:
        from oracle import weightmap_ref
        t0 = time.perf_counter()
        ref = [weightmap_ref.image_weight_map2(lab[i].astype(np.float32)) for i in range(2)]
        ct = time.perf_counter() - t0
        wn = w[:2].cpu().numpy()[..., 0]
        bg = ~lab[:2]
        err = np.abs(wn - np.stack([r[..., 0] for r in ref]))[bg]
        res["cpu_baseline"] = {"value": round(2 * TILE * TILE / ct / 1e6, 3), "unit": "Mpixels/s", "cores": 1, "kind": "port",
                               "sample": "vectorised numpy/scipy restatement with scipy's Delaunay + find_simplex "
                                         "(oracle/weightmap_ref.py) on 2 of the 16 tiles; the reference's per-pixel "
                                         "Python loop is ~20x slower still (2.7 s per tile)",
                               "mean_abs_diff_vs_cpu": float(err.mean()), "max_abs_diff_vs_cpu": float(err.max()),
                               "fraction_of_background_pixels_off_by_0.25": round(float((err > 0.25).mean()), 4),
                               "why_they_differ": "co-circular lattice points (Qhull's facet order picks the diagonal) and "
                                                  "pixels on simplex edges (scipy's walk is path dependent, the kernel takes "
                                                  "the longest candidate): tests/test_gpu_weightmap.py states the bounds"}
    print(json.dumps(res))


def main_frontend(args):
    """8 raw uint16 camera frames of 1200x1600 -> ImageNorm -> 512x512 tiles (margin 32) on the
    GPU, and the streamed end-to-end path (host frames -> masks on the host) through the default U-Net."""
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from sequitr_amd.frontend import FrameTiler, segment_frames
    from sequitr_amd.networks.unet import UNet2D
    F, H, W = 8, 1200, 1600
    fr = np.random.default_rng(4).integers(200, 4000, (F, H, W)).astype(np.uint16)
    fd = torch.from_numpy(fr).to(dev)
    tl = FrameTiler((H, W), TILE, 32, device=dev)
    for _ in range(args.warmup):
        tiles = tl.tiles(fd)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(args.steps):
        tiles = tl.tiles(fd)
    e.record()
    torch.cuda.synchronize()
    kms = s.elapsed_time(e) / args.steps
    npx = F * H * W
    alg = npx * 2 * 3 + tiles.numel() * 4                    # frame read by 3 passes (u16), tiles written once (f32)
    net = UNet2D({"shape": (TILE, TILE), "filters": FILTERS, "device": str(dev)}, "infer").initialize()
    segment_frames(net, fr[:4], tile=TILE, margin=32, frames_per_batch=2)
    reps = 4
    t0 = time.perf_counter()
    for _ in range(reps):
        masks = segment_frames(net, fr, tile=TILE, margin=32, frames_per_batch=2)
    dt = (time.perf_counter() - t0) / reps
    res = {"metric": "frames -> normalised tiles Mpixels/sec (1200x1600 uint16 frames, 512x512 tiles)",
           "value": round(npx / (kms * 1e-3) / 1e6, 1), "unit": "Mpixels/s", "n_gpus": 1, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(kms, 4), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "u16/f32", "data": "synthetic",
           "config": {"workload": "ImageNorm + tiling of 8 x 1200x1600 uint16 frames into %d tiles of 512x512" % tiles.shape[0],
                      "tiles_per_frame": tl.tiles_per_frame},
           "roofline": {"bound": "hbm", "achieved": round(alg / (kms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                        "frac": round(alg / (kms * 1e-3) / 8e12, 4), "traffic": pmc_step_traffic("frontend")[0], "traffic_source": pmc_step_traffic("frontend")[1]},
           "end_to_end": {"value": round(npx / dt / 1e6, 2), "unit": "Mpixels/s (frame pixels)",
                          "what": "host uint16 frames -> pinned -> H2D -> norm/tile -> U-Net -> stitch -> host masks, "
                                  "2 frames (24 tiles) per batch, uploads overlapped", "seconds": round(dt, 4),
                          "mask_shape": list(masks.shape)}}
    if not args.no_cpu_baseline:
        from oracle import frontend_ref
        t0 = time.perf_counter()
        ref = frontend_ref.tiles(fr[:2], tl.oy, tl.ox, TILE)
        ct = time.perf_counter() - t0
        same = np.array_equal(tiles[:2 * tl.tiles_per_frame].cpu().numpy(), ref)
        res["cpu_baseline"] = {"value": round(2 * H * W / ct / 1e6, 2), "unit": "Mpixels/s", "cores": 1, "kind": "port",
                               "sample": "numpy ImageNorm + slicing (oracle/frontend_ref.py) on 2 of the 8 frames",
                               "tiles_identical": bool(same)}
    print(json.dumps(res))


def main_infer_bf16(args):
    """Secondary line: the inference batch of config 2 through the bf16 graph (bf16 activations and multiplies, f32
    accumulation), with the IoU of its masks against the fp32 path's -- what BASELINE's metric asks beside the rate.
    Not the headline (config 2 is fp32); op by op, none of the fp32 path's fusions."""
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from sequitr_amd.networks.unet import UNet2D, UNet2DBf16, init_unet_weights
    params = {"shape": (TILE, TILE), "num_inputs": 1, "num_outputs": 2, "filters": FILTERS, "bridge": "eltwise_mul",
              "device": str(dev)}
    weights = o1_weights(params)          # logits of order one: the IoU is about bf16, not about 1e-5-sized near-ties
    x = torch.from_numpy(np.random.default_rng(1).standard_normal((BATCH, TILE, TILE, 1)).astype(np.float32)).to(dev)
    ref = UNet2D(params, "infer")
    ref.load_state_dict(weights)
    mref = ref.predict(x).cpu().numpy()
    lref = ref.logits().float().cpu()
    net = UNet2DBf16(params, "infer")
    net.load_state_dict(weights)
    for _ in range(max(args.warmup, 2)):
        net.predict(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        m = net.predict(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    iou = iou_per_class(m.cpu().numpy(), mref)
    ldiff = float((net.logits().float().cpu() - lref).abs().max())
    print(json.dumps({"metric": "segmented Mpixels/sec on 512x512 tiles; IoU vs reference", "value": round(BATCH * TILE * TILE / dt / 1e6, 3),
                      "unit": "Mpixels/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
                      "ms_per_step": round(dt * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                      "dtype": "bf16", "data": "synthetic",
                      "config": {"workload": "U-Net 2-class inference, batch=32 512x512x1 tiles, bf16 activations / f32 "
                                             "accumulation, per-layer launches (secondary line: config 2 is fp32)",
                                 "iou_vs_fp32_masks_per_class": [round(v, 6) for v in iou],
                                 "logits_max_abs_diff_vs_fp32": float("%.3e" % ldiff),
                                 "logits_max_abs_fp32": float("%.3e" % float(lref.abs().max()))}}))


class WorkCounter(object):
    """Counts the ALGORITHMIC work of everything launched through sequitr_amd.ops while active: FLOPs of the conv
    family (2 x pixels x filter elements for conv2d / dgrad / wgrad, 2 x M x K x N for dense) and bytes of every op
    (tensor arguments + tensor results).  Only the outermost ops call counts (conv2d re-enters itself for its mosaic
    / dense dispatch), so small-image layers count their real shape, not the padded mosaic.  Used on ONE eager
    iteration outside the timed region."""

    def __init__(self, ops_mod):
        self.ops, self.flops, self.bytes, self.calls, self.depth, self.orig = ops_mod, 0.0, 0.0, 0, 0, {}

    @staticmethod
    def _tbytes(obj):
        if isinstance(obj, torch.Tensor):
            return obj.numel() * obj.element_size()
        if isinstance(obj, (tuple, list)):
            return sum(WorkCounter._tbytes(o) for o in obj)
        return 0

    def _conv_flops(self, name, a, kw):
        if name in ("conv2d", "conv_dgrad_raw", "conv2d_dgrad"):
            x, w = a[0], a[1]
            return 2.0 * (x.numel() // x.shape[-1]) * w.numel()
        if name in ("conv_wgrad_raw", "conv2d_wgrad"):
            x, dy, K = a[0], a[1], a[2]
            return 2.0 * (x.numel() // x.shape[-1]) * K * K * x.shape[-1] * dy.shape[-1]
        if name == "dense":
            return 2.0 * a[0].shape[0] * a[1].numel()
        return 0.0

    def __enter__(self):
        import types
        for name, fn in list(vars(self.ops).items()):
            if name.startswith("_") or not isinstance(fn, types.FunctionType) or fn.__module__ != self.ops.__name__:
                continue
            if name in ("invalidate_packs",):
                continue
            self.orig[name] = fn

            def make(name, fn):
                def counted(*a, **kw):
                    top = self.depth == 0
                    self.depth += 1
                    try:
                        out = fn(*a, **kw)
                    finally:
                        self.depth -= 1
                    if top:
                        self.calls += 1
                        self.flops += self._conv_flops(name, a, kw)
                        self.bytes += self._tbytes(a) + self._tbytes(list(kw.values())) + self._tbytes(out)
                    return out
                return counted
            setattr(self.ops, name, make(name, fn))
        # weight gradients queued for a grouped launch (ops_bf16.WgradQueue) never pass through an ops.* call: count them
        # where they are queued -- 2 x pixels x K x K x Cin x Cout flops, the bytes of X, dY and the gradients written
        from sequitr_amd import ops_bf16 as ob
        self._push = ob.WgradQueue.push
        counter = self

        def push(q, x, dy, K, dw, db, *a, **kw):
            counter.calls += 1
            counter.flops += 2.0 * (x.numel() // x.shape[-1]) * K * K * x.shape[-1] * dy.shape[-1]
            counter.bytes += counter._tbytes([x, dy, dw, db])
            return counter._push(q, x, dy, K, dw, db, *a, **kw)
        ob.WgradQueue.push = push
        return self

    def __exit__(self, *exc):
        for name, fn in self.orig.items():
            setattr(self.ops, name, fn)
        from sequitr_amd import ops_bf16 as ob
        ob.WgradQueue.push = self._push


def cpu_baseline_gan(level=6, nb=2):
    """CPU leg of the GAN line: fp32 torch-CPU autograd restatement (oracle/torch_gan_ref.py, its dtype switched to
    float32) of one d_loss gradient + one g_loss gradient evaluation at level 6 on `nb` samples."""
    from oracle import torch_gan_ref as ref
    from sequitr_amd.networks.gan import GenerativeAdverserialNetwork
    g = GenerativeAdverserialNetwork({"num_levels": 7, "batch_size": nb, "device": "cuda:0", "seed": 0}, mode=None)
    g.build()
    sd = g.store.state_dict()
    filters = list(g.filters)
    d_names = [n for n, _ in g.get_training_variables(level)[0]]
    g_names = [n for n, _ in g.get_training_variables(level)[1]]
    del g
    old = ref.DT
    ref.DT = torch.float32
    try:
        W = ref.to_torch(sd)
        rng = np.random.default_rng(3)
        X = torch.as_tensor(rng.standard_normal((nb, 256, 256, 2)), dtype=torch.float32)
        Z = torch.as_tensor(rng.standard_normal((nb, 1, 1, 512)), dtype=torch.float32)
        r = torch.as_tensor(rng.random(nb), dtype=torch.float32)

        def one():
            _, d_loss, _ = ref.losses(X, Z, 1.0, r, W, filters, level)
            torch.autograd.grad(d_loss, [W[n] for n in d_names], allow_unused=True)
            _, _, g_loss = ref.losses(X, Z, 1.0, r, W, filters, level)
            torch.autograd.grad(g_loss, [W[n] for n in g_names], allow_unused=True)
        threads = best_cpu_threads(one)
        torch.set_num_threads(threads)
        times, warm = timed_passes(one, warm=1, timed=5, budget_s=15.0)
    finally:
        ref.DT = old
    rate = [nb * 256 * 256 / t / 1e6 for t in times]
    return {"value": round(float(np.median(rate)), 4), "unit": "Mpixels/s", "cores": threads, "kind": "port",
            "cpu_model": cpu_model_name(), "min": round(min(rate), 4), "max": round(max(rate), 4),
            "sample": "fp32 torch-CPU autograd restatement of the WGAN-GP level-6 losses and both solvers' gradients "
                      "(oracle/torch_gan_ref.py; penalty by create_graph double backward, the discriminator evaluated as "
                      "the reference does) on %d samples, %d warm-up + %d timed passes, %d threads; Adam excluded"
                      % (nb, warm, len(times), threads)}


def main_gan(args):
    """BASELINE configs[4]: progressive WGAN-GP at level 6 (256x256x2), batch 32 per GPU, alpha = 1;
    one iteration = one d_solver + one g_solver (sequitr/networks/gan.py:850-851).  Always weak scaling: the
    reference's batch_size (32, gan.py:431) is per replica, and the minibatch-stdev statistic is per replica."""
    world, rank, dist, dev = rank_setup()
    out = gan_line(args, world, rank, dist, dev)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def gan_line(args, world, rank, dist, dev):
    """the GAN line as a dict (rank 0; None elsewhere) -- printed by main_gan, or attached to the headline as `gan_bf16`"""
    from sequitr_amd import ops as sq_ops
    from sequitr_amd.networks.gan import GenerativeAdverserialNetwork
    nb, level = 32, 6
    g = GenerativeAdverserialNetwork({"num_levels": 7, "batch_size": nb, "repeat_batch": 1, "learning_rate": 1e-3,
                                      "device": str(dev), "seed": 0, "dtype": args.dtype,
                                      "graph": bool(args.graph),
                                      "batch_d": os.environ.get("SQ_GAN_BATCH_D", "1") != "0"}, mode=None)
    g.build()
    g.set_level(level)
    rng = np.random.default_rng(3 + rank)
    X = torch.from_numpy(rng.standard_normal((nb, 256, 256, 2)).astype(np.float32)).to(dev)
    Z = torch.from_numpy(rng.standard_normal((nb, 1, 1, 512)).astype(np.float32)).to(dev)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    shared_g = os.environ.get("SQ_GAN_SHARED_G", "1") != "0"     # A/B: 0 = the two solver calls, each with its own generator pass

    def it():
        if shared_g:
            g.iteration(X, Z, 1.0)                              # d_solver + g_solver on one feed, ONE generator forward pass
        else:
            g.d_solver(X, Z, 1.0)
            g.g_solver(X, Z, 1.0)

    with WorkCounter(sq_ops) as wc:    # the first iteration is eager in every mode (it warms the graphs up): count it
        it()
    for _ in range(max(args.warmup - 1, 2 if args.graph else 0)):
        it()
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()
        it()
        ev[i][1].record()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        samples = float(world) * nb * args.steps
        dev_ms = float(np.mean([s.elapsed_time(e) for s, e in ev]))
        peak_tf = PEAK_BF16_MFMA_TFLOPS if args.dtype in ("bf16", "mixed") else PEAK_F32_MFMA_TFLOPS
        f_mfma = wc.flops / (dev_ms * 1e-3) / 1e12 / peak_tf
        f_hbm = wc.bytes / (dev_ms * 1e-3) / 1e9 / PEAK_HBM_GBS
        traffic, src = pmc_step_traffic("gan_" + args.dtype)
        hbm = {"achieved": round(wc.bytes / (dev_ms * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
               "frac": round(f_hbm, 4)}
        mfma = {"achieved": round(wc.flops / (dev_ms * 1e-3) / 1e12, 2), "peak": peak_tf, "unit": "TFLOP/s",
                "frac": round(f_mfma, 4)}
        rl = dict(mfma)                                                # FLOPs / dense MFMA peak is the line's roofline
        n_par = sum(int(v.numel()) for _, v in g.get_training_variables(level)[0]) + \
            sum(int(v.numel()) for _, v in g.get_training_variables(level)[1])
        comp_bytes = float(X.numel() * 4 + Z.numel() * 4) + 7.0 * 4 * n_par      # inputs + Adam's reads / writes of the two var lists
        hbm["algorithmic_bytes_per_step"] = wc.bytes
        hbm["algorithmic_bytes_are"] = "the op-by-op schedule's tensor traffic, NOT a floor; compulsory bytes beside it"
        rl.update({"bound": "mfma", "hbm_schedule": hbm,
                   "compulsory_bytes_per_step": comp_bytes,
                   "compulsory": {"achieved": round(comp_bytes / (dev_ms * 1e-3) / 1e9, 2), "unit": "GB/s",
                                  "frac": round(comp_bytes / (dev_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 5)},
                   "kernel": "the whole iteration (d_solver + g_solver, two pairs of replayed hipGraphs); algorithmic "
                             "work counted op by op on one eager iteration: conv-family FLOPs (forward, dgrad, wgrad, "
                             "double-backward convs, dense) and the bytes of every operator's tensor arguments + results",
                   "traffic": traffic, "traffic_source": src, "device_ms_per_step": round(dev_ms, 4),
                   "flops_per_step": wc.flops, "ops_per_step": wc.calls,
                   "gflop_per_sample": round(wc.flops / nb / 1e9, 3)})
        out = {"metric": "GAN training Mpixels/sec (256x256 samples; one D step + one G step)",
               "value": round(samples * 256 * 256 / dt / 1e6, 3), "unit": "Mpixels/s",
               "samples_per_s": round(samples / dt, 2), "n_gpus": world,
               "ranks_seen": dist.get_world_size() if dist is not None else 1, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
               "data": "synthetic",
               "config": {"workload": "progressive WGAN-GP level 6 (256x256x2), filters "
                                      "[512,256,128,64,32,16,8], batch 32 per GPU, alpha 1; " +
                                      {"bf16": "bf16 feature maps and feature-map gradients in HBM, bf16-multiply / f32-accumulate "
                                               "convolutions, f32 parameters / images / losses",
                                       "mixed": "f32 tensors, bf16-multiply / f32-accumulate convolutions",
                                       "f32": "fp32"}[args.dtype] +
                                      ("; hipGraph replay" if args.graph else "; eager launches") +
                                      ("; one generator forward pass per iteration, shared by the two solver steps"
                                       if shared_g else "; each solver step runs its own generator pass"),
                          "d_loss": g.last_losses[0], "g_loss": g.last_losses[1]},
               "roofline": rl}
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline_gan()
            except Exception as e:                              # noqa: BLE001
                out["cpu_baseline"] = {"value": None, "unit": "Mpixels/s", "cores": 0, "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        return out
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)       # >= 20 warm-up + >= 100 timed iterations
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", choices=["f32", "bf16", "mixed"], default="bf16", help="--mode train / gan: compute dtype")
    ap.add_argument("--fuse-up", type=int, default=1, help="1 = convT+bridge of up0 inside its first conv (infer mode)")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the PCIe-inclusive rate (infer mode)")
    ap.add_argument("--no-side-lines", action="store_true",
                    help="infer mode: do not append the train_bf16 / gan_bf16 sub-objects (profiling passes)")
    ap.add_argument("--graph", type=int, default=1, help="--mode train / gan: replay the step as hipGraphs (1) or eager (0)")
    ap.add_argument("--fuse", type=int, default=1, help="0 = hook-by-hook kernels, 1 = fused inference kernels")
    ap.add_argument("--mode", choices=["infer", "infer-bf16", "train", "gan", "centroids", "weightmap", "weightmap2", "frontend"], default="infer",
                    help="infer = the headline metric (BASELINE configs[1]); train = configs[2]/[3] "
                         "(U-Net training step, batch 16 per GPU) for DESIGN.md, not the driver's line")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="strong (default for --gpus > 1): a FIXED global batch (--global-tiles; train: 128 tiles = "
                         "config 4) is sharded over the ranks; weak (default for one GPU): every rank its own "
                         "config-sized batch (32 tiles; train: 16)")
    ap.add_argument("--global-tiles", type=int, default=0, help="global batch of the strong mode (default 256; train 128)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if args.mode not in ("infer", "train", "gan"):
            ap.error("--gpus > 1 is for --mode infer | train | gan")
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.scaling is None:
        args.scaling = "strong" if int(os.environ.get("WORLD_SIZE", "1")) > 1 else "weak"
    if args.mode == "train":
        return main_train(args)
    if args.mode == "infer-bf16":
        return main_infer_bf16(args)
    if args.mode == "gan":
        return main_gan(args)
    if args.mode == "centroids":
        return main_centroids(args)
    if args.mode == "weightmap":
        return main_weightmap(args)
    if args.mode == "weightmap2":
        return main_weightmap2(args)
    if args.mode == "frontend":
        return main_frontend(args)

    world, rank, dist, dev = rank_setup()
    ranks_seen = dist.get_world_size() if dist is not None else 1

    from sequitr_amd import ops
    from sequitr_amd.networks.unet import UNet2D, init_unet_weights

    params = {"shape": (TILE, TILE), "num_inputs": 1, "num_outputs": 2, "filters": FILTERS,
              "bridge": "eltwise_mul", "device": str(dev), "fuse": bool(args.fuse), "fuse_up": bool(args.fuse_up)}
    weights = init_unet_weights(params, seed=0)
    net = UNet2D(params, "infer")
    net.load_state_dict(weights)
    # synthetic tiles, already resident in HBM.  weak: every rank its own 32-tile batch (seed 1 + rank).  strong: one
    # FIXED global batch (default 256 tiles = 8 blocks of 32; block b is default_rng(1000 + b)) sharded contiguously
    # over the ranks with parallel.shard_range; a rank walks its shard in launches of <= 32 tiles.  No data-path
    # collective either way: tiles are independent units.
    from sequitr_amd.parallel import shard_range
    if args.scaling == "strong":
        g_tiles = args.global_tiles or 8 * BATCH
        lo, hi = shard_range(g_tiles, rank, world)
        blocks = []
        for blk in range(lo // BATCH, (max(hi, lo + 1) - 1) // BATCH + 1):
            xb = np.random.default_rng(1000 + blk).standard_normal((BATCH, TILE, TILE, 1)).astype(np.float32)
            blocks.append(xb[max(lo - blk * BATCH, 0):min(hi - blk * BATCH, BATCH)])
        xs = np.concatenate(blocks) if blocks else np.zeros((0, TILE, TILE, 1), np.float32)
        total_tiles = g_tiles
    else:
        xs = np.random.default_rng(1 + rank).standard_normal((BATCH, TILE, TILE, 1)).astype(np.float32)
        total_tiles = world * BATCH
    x_all = torch.from_numpy(xs).to(dev)
    chunks = [x_all[i:i + BATCH] for i in range(0, x_all.shape[0], BATCH)]
    x = chunks[0]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        for c in chunks:
            net.predict(c)

    for _ in range(args.warmup):
        step()
    barrier()
    with ConvTimer(ops, expected_groups=(8 * args.steps + 8) * len(chunks)) as ct:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        ct.close()
        barrier()
        dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # PCIe-inclusive rate: EVERY rank streams at the same time (the collective inside is the harness's MAX of the times),
    # so all ranks must take the same branch -- a failure on any rank is agreed on before the line is built
    e2e = None
    if not args.no_end_to_end and x_all.shape[0] > 0:
        try:
            e2e = end_to_end_rate(net, x, dist=dist, world=world)
        except Exception as e:                                  # noqa: BLE001
            if dist is not None:
                raise
            e2e = {"value": None, "what": "failed: %r" % (e,)}
    if rank == 0:
        pix = float(total_tiles) * TILE * TILE * args.steps
        conv_ms, conv_flops, nlaunch = ct.summary()
        achieved = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        out = {
            "metric": "segmented Mpixels/sec on 512x512 tiles; IoU vs reference",
            "value": round(pix / dt / 1e6, 3),
            "unit": "Mpixels/s",
            "n_gpus": world,
            "ranks_seen": ranks_seen,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "U-Net 2-class inference, launches of 32 512x512x1 tiles, fp32 "
                                   "(BASELINE.json configs[1]: filters 16-32-64-128-256, eltwise_mul bridge); "
                                   "logits + uint8 argmax mask; " +
                                   ("strong scaling: a fixed global batch of %d tiles per step, sharded over the ranks"
                                    % total_tiles if args.scaling == "strong" else
                                    "one 32-tile batch per GPU per step"),
                       "tiles_per_step": int(total_tiles), "tiles_this_rank": int(x_all.shape[0]),
                       "parity": "logits and masks bit-exact vs oracle/sq_oracle.c (tests/test_gpu_unet.py); the oracle is parity "
                                 "UNPINNED against the reference (its U-Net leaf ops are abstract, unet.py:326-343)"},
            "roofline": {
                "bound": "mfma",
                "kernel": "the 3x3 implicit-GEMM family on v_mfma_f32_16x16x4_f32, all 17 launches of a step: "
                          "conv_mfma_f32_v2_kernel (levels 1-4) and conv_l0_kernel (the 16-channel level-0 launches: first block + "
                          "pool, convT-fused, 1x1-head-fused; the 16 -> 32 conv of level 1)",
                "achieved": round(achieved, 3),
                "peak": PEAK_F32_MFMA_TFLOPS,
                "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                # HBM bytes PER STEP over the conv launches (PMC) with the algorithmic bytes per step (in + out of
                # every conv launch) beside it; per-launch average kept under its own, labelled key
                "traffic": (pmc_conv_traffic() or {}).get("per_step"),
                "traffic_is": "HBM bytes per step (all conv_mfma / conv_l0 launches of one 32-tile pass), rocprofv3 PMC",
                "algorithmic_bytes_per_step": (pmc_conv_traffic() or {}).get("algorithmic_per_step"),
                "traffic_per_launch_avg": (pmc_conv_traffic() or {}).get("per_launch"),
                "compulsory_bytes_per_step": float(x_all.shape[0]) * TILE * TILE * (4 + 8 + 1) + 4.0 * 1744994,
                "mfma_busy_pmc": pmc_mfma_busy(),
                # PMC counters cannot be read inside the timed run: these come from committed rocprofv3 passes
                "pmc_source": {"traffic": pmc_conv_traffic.source, "mfma_busy_pmc": pmc_mfma_busy.source},
                "launches_per_step": nlaunch // max(1, args.steps),
                "kernel_ms_per_step": round(conv_ms / max(1, args.steps), 4),
                "flops_per_step": conv_flops / max(1, args.steps),
            },
        }
        # the two side measurements must never cost the driver its line: a failure is reported in place
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(weights, params, gpu_net=net)
            except Exception as e:                              # noqa: BLE001
                out["cpu_baseline"] = {"value": None, "unit": "Mpixels/s", "cores": 0, "kind": "port",
                                       "sample": "failed: %r" % (e,)}
            try:                                                # IoU / logits parity on logits of order one
                out["parity_o1"] = parity_on_o1_fixture(params, dev)
            except Exception as e:                              # noqa: BLE001
                out["parity_o1"] = {"failed": repr(e)}
        if e2e is not None:
            out["end_to_end"] = e2e
        if world == 1 and not args.no_side_lines:
            # BASELINE configs[2] and [4] in the driver's ONE command: the bf16 training step and
            # the bf16 GAN iteration, 20 timed steps each after capture, each with its own roofline + cpu_baseline.
            # Same guard as above: a side measurement never costs the headline.
            del net, x_all, chunks, x
            torch.cuda.empty_cache()
            side = argparse.Namespace(**vars(args))
            side.steps, side.warmup, side.dtype, side.graph, side.scaling, side.global_tiles = 20, 5, "bf16", 1, "weak", 0
            for key, fn in (("train_bf16", train_line), ("gan_bf16", gan_line)):
                try:
                    out[key] = fn(side, 1, 0, None, dev)
                except Exception as e:                          # noqa: BLE001
                    out[key] = {"value": None, "failed": repr(e)}
                torch.cuda.empty_cache()
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
