"""The driver's contract with bench.py (task statement "Maintain bench.py", tier section 4): flags, ONE JSON line,
the metric / roofline / cpu_baseline objects and the algorithmic-work figure of SURVEY 8(d)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_flags_and_work_figures_cpu():
    import bench
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], stdout=subprocess.PIPE, text=True).stdout
    for flag in ("--gpus", "--steps", "--warmup", "--mode", "--dtype"):
        assert flag in out, flag
    # 19,285,409,792 FLOP per 512x512 tile for filters 16..256, 2 classes, eltwise bridge
    f, t = bench.FILTERS, bench.TILE
    flops, cin = 0.0, 1
    for i, c in enumerate(f):                                               # encoder: conv1, conv2 per level
        h = t >> i
        flops += bench.mfma_conv_flops(1, h, h, cin, c, 3) + bench.mfma_conv_flops(1, h, h, c, c, 3)
        cin = c
    for i in reversed(range(len(f) - 1)):                                   # decoder: convT 2x2, conv1, conv2
        h = t >> i
        flops += 2.0 * (h // 2) * (h // 2) * 4 * f[i + 1] * f[i] + 2 * bench.mfma_conv_flops(1, h, h, f[i], f[i], 3)
    flops += bench.mfma_conv_flops(1, t, t, f[0], 2, 1)
    assert flops == 19285409792.0
    assert bench.PEAK_F32_MFMA_TFLOPS == 157.3
    a = np.array([[0, 1], [1, 1]])
    assert bench.iou_per_class(a, a) == [1.0, 1.0] and bench.iou_per_class(a, 1 - a) == [0.0, 0.0]
    # the committed PMC summary the line's roofline.traffic comes from
    t = bench.pmc_conv_traffic()
    assert t is None or (t["per_step"] > 1e9 and abs(t["per_step"] - t["per_launch"] * t["launches_per_step"]) < 1e-3 * t["per_step"])


def test_hwinfo_counts_gpus_without_a_runtime(tmp_path):
    """launchers (bench.py --gpus N, the job server) count GPUs from render nodes / KFD topology / *_VISIBLE_DEVICES,
    never through torch.cuda: fake trees stand in for /dev/dri and /sys/class/kfd."""
    from sequitr_amd import hwinfo
    dri, kfd = tmp_path / "dri", tmp_path / "nodes"
    dri.mkdir(), kfd.mkdir()
    for i in range(3):
        (dri / ("renderD%d" % (128 + i))).write_text("")
    (dri / "card0").write_text("")
    for i, simd in enumerate((0, 0, 256, 256, 256, 256)):                  # two CPU nodes, four GPUs
        (kfd / str(i)).mkdir()
        (kfd / str(i) / "properties").write_text("cpu_cores_count 0\nsimd_count %d\n" % simd)
    assert hwinfo.kfd_gpu_nodes(str(kfd)) == [2, 3, 4, 5]
    assert hwinfo.count_gpus({}, str(kfd), str(dri)) == 3                  # the container was handed 3 render nodes
    assert hwinfo.count_gpus({"ROCR_VISIBLE_DEVICES": "0,1"}, str(kfd), str(dri)) == 2
    assert hwinfo.count_gpus({"HIP_VISIBLE_DEVICES": ""}, str(kfd), str(dri)) == 0
    assert hwinfo.count_gpus({}, str(tmp_path / "none"), str(tmp_path / "none")) is None
    src = open(os.path.join(ROOT, "bench.py")).read()
    launcher = src[src.index("def launch_ranks"):src.index("def mfma_conv_flops")]
    code = [l.split("#")[0] for l in launcher.split('"""')[2].splitlines()]            # body without docstring / comments
    assert not any("torch.cuda" in l for l in code)


def test_gpus_n_launcher_fails_loudly_without_gpus():
    """`python bench.py --gpus N` starts N ranks itself.  Here there is no GPU: with the RCCL
    backend the launcher refuses before starting anything; in the gloo rehearsal mode the children start, fail at
    their first HIP call, and the parent must come back non-zero (never hang in a barrier, never print a line)."""
    b = os.path.join(ROOT, "bench.py")
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("box has GPUs: covered by the gpu-marked launcher test")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, b, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=dict(env, SQ_BENCH_BACKEND="nccl"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 2 and "GPU(s) visible" in r.stderr and not r.stdout.strip()
    if torch.cuda.device_count() == 0:
        r = subprocess.run([sys.executable, b, "--gpus", "2", "--steps", "1", "--warmup", "0"],
                           env=dict(env, SQ_BENCH_BACKEND="gloo"), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           text=True, timeout=300)
        assert r.returncode != 0 and not r.stdout.strip()


def test_work_figures_of_the_training_line():
    import bench
    f, e, first = bench.unet_work_per_tile()
    assert f == 19285409792.0 and e == 108003328 and first == 2.0 * 37748736        # SURVEY A.6 totals
    lab = bench.disk_labels(np.random.default_rng(2), 2, tile=128, disks=10)
    assert lab.shape == (2, 128, 128) and lab.dtype == np.bool_ and 0.01 < lab.mean() < 0.9


@pytest.mark.gpu
def test_gpus_2_launches_two_ranks_and_reports_strong_scaling():
    """one-GPU box: SQ_BENCH_BACKEND=gloo puts both ranks on GPU 0 -- it rehearses the N-rank control flow (spawn,
    barriers, max-over-ranks time, rank 0's single line), not the interconnect."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["SQ_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--global-tiles", "64"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["tiles_per_step"] == 64
    assert d["config"]["tiles_this_rank"] == 32 and d["steps"] == 2
    assert abs(d["value"] - 64 * 512 * 512 / (d["ms_per_step"] * 1e-3) / 1e6) <= 1e-3 * d["value"]
    assert "cpu_baseline" not in d and d["roofline"]["frac"] > 0.2
    # the multi-rank line explains itself: the world size as the process group saw it, and the
    # PCIe-inclusive rate with EVERY rank streaming at once (max-over-ranks time)
    assert d["ranks_seen"] == 2
    e = d["end_to_end"]
    assert e["ranks_streaming"] == 2 and e["masks_equal_predict"] is True
    assert e["value"] > 0 and e["from_pageable"]["value"] > 0 and e["serial"]["value"] > 0


@pytest.mark.gpu
def test_gpus_2_training_line_times_the_allreduce_apart():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["SQ_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "train", "--gpus", "2", "--steps", "3",
                        "--warmup", "1", "--global-tiles", "32"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["scaling"] == "strong" and d["config"]["tiles_per_step"] == 32
    ar = d["allreduce_ms"]
    assert ar["bytes"] >= 4 * 1744994 and ar["backend"] == "gloo"
    assert 0 < ar["min_over_ranks"] <= ar["max_over_ranks"] < d["ms_per_step"]
    rl = d["roofline"]
    assert rl["bound"] == "mfma" and rl["peak"] == 2500.0 and abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-3
    assert rl["hbm_schedule"]["frac"] > rl["frac"]              # the schedule's HBM fraction sits under its own key


@pytest.mark.gpu
def test_default_line_has_every_contract_field():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                                            # exactly ONE JSON line on stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert d["unit"] == "Mpixels/s" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 32 * 512 * 512 / (d["ms_per_step"] * 1e-3) / 1e6) <= 1e-3 * d["value"]
    rl = d["roofline"]
    assert rl["bound"] == "mfma" and rl["unit"] == "TFLOP/s" and rl["peak"] == 157.3
    assert abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-3 and 0.3 < rl["frac"] < 1.0
    assert rl["traffic"] is None or rl["traffic"] > 1e8
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    assert min(cb["iou_gpu_vs_cpu_per_class"]) >= 0.9999                     # oneDNN sums in another order: near-ties may flip
