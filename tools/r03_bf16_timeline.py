#!/usr/bin/env python3
"""Per-phase timeline (s_memtime stamps) of conv_mfma_bf16_kernel's work items on the deep layers (round 3): where do
the ~2.5 us per 32-channel chunk go when the MFMA work is 0.48 us?
  python tools/r03_bf16_timeline.py build     # here: tools/_exp/bf16tl/libsequitr_hip.so (a patched copy of sq_conv_bf16.hip)
  python tools/r03_bf16_timeline.py run       # on the GPU box
Stamps per item (wave-0 lane 0 of the first 64 blocks): T0 next item's loads issued | T1 MFMA phase done | T2 vmcnt(0)
passed | T3 barrier + commit done | T4 epilogue issued | T5 second barrier passed."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "sequitr_amd", "csrc", "sq_conv_bf16.hip")
OUT = os.path.join(ROOT, "tools", "_exp", "bf16tl")
STAMP = ("if (dbg_ && vb < 64 && lane == 0 && it < 32) dbg_[(((size_t)(vb * gridDim.y + blockIdx.y) * 4 + wv) * 32 + it) * 6 + %d] = "
         "(long long)__builtin_readcyclecounter();\n")
PATCHES = [
    ("    return dispatch_kc<__bf16>(xb, wb, bias, yb, N, H, W, Cin, Cout, K, act, st, gb, drop);\n}",
     "    SqDropEpi d2 = drop;\n    if (const char *e_ = getenv(\"SQ_DBG_PTR\")) d2.gate_f32 = (const float *)strtoull(e_, 0, 16);\n"
     "    return dispatch_kc<__bf16>(xb, wb, bias, yb, N, H, W, Cin, Cout, K, act, st, gb, d2);\n}"),
    ("    for (int it = 0; it < nitems; ++it) {\n        int ntile = tile, nchk = chunk;\n",
     "    long long *dbg_ = (FORM == FORM_PLAIN && !F32IO && gridDim.x * gridDim.y <= 1024) ? (long long *)drop.gate_f32 : nullptr;\n"
     "    for (int it = 0; it < nitems; ++it) {\n        int ntile = tile, nchk = chunk;\n"),
    ("        if (!EARLY && has_next) issue(ntile, nchk, restage_w);\n",
     "        if (!EARLY && has_next) issue(ntile, nchk, restage_w);\n        " + STAMP % 0),
    ("        __builtin_amdgcn_s_setprio(3);\n        // the prefetch has landed",
     "        __builtin_amdgcn_s_setprio(3);\n        " + STAMP % 1 + "        // the prefetch has landed"),
    ("        __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0); expcnt / lgkmcnt untouched\n        if (has_next) {\n            __syncthreads();\n            commit(restage_w);\n        }\n",
     "        __builtin_amdgcn_s_waitcnt(0x0F70);\n        " + STAMP % 2 +
     "        if (has_next) {\n            __syncthreads();\n            commit(restage_w);\n            __builtin_amdgcn_s_waitcnt(0);\n        }\n        " + STAMP % 3),
    ("        if (chunk == nchunk - 1) epilogue(tile);\n        if (has_next) __syncthreads();\n        tile = ntile;",
     "        if (chunk == nchunk - 1) epilogue(tile);\n        " + STAMP % 4 + "        if (has_next) __syncthreads();\n        " + STAMP % 5 + "        tile = ntile;"),
]


DB_PATCHES = [
    PATCHES[0],
    ("        int tile = t_begin, chunk = 0;\n        for (int it = 0; it < nitems; ++it) {\n            const int cur = it & 1;\n",
     "        long long *dbg_ = (FORM == FORM_PLAIN && !F32IO && gridDim.x * gridDim.y <= 1024) ? (long long *)drop.gate_f32 : nullptr;\n"
     "        int tile = t_begin, chunk = 0;\n        for (int it = 0; it < nitems; ++it) {\n            const int cur = it & 1;\n            " + STAMP % 0),
    ("            __builtin_amdgcn_s_setprio(1);\n            bf16x8 fa[2][NR], fb[2][4];\n",
     "            " + STAMP % 1 + "            __builtin_amdgcn_s_setprio(1);\n            bf16x8 fa[2][NR], fb[2][4];\n"),
    ("            step(ti, ci);\n            __builtin_amdgcn_s_setprio(3);\n            if (chunk == nchunk - 1) epilogue(tile);\n            __syncthreads();                                    // the one barrier of an item",
     "            step(ti, ci);\n            __builtin_amdgcn_s_setprio(3);\n            " + STAMP % 2 + "            if (chunk == nchunk - 1) epilogue(tile);\n            " + STAMP % 3 +
     "            __syncthreads();\n            " + STAMP % 4 + "            " + STAMP % 5 + "            // the one barrier of an item"),
]


def build():
    s = open(SRC).read()
    db = os.environ.get("TL_DB", "0") == "1"
    for a, b in (DB_PATCHES if db else PATCHES):
        assert a in s, a
        s = s.replace(a, b, 1)
    os.makedirs(OUT, exist_ok=True)
    fn = os.path.join(ROOT, "sequitr_amd", "csrc", "_exp_bf16tl.hip")
    open(fn, "w").write(s)
    try:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-value",
                               "-c", fn, "-o", os.path.join(OUT, "conv.o")])
    finally:
        os.remove(fn)
    objs = [o for o in glob.glob(os.path.join(ROOT, "sequitr_amd", "_build", "*.o")) if not o.endswith("sq_conv_bf16.o")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o",
                           os.path.join(OUT, "libsequitr_hip.so"), os.path.join(OUT, "conv.o")] + objs)
    print("built", OUT)


def run():
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from sequitr_amd import _lib
    _lib.LIB_PATH = os.path.join(OUT, "libsequitr_hip.so")
    dbg = torch.zeros((1024 * 4 * 32 * 6,), dtype=torch.int64, device="cuda:0")
    os.environ["SQ_DBG_PTR"] = "%x" % dbg.data_ptr()
    from sequitr_amd import ops_bf16 as ob
    N = int(os.environ.get("BENCH_N", 16))
    for (h, ci, co) in [(32, 256, 256), (64, 128, 128), (128, 64, 64), (32, 128, 256)]:
        x = torch.randn(N, h, h, ci, device="cuda:0").to(torch.bfloat16)
        wp = ob.pack_weights(torch.randn(3, 3, ci, co, device="cuda:0") * 0.05)
        b = torch.zeros(co, device="cuda:0")
        for _ in range(200):
            ob.conv2d(x, wp, b, 3, co, act="relu")
        dbg.zero_()
        ob.conv2d(x, wp, b, 3, co, act="relu")
        torch.cuda.synchronize()
        t = dbg.cpu().numpy().reshape(1024, 4, 32, 6).astype(np.float64)
        used = t[:, 0, 0, 0] > 0
        t = t[used]
        nit = int((t[0, 0, :, 0] > 0).sum())
        first, last = t[..., 0, 0].min(), t[:, :, nit - 1, 5].max()
        names = ["mfma phase", "wait vmcnt(0)", "barrier+commit(+lgkm)", "epilogue", "barrier 2"]
        if os.environ.get("TL_DB", "0") == "1":
            names = ["tile offsets", "mfma + staging", "epilogue", "barrier", "-"]
        mid = t[:, :, 1:nit - 1, :] if nit > 2 else t[:, :, :nit, :]
        d = [np.median(mid[..., k + 1] - mid[..., k]) for k in range(5)]
        period = np.median(t[:, :, 1:nit, 0] - t[:, :, :nit - 1, 0]) if nit > 1 else float("nan")
        starts = t[:, 0, 0, 0] - first
        print("%3d->%3d @%3d^2: blocks stamped %d, items/block %d, kernel span %.0f clk (100 MHz counter?), item period %.0f | "
              % (ci, co, h, int(used.sum()), nit, last - first, period) + " | ".join("%s %.0f" % (nm, v) for nm, v in zip(names, d))
              + " | block start spread p50 %.0f p95 %.0f" % (np.median(starts), np.percentile(starts, 95)), flush=True)


if __name__ == "__main__":
    build() if sys.argv[1] == "build" else run()
