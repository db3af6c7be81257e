#!/usr/bin/env python3
"""Ablation builds of the f32 conv kernel (timing only -- results are wrong by construction): where does the
gap between the loop skeleton (91-95 % of the MFMA peak, tools/mfma_probe.hip) and the real kernel go?
  python tools/conv_ablation.py build     # here (hipcc cross-compiles): tools/_exp/<variant>/libsequitr_hip.so
  python tools/conv_ablation.py run       # on the GPU box: times the level-0/1/2 layers with every variant
"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "sequitr_amd", "csrc", "sq_conv_f32_v2.hip")
WSRC = os.path.join(ROOT, "sequitr_amd", "csrc", "sq_conv_wgrad_f32.hip")
WGRAD_VARIANTS = {
    "wg_base": [],
    "wg_noload": [("raw_buffer_load_b128(xrsrc, inb ? (unsigned)(xbase + xrel[sl]) : OOB, 0, 0)",
                   "raw_buffer_load_b128(xrsrc, inb ? OOB : OOB, 0, 0)"),
                  ("raw_buffer_load_b128(yrsrc, inb ? (unsigned)(ybase + yrel[sl]) : OOB, 0, 0)",
                   "raw_buffer_load_b128(yrsrc, inb ? OOB : OOB, 0, 0)")],
}
VARIANTS = {
    "base": [],
    "nostore": [("const unsigned off = ok ? (unsigned)((((n * H + gy) * W + gx) * Cout + co) * 4) : OOB;",
                 "const unsigned off = OOB; (void)ok;")],
    "noload": [("const unsigned off = inb ? (unsigned)(base + xrel[sl]) : OOB;", "const unsigned off = OOB; (void)inb; (void)base;")],
}
VARIANTS["noload_nostore"] = VARIANTS["nostore"] + VARIANTS["noload"]
VARIANTS["store_lane_order"] = [("#define SQ_STORE_PERMUTE 0", "#define SQ_STORE_PERMUTE 1")]
VARIANTS["tiles_contiguous"] = [("#define SQ_TILE_INTERLEAVE 1", "#define SQ_TILE_INTERLEAVE 0")]
# convoy hypothesis: persistent blocks that share a CU start together and stay in phase (all staging, then all
# MFMA); delay residency slot s by s * k * 64 clocks so their phases interleave
for _k in (24, 48, 96):
    VARIANTS["stagger%d" % _k] = [("    if (t_begin >= t_end) return;\n",
                                   "    if (t_begin >= t_end) return;\n"
                                   "    for (int s_ = (blockIdx.x >> 8) %% C::OCC; s_ > 0; --s_) __builtin_amdgcn_s_sleep(%d);\n" % _k)]
    # residency slots are not known from blockIdx: a hashed delay desynchronises whatever shares a CU
    VARIANTS["hstagger%d" % _k] = [("    if (t_begin >= t_end) return;\n",
                                    "    if (t_begin >= t_end) return;\n"
                                    "    for (int s_ = (int)((blockIdx.x * 2654435761u) >> 30); s_ > 0; --s_) __builtin_amdgcn_s_sleep(%d);\n" % _k)]


def build():
    _build(SRC, "sq_conv_f32_v2.o", VARIANTS)
    _build(WSRC, "sq_conv_wgrad_f32.o", WGRAD_VARIANTS)


def _build(src_file, obj_name, variants):
    objs = [o for o in glob.glob(os.path.join(ROOT, "sequitr_amd", "_build", "*.o")) if not o.endswith(obj_name)]
    src = open(src_file).read()
    for name, patches in variants.items():
        d = os.path.join(ROOT, "tools", "_exp", name)
        os.makedirs(d, exist_ok=True)
        s = src
        for a, b in patches:
            assert a in s, (name, a)
            s = s.replace(a, b)
        fn = os.path.join(ROOT, "sequitr_amd", "csrc", "_exp_%s.hip" % name)
        open(fn, "w").write(s)
        try:
            subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
                                   "-Wno-unused-value", "-c", fn, "-o", os.path.join(d, "v2.o")])
        finally:
            os.remove(fn)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o",
                               os.path.join(d, "libsequitr_hip.so"), os.path.join(d, "v2.o")] + objs)
        print("built", name)


# per-phase timestamps (s_memtime) of the first 32 blocks x 4 waves: where a tile's time goes.
#   T0 loop top (next item's loads issued) | T1 MFMA phase done | T2 barrier passed | T3 commit done (loads landed)
#   T4 epilogue issued | T5 second barrier passed.     python tools/conv_ablation.py timeline
_STAMP = ("if (dbg_ && vb < 32 && lane == 0 && it < 64) dbg_[(((size_t)vb * 4 + wv) * 64 + it) * 6 + %d] = "
          "(long long)__builtin_readcyclecounter();\n")
VARIANTS["timeline"] = [
    ("    SqConvEpi epi = {};\n    epi.store_y = 1;\n    if (Cin % 16 == 0)\n        return K == 3 ? dispatch_bn<3, 16>(x, w, bias, y, N, H, W, Cin, Cout, wscale, act, epi, st)",
     "    SqConvEpi epi = {};\n    epi.store_y = 1;\n    if (const char *e_ = getenv(\"SQ_DBG_PTR\")) epi.first_b = (const float *)strtoull(e_, 0, 16);\n"
     "    if (Cin % 16 == 0)\n        return K == 3 ? dispatch_bn<3, 16>(x, w, bias, y, N, H, W, Cin, Cout, wscale, act, epi, st)"),
    ('#include "sq_common.h"\n', '#include "sq_common.h"\n#include <stdlib.h>\n'),
    ("    int tile = t_begin, chunk = 0;\n    for (int it = 0; it < nitems; ++it) {\n",
     "    long long *dbg_ = MODE == 0 ? (long long *)epi.first_b : nullptr;\n"
     "    int tile = t_begin, chunk = 0;\n    for (int it = 0; it < nitems; ++it) {\n"),
    ("        if (has_next) issue(ntile, nchk * KC, restage_w, true);\n",
     "        if (has_next) issue(ntile, nchk * KC, restage_w, true);\n        " + _STAMP % 0),
    ("            __builtin_amdgcn_s_setprio(3);\n        }\n",
     "            __builtin_amdgcn_s_setprio(3);\n        }\n        " + _STAMP % 1),
    ("        __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0); expcnt / lgkmcnt untouched\n        if (has_next) {\n"
     "            __syncthreads();            // every wave is done reading this item's LDS image\n            commit(restage_w);\n",
     "        __builtin_amdgcn_s_waitcnt(0x0F70);\n        " + _STAMP % 2 +
     "        if (has_next) {\n            __syncthreads();\n            commit(restage_w);\n            __builtin_amdgcn_s_waitcnt(0);\n            " + _STAMP % 3),
    ("        if (chunk == nchunk - 1) epilogue(tile);\n        if (has_next) __syncthreads();\n",
     "        if (chunk == nchunk - 1) epilogue(tile);\n        " + _STAMP % 4 + "        if (has_next) __syncthreads();\n        " + _STAMP % 5),
]


def timeline():
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from sequitr_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "_exp", "timeline", "libsequitr_hip.so")
    dbg = torch.zeros((32 * 4 * 64 * 6,), dtype=torch.int64, device="cuda:0")
    os.environ["SQ_DBG_PTR"] = "%x" % dbg.data_ptr()
    from sequitr_amd import ops
    for (n, h, ci, co) in [(32, 512, 16, 16), (32, 256, 32, 32), (32, 128, 64, 64)]:
        x = torch.randn(n, h, h, ci, device="cuda:0")
        w = torch.randn(3, 3, ci, co, device="cuda:0") * 0.05
        b = torch.zeros(co, device="cuda:0")
        for _ in range(WARM):
            ops.conv2d(x, w, b, act="relu")
        dbg.zero_()
        ops.conv2d(x, w, b, act="relu")                          # no host synchronisation before it: the chip stays loaded
        torch.cuda.synchronize()
        t = dbg.cpu().numpy().reshape(32, 4, 64, 6).astype(np.float64)
        nit = int((t[0, 0, :, 0] > 0).sum())
        t = t[:, :, 2:nit - 1, :]                                # steady state: drop the first two and the last item
        names = ["mfma phase", "wait vmcnt(0)", "barrier+commit", "epilogue", "barrier 2"]
        d = [np.median(t[..., k + 1] - t[..., k]) for k in range(5)]
        period = np.median(t[:, :, 1:, 0] - t[:, :, :-1, 0])
        print("C=%d %dx%d: items/block %d, period %.0f clk | " % (ci, h, h, nit, period) +
              " | ".join("%s %.0f" % (nm, v) for nm, v in zip(names, d)), flush=True)


WARM, REPS = int(os.environ.get("SQ_ABL_WARM", "300")), int(os.environ.get("SQ_ABL_REPS", "100"))


def run_one(name):
    sys.path.insert(0, ROOT)
    import torch
    from sequitr_amd import _lib
    if name != "prod":                                     # "prod" = the library as built in sequitr_amd/_build
        _lib.LIB_PATH = os.path.join(ROOT, "tools", "_exp", name, "libsequitr_hip.so")
    from sequitr_amd import ops
    dev = "cuda:0"
    out = []
    if name.startswith("wg_"):
        for (n, h, ci, co) in [(16, 512, 16, 16), (16, 256, 32, 32), (16, 128, 64, 64), (16, 32, 256, 256)]:
            x = torch.randn(n, h, h, ci, device=dev)
            dy = torch.randn(n, h, h, co, device=dev)
            for _ in range(3):
                ops.conv2d_wgrad(x, dy, 3)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                ops.conv2d_wgrad(x, dy, 3)
            e.record()
            torch.cuda.synchronize()
            us = s.elapsed_time(e) / 20 * 1e3
            out.append("%6.1f us %5.1f TF" % (us, 2.0 * n * h * h * 9 * ci * co / us / 1e6))
        print("%-16s %s" % (name, " | ".join(out)), flush=True)
        return
    for (n, h, ci, co) in [(32, 512, 16, 16), (32, 256, 32, 32), (32, 128, 64, 64), (32, 32, 256, 256)]:
        x = torch.randn(n, h, h, ci, device=dev)
        w = torch.randn(3, 3, ci, co, device=dev) * 0.05
        b = torch.zeros(co, device=dev)
        # the clocks take tens of milliseconds to settle after the (HBM-bound) tensor initialisation: a 3 + 20 launch
        # measurement reads level 0 at 446 us where the settled rate is ~350 us (tools/clock_probe.py)
        for _ in range(WARM):
            ops.conv2d(x, w, b, act="relu")
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(REPS):
            ops.conv2d(x, w, b, act="relu")
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) / REPS * 1e3
        tf = 2.0 * n * h * h * 9 * ci * co / us / 1e6
        out.append("%6.1f us %5.1f TF" % (us, tf))
    print("%-16s %s" % (name, " | ".join(out)), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    elif sys.argv[1] == "timeline":
        timeline()
    elif sys.argv[1] == "run":
        names = sys.argv[2:] or (["prod"] + list(VARIANTS) + list(WGRAD_VARIANTS))
        for v in names:   # one process per variant: the library is loaded once
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "one", v])
    else:
        run_one(sys.argv[2])
