#!/usr/bin/env python3
"""Generates tests/golden/pipeline_golden.npz by importing the REFERENCE's
sequitr/pipeline.py in the build container (never on the GPU box; the reference does
not travel).  Recipe from SURVEY.md 8c: stub `skimage.transform` in sys.modules (only
`rotate`/`resize` come from it, pipeline.py:29) and inject the Python-2 builtins the
module relies on (`xrange`, list-returning `zip`) into its namespace -- no file edits.

The fixture is data only: seeded inputs and the arrays the reference returned.

Run:  PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_pipeline_golden.py
"""
import builtins
import io
import json
import os
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference/sequitr"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pipeline_golden.npz")


def disks(seed, size, n, rmin, rmax):
    """union of n random disks -> binary float32 label (BASELINE.md config 3 generator)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:size, 0:size]
    img = np.zeros((size, size), np.float32)
    for _ in range(n):
        cy, cx = rng.integers(0, size, 2)
        r = rng.integers(rmin, rmax + 1)
        img[(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = 1.0
    return img


def main():
    sys.dont_write_bytecode = True
    stub = types.ModuleType("skimage")
    tr = types.ModuleType("skimage.transform")
    tr.rotate = tr.resize = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError("skimage absent"))
    stub.transform = tr
    sys.modules["skimage"], sys.modules["skimage.transform"] = stub, tr
    sys.path.insert(0, REF)
    import pipeline as ref                                     # the reference module
    ref.xrange = range
    ref.zip = lambda *a: list(builtins.zip(*a))

    out = {}
    for seed in range(3):
        lab = disks(seed, 64, 8, 3, 8)
        out["wm_in_%d" % seed] = lab
        out["wm1_out_%d" % seed] = ref.ImageWeightMap(w0=10., sigma=5.)(lab.copy())
        out["wm2_out_%d" % seed] = ref.ImageWeightMap2(w0=10., sigma=5.)(lab.copy())
        out["wm1b_out_%d" % seed] = ref.ImageWeightMap(w0=30., sigma=3.)(lab.copy())
    lab = disks(7, 512, 60, 6, 15)
    out["wm_in_512"] = lab.astype(np.uint8)
    out["wm1_out_512"] = ref.ImageWeightMap(w0=10., sigma=5.)(lab.copy()).astype(np.float32)
    # ImageWeightMap2 of the same 512x512 label (2.7 s of per-pixel Python in the reference); the fixture keeps the
    # central 256x256 window of the float32 map (the corners are the 1024-sentinel plateau) to stay small
    out["wm2_out_512_centre"] = ref.ImageWeightMap2(w0=10., sigma=5.)(lab.copy())[128:384, 128:384, 0].astype(np.float32)

    rng = np.random.default_rng(11)
    img = (rng.standard_normal((48, 40)) * 30 + 100).astype(np.float32)
    img[5, 7] += 500.0                                         # a hot pixel for ImageOutliers
    out["img_in"] = img
    out["norm_out"] = ref.ImageNorm()(img.copy())
    out["blur_out"] = ref.ImageBlur(sigma=1.5).pipe(img.copy()[..., np.newaxis].astype("float32"))
    out["outliers_out"] = ref.ImageOutliers(sigma=2, threshold=50.)(img.copy())
    out["bgsub_out"] = ref.ImageBGSubtract()(img.copy())
    flip = ref.ImageFlip()
    for i in range(len(flip)):
        out["flip_out_%d" % i] = flip(img.copy())
        flip.update()
    multi = rng.standard_normal((32, 32, 2)).astype(np.float32)
    out["img2_in"] = multi
    out["norm2_out"] = ref.ImageNorm()(multi.copy())

    # a chained pipeline, its multiplicity and its JSON form
    p = ref.ImagePipeline([ref.ImageOutliers(sigma=2, threshold=50.), ref.ImageNorm(), ref.ImageFlip()])
    out["chain_len"] = np.array(len(p))
    out["chain_out_0"] = p(img.copy())
    p.update()
    out["chain_out_1"] = p(img.copy())
    ref.inspect.getargspec = lambda f: ref.inspect.getfullargspec(f)     # py3 name of the same call
    with tempfile.TemporaryDirectory() as d:
        fn = os.path.join(d, "pipe.json")
        p.save(fn)
        out["chain_json"] = np.array(open(fn).read())
    np.random.seed(5)
    samp = ref.ImageSample(samples=3, ROI_size=(16, 16))
    samp.im_size = img[..., np.newaxis].shape
    out["sample_out"] = samp(img.copy())
    out["sample_coords"] = np.array(samp.coords)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
