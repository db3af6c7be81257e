"""GPU EDT weight maps (ImageWeightMap.pipe, sequitr/pipeline.py:475-479) against the
reference-generated vectors in tests/golden/pipeline_golden.npz and against scipy's exact transform."""
import os

import numpy as np
import pytest
import torch

from oracle import weightmap_ref
from sequitr_amd import ops

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "pipeline_golden.npz"))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def ulps64(a, b):
    return np.abs(a.view(np.int64) - b.view(np.int64)).max()


def test_reference_vectors_64px():
    """float64 maps: every operation of the reference's expression in the same order; only exp() may differ
    from numpy's in the last bit (both are < 1 ulp implementations) => <= 2 ulp in float64."""
    labs = np.stack([G["wm_in_%d" % s] for s in (0, 1, 2)]).astype(np.float32)
    for w0, sigma, key in ((10., 5., "wm1_out_%d"), (30., 3., "wm1b_out_%d")):
        got = ops.weightmap_edt(dev(labs), w0, sigma, dtype=torch.float64).cpu().numpy()
        for i in range(3):
            ref = G[key % i][..., 0]
            assert ulps64(got[i], ref) <= 2, (key % i, ulps64(got[i], ref))


def test_reference_vector_512px_float32():
    lab = G["wm_in_512"].astype(np.float32)[None]
    got = ops.weightmap_edt(dev(lab), 10., 5., dtype=torch.float32).cpu().numpy()[0]
    ref = G["wm1_out_512"][..., 0]
    bad = got.view(np.int32) != ref.view(np.int32)
    # a float64 last-bit difference in exp() can flip a float32 rounding only on an exact tie: none expected
    assert bad.sum() == 0, (int(bad.sum()), np.abs(got - ref).max())


@pytest.mark.parametrize("shape,p", [((3, 40, 70), 0.02), ((2, 128, 65), 0.3), ((1, 200, 333), 0.001),
                                     ((4, 17, 9), 0.5), ((1, 1, 130), 0.05), ((1, 90, 1), 0.1)])
def test_squared_distances_are_exact(shape, p):
    rng = np.random.default_rng(shape[1])
    img = (rng.random(shape) < p).astype(np.float32)
    for i in range(shape[0]):
        if img[i].sum() == 0:
            img[i, shape[1] // 2, shape[2] // 2] = 1
    d2 = ops.edt_squared(dev(img)).cpu().numpy()
    for i in range(shape[0]):
        assert np.array_equal(d2[i].astype(np.int64), weightmap_ref.edt_squared(img[i])), i


def test_image_without_any_cell_reproduces_scipy():
    """No feature pixel: scipy's transform measures to index (-1, 0); the reference would write that map."""
    img = np.zeros((2, 12, 20), np.float32)
    img[1, 3, 4] = 1
    got = ops.weightmap_edt(dev(img), 10., 5., dtype=torch.float64).cpu().numpy()
    for i in range(2):
        ref = weightmap_ref.image_weight_map(img[i])[..., 0]
        assert ulps64(got[i], ref) <= 2


def test_full_batch_feeds_the_loss_kernel():
    """BASELINE config 3 shape: 16 x 512 x 512 label tiles -> f32 weights in HBM, used directly as the
    `weights` operand of the loss (no TIFF round trip, weightmap.py:171-205)."""
    rng = np.random.default_rng(2)
    yy, xx = np.mgrid[0:512, 0:512]
    lab = np.zeros((16, 512, 512), np.float32)
    for i in range(16):
        for _ in range(60):
            cy, cx, r = rng.integers(0, 512), rng.integers(0, 512), rng.integers(6, 16)
            lab[i][(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = 1
    w = ops.weightmap_edt(dev(lab), 10., 5.)
    assert w.shape == (16, 512, 512) and w.dtype == torch.float32
    wn = w.cpu().numpy()
    assert wn.min() >= 1.0 and wn.max() <= 11.0 and np.all(wn[lab == 1] == 2.0)
    for i in (0, 9):
        ref = weightmap_ref.image_weight_map(lab[i])[..., 0].astype(np.float32)
        assert np.array_equal(wn[i], ref)
    logits = torch.randn(16, 512, 512, 2, device="cuda:0")
    onehot = dev(np.stack([1 - lab, lab], -1).astype(np.uint8))
    loss, dl = ops.wsoftmax_ce(logits, onehot, w.reshape(16, 512, 512, 1))
    assert np.isfinite(float(loss))
