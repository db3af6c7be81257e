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


# ---- ImageWeightMap2 (sequitr/pipeline.py:482-571): host Delaunay + per-pixel work on the GPU -----------------------------
def _wm2_check(lab, ref, w0=10., sigma=5., ref_is_f32=False, window=None):
    """GPU map vs (a) the rasterising CPU restatement of the SAME rule (oracle/weightmap_ref.py: exact emulation, so
    1e-12: it pins the kernels) and (b) the reference-generated vector: equal to 1e-12 (1e-6 for a float32 vector)
    wherever the reference's answer is determined by the geometry, i.e. at every pixel whose 9x9 Gaussian window holds
    no background pixel lying on a simplex edge / vertex (there scipy's find_simplex returns whichever incident simplex
    its walk reaches first and the kernel takes the one with the longest edge); elsewhere within the map's range w0."""
    from scipy.ndimage import maximum_filter
    from sequitr_amd.weightmap import device_weightmaps2
    got = device_weightmaps2(lab[None], w0, sigma, device="cuda:0", dtype=torch.float64,
                             triangulation="scipy").cpu().numpy()[0]
    emu, count = weightmap_ref.image_weight_map2_raster(lab, w0, sigma)
    assert got.shape == emu.shape and np.abs(got - emu).max() <= 1e-12, np.abs(got - emu).max()
    tie = (count >= 2) & (lab == 0)
    clean = maximum_filter(tie.astype(np.uint8), size=9) == 0
    g, r, c = got[..., 0], ref, clean
    if window is not None:
        g, c = g[window], c[window]
    err = np.abs(g - r)
    tol = 1e-6 if ref_is_f32 else 1e-12
    assert err[c].max() <= tol, err[c].max()
    assert err.max() <= w0 + 1e-6
    return float(tie.mean()), float(c.mean()), float(err.max()), float((err > 1e-6).mean())


def test_weightmap2_reference_vectors_64px_and_512px():
    for s in (0, 1, 2):
        _wm2_check(G["wm_in_%d" % s], G["wm2_out_%d" % s][..., 0])
    lab = G["wm_in_512"].astype(np.float32)
    tie, clean, emax, efrac = _wm2_check(lab, G["wm2_out_512_centre"], ref_is_f32=True,
                                         window=(slice(128, 384), slice(128, 384)))
    # at the benchmark size the lattice ties are a few per cent of the pixels and most of the map is determined
    assert tie < 0.08 and clean > 0.5 and efrac < 0.25, (tie, clean, emax, efrac)
    # float32 maps for the training step: the float64 map rounded once
    from sequitr_amd.weightmap import device_weightmaps2
    w32 = device_weightmaps2(lab[None], 10., 5., device="cuda:0", triangulation="scipy")
    w64 = device_weightmaps2(lab[None], 10., 5., device="cuda:0", dtype=torch.float64, triangulation="scipy")
    assert w32.dtype == torch.float32 and tuple(w32.shape) == (1, 512, 512, 1) and torch.equal(w32, w64.float())


def test_weightmap2_native_triangulation_no_scipy_in_the_path():
    """Round 3: ImageWeightMap2 with the library's own triangulation (opt-in: triangulation='native';
    the default is the reference-equal scipy path, pinned exactly by the tests above).
      * the boundary-point kernel equals scipy's morphology (pipeline.py:516-528) bit for bit, also at the tile border;
      * the device map equals the CPU rasterisation of the SAME native triangulation to 1e-12 (pins the kernels);
      * against the reference-generated vectors: the native Delaunay triangulation is the reference's wherever the
        triangulation is unique; among co-circular lattice points Qhull's choice is an artefact of its facet order, so
        a pixel inside such a quadrilateral can get the other diagonal.  Measured and asserted on the 512x512 vector:
        mean |dw| <= 0.05 and at most 3 % of the background pixels off by more than 0.25 (range of the map: 1 .. 11);
        scipy's own joggled run ('QJ') of the same points differs from its default run by mean 0.021 / 1.4 %
        (profiles/r03_wm2_notes.txt) -- the bound is the tie-breaking, not this implementation."""
    from sequitr_amd.weightmap import device_weightmaps2
    from sequitr_amd import ops
    labs = [G["wm_in_%d" % s] for s in (0, 1, 2)]
    edge = np.zeros((64, 64), np.float32)
    edge[0:5, 10:30] = 1
    edge[30:40, 58:64] = 1
    edge[61:64, 0:3] = 1                                                   # objects cut by the tile border
    for lab in labs + [edge]:
        pts = ops.wm2_boundary_points(dev(lab[None].astype(np.float32))).cpu().numpy()[0]
        assert np.array_equal(pts.astype(bool), weightmap_ref.boundary_points(lab))
    big = G["wm_in_512"].astype(np.float32)
    pts = ops.wm2_boundary_points(dev(big[None])).cpu().numpy()[0]
    assert np.array_equal(pts.astype(bool), weightmap_ref.boundary_points(big))
    # kernels pinned: device map == CPU rasterisation of the same (native) triangulation
    for lab in labs:
        P = np.column_stack(np.where(weightmap_ref.boundary_points(lab))).astype(np.int32)
        simp, _ = ops.delaunay2d_batch(torch.from_numpy(P), torch.tensor([0, len(P)], dtype=torch.int64), compact=True)
        emu, _ = weightmap_ref.image_weight_map2_raster(lab, 10., 5., vertices=simp.numpy()[:, 1:].reshape(-1, 3, 2))
        got = device_weightmaps2(lab[None], 10., 5., device="cuda:0", dtype=torch.float64, triangulation="native").cpu().numpy()[0]
        assert np.abs(got - emu).max() <= 1e-12
    # against the reference-generated vector at the benchmark size
    got = device_weightmaps2(big[None], 10., 5., device="cuda:0", dtype=torch.float64, triangulation="native").cpu().numpy()[0, 128:384, 128:384, 0]
    bg = big[128:384, 128:384] == 0
    err = np.abs(got - G["wm2_out_512_centre"].astype(np.float64))[bg]
    print("native WM2 vs reference: mean %.4f  frac > 0.25: %.4f  frac > 1e-5: %.4f  max %.3f"
          % (err.mean(), (err > 0.25).mean(), (err > 1e-5).mean(), err.max()))
    # observed on MI355X (round 3): mean 0.0354, 2.32 % > 0.25, max 6.52 (one other diagonal next to a cell)
    assert err.mean() <= 0.04 and (err > 0.25).mean() <= 0.026 and err.max() <= 6.6
    # a batch: every tile equals the tile run alone; tiles with fewer than three boundary points are refused loudly
    stack = np.stack([big, np.roll(big, 37, axis=1), big[::-1].copy()])
    wb = device_weightmaps2(stack, 10., 5., device="cuda:0", dtype=torch.float64, triangulation="native").cpu().numpy()
    for k in range(3):
        assert np.array_equal(wb[k], device_weightmaps2(stack[k:k + 1], 10., 5., device="cuda:0", dtype=torch.float64,
                                                        triangulation="native").cpu().numpy()[0])
    with pytest.raises(ValueError, match="three boundary points"):
        device_weightmaps2(np.zeros((1, 64, 64), np.float32), 10., 5., device="cuda:0", triangulation="native")


def test_create_weightmaps_gpu_methods_write_the_reference_layout(tmp_path):
    """create_weightmaps (weightmap.py:171-205) with both GPU methods: folder weights_w0-.._sigma-.., file
    <stem>_weights.tif, float32 -- 'edt' = ImageWeightMap, 'delaunay_gpu' = ImageWeightMap2 on the reference's (scipy)
    triangulation, 'delaunay_gpu_native' = the same on the library's own -- against the host restatements."""
    from sequitr_amd import weightmap as wmod
    from sequitr_amd import pipeline
    lab = (G["wm_in_512"][128:384, 128:384] > 0).astype(np.uint8)
    d = tmp_path / "set1" / "label"
    d.mkdir(parents=True)
    wmod.imsave(str(d / "img_0001_label.tif"), lab)
    for method, pipe in (("edt", pipeline.ImageWeightMap(10., 5.)), ("delaunay_gpu", "scipy"), ("delaunay_gpu_native", None)):
        files = wmod.create_weightmaps(str(tmp_path), ["set1"], w0=10., sigma=5., method=method)
        assert files == [str(tmp_path / "set1" / "weights_w0-10.00_sigma-5.00" / "img_0001_weights.tif")]
        got = wmod.imread(files[0])
        assert got.dtype == np.float32 and got.shape == (256, 256)
        if pipe == "scipy":                                     # the default: the reference's own triangulation
            want = weightmap_ref.image_weight_map2_raster(lab.astype(np.float32), 10., 5.)[0][..., 0].astype(np.float32)
        elif pipe is not None:
            want = np.squeeze(pipe(lab.astype(np.float32)[..., None])).astype(np.float32)
        else:
            from sequitr_amd import ops
            P = np.column_stack(np.where(weightmap_ref.boundary_points(lab))).astype(np.int32)
            simp, _ = ops.delaunay2d_batch(torch.from_numpy(P), torch.tensor([0, len(P)], dtype=torch.int64), compact=True)
            want = weightmap_ref.image_weight_map2_raster(lab.astype(np.float32), 10., 5.,
                                                          vertices=simp.numpy()[:, 1:].reshape(-1, 3, 2))[0][..., 0].astype(np.float32)
        assert np.abs(got - want).max() <= 1e-6, method
        os.remove(files[0])
