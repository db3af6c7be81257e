"""oracle/weightmap_ref.py against the vectors generated from the reference's own pipeline.py
(tests/golden/make_pipeline_golden.py): bit-exact."""
import os

import numpy as np

from oracle import weightmap_ref

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "pipeline_golden.npz"))


def test_image_weight_map_matches_reference_vectors():
    for seed in (0, 1, 2):
        lab = G["wm_in_%d" % seed]
        assert np.array_equal(weightmap_ref.image_weight_map(lab.copy(), 10., 5.), G["wm1_out_%d" % seed])
        assert np.array_equal(weightmap_ref.image_weight_map(lab.copy(), 30., 3.), G["wm1b_out_%d" % seed])
    big = weightmap_ref.image_weight_map(G["wm_in_512"].astype(np.float32))
    assert np.array_equal(big.astype(np.float32), G["wm1_out_512"])


def test_weight_map2_restatements_against_the_reference_vectors():
    """ImageWeightMap2 (pipeline.py:482-571).  (i) the restatement with scipy's own find_simplex reproduces the
    reference-generated vectors wm2_out_* to 1e-12; (ii) the rasterising restatement (the rule of the GPU kernel: on a
    simplex edge / vertex take the largest candidate) equals the reference wherever the answer is determined by the
    geometry -- every pixel whose 9x9 filter window holds no background tie pixel -- and differs elsewhere by at most
    w0 (the map's range); the tie pixels are a few per cent of the image."""
    import os
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "pipeline_golden.npz"))
    from scipy.ndimage import maximum_filter
    for s in (0, 1, 2):
        lab, ref = G["wm_in_%d" % s], G["wm2_out_%d" % s]
        assert np.abs(weightmap_ref.image_weight_map2(lab) - ref).max() <= 1e-12
        got, count = weightmap_ref.image_weight_map2_raster(lab)
        tie = (count >= 2) & (lab == 0)
        clean = maximum_filter(tie.astype(np.uint8), size=9) == 0
        assert np.abs(got - ref)[clean].max() <= 1e-12
        assert np.abs(got - ref).max() <= 10.0 + 1e-9 and 0 < tie.mean() < 0.2          # 64x64 tiles: 10-16 % lattice ties
