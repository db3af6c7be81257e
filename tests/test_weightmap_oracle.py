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
