"""CPU check of oracle/centroids_ref.py (restatement of CentroidWriter.write, sequitr/utils.py:531-578)
against hand-computed centres of mass, label order and class order."""
import numpy as np

from oracle import centroids_ref


def test_hand_computed_frame():
    m = np.zeros((2, 8, 10), np.uint8)
    m[0, 1:3, 1:4] = 1            # 2x3 block: centre (1.5, 2.0)
    m[0, 6, 7:10] = 1             # later in raster order: (6.0, 8.0)
    m[0, 0, 9] = 2                # class 2 comes after every class-1 row although it is first in raster order
    m[0, 4, 0] = 1
    m[0, 5, 1] = 1                # diagonal neighbour: NOT connected (4-connectivity)
    out = centroids_ref.mask_centroids(m)
    assert out[1].shape == (0, 5) and out[0].dtype == np.float32
    assert np.array_equal(out[0], np.array([[0, 1.5, 2, 0, 1], [0, 4, 0, 0, 1], [0, 5, 1, 0, 1], [0, 6, 8, 0, 1],
                                            [0, 0, 9, 0, 2]], np.float32))
