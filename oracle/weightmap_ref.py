"""EDT weight map on the CPU -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

ImageWeightMap.pipe restated from sequitr/pipeline.py:475-479 with the same scipy call the reference
makes.  Pinned by the reference-generated vectors wm1_out_* / wm1b_out_* of
tests/golden/pipeline_golden.npz (tests/test_weightmap_oracle.py)."""
import numpy as np
from scipy.ndimage import distance_transform_edt


def image_weight_map(image, w0=10., sigma=5.):
    if image.ndim < 3:                                                         # ImagePipe.__call__, pipeline.py:174-180
        image = image[..., np.newaxis].astype('float32')
    weight_map = distance_transform_edt(1. - image)                            # pipeline.py:476
    weight_map = w0 * (1. - image) * np.exp(-(weight_map * weight_map) / (2. * sigma ** 2 + 1e-99))
    return weight_map + image + 1.                                             # pipeline.py:479


def edt_squared(image):
    """exact integer squared distances (the transform is exact; sqrt is the only rounding)."""
    d = distance_transform_edt(1. - np.asarray(image, np.float64))
    return np.rint(d * d).astype(np.int64)
