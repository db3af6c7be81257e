"""Mask -> centroids on the CPU -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates CentroidWriter.write (sequitr/utils.py:505-578) for planar (N,H,W) input with the SAME
third-party calls the reference makes (scipy.ndimage label + center_of_mass, which are importable
here): per frame, classes = sorted unique values > 0; per class label(out == c) with scipy's default
4-connectivity; center_of_mass(out, matrix, labels); rows [frame, x, y, 0, class] as float32,
classes concatenated in ascending order (utils.py:553-566).  The reference stores each frame's rows
as the HDF5 dataset frames/frame_<i>/coords (utils.py:569-578).
"""
import numpy as np
from scipy.ndimage import label, center_of_mass


def _cartesian(coords, volumetric):
    if volumetric:                                              # utils.py:514-516
        x, y, z = zip(*coords)
        return x, y, z
    x, y = zip(*coords)                                         # utils.py:524-527
    return x, y, [0.0] * len(x)


def frame_centroids(out, i):
    classes = [x for x in np.unique(out) if x > 0]                     # utils.py:541
    this_frame = []
    for c in classes:
        matrix, _ = label(out == c)                                    # utils.py:547
        labels = [l for l in np.unique(matrix) if l > 0]
        coords = center_of_mass(out, matrix, labels)                   # utils.py:550
        if len(coords) < 1:
            continue
        x, y, z = _cartesian(coords, out.ndim == 3)
        this_class = np.zeros((len(x), 5), dtype='float32')
        this_class[:, 0] = i
        this_class[:, 1] = x
        this_class[:, 2] = y
        this_class[:, 3] = z
        this_class[:, 4] = c
        this_frame.append(this_class)
    if this_frame:
        return np.concatenate(this_frame, axis=0)
    return np.zeros((0, 5), np.float32)


def mask_centroids(segmented):
    """segmented (N,H,W), or volumetric (N,Z,X,Y) as CentroidWriter.write receives it -> list of N arrays
    (k_i, 5) float32.  Volumes are axis-swapped exactly as the reference does (utils.py:519-521)."""
    segmented = np.asarray(segmented)
    if segmented.ndim == 4:
        segmented = np.swapaxes(segmented, 1, -1)
    elif segmented.ndim != 3:
        raise ValueError("Incorrect image data shape.")
    return [frame_centroids(segmented[i], i) for i in range(segmented.shape[0])]
