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


# ---- ImageWeightMap2 (sequitr/pipeline.py:482-571) -----------------------------------------------------------------
def _boundary_delaunay(label):
    """pipeline.py:514-537 literally (von Neumann element, erosion outline XOR outline of the 3x-dilated mask)."""
    from scipy.ndimage import binary_erosion, binary_dilation
    from scipy.spatial import Delaunay
    s = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]])
    b = np.squeeze(np.asarray(label).astype('bool'))
    b_erode_outline = np.logical_xor(binary_erosion(b, iterations=1, structure=s), b)
    b_dilate = binary_dilation(b, iterations=3, structure=s)
    b_dilate_outline = np.logical_xor(binary_erosion(b_dilate, iterations=1, structure=s), b_dilate)
    x, y = np.where(np.logical_xor(b_erode_outline, b_dilate_outline))
    return b, Delaunay(np.column_stack((x, y)))


def _longest_edges(tri):
    v = tri.points[tri.simplices]
    e = v - np.roll(v, -1, axis=1)
    return np.sqrt((e ** 2).sum(-1)).max(-1)                                    # max of edist, pipeline.py:545


def _finish2(b, d, w0, sigma):
    from scipy.ndimage import gaussian_filter
    wm = np.zeros(b.shape + (1,))
    fx, fy = np.where(np.logical_not(b))
    wm[fx, fy, 0] = d[fx, fy]
    mask = b[..., np.newaxis].astype('float32')
    wm = gaussian_filter(wm, 1.)                                                # pipeline.py:548 (sigma fixed at 1)
    return w0 * (1. - mask) * np.exp(-(wm * wm) / (2. * sigma ** 2 + 1e-99)) + 1. + mask


def image_weight_map2(label, w0=10., sigma=5.):
    """the reference's map with scipy's own find_simplex (its choice on simplex edges is walk-path dependent)"""
    b, tri = _boundary_delaunay(label)
    longest = _longest_edges(tri)
    d = np.zeros(b.shape)
    fx, fy = np.where(np.logical_not(b))
    sim = tri.find_simplex(np.column_stack((fx, fy)))
    d[fx, fy] = np.where(sim >= 0, longest[np.maximum(sim, 0)], 1024.)
    return _finish2(b, d, w0, sigma)


def boundary_points(label):
    """the boundary-point mask of pipeline.py:516-528 (what _boundary_delaunay triangulates), for the device kernel"""
    from scipy.ndimage import binary_erosion, binary_dilation
    s = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]])
    b = np.squeeze(np.asarray(label).astype('bool'))
    out = lambda m: np.logical_xor(binary_erosion(m, iterations=1, structure=s), m)
    return np.logical_xor(out(b), out(binary_dilation(b, iterations=3, structure=s)))


def image_weight_map2_raster(label, w0=10., sigma=5., vertices=None):
    """The same map with point location by RASTERISATION and the rule of sq_weightmap2_delaunay_f32: a pixel
    covered by several simplices (it lies on an edge / a vertex) takes the largest longest-edge.  Returns
    (map (H,W,1) float64, cover count (H,W) int: 0 = outside the hull, 1 = the reference's answer is determined,
    >= 2 = tie pixel, where the reference returns whichever incident simplex scipy's walk reaches first).
    vertices (S,3,2) int: rasterise THIS triangulation of the boundary points instead of scipy's (the library's own
    exact Delaunay breaks co-circular ties differently from Qhull)."""
    if vertices is None:
        b, tri = _boundary_delaunay(label)
        longest = _longest_edges(tri)
        V = tri.points[tri.simplices].astype(np.int64)
    else:
        b = np.squeeze(np.asarray(label).astype('bool'))
        V = np.asarray(vertices).astype(np.int64)
        e = (V - np.roll(V, -1, axis=1)).astype(np.float64)
        longest = np.sqrt((e ** 2).sum(-1)).max(-1)
    H, W = b.shape
    best = np.zeros((H, W))
    count = np.zeros((H, W), np.int32)
    for k in range(len(V)):
        (x0, y0), (x1, y1), (x2, y2) = V[k]
        xa, xb, ya, yb = min(x0, x1, x2), max(x0, x1, x2), min(y0, y1, y2), max(y0, y1, y2)
        X, Y = np.mgrid[xa:xb + 1, ya:yb + 1]
        e0 = (x1 - x0) * (Y - y0) - (y1 - y0) * (X - x0)
        e1 = (x2 - x1) * (Y - y1) - (y2 - y1) * (X - x1)
        e2 = (x0 - x2) * (Y - y2) - (y0 - y2) * (X - x2)
        ins = ((e0 >= 0) & (e1 >= 0) & (e2 >= 0)) | ((e0 <= 0) & (e1 <= 0) & (e2 <= 0))
        sub = best[xa:xb + 1, ya:yb + 1]
        np.maximum(sub, np.where(ins, longest[k], 0.0), out=sub)
        count[xa:xb + 1, ya:yb + 1] += ins
    d = np.where(count > 0, best, 1024.)
    return _finish2(b, d, w0, sigma), count
