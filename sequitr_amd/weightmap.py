"""Offline weight-map creation -- mirror of sequitr/weightmap.py (rows a15-a16 of SURVEY.md 8a).

``ImageLabels`` (weightmap.py:31-73) turns a label image (2-D, or a 3-D stack with one binary plane
per class) into a uint8 class-index map with at most 5 classes; ``create_weightmaps``
(weightmap.py:171-205) walks ``<path>/<folder>/label/*.tif``, computes
``ImageWeightMap2(w0, sigma)`` of the binarised labels and writes float32 TIFFs to
``weights_w0-%2.2f_sigma-%2.2f/`` with the reference's file-name rule.  The reference's own copy of
ImageWeightMap2 in this file is broken; the working class lives in pipeline.py.

TIFF IO uses Pillow (the reference's ``tifffile`` is a git-ignored third-party file that is not in
the tree); ``.npy`` label files are accepted too.
"""
import argparse
import os
import re

import numpy as np

from . import utils
from .pipeline import ImageWeightMap2


def imread(filename):
    if filename.endswith('.npy'):
        return np.load(filename, allow_pickle=False)
    from PIL import Image, ImageSequence
    with Image.open(filename) as im:
        pages = [np.array(p) for p in ImageSequence.Iterator(im)]
    return pages[0] if len(pages) == 1 else np.stack(pages)


def imsave(filename, array):
    if filename.endswith('.npy'):
        np.save(filename, array)
        return
    from PIL import Image
    Image.fromarray(np.ascontiguousarray(array)).save(filename, format='TIFF')


class ImageLabels(object):
    """Label image -> class indices 0..n (weightmap.py:31-73)."""

    def __init__(self, filename, thresh_fn=lambda x: x > 0):
        self._raw_data = filename if isinstance(filename, np.ndarray) else imread(filename)
        assert (self._raw_data.ndim > 1 and self._raw_data.ndim < 4)
        if self._raw_data.ndim == 3:
            l_data = np.zeros(self._raw_data.shape[1:], dtype='uint8')
            for l in range(self._raw_data.shape[0]):
                l_data[thresh_fn(self._raw_data[l, ...])] = l + 1
            raw_labels = list(range(self._raw_data.shape[0] + 1))
        else:
            l_data = thresh_fn(self._raw_data).astype('uint8')
            raw_labels = [0, 1]
        self._outputs = len(raw_labels)
        if self.outputs > 5:
            raise ValueError('More that five output classes!')
        self._labels = l_data

    def labels(self):
        return self._labels

    @property
    def outputs(self):
        return self._outputs


def weights_folder_name(w0, sigma, name_weights_folder=True):
    base = 'weights'
    if name_weights_folder:
        base += '_w0-{0:2.2f}_sigma-{1:2.2f}'.format(w0, sigma)
    return base


def weights_file_name(label_file):
    """'a_b_label.tif' -> 'a_b_weights.tif' (weightmap.py:198-199)."""
    return re.match('([a-zA-Z0-9()]+)_([a-zA-Z0-9()]+_)*', label_file).group(0) + 'weights.tif'


def device_weightmaps(labels, w0=10., sigma=5., device=None):
    """EDT weight maps (ImageWeightMap, pipeline.py:455-479) of a stack of binary label images computed on
    the GPU and LEFT THERE as the (N,H,W,1) float32 `weights` tensor the training step takes -- no TIFF
    round trip.  labels: (N,H,W) array or device tensor, non-zero = cell."""
    import torch
    from . import ops
    if not isinstance(labels, torch.Tensor):
        dev = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        labels = torch.from_numpy(np.ascontiguousarray(np.asarray(labels) > 0, dtype=np.float32)).to(dev)
    else:
        labels = (labels > 0).to(torch.float32).contiguous()
    w = ops.weightmap_edt(labels, w0, sigma, dtype=torch.float32)
    return w.reshape(tuple(w.shape) + (1,))


def boundary_triangulation(label):
    """Host half of ImageWeightMap2 (pipeline.py:514-537), literally: boundary points = erosion outline of the mask
    XOR outline of the 3x-dilated mask (von Neumann element), scipy's Delaunay of them in np.where order.  Returns
    (vertices (S,3,2) int32, longest edge of every simplex (S) float64 -- what edist's max is, pipeline.py:545,559-566)."""
    from scipy import ndimage
    from scipy.spatial import Delaunay
    cross = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]])
    b = np.squeeze(np.asarray(label).astype('bool'))

    def outline(m):
        return np.logical_xor(ndimage.binary_erosion(m, iterations=1, structure=cross), m)

    pts_mask = np.logical_xor(outline(b), outline(ndimage.binary_dilation(b, iterations=3, structure=cross)))
    px, py = np.where(pts_mask)
    tri = Delaunay(np.column_stack((px, py)))
    verts = tri.points[tri.simplices]                                  # (S,3,2), exact integers in float64
    edges = verts - np.roll(verts, -1, axis=1)
    longest = np.sqrt((edges ** 2).sum(-1)).max(-1)
    return verts.astype(np.int32), longest


def device_weightmaps2(labels, w0=10., sigma=5., device=None, dtype=None, triangulation='scipy'):
    """ImageWeightMap2 (pipeline.py:482-571) of a stack of binary label images, returned as the (N,H,W,1) tensor the
    training step takes (float32; dtype=torch.float64 for the reference's own precision), left in HBM.

    triangulation='scipy' (default): the REFERENCE-EQUAL path -- scipy's morphology and Qhull on the host, one tile at
    a time, exactly the calls of pipeline.py:516-537; the per-pixel part on the device.  ~12 Mpix/s.
    triangulation='native' (opt-in, ~1 Gpix/s; PARITY RELAXED at Delaunay ties): no scipy anywhere -- the boundary points are found on the device
    (sq_wm2_boundary_points_u8), compacted there, copied to the host (~50 KB per tile), triangulated by the library's own
    exact integer Delaunay on a pool of host threads (sq_delaunay2d_batch_i32, ~4 ms per 512x512 tile per thread against
    Qhull's 22-38 ms under the GIL), and the simplices go back for the per-pixel part (sq_weightmap2_delaunay_f32:
    point location by rasterisation, scipy-order Gaussian, the reference's float64 expression).
    Which one equals the reference: boundary pixels are lattice points, so a third of a real tile's simplices lie in
    co-circular polygons (2030 of them on the golden 512x512 tile) where the Delaunay triangulation is not unique.
    Qhull ('Qt') fans each such polygon from the vertex it happened to add last; no rule on coordinates or input order
    reproduces that choice (best of six tried: 34 % of the polygons, profiles/r04_wm2_ties.txt), so only 'scipy' gives
    the reference's map there; 'native' takes another valid diagonal: mean |dw| 0.035, 2.3 % of the background pixels
    off by > 0.25 on the reference vector.

    Exact wherever the Delaunay triangulation is unique and one simplex covers the pixel.  On simplex edges / vertices
    the reference's answer depends on the path of scipy's walk (here the largest candidate is taken), and among
    co-circular lattice points on Qhull's facet order (the native triangulation breaks those ties its own way);
    tests/test_gpu_weightmap.py states both bounds against the reference-generated vectors."""
    import torch
    from . import ops
    dev = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
    if triangulation not in ('native', 'scipy'):
        raise ValueError("triangulation must be 'native' or 'scipy'")
    if triangulation == 'native':
        if isinstance(labels, torch.Tensor):
            lab = labels.to(dev)
        else:
            lab = torch.from_numpy(np.ascontiguousarray(np.asarray(labels))).to(dev)
        if lab.dim() == 4 and lab.shape[-1] == 1:
            lab = lab[..., 0]
        img = (lab > 0).to(torch.float32).contiguous()
        N = img.shape[0]
        # (P,3) [tile, row, column] in scan order.  torch.nonzero hands back a COLUMN-major (P,3) tensor: the (row, column) pairs
        # are made contiguous on the device (a 0.8 MB copy there; 0.32 ms per call when it was left to the host) and the per-tile
        # counts are taken there too
        nz = torch.nonzero(ops.wm2_boundary_points(img))
        counts = torch.bincount(nz[:, 0], minlength=N).cpu().numpy()
        xy = nz[:, 1:].to(torch.int32).contiguous().cpu()
        offsets = torch.zeros(N + 1, dtype=torch.int64)
        offsets[1:] = torch.from_numpy(np.cumsum(counts))
        if int(counts.min()) < 3:
            raise ValueError('ImageWeightMap2 needs at least three boundary points per tile (tile %d has %d): the '
                             'reference\'s Delaunay call fails there too' % (int(counts.argmin()), int(counts.min())))
        simp, lng = ops.delaunay2d_batch(xy, offsets)
        simp_d, lng_d = simp.to(dev, non_blocking=True), lng.to(dev, non_blocking=True)
        w = ops.weightmap_delaunay(img, simp_d, lng_d, w0, sigma, dtype=dtype or torch.float32)
        torch.cuda.current_stream(dev).synchronize()            # the pinned staging buffer is reused by the next call
        return w.reshape(tuple(w.shape) + (1,))
    if isinstance(labels, torch.Tensor):
        labels = labels.detach().cpu().numpy()
    lab = np.asarray(labels)
    if lab.ndim == 4 and lab.shape[-1] == 1:
        lab = lab[..., 0]
    lab = lab > 0
    rows, longest = [], []
    for n in range(lab.shape[0]):
        v, l = boundary_triangulation(lab[n])
        rows.append(np.concatenate([np.full((len(v), 1), n, np.int32), v.reshape(len(v), 6)], axis=1))
        longest.append(l)
    simp = torch.from_numpy(np.ascontiguousarray(np.concatenate(rows))).to(dev)
    lng = torch.from_numpy(np.ascontiguousarray(np.concatenate(longest))).to(dev)
    img = torch.from_numpy(np.ascontiguousarray(lab, dtype=np.float32)).to(dev)
    w = ops.weightmap_delaunay(img, simp, lng, w0, sigma, dtype=dtype or torch.float32)
    return w.reshape(tuple(w.shape) + (1,))


def create_weightmaps(path, folders, w0=10., sigma=3., thresh_fn=lambda x: x > 0, name_weights_folder=True,
                      method='delaunay'):
    """ Generate weightmaps for the images using the binary masks; returns the files written.
    method='delaunay' is the reference's ImageWeightMap2 on the host (weightmap.py:181); 'delaunay_gpu' the same map
    (the reference's own scipy triangulation) with its per-pixel part on the GPU (device_weightmaps2);
    'delaunay_gpu_native' the ~80x faster form on the library's triangulation, which differs from the reference at
    Delaunay ties (device_weightmaps2's docstring); 'edt' computes ImageWeightMap on the GPU (sq_weightmap_edt_f32).
    All write the same float32 TIFFs. """
    if method not in ('delaunay', 'delaunay_gpu', 'delaunay_gpu_native', 'edt'):
        raise ValueError("method must be 'delaunay', 'delaunay_gpu', 'delaunay_gpu_native' or 'edt'")
    w_pipe = ImageWeightMap2(w0=w0, sigma=sigma)
    written = []
    for d in folders:
        r_dir = os.path.join(path, d)
        f_labels = sorted(l for l in os.listdir(os.path.join(r_dir, 'label')) if l.endswith('.tif'))
        w_dir = os.path.join(r_dir, weights_folder_name(w0, sigma, name_weights_folder))
        utils.check_and_makedir(w_dir)
        for f in f_labels:
            im_label = ImageLabels(os.path.join(r_dir, 'label', f), thresh_fn=thresh_fn).labels()
            if method == 'edt':
                im_weights = device_weightmaps(im_label[np.newaxis], w0, sigma).cpu().numpy()[0, ..., 0]
            elif method in ('delaunay_gpu', 'delaunay_gpu_native'):
                im_weights = device_weightmaps2(im_label[np.newaxis], w0, sigma, triangulation='native' if method.endswith(
                    'native') else 'scipy').cpu().numpy()[0, ..., 0]
            else:
                im_weights = np.squeeze(w_pipe(im_label.astype('bool')))
            out = os.path.join(w_dir, weights_file_name(f))
            imsave(out, im_weights.astype('float32'))
            written.append(out)
    return written


def main(argv=None):
    p = argparse.ArgumentParser(description='Sequitr: weightmap calculation')
    p.add_argument('-p', '--workdir', default="/media/lowe-sn00/TrainingData/", help='Path to the image data')
    p.add_argument('-f', '--folders', nargs='+', required=True, help='Specify the sub-folders of image data')
    p.add_argument('--w0', type=float, default=30., help='Specify the amplitude')
    p.add_argument('--sigma', type=float, default=3., help='Specify the sigma')
    args = p.parse_args(argv)
    create_weightmaps(args.workdir, args.folders, w0=args.w0, sigma=args.sigma)


if __name__ == '__main__':
    main()
