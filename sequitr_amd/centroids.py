"""Mask -> connected components -> centroids on the GPU, behind the reference's
``CentroidWriter`` interface (sequitr/utils.py:479-578).

``mask_centroids(mask)`` is the device path: one C-ABI call (sq_mask_centroids_u8: union-find labelling
of every class at once + exact integer centre-of-mass sums), then the rows are put into the reference's
order -- frame, class ascending, scipy label order = first pixel in raster order -- from the keys the
kernel returns.  Results equal the reference's scipy loop bit for bit (tests/test_gpu_centroids.py).
There is no CPU path: the mask must live in GPU memory (UNet2D.predict returns it there).
"""
import logging
import os

import numpy as np
import torch

from . import _lib

logger = logging.getLogger('worker_process')

_MAX_OUT = 1 << 18


def mask_centroids(mask, as_numpy=True):
    """mask: uint8 class labels on the GPU, planar (N,H,W) or volumetric (N,D0,D1,D2) -- for volumes pass the
    array as CentroidWriter.write sees it after its swapaxes(1,-1) (utils.py:521).  Returns a list of N
    (k_i, 5) float32 arrays [frame, x, y, z, class] (planar: x = row centre, y = column centre, z = 0;
    volumetric: centres along D0, D1, D2), ordered as CentroidWriter.write orders them."""
    if not isinstance(mask, torch.Tensor):
        raise TypeError("mask must be a torch.Tensor in GPU memory")
    if not mask.is_cuda:
        raise _lib.SequitrHipError("mask must live in GPU memory (no CPU fallback exists)")
    if mask.dtype != torch.uint8 or mask.dim() not in (3, 4) or not mask.is_contiguous():
        raise ValueError("mask must be a contiguous (N,H,W) or (N,D0,D1,D2) uint8 tensor")
    volumetric = mask.dim() == 4
    N = mask.shape[0]
    planes = mask.shape[1] if volumetric else 1
    H, W = mask.shape[-2], mask.shape[-1]
    lib = _lib.load()
    nbytes = lib.sq_mask_centroids_workspace(N * planes, H, W)
    if nbytes < 0:
        raise ValueError("mask %s is too large for one call" % (tuple(mask.shape),))
    dev = mask.device
    ws = torch.empty((nbytes + 15) // 16 * 4, dtype=torch.int32, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    max_out = _MAX_OUT
    while True:
        out = torch.empty((max_out, 5), dtype=torch.float32, device=dev)
        keys = torch.empty((max_out,), dtype=torch.int32, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        if volumetric:
            _lib.check(lib.sq_volume_centroids_u8(mask.data_ptr(), N, planes, H, W, ws.data_ptr(), count.data_ptr(),
                                                  out.data_ptr(), keys.data_ptr(), max_out, st), "sq_volume_centroids_u8")
        else:
            _lib.check(lib.sq_mask_centroids_u8(mask.data_ptr(), N, H, W, ws.data_ptr(), count.data_ptr(), out.data_ptr(),
                                                keys.data_ptr(), max_out, st), "sq_mask_centroids_u8")
        n = int(count.item())
        if n <= max_out:
            break
        max_out = n                                             # more components than room: once more
    rows = out[:n].cpu().numpy()
    k = keys[:n].cpu().numpy()
    order = np.lexsort((k, rows[:, 4], rows[:, 0]))            # frame, then class, then first pixel
    rows = rows[order]
    bounds = np.searchsorted(rows[:, 0], np.arange(N + 1))
    return [rows[bounds[i]:bounds[i + 1]] for i in range(N)]


class CentroidWriter(object):
    """sequitr/utils.py:479-578.  ``write(segmented)`` takes the (N,H,W) segmentation (GPU uint8 tensor, or
    a numpy array that is uploaded) and stores frames/frame_<i>/coords = (k,5) float32 per frame.  The
    reference writes HDF5 through h5py; when h5py is not importable the same keys go into an ``.npz``."""

    def __init__(self, filename=None):
        if not isinstance(filename, str):
            raise TypeError('Filename must be specified as a string')
        pth, f = os.path.split(filename)
        if pth and not os.path.exists(pth):
            raise IOError('Destination path {0:s} doesn\'t exist'.format(pth))
        try:
            import h5py
        except ImportError:
            h5py = None
        self._h5py = h5py
        base = os.path.splitext(filename)[0]
        self.filename = base + ('.hdf5' if h5py is not None else '.npz')
        self._frames = {}
        self._hdf = None
        if h5py is not None:
            logger.info('Opening HDF file: {0:s}'.format(self.filename))
            self._hdf = h5py.File(self.filename, 'w')
            self._hdf.create_group('frames')

    def write(self, segmented, device=None):
        if isinstance(segmented, np.ndarray):
            dev = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
            segmented = torch.from_numpy(np.ascontiguousarray(segmented, dtype=np.uint8)).to(dev)
        if segmented.dim() == 4:
            im_type = "Volumetric"                               # default input is N,Z,X,Y (utils.py:519-521)
            segmented = segmented.transpose(1, 3).contiguous()
        elif segmented.dim() == 3:
            im_type = "Image"
        else:
            logger.error("Incorrect image data shape.")
            raise ValueError("Incorrect image data shape.")
        frames = mask_centroids(segmented)
        for i, coords in enumerate(frames):
            if i % 100 == 0:
                logger.info('Written out {0:d} of {1:d} frames ({2:s})...'.format(i, len(frames), im_type))
            self.add_frame(i, coords)
        return frames

    def add_frame(self, i, coords):
        """store one frame's (k,5) rows as frames/frame_<i>/coords (utils.py:569-578)"""
        if self._hdf is not None:
            grp = self._hdf['frames'].create_group('frame_' + str(i))
            grp.create_dataset('coords', data=coords, dtype='float32')
        else:
            self._frames['frames/frame_' + str(i) + '/coords'] = coords

    def close(self):
        if self._hdf is not None:
            logger.info('Closing HDF file.')
            self._hdf.close()
            self._hdf = None
        elif self._frames is not None:
            np.savez(self.filename, **self._frames)
            self._frames = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
