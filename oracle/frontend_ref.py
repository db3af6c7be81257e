"""Tile front end on the CPU -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

image_norm restates ImagePipe.__call__ + ImageNorm.pipe (sequitr/pipeline.py:174-180, 350-356) with numpy,
i.e. with numpy's own float32 reduction order; pinned by the reference-generated vectors img_in / norm_out of
tests/golden/pipeline_golden.npz.  tiles / stitch restate the build-defined tiling of sequitr_amd/frontend.py
with plain slicing."""
import numpy as np


def image_norm(image):
    if image.ndim < 3:
        image = image[..., np.newaxis].astype('float32')       # pipeline.py:178-179
    for chnl in range(image.shape[-1]):                        # pipeline.py:351-355
        image[..., chnl] = (image[..., chnl] - np.mean(image[..., chnl])) / (1e-99 + np.std(image[..., chnl]))
    return image


def tiles(frames, oy, ox, T, normalise=True):
    out = []
    for f in frames:
        g = image_norm(np.array(f, dtype='float'))[..., 0] if normalise else np.asarray(f, np.float32)
        for y in oy:
            for x in ox:
                out.append(g[y:y + T, x:x + T])
    return np.stack(out)[..., None].astype(np.float32)


def stitch(tile_masks, oy, ox, ymap, xmap, H, W):
    TR, TC = len(oy), len(ox)
    F = tile_masks.shape[0] // (TR * TC)
    out = np.empty((F, H, W), tile_masks.dtype)
    ty, ly, tx, lx = ymap >> 16, ymap & 0xffff, xmap >> 16, xmap & 0xffff
    for f in range(F):
        t = (f * TR + ty[:, None]) * TC + tx[None, :]
        out[f] = tile_masks[t, ly[:, None], lx[None, :]]
    return out
