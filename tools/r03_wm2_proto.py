#!/usr/bin/env python3
"""Round 3 prototype (CPU, numpy): ImageWeightMap2 WITHOUT a triangulation -- the 'medial chord' form.
For a background pixel p with nearest boundary point s (exact feature transform), the Delaunay triangle that holds
p is a thin one whose long edges join s (or its curve neighbour) to the boundary point t on the far side of the gap:
march from p along s->p until the nearest-site map changes to a site that is not a curve neighbour of s; d = |s - t|.
Compares the resulting weight map with the reference's (sequitr_amd.pipeline.ImageWeightMap2 == reference to 1e-12)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy import ndimage


def sites_of(b):
    cross = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]])
    out = lambda m: np.logical_xor(ndimage.binary_erosion(m, iterations=1, structure=cross), m)
    return np.logical_xor(out(b), out(ndimage.binary_dilation(b, iterations=3, structure=cross)))


def chord_map(b, kmax=96, near=2.9, step=1.0):
    H, W = b.shape
    sites = sites_of(b)
    _, (iy, ix) = ndimage.distance_transform_edt(~sites, return_indices=True)
    py, px = np.where(~b)
    sy, sx = iy[py, px].astype(np.float64), ix[py, px].astype(np.float64)
    vy, vx = py - sy, px - sx
    n = np.sqrt(vy * vy + vx * vx)
    d = np.full(len(py), 1024.0)
    live = n > 0
    uy, ux = np.where(live, vy / np.maximum(n, 1e-9), 0), np.where(live, vx / np.maximum(n, 1e-9), 0)
    done = ~live
    for k in range(1, kmax):
        qy, qx = py + k * step * uy, px + k * step * ux
        ry, rx = np.rint(qy).astype(int), np.rint(qx).astype(int)
        inside = (ry >= 0) & (ry < H) & (rx >= 0) & (rx < W)
        ryc, rxc = np.clip(ry, 0, H - 1), np.clip(rx, 0, W - 1)
        ty, tx = iy[ryc, rxc], ix[ryc, rxc]
        dist = np.sqrt((ty - sy) ** 2 + (tx - sx) ** 2)
        hit = inside & ~done & (dist > near)
        d[hit] = dist[hit]
        done |= hit | ~inside
        if done.all():
            break
    return py, px, d, live


def weightmap_from_d(b, py, px, d, w0=10., sigma=5.):
    wm = np.zeros(b.shape)
    wm[py, px] = d
    wm = ndimage.gaussian_filter(wm, 1.)
    mask = b.astype(np.float64)
    return w0 * (1. - mask) * np.exp(-(wm * wm) / (2. * sigma ** 2 + 1e-99)) + 1. + mask


def report(name, got, ref, sel=None):
    e = np.abs(got - ref)
    if sel is not None:
        e = e[sel]
    print("%-10s mean|e| %.4f  p90 %.3f p99 %.3f max %.3f  frac>0.25 %.4f frac>1 %.4f" % (
        name, e.mean(), np.percentile(e, 90), np.percentile(e, 99), e.max(), (e > 0.25).mean(), (e > 1).mean()), flush=True)


if __name__ == "__main__":
    from sequitr_amd.pipeline import ImageWeightMap2
    G = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "pipeline_golden.npz"))
    near = float(sys.argv[1]) if len(sys.argv) > 1 else 2.9
    for s in range(3):
        b = G["wm_in_%d" % s] > 0
        py, px, d, live = chord_map(b, near=near)
        report("golden%d" % s, weightmap_from_d(b, py, px, d), G["wm2_out_%d" % s][..., 0], sel=~b)
    b = G["wm_in_512"] > 0
    py, px, d, live = chord_map(b, near=near)
    got = weightmap_from_d(b, py, px, d)[128:384, 128:384]
    report("512centre", got, G["wm2_out_512_centre"].astype(np.float64), sel=~b[128:384, 128:384])
