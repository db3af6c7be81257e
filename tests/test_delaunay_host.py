"""CPU: the library's host-side exact Delaunay triangulation (sq_delaunay2d_batch_i32, sequitr_amd/csrc/sq_delaunay.hip),
which replaces scipy.spatial.Delaunay in ImageWeightMap2 (sequitr/pipeline.py:531-537).  A host function of the C-ABI:
it runs without a GPU."""
import numpy as np
import pytest
import torch
from scipy.spatial import ConvexHull, Delaunay

from oracle import weightmap_ref
from sequitr_amd import ops

G = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "pipeline_golden.npz"))


def triangulate(sets):
    xy = torch.from_numpy(np.ascontiguousarray(np.concatenate(sets).astype(np.int32)))
    off = torch.zeros(len(sets) + 1, dtype=torch.int64)
    off[1:] = torch.cumsum(torch.tensor([len(s) for s in sets]), 0)
    simp, lng = ops.delaunay2d_batch(xy, off, compact=True)        # without the per-tile padding rows
    return simp.numpy(), lng.numpy()


def canon(rows):
    return set(tuple(sorted([(r[1], r[2]), (r[3], r[4]), (r[5], r[6])])) for r in rows.tolist())


def test_general_position_points_give_scipys_triangulation():
    rng = np.random.default_rng(0)
    for n, hi in ((3000, 30000), (500, 32767), (40, 1000)):
        pts = np.unique(rng.integers(0, hi, (n, 2)), axis=0)
        simp, lng = triangulate([pts])
        ref = set(tuple(sorted(map(tuple, pts[s].tolist()))) for s in Delaunay(pts).simplices)
        assert canon(simp) == ref
        v = simp[:, 1:].reshape(-1, 3, 2).astype(np.float64)
        e = v - np.roll(v, -1, axis=1)
        assert np.array_equal(lng, np.sqrt((e ** 2).sum(-1)).max(-1))          # the longest edge, as numpy computes it


def test_lattice_points_valid_delaunay_covering_the_hull():
    """boundary pixels are lattice points: co-circular quadruples make the triangulation non-unique, so the check is
    the definition -- same simplex count as scipy, the triangles tile the convex hull exactly, and no point lies
    strictly inside any circumcircle (exact integer arithmetic)."""
    lab = G["wm_in_512"] > 0
    P = np.column_stack(np.where(weightmap_ref.boundary_points(lab))).astype(np.int32)
    simp, _ = triangulate([P, P[::-1].copy(), P])                              # a batch; insertion order must not matter ...
    assert set(np.unique(simp[:, 0]).tolist()) == {0, 1, 2}
    a, b, c = (canon(simp[simp[:, 0] == k]) for k in range(3))
    assert a == b == c                                                         # ... nor the thread a tile runs on
    tri = simp[simp[:, 0] == 0][:, 1:].reshape(-1, 3, 2).astype(object)
    assert len(tri) == len(Delaunay(P).simplices)
    area2 = sum(abs((t[1][0] - t[0][0]) * (t[2][1] - t[0][1]) - (t[1][1] - t[0][1]) * (t[2][0] - t[0][0])) for t in tri)
    assert area2 == round(2 * ConvexHull(P).volume) and all(
        (t[1][0] - t[0][0]) * (t[2][1] - t[0][1]) - (t[1][1] - t[0][1]) * (t[2][0] - t[0][0]) != 0 for t in tri)
    px, py = P[:, 0].astype(object), P[:, 1].astype(object)
    for k in np.random.default_rng(1).choice(len(tri), 200, replace=False):
        (ax, ay), (bx, by), (cx, cy) = tri[k]
        if (bx - ax) * (cy - ay) - (by - ay) * (cx - ax) < 0:
            (bx, by), (cx, cy) = (cx, cy), (bx, by)
        a0, a1, b0, b1, c0, c1 = ax - px, ay - py, bx - px, by - py, cx - px, cy - py
        det = (a0 * (b1 * (c0 * c0 + c1 * c1) - (b0 * b0 + b1 * b1) * c1) - a1 * (b0 * (c0 * c0 + c1 * c1) - (b0 * b0 + b1 * b1) * c0)
               + (a0 * a0 + a1 * a1) * (b0 * c1 - b1 * c0))
        assert max(det) <= 0, k


def test_degenerate_inputs():
    simp, _ = triangulate([np.array([[0, 0], [0, 5], [7, 0], [7, 5]])])        # one co-circular quadruple: two triangles
    assert len(simp) == 2
    simp, _ = triangulate([np.array([[0, 0], [1, 1], [2, 2], [3, 3]]), np.array([[5, 5], [6, 9], [9, 6]])])
    assert len(simp) == 1 and simp[0, 0] == 1                                  # collinear set: no triangle; the other tile has one
    with pytest.raises(Exception, match="out of range|could not be triangulated"):
        triangulate([np.array([[0, 0], [40000, 1], [2, 7]])])


@pytest.mark.parametrize("k", [1000, 2000, 8000, 16000])
def test_hull_slivers_with_huge_circumcircles_are_kept(k):
    """the chain (0,0), (k,1), (2k-1,2) is an area-1/2 lattice triple with a circumradius of ~k^2 / 2 (up to 2^27 here,
    beyond any finite super triangle round 3 could have used).  The super vertices are symbolic now: the triangles must tile
    the convex hull exactly and agree with scipy's count."""
    rng = np.random.default_rng(k)
    inner = np.column_stack((rng.integers(1, 2 * k - 2, 40), rng.integers(3, 60, 40)))
    pts = np.unique(np.concatenate([np.array([[0, 0], [k, 1], [2 * k - 1, 2]]), inner]), axis=0)
    simp, _ = triangulate([pts])
    tri = simp[:, 1:].reshape(-1, 3, 2).astype(object)
    area2 = sum(abs((t[1][0] - t[0][0]) * (t[2][1] - t[0][1]) - (t[1][1] - t[0][1]) * (t[2][0] - t[0][0])) for t in tri)
    assert area2 == round(2 * ConvexHull(pts).volume)
    assert len(tri) == len(Delaunay(pts).simplices)
    assert canon(simp) == set(tuple(sorted(map(tuple, pts[s].tolist()))) for s in Delaunay(pts).simplices)
