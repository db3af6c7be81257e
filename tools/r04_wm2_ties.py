"""on the co-circular lattice polygons of the golden tile wm_in_512, which vertex does scipy's
(Qhull 'Qt') triangulation fan from, and does a deterministic rule reproduce it?  CPU only (scipy)."""
import sys
from collections import defaultdict
from fractions import Fraction
import numpy as np
from scipy.spatial import Delaunay
sys.path.insert(0, '.')
from sequitr_amd.weightmap import boundary_triangulation  # noqa: E402
from scipy import ndimage

z = np.load('tests/golden/pipeline_golden.npz')
lab = z['wm_in_512'] > 0
cross = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]])
def outline(m):
    return np.logical_xor(ndimage.binary_erosion(m, iterations=1, structure=cross), m)
pm = np.logical_xor(outline(lab), outline(ndimage.binary_dilation(lab, iterations=3, structure=cross)))
px, py = np.where(pm)
P = np.column_stack((px, py)).astype(np.int64)
tri = Delaunay(P)
S = tri.simplices
print('points', len(P), 'simplices', len(S))

def circum(a, b, c):
    ax, ay = map(int, a); bx, by = map(int, b); cx, cy = map(int, c)
    d = 2 * (ax * (by - cy) + bx * (cy - ay) + cx * (ay - by))
    ux = Fraction((ax*ax+ay*ay)*(by-cy) + (bx*bx+by*by)*(cy-ay) + (cx*cx+cy*cy)*(ay-by), d)
    uy = Fraction((ax*ax+ay*ay)*(cx-bx) + (bx*bx+by*by)*(ax-cx) + (cx*cx+cy*cy)*(bx-ax), d)
    return ux, uy
cc = [circum(P[s[0]], P[s[1]], P[s[2]]) for s in S]
# union adjacent simplices with identical circumcentre
parent = list(range(len(S)))
def find(i):
    while parent[i] != i:
        parent[i] = parent[parent[i]]; i = parent[i]
    return i
for i, nb in enumerate(tri.neighbors):
    for j in nb:
        if j >= 0 and cc[i] == cc[j]:
            parent[find(i)] = find(j)
groups = defaultdict(list)
for i in range(len(S)):
    groups[find(i)].append(i)
polys = [g for g in groups.values() if len(g) > 1]
print('co-circular polygons', len(polys), 'sizes', np.bincount([len(g) + 2 for g in polys]))
stats = defaultdict(int)
nfan = 0
for g in polys:
    verts = sorted(set(int(v) for i in g for v in S[i]))
    cnt = defaultdict(int)
    for i in g:
        for v in S[i]:
            cnt[int(v)] += 1
    apexes = [v for v in verts if cnt[v] == len(g)]           # a fan: one vertex in every triangle
    if not apexes:
        stats['not a fan'] += 1
        continue
    nfan += 1
    # candidate rules
    lex = sorted(verts, key=lambda v: (P[v][0], P[v][1]))
    rules = {'max index': max(verts), 'min index': min(verts), 'lex min': lex[0], 'lex max': lex[-1],
             'min y then x': sorted(verts, key=lambda v: (P[v][1], P[v][0]))[0],
             'max y then x': sorted(verts, key=lambda v: (P[v][1], P[v][0]))[-1]}
    for k, v in rules.items():
        if v in apexes:
            stats[k] += 1
print('fans', nfan)
for k, v in sorted(stats.items(), key=lambda kv: -kv[1]):
    print('%-16s %6d  %.3f' % (k, v, v / max(nfan, 1)))
