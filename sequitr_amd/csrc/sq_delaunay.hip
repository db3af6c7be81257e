// ImageWeightMap2's triangulation without scipy (sequitr/pipeline.py:514-545), round 3.
//
//   * sq_wm2_boundary_points_u8 (device): the boundary-point mask of :516-528 -- erosion outline of the label XOR
//     outline of the label dilated three times, von Neumann element, scipy's border_value = 0 -- one pass per pixel.
//   * sq_delaunay2d_batch_i32 (HOST, native, threaded): an exact Delaunay triangulation of each tile's boundary
//     pixels.  The reference calls Qhull through scipy (general-dimension, ~22-38 ms per 512x512 tile, one tile at a
//     time under the GIL: 352 of the 352.2 ms of a 16-tile batch in round 2).  The points are distinct integer
//     pixels, so the in-circle and orientation predicates are evaluated EXACTLY in 128-bit integers and a plain
//     incremental Bowyer-Watson insertion along a Z curve (the next point is next to the last one, so the point-location
//     walk is a few steps) triangulates a tile in ~2 ms; tiles are independent and run on a small
//     pool of host threads.  Output = the (tile, x0, y0, x1, y1, x2, y2) rows and longest edges the raster kernel
//     (sq_weightmap2_delaunay_f32) takes.
//
// Parity: wherever the Delaunay triangulation is unique this IS the reference's triangulation.  Boundary pixels are
// lattice points, so co-circular quadruples are common; there the triangulation is not unique, Qhull's choice depends
// on its internal facet order (not reproducible by any other implementation, device or host) and a pixel inside such
// a quadrilateral may get the other diagonal's longest edge.  tests/test_gpu_weightmap.py states the measured bound
// against the reference-generated vectors; scipy's own joggled triangulation ("QJ") of the same points differs from
// its default one by the same amount (profiles/r03_wm2_notes.txt).
#include "sq_common.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#define SQ_GRID_STRIDE(i, n) \
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (n); i += (int64_t)gridDim.x * 256)

namespace {

// ---- device: boundary points ------------------------------------------------------------------------------------
__device__ __forceinline__ bool wm2_lab(const float *__restrict__ img, int H, int W, int y, int x) {
    return (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W && img[(size_t)y * W + x] != 0.0f;   // border_value = 0
}
// binary_dilation(iterations = 3, cross) = some label pixel within L1 distance 3
__device__ __forceinline__ bool wm2_dil3(const float *__restrict__ img, int H, int W, int y, int x) {
    if ((unsigned)y >= (unsigned)H || (unsigned)x >= (unsigned)W) return false;   // the dilated image lives on the tile; outside = 0
#pragma unroll
    for (int dy = -3; dy <= 3; ++dy) {
        const int r = 3 - (dy < 0 ? -dy : dy);
        for (int dx = -r; dx <= r; ++dx)
            if (wm2_lab(img, H, W, y + dy, x + dx)) return true;
    }
    return false;
}

__global__ __launch_bounds__(256) void wm2_points_kernel(const float *__restrict__ img, uint8_t *__restrict__ pts, int N, int H,
                                                         int W) {
    const int64_t total = (int64_t)N * H * W;
    SQ_GRID_STRIDE(i, total) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        const float *im = img + (i / ((int64_t)H * W)) * (int64_t)H * W;
        // outline(m) = m xor erosion(m): the pixels of m with a 4-neighbour outside m (the tile border counts as outside)
        const bool b = wm2_lab(im, H, W, y, x);
        const bool b_out = b && !(wm2_lab(im, H, W, y - 1, x) && wm2_lab(im, H, W, y + 1, x) && wm2_lab(im, H, W, y, x - 1) &&
                                  wm2_lab(im, H, W, y, x + 1));
        const bool d = wm2_dil3(im, H, W, y, x);
        const bool d_out = d && !(wm2_dil3(im, H, W, y - 1, x) && wm2_dil3(im, H, W, y + 1, x) && wm2_dil3(im, H, W, y, x - 1) &&
                                  wm2_dil3(im, H, W, y, x + 1));
        pts[i] = (uint8_t)(b_out != d_out);
    }
}

// ---- host: exact incremental Delaunay -----------------------------------------------------------------------------
typedef __int128 i128;
struct Pt { long long x, y; };

inline i128 orient(const Pt &a, const Pt &b, const Pt &c) {     // > 0: a, b, c counter-clockwise
    return (i128)(b.x - a.x) * (c.y - a.y) - (i128)(b.y - a.y) * (c.x - a.x);
}
inline long long orient_real(const Pt &a, const Pt &b, const Pt &c) {   // all three below 2^15: 64 bits hold it
    return (b.x - a.x) * (c.y - a.y) - (b.y - a.y) * (c.x - a.x);
}
// > 0: d strictly inside the circle through the counter-clockwise a, b, c.  Exact: with every coordinate below 2^15 the
// 2x2 minors fit 64 bits (differences < 2^16, squared lengths < 2^33, minors < 2^51) and only the last three products
// need 128; a triangle with a super vertex takes the all-128-bit form.
template <bool TINY>
inline bool in_circle(const Pt &a, const Pt &b, const Pt &c, const Pt &d, bool real) {
    if (TINY && real) {
        // every coordinate below 2^12 (a 512 .. 4096 pixel tile): differences < 2^13, squared lengths < 2^27, minors < 2^41,
        // products < 2^54 -- the whole determinant in 64 bits
        const long long ax = a.x - d.x, ay = a.y - d.y, bx = b.x - d.x, by = b.y - d.y, cx = c.x - d.x, cy = c.y - d.y;
        const long long a2 = ax * ax + ay * ay, b2 = bx * bx + by * by, c2 = cx * cx + cy * cy;
        return ax * (by * c2 - b2 * cy) - ay * (bx * c2 - b2 * cx) + a2 * (bx * cy - by * cx) > 0;
    }
    if (real) {
        const long long ax = a.x - d.x, ay = a.y - d.y, bx = b.x - d.x, by = b.y - d.y, cx = c.x - d.x, cy = c.y - d.y;
        const long long a2 = ax * ax + ay * ay, b2 = bx * bx + by * by, c2 = cx * cx + cy * cy;
        const i128 det = (i128)ax * (by * c2 - b2 * cy) - (i128)ay * (bx * c2 - b2 * cx) + (i128)a2 * (bx * cy - by * cx);
        return det > 0;
    }
    const i128 ax = a.x - d.x, ay = a.y - d.y, bx = b.x - d.x, by = b.y - d.y, cx = c.x - d.x, cy = c.y - d.y;
    const i128 a2 = ax * ax + ay * ay, b2 = bx * bx + by * by, c2 = cx * cx + cy * cy;
    const i128 det = ax * (by * c2 - b2 * cy) - ay * (bx * c2 - b2 * cx) + a2 * (bx * cy - by * cx);
    return det > 0;
}

struct Tri {
    int v[3];          // counter-clockwise
    int n[3];          // n[i]: triangle across the edge opposite v[i] (-1: none)
    bool alive;
};

// Triangulates pts[0..n) (distinct, 0 <= coordinate < 2^15).  Returns the triangles without a super vertex as vertex
// index triples appended to `out`; false on an internal inconsistency (the caller reports an error, never a wrong map).
template <bool TINY>
bool delaunay_t(const int32_t *xy, int n, std::vector<int> &out) {
    if (n < 3) return true;
    const long long M = (long long)1 << 28;                     // super triangle: beyond every circumcircle that can border the hull
    // insertion order: along a Z curve, so that the next point is next to the last one and the walk stays short (scan
    // order jumps from cell to cell along a row: ~100 triangles per walk); vertex indices below are positions in `p`,
    // mapped back through `order` on output
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    auto zkey = [&](int i) {
        unsigned long long k = 0;
        const unsigned x = (unsigned)xy[2 * i], y = (unsigned)xy[2 * i + 1];
        for (int b = 0; b < 15; ++b) k |= ((unsigned long long)((x >> b) & 1u) << (2 * b + 1)) | ((unsigned long long)((y >> b) & 1u) << (2 * b));
        return k;
    };
    std::vector<unsigned long long> keys(n);
    for (int i = 0; i < n; ++i) keys[i] = zkey(i);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return keys[a] < keys[b]; });
    std::vector<Pt> p(n + 3);
    for (int i = 0; i < n; ++i) p[i] = {xy[2 * order[i]], xy[2 * order[i] + 1]};
    p[n] = {-4 * M, -4 * M};
    p[n + 1] = {4 * M, -4 * M};
    p[n + 2] = {0, 4 * M};
    std::vector<Tri> t;
    t.reserve(2 * n + 16);
    t.push_back({{n, n + 1, n + 2}, {-1, -1, -1}, true});
    std::vector<int> freelist, cavity, stack, bnd_a, bnd_b, bnd_out, bnd_new;
    std::vector<unsigned> mark((size_t)2 * n + 1024, 0);
    unsigned stamp = 0;
    int cur = 0;
    for (int ip = 0; ip < n; ++ip) {
        const Pt &q = p[ip];
        // ---- locate: visibility walk from the last triangle touched --------------------------------------------
        int guard = 0;
        for (;;) {
            const Tri &c = t[cur];
            int go = -1;
            bool leave = false;
            for (int i = 0; i < 3; ++i) {
                const int ia = c.v[(i + 1) % 3], ib = c.v[(i + 2) % 3];
                const bool neg = (ia < n && ib < n) ? orient_real(p[ia], p[ib], q) < 0 : orient(p[ia], p[ib], q) < 0;
                if (neg) { go = c.n[i]; leave = true; break; }
            }
            if (!leave) break;
            if (go < 0) return false;                           // outside the super triangle: never
            cur = go;
            if (++guard > 4 * (int)t.size() + 64) return false;
        }
        // ---- cavity: every triangle whose circumcircle strictly contains q (connected, holds `cur`) ----------------
        if (mark.size() < t.size()) mark.resize(t.size() + 1024, 0);
        ++stamp;
        cavity.clear();
        stack.clear();
        stack.push_back(cur);
        mark[cur] = stamp;
        while (!stack.empty()) {
            const int k = stack.back();
            stack.pop_back();
            cavity.push_back(k);
            for (int i = 0; i < 3; ++i) {
                const int m = t[k].n[i];
                if (m < 0 || mark[m] == stamp) continue;
                if (in_circle<TINY>(p[t[m].v[0]], p[t[m].v[1]], p[t[m].v[2]], q, t[m].v[0] < n && t[m].v[1] < n && t[m].v[2] < n)) {
                    mark[m] = stamp;
                    stack.push_back(m);
                }
            }
        }
        // ---- boundary edges (a -> b counter-clockwise round the cavity), then the fan of new triangles ------------------
        bnd_a.clear(); bnd_b.clear(); bnd_out.clear();
        for (int k : cavity)
            for (int i = 0; i < 3; ++i) {
                const int m = t[k].n[i];
                if (m >= 0 && mark[m] == stamp) continue;
                bnd_a.push_back(t[k].v[(i + 1) % 3]);
                bnd_b.push_back(t[k].v[(i + 2) % 3]);
                bnd_out.push_back(m);
            }
        for (int k : cavity) {
            t[k].alive = false;
            freelist.push_back(k);
        }
        const int nb = (int)bnd_a.size();
        bnd_new.assign(nb, -1);
        for (int e = 0; e < nb; ++e) {
            if (orient(p[bnd_a[e]], p[bnd_b[e]], q) <= 0) return false;    // the cavity is star-shaped round q: never
            int id;
            if (!freelist.empty()) { id = freelist.back(); freelist.pop_back(); }
            else { id = (int)t.size(); t.push_back(Tri()); if (mark.size() < t.size()) mark.resize(t.size() + 1024, 0); }
            t[id] = {{bnd_a[e], bnd_b[e], ip}, {-1, -1, bnd_out[e]}, true};
            mark[id] = 0;
            bnd_new[e] = id;
            const int m = bnd_out[e];
            if (m >= 0)
                for (int i = 0; i < 3; ++i)                     // the outside triangle's edge (b, a) now borders the new one
                    if (t[m].v[(i + 1) % 3] == bnd_b[e] && t[m].v[(i + 2) % 3] == bnd_a[e]) t[m].n[i] = id;
        }
        // new triangle (a, b, q): edge (b, q) is opposite a -> n[0] = the new triangle that starts at b;
        //                          edge (q, a) is opposite b -> n[1] = the new triangle that ends at a
        for (int e = 0; e < nb; ++e) {
            int nxt = -1, prv = -1;
            for (int f = 0; f < nb; ++f) {
                if (bnd_a[f] == bnd_b[e]) nxt = bnd_new[f];
                if (bnd_b[f] == bnd_a[e]) prv = bnd_new[f];
            }
            if (nxt < 0 || prv < 0) return false;
            t[bnd_new[e]].n[0] = nxt;
            t[bnd_new[e]].n[1] = prv;
        }
        cur = bnd_new[0];
    }
    for (const Tri &c : t)
        if (c.alive && c.v[0] < n && c.v[1] < n && c.v[2] < n) {
            out.push_back(order[c.v[0]]);
            out.push_back(order[c.v[1]]);
            out.push_back(order[c.v[2]]);
        }
    return true;
}

bool delaunay(const int32_t *xy, int n, std::vector<int> &out) {
    int hi = 0;
    for (int i = 0; i < 2 * n; ++i) hi = xy[i] > hi ? xy[i] : hi;
    return hi < (1 << 12) ? delaunay_t<true>(xy, n, out) : delaunay_t<false>(xy, n, out);
}

int host_threads(int jobs) {
    const char *e = getenv("SQ_HOST_THREADS");
    int want = e ? atoi(e) : 16;                                // a 1-GPU box owns a 16-core share of the host
    const int hw = (int)std::thread::hardware_concurrency();
    if (hw > 0 && want > hw) want = hw;
    if (want > jobs) want = jobs;
    return want < 1 ? 1 : want;
}

}  // namespace

extern "C" int sq_wm2_boundary_points_u8(const float *img, uint8_t *points, int N, int H, int W, void *stream) {
    SQ_REQUIRE(img && points && N > 0 && H > 0 && W > 0, "sq_wm2_boundary_points_u8: bad arguments");
    const int64_t total = (int64_t)N * H * W;
    int64_t nb = (total + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(wm2_points_kernel, dim3((unsigned)nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), img, points, N, H,
                       W);
    return sq_check_launch("sq_wm2_boundary_points_u8");
}

// HOST function (no GPU work).  xy: the boundary points of `nsets` tiles back to back, (row, column) int32 pairs, tile s
// = points offsets[s] .. offsets[s + 1]; simplices: room for `cap` rows of 7 int32 {tile, x0, y0, x1, y1, x2, y2};
// longest: `cap` doubles.  A Delaunay triangulation of n points has < 2 n triangles, so tile s owns rows 2 offsets[s] ..
// 2 offsets[s + 1] - 1: every worker writes its tile's rows as soon as the tile is triangulated (no second pass, no
// barrier between the tiles), and pads the few rows its tile does not need with tile = -1 (the raster kernel skips those).
// Returns the number of rows = 2 * offsets[nsets] (cap must hold them), or a negative SQ_E* code.
extern "C" int64_t sq_delaunay2d_batch_i32(const int32_t *xy, const int64_t *offsets, int nsets, int32_t *simplices,
                                           double *longest, int64_t cap) {
    if (!xy || !offsets || !simplices || !longest || nsets <= 0) {
        sq_set_error("sq_delaunay2d_batch_i32: bad arguments");
        return SQ_EINVAL;
    }
    const int64_t rows = 2 * offsets[nsets];
    if (rows > cap) {
        sq_set_error("sq_delaunay2d_batch_i32: %lld rows (2 per point) exceed the capacity %lld", (long long)rows, (long long)cap);
        return SQ_EINVAL;
    }
    std::atomic<int> next(0), failed(-1);
    const bool prof = getenv("SQ_DL_PROF") != nullptr;
    auto work = [&]() {
        std::vector<int> tv;
        for (;;) {
            const int s = next.fetch_add(1);
            if (s >= nsets) return;
            const int64_t b = offsets[s], e = offsets[s + 1];
            bool ok = e >= b && e - b < ((int64_t)1 << 24);
            for (int64_t i = 2 * b; ok && i < 2 * e; ++i) ok = xy[i] >= 0 && xy[i] < (1 << 15);
            const auto d0 = std::chrono::steady_clock::now();
            tv.clear();
            if (ok) ok = delaunay(xy + 2 * b, (int)(e - b), tv);
            const auto d1 = std::chrono::steady_clock::now();
            int64_t r = 2 * b;
            const int64_t rend = 2 * e;
            if (ok && (int64_t)tv.size() / 3 > rend - r) ok = false;
            if (!ok) failed.store(s);
            const int32_t *pts = xy + 2 * b;
            if (ok)
                for (size_t k = 0; k + 2 < tv.size(); k += 3, ++r) {
                    int32_t *row = simplices + 7 * r;
                    row[0] = s;
                    long long best = 0;
                    for (int j = 0; j < 3; ++j) {
                        const int a = tv[k + j], c = tv[k + (j + 1) % 3];
                        row[1 + 2 * j] = pts[2 * a];
                        row[2 + 2 * j] = pts[2 * a + 1];
                        const long long dx = (long long)pts[2 * a] - pts[2 * c], dy = (long long)pts[2 * a + 1] - pts[2 * c + 1];
                        best = std::max(best, dx * dx + dy * dy);
                    }
                    longest[r] = std::sqrt((double)best);      // = np.sqrt(dx^2 + dy^2).max(): sqrt is monotone, the integers exact
                }
            for (; r < rend; ++r) {                             // rows this tile does not need
                int32_t *row = simplices + 7 * r;
                row[0] = -1;
                for (int j = 1; j < 7; ++j) row[j] = 0;
                longest[r] = 0.0;
            }
            if (prof)
                fprintf(stderr, "  tile %d: %lld points, triangulate %.3f ms, rows %.3f ms\n", s, (long long)(e - b),
                        std::chrono::duration<double, std::milli>(d1 - d0).count(),
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - d1).count());
        }
    };
    const int nt = host_threads(nsets);
    const auto c0 = std::chrono::steady_clock::now();
    {
        std::vector<std::thread> pool;
        for (int i = 1; i < nt; ++i) pool.emplace_back(work);
        work();
        for (auto &th : pool) th.join();
    }
    if (failed.load() >= 0) {
        sq_set_error("sq_delaunay2d_batch_i32: tile %d could not be triangulated (coordinates out of range or an "
                     "internal inconsistency)", failed.load());
        return SQ_EINVAL;
    }
    if (prof)
        fprintf(stderr, "sq_delaunay2d_batch_i32: %d tiles, %d threads: %.3f ms\n", nsets, nt,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - c0).count());
    return rows;
}
