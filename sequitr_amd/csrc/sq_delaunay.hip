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
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <functional>
#include <condition_variable>
#include <mutex>
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
// need 128; a triangle with a super vertex takes the symbolic form (in_circle_sym).
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

// ---- the three super vertices are AT INFINITY, symbolically ------------------------------------------------------------
// Super vertex k sits at M * SUPER_DIR[k] for an M larger than any number that occurs: a predicate with super vertices is a
// polynomial in M with exact integer coefficients and its sign is the sign of the highest non-zero coefficient.  (Round 3
// used a finite M = 2^30: a hull sliver whose circumradius exceeded it -- area-1/2 lattice triples with edges beyond ~1000
// pixels -- had a super vertex inside its circumcircle and was dropped; tiles of 512 pixels were not affected.)
const long long SUPER_DIR[3][2] = {{-1, -1}, {1, -1}, {0, 1}};
struct PolyM {                      // c[k] M^k, k <= 6
    i128 c[7];
    PolyM() { for (int k = 0; k < 7; ++k) c[k] = 0; }
    PolyM(i128 c0, i128 c1 = 0, i128 c2 = 0) { for (int k = 0; k < 7; ++k) c[k] = 0; c[0] = c0; c[1] = c1; c[2] = c2; }
    int sign() const {
        for (int k = 6; k >= 0; --k)
            if (c[k] != 0) return c[k] > 0 ? 1 : -1;
        return 0;
    }
};
inline PolyM operator*(const PolyM &a, const PolyM &b) {
    PolyM r;
    for (int i = 0; i < 7; ++i) {
        if (a.c[i] == 0) continue;
        for (int j = 0; i + j < 7; ++j) r.c[i + j] += a.c[i] * b.c[j];
    }
    return r;
}
inline PolyM operator-(const PolyM &a, const PolyM &b) {
    PolyM r;
    for (int k = 0; k < 7; ++k) r.c[k] = a.c[k] - b.c[k];
    return r;
}
inline PolyM operator+(const PolyM &a, const PolyM &b) {
    PolyM r;
    for (int k = 0; k < 7; ++k) r.c[k] = a.c[k] + b.c[k];
    return r;
}
// coordinate `axis` of vertex i (real: p[i]; super: M * direction) minus the real point q's, as a polynomial in M
inline PolyM rel(const std::vector<Pt> &p, int n, int i, int axis, const Pt &q) {
    const long long qv = axis ? q.y : q.x;
    if (i < n) return PolyM((i128)((axis ? p[i].y : p[i].x) - qv));
    return PolyM((i128)(-qv), (i128)SUPER_DIR[i - n][axis]);
}
// sign of orient(a, b, q), q real, a / b real or super
inline int orient_sym(const std::vector<Pt> &p, int n, int ia, int ib, const Pt &q) {
    // orient(a, b, q) = (a - q) x (b - q)
    const PolyM ax = rel(p, n, ia, 0, q), ay = rel(p, n, ia, 1, q), bx = rel(p, n, ib, 0, q), by = rel(p, n, ib, 1, q);
    return (ax * by - ay * bx).sign();
}
// q strictly inside the circle through the counter-clockwise (a, b, c), at least one of them a super vertex
inline bool in_circle_sym(const std::vector<Pt> &p, int n, int ia, int ib, int ic, const Pt &q) {
    const PolyM ax = rel(p, n, ia, 0, q), ay = rel(p, n, ia, 1, q), bx = rel(p, n, ib, 0, q), by = rel(p, n, ib, 1, q),
                cx = rel(p, n, ic, 0, q), cy = rel(p, n, ic, 1, q);
    const PolyM a2 = ax * ax + ay * ay, b2 = bx * bx + by * by, c2 = cx * cx + cy * cy;
    return (ax * (by * c2 - b2 * cy) - ay * (bx * c2 - b2 * cx) + a2 * (bx * cy - by * cx)).sign() > 0;
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
    p[n] = p[n + 1] = p[n + 2] = {0, 0};                        // never read: super vertices are symbolic (orient_sym / in_circle_sym)
    std::vector<Tri> t;
    t.reserve(2 * n + 16);
    t.push_back({{n, n + 1, n + 2}, {-1, -1, -1}, true});
    std::vector<int> freelist, cavity, stack, bnd_a, bnd_b, bnd_out, bnd_new;
    std::vector<unsigned> mark((size_t)2 * n + 1024, 0);
    std::vector<int> starts((size_t)n + 3), ends((size_t)n + 3);
    std::vector<unsigned> sstamp((size_t)n + 3, 0), estamp((size_t)n + 3, 0);
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
                const bool neg = (ia < n && ib < n) ? orient_real(p[ia], p[ib], q) < 0 : orient_sym(p, n, ia, ib, q) < 0;
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
                const bool real = t[m].v[0] < n && t[m].v[1] < n && t[m].v[2] < n;
                if (real ? in_circle<TINY>(p[t[m].v[0]], p[t[m].v[1]], p[t[m].v[2]], q, true)
                         : in_circle_sym(p, n, t[m].v[0], t[m].v[1], t[m].v[2], q)) {
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
            // the cavity is star-shaped round q: never (64-bit where no super vertex is involved)
            if ((bnd_a[e] < n && bnd_b[e] < n) ? orient_real(p[bnd_a[e]], p[bnd_b[e]], q) <= 0 : orient_sym(p, n, bnd_a[e], bnd_b[e], q) <= 0)
                return false;
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
        // The boundary is a cycle round q: per vertex, the fan triangle that STARTS there and the one that ENDS there (two
        // stamped scratch arrays).  Searching all boundary edges for every edge's two neighbours was a third of the run time.
        for (int e = 0; e < nb; ++e) {
            starts[bnd_a[e]] = bnd_new[e]; sstamp[bnd_a[e]] = stamp;
            ends[bnd_b[e]] = bnd_new[e]; estamp[bnd_b[e]] = stamp;
        }
        for (int e = 0; e < nb; ++e) {
            const int vb = bnd_b[e], va = bnd_a[e];
            if (sstamp[vb] != stamp || estamp[va] != stamp) return false;
            t[bnd_new[e]].n[0] = starts[vb];
            t[bnd_new[e]].n[1] = ends[va];
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

// A small persistent pool for the per-tile work: creating and joining 15 threads cost ~0.2 ms of a 2.5 ms call.  Workers
// sleep on a condition variable between calls; the pool object is leaked on purpose (a worker may still be blocked on it
// when the process exits).  One caller at a time (calls are serialised by `busy`); the caller takes part in the work.
struct HostPool {
    std::mutex m, busy;
    std::condition_variable wake, done;
    std::vector<std::thread> threads;
    const std::function<void()> *job = nullptr;
    unsigned long long gen = 0;
    int want = 0, pending = 0;
    void worker(int id) {
        unsigned long long seen = 0;
        for (;;) {
            const std::function<void()> *f;
            {
                std::unique_lock<std::mutex> lk(m);
                wake.wait(lk, [&] { return gen != seen; });
                seen = gen;
                if (id >= want) continue;                       // this call uses fewer workers
                f = job;
            }
            (*f)();
            {
                std::lock_guard<std::mutex> lk(m);
                if (--pending == 0) done.notify_one();
            }
        }
    }
};

void run_on_pool(int nt, const std::function<void()> &work) {
    static HostPool *pool = nullptr;                            // never destroyed
    static int pool_pid = 0;
    static std::mutex create;
    if (nt <= 1) { work(); return; }
    {
        std::lock_guard<std::mutex> lk(create);
        const int pid = (int)getpid();
        if (!pool || pool_pid != pid) {                         // first call, or a forked child: the parent's workers are not here
            pool = new HostPool();
            pool_pid = pid;
        }
    }
    std::lock_guard<std::mutex> serial(pool->busy);
    {
        std::lock_guard<std::mutex> lk(pool->m);
        while ((int)pool->threads.size() < nt - 1) {
            const int id = (int)pool->threads.size();
            pool->threads.emplace_back([id] { pool->worker(id); });
            pool->threads.back().detach();
        }
        pool->job = &work;
        pool->want = nt - 1;
        pool->pending = nt - 1;
        ++pool->gen;
    }
    pool->wake.notify_all();
    work();
    std::unique_lock<std::mutex> lk(pool->m);
    pool->done.wait(lk, [&] { return pool->pending == 0; });
    pool->job = nullptr;
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
        std::vector<int32_t> rowbuf;
        std::vector<double> lngbuf;
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
            // rows are assembled in ordinary (cached) memory and copied to their slot in two wide copies: the destination is
            // pinned staging memory, where 4-byte stores scattered over 36-byte rows ran at ~1 GB/s (0.45 ms per tile)
            const int64_t cnt = ok ? (int64_t)tv.size() / 3 : 0, room = rend - r;
            rowbuf.resize((size_t)7 * room);
            lngbuf.resize((size_t)room);
            for (int64_t k = 0; k < cnt; ++k) {
                int32_t *row = rowbuf.data() + 7 * k;
                const int a = tv[3 * k], b2 = tv[3 * k + 1], c = tv[3 * k + 2];
                const int ax = pts[2 * a], ay = pts[2 * a + 1], bx = pts[2 * b2], by = pts[2 * b2 + 1], cx = pts[2 * c], cy = pts[2 * c + 1];
                row[0] = s; row[1] = ax; row[2] = ay; row[3] = bx; row[4] = by; row[5] = cx; row[6] = cy;
                const long long d0 = (long long)(ax - bx) * (ax - bx) + (long long)(ay - by) * (ay - by);
                const long long d1 = (long long)(bx - cx) * (bx - cx) + (long long)(by - cy) * (by - cy);
                const long long d2 = (long long)(cx - ax) * (cx - ax) + (long long)(cy - ay) * (cy - ay);
                lngbuf[(size_t)k] = std::sqrt((double)std::max(d0, std::max(d1, d2)));   // = np.sqrt(dx^2 + dy^2).max(): sqrt is monotone, the integers exact
            }
            for (int64_t k = cnt; k < room; ++k) {              // rows this tile does not need
                int32_t *row = rowbuf.data() + 7 * k;
                row[0] = -1;
                for (int j = 1; j < 7; ++j) row[j] = 0;
                lngbuf[(size_t)k] = 0.0;
            }
            memcpy(simplices + 7 * r, rowbuf.data(), (size_t)room * 7 * sizeof(int32_t));
            memcpy(longest + r, lngbuf.data(), (size_t)room * sizeof(double));
            if (prof)
                fprintf(stderr, "  tile %d: %lld points, triangulate %.3f ms, rows %.3f ms\n", s, (long long)(e - b),
                        std::chrono::duration<double, std::milli>(d1 - d0).count(),
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - d1).count());
        }
    };
    const int nt = host_threads(nsets);
    const auto c0 = std::chrono::steady_clock::now();
    run_on_pool(nt, work);
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
