// EDT weight maps on the GPU (the step in front of the training hot path).
// ImageWeightMap.pipe, sequitr/pipeline.py:475-479:
//     d   = scipy.ndimage.distance_transform_edt(1 - image)        (distance to the nearest pixel == 1)
//     out = w0 * (1 - image) * exp(-(d*d) / (2 sigma^2 + 1e-99)) + image + 1          (float64)
//
// Exact Euclidean distance transform, separable:
//   pass 1 (rows)   : g(y,x) = distance along the row to the nearest feature pixel; one wave per row,
//                     64-pixel segments, nearest set bit of the feature ballot to the left / right
//                     (clz / ffs) with a carry between segments -- no loops over pixels
//   pass 2 (columns): D2(y,x) = min over y' of g(y',x)^2 + (y-y')^2, searched outwards from y and stopped
//                     as soon as (y-y')^2 >= the best so far; lanes run along x => every read of g is a
//                     coalesced row segment
// D2 is an exact integer, d = sqrt((double)D2) is correctly rounded => identical to scipy's
// sqrt(sum(delta^2)).  The map then follows the reference's float64 expression operation by operation.
// Integer / streaming work: 4 B/pixel in, 2 B/pixel of g written and re-read, 8 (+4) B/pixel out.
#include "sq_common.h"

// every multiply and add below is a separate, correctly rounded operation as in numpy: no fused contraction
#pragma clang fp contract(off)

namespace {

typedef unsigned long long u64;
constexpr int G_INF = 30000;           // "no feature in this row": 30000^2 + 32767^2 < 2^31

__global__ __launch_bounds__(256) void edt_rows_kernel(const float *__restrict__ img, unsigned short *__restrict__ g,
                                                       int *__restrict__ has_feature, int rows, int H, int W) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float *row = img + (size_t)r * W;
    unsigned short *grow = g + (size_t)r * W;
    // left-to-right: distance to the nearest feature at or before this pixel
    int last = -G_INF;                                          // column of the last feature seen so far
    bool any = false;
    for (int c0 = 0; c0 < W; c0 += 64) {
        const int col = c0 + lane;
        const bool f = col < W && (1.0 - (double)row[col]) == 0.0;     // feature <=> (1 - image) == 0
        const u64 F = __ballot(f);
        const u64 upto = lane == 63 ? ~0ULL : ((2ULL << lane) - 1ULL);
        const u64 m = F & upto;
        const int near = m ? c0 + 63 - __clzll((long long)m) : last;
        if (col < W) {
            const int d = col - near;
            grow[col] = (unsigned short)(d < G_INF ? d : G_INF);
        }
        if (F) {
            last = c0 + 63 - __clzll((long long)F);
            any = true;
        }
    }
    // right-to-left: combine with the nearest feature at or after this pixel
    int next = 2 * G_INF;
    const int nseg = (W + 63) / 64;
    for (int sgm = nseg - 1; sgm >= 0; --sgm) {
        const int c0 = sgm * 64, col = c0 + lane;
        const bool f = col < W && (1.0 - (double)row[col]) == 0.0;
        const u64 F = __ballot(f);
        const u64 from = ~0ULL << lane;                         // lanes >= mine
        const u64 m = F & from;
        const int near = m ? c0 + __ffsll((long long)m) - 1 : next;
        if (col < W) {
            const int d = near - col;
            const int old = grow[col];
            if (d < old) grow[col] = (unsigned short)d;
        }
        if (F) next = c0 + __ffsll((long long)F) - 1;
    }
    if (any && lane == 0) atomicOr(&has_feature[r / H], 1);
}

template <bool WANT_D2>
__global__ __launch_bounds__(256) void edt_cols_weight_kernel(const float *__restrict__ img,
                                                              const unsigned short *__restrict__ g,
                                                              const int *__restrict__ has_feature,
                                                              double *__restrict__ out64, float *__restrict__ out32,
                                                              int *__restrict__ d2out, int N, int H, int W, double w0,
                                                              double denom) {
    const int64_t total = (int64_t)N * H * W;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(p % W), y = (int)((p / W) % H), n = (int)(p / ((int64_t)H * W));
        const unsigned short *gc = g + (size_t)n * H * W + x;
        int best;
        if (!has_feature[n]) {
            // scipy's feature transform with no background pixel at all points every pixel at index
            // (-1, 0): reproduce the artefact rather than invent a value
            best = (y + 1) * (y + 1) + x * x;
        } else {
            const int g0 = gc[(size_t)y * W];
            best = g0 * g0;
            for (int dy = 1; dy * dy < best; ++dy) {
                const int ya = y - dy, yb = y + dy;
                if (ya < 0 && yb >= H) break;
                if (ya >= 0) {
                    const int v = gc[(size_t)ya * W];
                    const int c = v * v + dy * dy;
                    best = c < best ? c : best;
                }
                if (yb < H) {
                    const int v = gc[(size_t)yb * W];
                    const int c = v * v + dy * dy;
                    best = c < best ? c : best;
                }
            }
        }
        if (WANT_D2) {
            d2out[p] = best;
        } else {
            const double image = (double)img[p];
            const double bg = 1.0 - image;
            const double d = sqrt((double)best);
            // self.w0 * (1.-image) * np.exp(-(d*d) / (2.*sigma**2 + 1e-99)) + image + 1.
            const double v = w0 * bg * exp(-(d * d) / denom) + image + 1.0;
            if (out64) out64[p] = v;
            if (out32) out32[p] = (float)v;
        }
    }
}

// ---- round-2 forms of the two passes (W <= 1024, H <= 2048; larger images keep the kernels above) ----------------------
// rows: the feature ballots of the WHOLE row are taken first (all loads of the row in flight at once), the nearest feature
// to the left / right of every pixel then comes out of registers and g is written once (the kernel above re-reads the row
// and read-modify-writes g, one dependent round trip per 64-pixel segment and direction).
constexpr int ROW_SEGS = 16;
__global__ __launch_bounds__(256) void edt_rows_v2_kernel(const float *__restrict__ img, unsigned short *__restrict__ g,
                                                          int rows, int W) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float *row = img + (size_t)r * W;
    const int nseg = (W + 63) / 64;
    float v[ROW_SEGS];
#pragma unroll
    for (int sg = 0; sg < ROW_SEGS; ++sg) {
        const int col = sg * 64 + lane;
        v[sg] = (sg < nseg && col < W) ? row[col] : 0.f;
    }
    u64 F[ROW_SEGS];
#pragma unroll
    for (int sg = 0; sg < ROW_SEGS; ++sg)
        F[sg] = (sg < nseg) ? __ballot(sg * 64 + lane < W && (1.0 - (double)v[sg]) == 0.0) : 0ULL;   // feature <=> (1 - image) == 0
    // right-to-left sweep over the segments: column of the first feature after each segment
    int nxt[ROW_SEGS];
    int next = 2 * G_INF;
#pragma unroll
    for (int sg = ROW_SEGS - 1; sg >= 0; --sg) {
        nxt[sg] = next;
        if (F[sg]) next = sg * 64 + __ffsll((long long)F[sg]) - 1;
    }
    int last = -G_INF;                                          // column of the last feature before the segment
    const u64 upto = lane == 63 ? ~0ULL : ((2ULL << lane) - 1ULL), from = ~0ULL << lane;
#pragma unroll
    for (int sg = 0; sg < ROW_SEGS; ++sg) {
        if (sg < nseg) {
            const int col = sg * 64 + lane;
            const u64 ml = F[sg] & upto, mr = F[sg] & from;
            const int left = ml ? sg * 64 + 63 - __clzll((long long)ml) : last;
            const int right = mr ? sg * 64 + __ffsll((long long)mr) - 1 : nxt[sg];
            int d = col - left;
            d = d < G_INF ? d : G_INF;
            const int dr = right - col;
            d = dr < d ? dr : d;
            if (col < W) g[(size_t)r * W + col] = (unsigned short)d;
            if (F[sg]) last = sg * 64 + 63 - __clzll((long long)F[sg]);
        }
    }
}

// columns + map: a block owns a strip of CS columns of one image with g for ALL rows in LDS (H x CS x 2 bytes); the outward
// search of the kernel above then runs on LDS (it issued two dependent global loads per step), lanes along x.  "No feature
// in this image" is a block-local fact: every row with a feature has a finite g in every column.
constexpr int CS = 32, RSPLIT = 8;                           // RSPLIT blocks share a strip, each maps H / RSPLIT of its rows (the
                                                              // strip is re-read from L2 by each: parallelism over bytes)
template <bool WANT_D2>
__global__ __launch_bounds__(256) void edt_cols_weight_v2_kernel(const float *__restrict__ img,
                                                                 const unsigned short *__restrict__ g,
                                                                 double *__restrict__ out64, float *__restrict__ out32,
                                                                 int *__restrict__ d2out, int H, int W, double w0, double denom) {
    extern __shared__ __attribute__((aligned(16))) unsigned short gs[];       // [H][CS], then the block minima [ceil(H/8)][CS]
    __shared__ int any_feature;
    const int nblk8 = (H + 7) / 8;
    unsigned short *m8 = gs + (size_t)H * CS;                   // m8[b][x] = min of g over rows 8b .. 8b+7 of column x
    const int strips = (W + CS - 1) / CS;
    const int part = blockIdx.x % RSPLIT, sb = blockIdx.x / RSPLIT;
    const int n = sb / strips, x0 = (sb % strips) * CS;
    const int rows_per = (H + RSPLIT - 1) / RSPLIT, y_lo = part * rows_per, y_hi = min(H, y_lo + rows_per);
    const unsigned short *gi = g + (size_t)n * H * W;
    if (threadIdx.x == 0) any_feature = 0;
    __syncthreads();
    int seen = 0;
    for (int i = threadIdx.x; i < H * CS; i += 256) {
        const int y = i / CS, x = x0 + i % CS;
        const unsigned short v = x < W ? gi[(size_t)y * W + x] : (unsigned short)G_INF;
        gs[i] = v;
        seen |= v < G_INF;
    }
    if (__any(seen) && (threadIdx.x & 63) == 0) any_feature = 1;             // benign race: every writer stores 1
    __syncthreads();
    for (int i = threadIdx.x; i < nblk8 * CS; i += 256) {
        const int b = i / CS, xx = i % CS;
        int m = G_INF;
        for (int r = 8 * b; r < min(H, 8 * b + 8); ++r) m = min(m, (int)gs[r * CS + xx]);
        m8[i] = (unsigned short)m;
    }
    __syncthreads();
    const bool has = any_feature != 0;
    const int xl = threadIdx.x % CS, x = x0 + xl;
    for (int y = y_lo + threadIdx.x / CS; y < y_hi; y += 256 / CS) {
        if (x >= W) continue;
        int best;
        if (!has) {
            best = (y + 1) * (y + 1) + x * x;                   // scipy's artefact for an image without background (see above)
        } else {
            // min over y' of g(y')^2 + (y - y')^2, searched outwards in blocks of 8 rows: a block is skipped when its nearest
            // row is already too far (and with it every block beyond) or when (nearest row)^2 + (its smallest g)^2 cannot
            // beat the best so far; the blocks that can are scanned exactly, eight independent candidates at a time.
            // Same minimum as the row-by-row search above: only candidates that cannot lower it are left out.
            const unsigned short *gc = gs + xl, *mc = m8 + xl;
            const int g0 = gc[y * CS];
            best = g0 * g0;
            auto scan = [&](int b) {
                const int r0 = 8 * b;
                if (r0 + 8 <= H) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int v = gc[(r0 + k) * CS], dy = y - (r0 + k);
                        const int c = v * v + dy * dy;
                        best = c < best ? c : best;
                    }
                } else {
                    for (int r = r0; r < H; ++r) {
                        const int v = gc[r * CS], dy = y - r;
                        const int c = v * v + dy * dy;
                        best = c < best ? c : best;
                    }
                }
            };
            const int b0 = y >> 3;
            scan(b0);
            bool up = true, dn = true;
            for (int db = 1; up || dn; ++db) {
                const int bu = b0 - db, bd = b0 + db;
                if (up) {
                    const int du = y - (8 * bu + 7);            // distance to the nearest row of the block above
                    if (bu < 0 || du * du >= best) up = false;
                    else { const int m = mc[bu * CS]; if (du * du + m * m < best) scan(bu); }
                }
                if (dn) {
                    const int dd = 8 * bd - y;
                    if (bd >= nblk8 || dd * dd >= best) dn = false;
                    else { const int m = mc[bd * CS]; if (dd * dd + m * m < best) scan(bd); }
                }
            }
        }
        const size_t p = ((size_t)n * H + y) * W + x;
        if (WANT_D2) {
            d2out[p] = best;
        } else {
            const double image = (double)img[p];
            const double bg = 1.0 - image;
            const double d = sqrt((double)best);
            const double v = w0 * bg * exp(-(d * d) / denom) + image + 1.0;   // the reference's expression, as above
            if (out64) out64[p] = v;
            if (out32) out32[p] = (float)v;
        }
    }
}

inline bool edt_v2_fits(int H, int W) {
    static const bool on = [] { const char *e = getenv("SQ_EDT_V2"); return !(e && e[0] == '0'); }();
    return on && W <= 64 * ROW_SEGS && H <= 2048;
}

template <bool WANT_D2>
int edt_v2(const float *img, unsigned short *g, double *out64, float *out32, int *d2, int N, int H, int W, double w0,
           double denom, hipStream_t st, const char *who) {
    const int rows = N * H;
    hipLaunchKernelGGL(edt_rows_v2_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, img, g, rows, W);
    int rc = sq_check_launch(who);
    if (rc) return rc;
    const int lds = (H + (H + 7) / 8) * CS * 2;
    auto kern = edt_cols_weight_v2_kernel<WANT_D2>;
    if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               lds) != hipSuccess) {
        sq_set_error("%s: cannot reserve %d bytes of LDS", who, lds);
        return SQ_ELAUNCH;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(N * ((W + CS - 1) / CS) * RSPLIT)), dim3(256), lds, st, img, g, out64, out32, d2, H, W,
                       w0, denom);
    return sq_check_launch(who);
}

inline unsigned wm_grid(int64_t items) {
    int64_t b = (items + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

int edt_common(const float *img, void *workspace, int N, int H, int W, unsigned short **g_out, int **flag_out,
               hipStream_t st, const char *who) {
    SQ_REQUIRE(img && workspace, "%s: null pointer", who);
    SQ_REQUIRE(N > 0 && H > 0 && W > 0 && H < G_INF && W < G_INF && (int64_t)N * H * W < ((int64_t)1 << 31),
               "%s: need 0 < H, W < %d and N*H*W < 2^31", who, G_INF);
    SQ_REQUIRE((((uintptr_t)workspace) & 15u) == 0, "%s: workspace must be 16-byte aligned", who);
    int *flag = reinterpret_cast<int *>(workspace);
    unsigned short *g = reinterpret_cast<unsigned short *>(flag + ((N + 3) / 4) * 4);
    if (hipMemsetAsync(flag, 0, sizeof(int) * N, st) != hipSuccess) {
        sq_set_error("%s: cannot clear the feature flags", who);
        return SQ_ELAUNCH;
    }
    const int rows = N * H;
    hipLaunchKernelGGL(edt_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, img, g, flag, rows, H, W);
    *g_out = g;
    *flag_out = flag;
    return sq_check_launch(who);
}

}  // namespace

extern "C" int64_t sq_weightmap_workspace(int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0 || (int64_t)N * H * W >= ((int64_t)1 << 31)) return -1;
    return (int64_t)((N + 3) / 4) * 16 + (int64_t)N * H * W * 2;
}

extern "C" int sq_edt_sq_f32(const float *img, int32_t *d2, void *workspace, int N, int H, int W, void *stream) {
    SQ_REQUIRE(d2, "sq_edt_sq_f32: null output");
    unsigned short *g;
    int *flag;
    hipStream_t st = (hipStream_t)stream;
    if (img && workspace && N > 0 && H > 0 && W > 0 && edt_v2_fits(H, W) && (int64_t)N * H * W < ((int64_t)1 << 31) &&
        (((uintptr_t)workspace) & 15u) == 0)
        return edt_v2<true>(img, reinterpret_cast<unsigned short *>(reinterpret_cast<int *>(workspace) + ((N + 3) / 4) * 4),
                            nullptr, nullptr, d2, N, H, W, 0.0, 1.0, st, "sq_edt_sq_f32");
    int rc = edt_common(img, workspace, N, H, W, &g, &flag, st, "sq_edt_sq_f32");
    if (rc) return rc;
    hipLaunchKernelGGL(edt_cols_weight_kernel<true>, dim3(wm_grid((int64_t)N * H * W)), dim3(256), 0, st, img, g, flag,
                       (double *)nullptr, (float *)nullptr, d2, N, H, W, 0.0, 1.0);
    return sq_check_launch("sq_edt_sq_f32");
}

extern "C" int sq_weightmap_edt_f32(const float *img, double *out64, float *out32, void *workspace, int N, int H, int W,
                                    double w0, double sigma, void *stream) {
    SQ_REQUIRE(out64 || out32, "sq_weightmap_edt_f32: no output requested");
    unsigned short *g;
    int *flag;
    hipStream_t st = (hipStream_t)stream;
    const double denom = 2.0 * (sigma * sigma) + 1e-99;         // 2.*self.sigma**2 + 1e-99
    if (img && workspace && N > 0 && H > 0 && W > 0 && edt_v2_fits(H, W) && (int64_t)N * H * W < ((int64_t)1 << 31) &&
        (((uintptr_t)workspace) & 15u) == 0)
        return edt_v2<false>(img, reinterpret_cast<unsigned short *>(reinterpret_cast<int *>(workspace) + ((N + 3) / 4) * 4),
                             out64, out32, nullptr, N, H, W, w0, denom, st, "sq_weightmap_edt_f32");
    int rc = edt_common(img, workspace, N, H, W, &g, &flag, st, "sq_weightmap_edt_f32");
    if (rc) return rc;
    hipLaunchKernelGGL(edt_cols_weight_kernel<false>, dim3(wm_grid((int64_t)N * H * W)), dim3(256), 0, st, img, g, flag,
                       out64, out32, (int *)nullptr, N, H, W, w0, denom);
    return sq_check_launch("sq_weightmap_edt_f32");
}

// ---------------------------------------------------------------------------------------------------------------
// ImageWeightMap2 (sequitr/pipeline.py:482-571): the Delaunay "narrowness" map.  The triangulation of the <= ~10^4
// boundary points is scipy's (Qhull) on the host -- 1 % of the reference's 2.7 s per tile and the only way to get
// its choice among co-circular lattice points; the per-PIXEL part that is the other 99 % runs here:
//   1. point location = rasterisation of the simplices: one wave per simplex walks its bounding box, a pixel is
//      inside when the three integer edge functions agree in sign (exact, 64-bit), and takes the simplex's longest
//      edge.  Where ONE simplex covers a pixel this is tri.find_simplex + edist of the reference exactly.  A pixel on
//      an edge or vertex is covered by several; scipy's walk returns whichever it reaches first (path-dependent), here
//      the largest longest-edge wins (atomicMax on the bit pattern of a positive double: order-independent);
//   2. weight_map = longest edge on background pixels (1024 where no simplex covers, :556), 0 on foreground;
//      gaussian_filter(sigma = 1) as scipy applies it: 9 taps (truncate 4), 'reflect' borders, axis 0 then axis 1;
//   3. w0 * (1 - mask) * exp(-(wm * wm) / (2 sigma^2 + 1e-99)) + 1 + mask, float64, operation by operation.
namespace {

__global__ __launch_bounds__(256) void wm2_raster_kernel(const int *__restrict__ simp, const double *__restrict__ longest,
                                                         int nsimp, u64 *__restrict__ cover, int H, int W) {
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= nsimp) return;
    const int *v = simp + (size_t)s * 7;                        // tile, x0, y0, x1, y1, x2, y2 (x = row, y = column as np.where)
    const int n = v[0];
    if (n < 0) return;                                          // a padding row of sq_delaunay2d_batch_i32
    const long long x0 = v[1], y0 = v[2], x1 = v[3], y1 = v[4], x2 = v[5], y2 = v[6];
    const int xmin = (int)min(x0, min(x1, x2)), xmax = (int)max(x0, max(x1, x2));
    const int ymin = (int)min(y0, min(y1, y2)), ymax = (int)max(y0, max(y1, y2));
    const int bw = ymax - ymin + 1, bh = xmax - xmin + 1;
    const u64 key = (u64)__double_as_longlong(longest[s]);
    u64 *img = cover + (size_t)n * H * W;
    for (int i = lane; i < bw * bh; i += 64) {
        const long long X = xmin + i / bw, Y = ymin + i % bw;
        if (X < 0 || X >= H || Y < 0 || Y >= W) continue;
        const long long e0 = (x1 - x0) * (Y - y0) - (y1 - y0) * (X - x0);
        const long long e1 = (x2 - x1) * (Y - y1) - (y2 - y1) * (X - x1);
        const long long e2 = (x0 - x2) * (Y - y2) - (y0 - y2) * (X - x2);
        if ((e0 >= 0 && e1 >= 0 && e2 >= 0) || (e0 <= 0 && e1 <= 0 && e2 <= 0)) atomicMax(img + X * W + Y, key);
    }
}

__device__ __forceinline__ int wm2_reflect(int i, int n) {     // scipy 'reflect': d c b a | a b c d | d c b a
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i - 1 : 2 * n - 1 - i;
    return i;
}

// pass along axis 0 (rows): builds the pre-filter map from (image, cover) on the fly, 9 taps, symmetric accumulation
// as ndimage.correlate1d does for symmetric weights: centre first, then pairs k = 1 .. 4
__global__ __launch_bounds__(256) void wm2_gauss0_kernel(const float *__restrict__ img, const u64 *__restrict__ cover,
                                                         double *__restrict__ tmp, int N, int H, int W, double w_0,
                                                         double w_1, double w_2, double w_3, double w_4) {
    const double wk[5] = {w_0, w_1, w_2, w_3, w_4};
    const int64_t total = (int64_t)N * H * W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int y = (int)(i % W);
        const int x = (int)((i / W) % H);
        const int64_t base = (i / ((int64_t)H * W)) * H * W;
        auto val = [&](int xx) {
            const int64_t j = base + (int64_t)wm2_reflect(xx, H) * W + y;
            if (img[j] != 0.f) return 0.0;                      // foreground: weight_map stays 0
            const u64 c = cover[j];
            return c ? __longlong_as_double((long long)c) : 1024.0;
        };
        double acc = val(x) * wk[0];
        for (int k = 4; k >= 1; --k) acc += (val(x - k) + val(x + k)) * wk[k];    // ni_filters.c: ii = -size1 .. -1
        tmp[i] = acc;
    }
}

__global__ __launch_bounds__(256) void wm2_gauss1_weight_kernel(const float *__restrict__ img, const double *__restrict__ tmp,
                                                                double *__restrict__ out64, float *__restrict__ out32, int N,
                                                                int H, int W, double w_0, double w_1, double w_2, double w_3,
                                                                double w_4, double wsum, double w0, double denom) {
    const double wk[5] = {w_0, w_1, w_2, w_3, w_4};
    const int64_t total = (int64_t)N * H * W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int y = (int)(i % W);
        const int64_t row = i - y;
        double acc = tmp[i] * wk[0];
        for (int k = 4; k >= 1; --k) acc += (tmp[row + wm2_reflect(y - k, W)] + tmp[row + wm2_reflect(y + k, W)]) * wk[k];
        const double wm = acc * wsum;                           // the singleton channel axis: every tap reads the same value
        const double mask = (double)(img[i] != 0.f ? 1.0f : 0.0f);
        const double r = w0 * (1.0 - mask) * exp(-(wm * wm) / denom) + 1.0 + mask;
        if (out64) out64[i] = r;
        if (out32) out32[i] = (float)r;
    }
}

}  // namespace

extern "C" int64_t sq_weightmap2_workspace(int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0 || (int64_t)N * H * W >= ((int64_t)1 << 31)) return -1;
    return (int64_t)N * H * W * 16;                             // cover (u64) + one filter pass (f64)
}

extern "C" int sq_weightmap2_delaunay_f32(const float *img, const int32_t *simplices, const double *longest, int nsimp,
                                          double *out64, float *out32, void *workspace, int N, int H, int W, double w0,
                                          double sigma, void *stream) {
    SQ_REQUIRE(img && simplices && longest && workspace, "sq_weightmap2_delaunay_f32: null pointer");
    SQ_REQUIRE(out64 || out32, "sq_weightmap2_delaunay_f32: no output requested");
    SQ_REQUIRE(nsimp > 0 && sq_weightmap2_workspace(N, H, W) > 0, "sq_weightmap2_delaunay_f32: bad shape");
    hipStream_t st = (hipStream_t)stream;
    u64 *cover = (u64 *)workspace;
    double *tmp = (double *)workspace + (size_t)N * H * W;
    const hipError_t e = hipMemsetAsync(cover, 0, (size_t)N * H * W * 8, st);
    SQ_REQUIRE(e == hipSuccess, "sq_weightmap2_delaunay_f32: memset failed: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(wm2_raster_kernel, dim3((nsimp + 3) / 4), dim3(256), 0, st, simplices, longest, nsimp, cover, H, W);
    int rc = sq_check_launch("sq_weightmap2_delaunay_f32(raster)");
    if (rc) return rc;
    // scipy.ndimage.gaussian_filter(sigma = 1.0): radius int(4.0 * 1.0 + 0.5) = 4, exp(-0.5 x^2) normalised by its sum
    double w[5], sum = 0.0;
    for (int k = -4; k <= 4; ++k) sum += exp(-0.5 * (double)(k * k));
    for (int k = 0; k <= 4; ++k) w[k] = exp(-0.5 * (double)(k * k)) / sum;
    double wsum = 0.0;                                          // what the pass over the length-1 channel axis multiplies by
    for (int k = -4; k <= 4; ++k) wsum += w[k < 0 ? -k : k];
    const unsigned grid = wm_grid((int64_t)N * H * W);
    hipLaunchKernelGGL(wm2_gauss0_kernel, dim3(grid), dim3(256), 0, st, img, cover, tmp, N, H, W, w[0], w[1], w[2], w[3], w[4]);
    rc = sq_check_launch("sq_weightmap2_delaunay_f32(gauss axis 0)");
    if (rc) return rc;
    const double denom = 2.0 * (sigma * sigma) + 1e-99;
    hipLaunchKernelGGL(wm2_gauss1_weight_kernel, dim3(grid), dim3(256), 0, st, img, tmp, out64, out32, N, H, W, w[0], w[1], w[2],
                       w[3], w[4], wsum, w0, denom);
    return sq_check_launch("sq_weightmap2_delaunay_f32(gauss axis 1 + weights)");
}
