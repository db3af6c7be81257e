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
    int rc = edt_common(img, workspace, N, H, W, &g, &flag, st, "sq_weightmap_edt_f32");
    if (rc) return rc;
    const double denom = 2.0 * (sigma * sigma) + 1e-99;         // 2.*self.sigma**2 + 1e-99
    hipLaunchKernelGGL(edt_cols_weight_kernel<false>, dim3(wm_grid((int64_t)N * H * W)), dim3(256), 0, st, img, g, flag,
                       out64, out32, (int *)nullptr, N, H, W, w0, denom);
    return sq_check_launch("sq_weightmap_edt_f32");
}
