// HBM-bound NHWC f32 ops of the sequitr hot path: pooling, bridge, argmax, pixel-norm,
// nearest-neighbour up-sampling.  16 B per lane everywhere (float4), grid-stride,
// capped at 256 CUs x 8 blocks.
#include <stdarg.h>
#include <string.h>
#include "sq_common.h"

// ---- error string (thread-local) ---------------------------------------------------
static thread_local char g_err[512] = "";

void sq_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char *sq_last_error(void) { return g_err; }
extern "C" int sq_version(void) { return 100; }  // 0.1.0

namespace {

inline unsigned grid_for(int64_t work_items) {
    int64_t b = (work_items + 255) / 256;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (unsigned)b;
}

__device__ __forceinline__ float4 max4(float4 a, float4 b) {
    return make_float4(b.x > a.x ? b.x : a.x, b.y > a.y ? b.y : a.y, b.z > a.z ? b.z : a.z,
                       b.w > a.w ? b.w : a.w);
}

// 2x2 stride-2 pooling; one thread = one output pixel x 4 channels.
template <bool AVG>
__global__ __launch_bounds__(256) void pool2x2_f32_kernel(const float4 *__restrict__ x,
                                                           float4 *__restrict__ y, int N, int H,
                                                           int W, int C4) {
    const int Ho = H >> 1, Wo = W >> 1;
    const int64_t total = (int64_t)N * Ho * Wo * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C4);
        int64_t t = i / C4;
        const int xo = (int)(t % Wo);
        t /= Wo;
        const int yo = (int)(t % Ho);
        const int n = (int)(t / Ho);
        const int64_t base = (((int64_t)n * H + 2 * yo) * W + 2 * xo) * C4 + c;
        const float4 a = x[base], b = x[base + C4], d = x[base + (int64_t)W * C4],
                     e = x[base + (int64_t)W * C4 + C4];
        float4 r;
        if (AVG) {
            r = make_float4(((a.x + b.x) + (d.x + e.x)) * 0.25f, ((a.y + b.y) + (d.y + e.y)) * 0.25f,
                            ((a.z + b.z) + (d.z + e.z)) * 0.25f, ((a.w + b.w) + (d.w + e.w)) * 0.25f);
        } else {
            r = max4(max4(max4(a, b), d), e);
        }
        y[i] = r;
    }
}

__global__ __launch_bounds__(256) void bridge_f32_kernel(const float4 *__restrict__ a,
                                                          const float4 *__restrict__ b,
                                                          float4 *__restrict__ y, int64_t n4, int op) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 u = a[i], v = b[i];
        float4 r;
        if (op == SQ_BRIDGE_ADD) r = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
        else if (op == SQ_BRIDGE_MUL) r = make_float4(u.x * v.x, u.y * v.y, u.z * v.z, u.w * v.w);
        else if (op == SQ_BRIDGE_SUB) r = make_float4(u.x - v.x, u.y - v.y, u.z - v.z, u.w - v.w);
        else r = u;
        y[i] = r;
    }
}

__global__ __launch_bounds__(256) void argmax_u8_kernel(const float *__restrict__ z,
                                                         uint8_t *__restrict__ m, int64_t npix, int C) {
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (int64_t)gridDim.x * 256) {
        const float *zp = z + p * C;
        float bv = zp[0];
        int best = 0;
        for (int c = 1; c < C; ++c) {
            const float v = zp[c];
            if (v > bv) { bv = v; best = c; }
        }
        m[p] = (uint8_t)best;
    }
}

// pixel_norm: one GL-lane group per pixel, GL = the power of two >= C / 4, at most 16 (64 / GL pixels per wave: with
// 16 lanes per pixel the 8- and 16-channel layers of the GAN, its largest tensors, ran on 2 or 4 lanes in 16);
// sequential-in-c f32 sum is
// NOT reproduced lane-parallel, so the sum order is fixed as: each lane sums its own
// channels c = 4*(lane in group) + 4*GL*k + {0..3} in order, then an xor butterfly over the group (the same value
// whatever GL: lanes without channels contribute zeros).
// The oracle comparison for this op is therefore a tolerance (1e-6 rel), not bit-exact.
template <int GL>
__global__ __launch_bounds__(256) void pixelnorm_f32_kernel(const float *__restrict__ x,
                                                             float *__restrict__ y, int64_t npix,
                                                             int C, float eps) {
    constexpr int PPW = 64 / GL;
    const int lane = threadIdx.x & 63, l16 = lane & (GL - 1), sub = lane / GL;
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * 256) >> 6;
    // wave-uniform trip count: the shuffles below always see a full wave
    for (int64_t pb = wave * PPW; pb < npix; pb += nwaves * PPW) {
        const int64_t p = pb + sub;
        const bool live = p < npix;
        float s = 0.f;
        if (live)
            for (int c = 4 * l16; c < C; c += 4 * GL) {
                const float4 v = *reinterpret_cast<const float4 *>(x + p * C + c);
                s = __builtin_fmaf(v.x, v.x, s); s = __builtin_fmaf(v.y, v.y, s);
                s = __builtin_fmaf(v.z, v.z, s); s = __builtin_fmaf(v.w, v.w, s);
            }
#pragma unroll
        for (int m = 1; m < GL; m <<= 1) s += __shfl_xor(s, m);
        const float r = 1.0f / __builtin_sqrtf(s / (float)C + eps);
        if (live)
            for (int c = 4 * l16; c < C; c += 4 * GL) {
                float4 v = *reinterpret_cast<const float4 *>(x + p * C + c);
                v.x *= r; v.y *= r; v.z *= r; v.w *= r;
                *reinterpret_cast<float4 *>(y + p * C + c) = v;
            }
    }
}

__global__ __launch_bounds__(256) void upsample_nn2x_f32_kernel(const float4 *__restrict__ x,
                                                                 float4 *__restrict__ y, int N, int H,
                                                                 int W, int C4) {
    const int Ho = 2 * H, Wo = 2 * W;
    const int64_t total = (int64_t)N * Ho * Wo * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C4);
        int64_t t = i / C4;
        const int xo = (int)(t % Wo);
        t /= Wo;
        const int yo = (int)(t % Ho);
        const int n = (int)(t / Ho);
        y[i] = x[(((int64_t)n * H + (yo >> 1)) * W + (xo >> 1)) * C4 + c];
    }
}

}  // namespace

extern "C" int sq_maxpool2x2_fwd_f32(const float *x, float *y, int N, int H, int W, int C, void *stream) {
    SQ_REQUIRE(x && y, "sq_maxpool2x2_fwd_f32: null tensor pointer");
    SQ_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && H % 2 == 0 && W % 2 == 0 && C % 4 == 0,
               "sq_maxpool2x2_fwd_f32: need even H,W and C %% 4 == 0 (got %d,%d,%d)", H, W, C);
    SQ_REQUIRE_ALIGNED(x); SQ_REQUIRE_ALIGNED(y);
    const int64_t total = (int64_t)N * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(pool2x2_f32_kernel<false>, dim3(grid_for(total)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const float4 *>(x),
                       reinterpret_cast<float4 *>(y), N, H, W, C / 4);
    return sq_check_launch("sq_maxpool2x2_fwd_f32");
}

extern "C" int sq_avgpool2x2_fwd_f32(const float *x, float *y, int N, int H, int W, int C, void *stream) {
    SQ_REQUIRE(x && y, "sq_avgpool2x2_fwd_f32: null tensor pointer");
    SQ_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && H % 2 == 0 && W % 2 == 0 && C % 4 == 0,
               "sq_avgpool2x2_fwd_f32: need even H,W and C %% 4 == 0 (got %d,%d,%d)", H, W, C);
    SQ_REQUIRE_ALIGNED(x); SQ_REQUIRE_ALIGNED(y);
    const int64_t total = (int64_t)N * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(pool2x2_f32_kernel<true>, dim3(grid_for(total)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const float4 *>(x),
                       reinterpret_cast<float4 *>(y), N, H, W, C / 4);
    return sq_check_launch("sq_avgpool2x2_fwd_f32");
}

extern "C" int sq_bridge_fwd_f32(const float *a, const float *b, float *y, int64_t n, int bridge, void *stream) {
    SQ_REQUIRE(a && b && y, "sq_bridge_fwd_f32: null tensor pointer");
    SQ_REQUIRE(n > 0 && n % 4 == 0, "sq_bridge_fwd_f32: n must be a positive multiple of 4");
    SQ_REQUIRE(bridge >= SQ_BRIDGE_NONE && bridge <= SQ_BRIDGE_SUB, "sq_bridge_fwd_f32: bad bridge %d", bridge);
    SQ_REQUIRE_ALIGNED(a); SQ_REQUIRE_ALIGNED(b); SQ_REQUIRE_ALIGNED(y);
    hipLaunchKernelGGL(bridge_f32_kernel, dim3(grid_for(n / 4)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const float4 *>(a),
                       reinterpret_cast<const float4 *>(b), reinterpret_cast<float4 *>(y), n / 4, bridge);
    return sq_check_launch("sq_bridge_fwd_f32");
}

extern "C" int sq_argmax_u8(const float *logits, uint8_t *mask, int64_t npix, int C, void *stream) {
    SQ_REQUIRE(logits && mask, "sq_argmax_u8: null tensor pointer");
    SQ_REQUIRE(npix > 0 && C > 0 && C <= 255, "sq_argmax_u8: bad shape");
    hipLaunchKernelGGL(argmax_u8_kernel, dim3(grid_for(npix)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), logits, mask, npix, C);
    return sq_check_launch("sq_argmax_u8");
}

extern "C" int sq_pixelnorm_fwd_f32(const float *x, float *y, int64_t npix, int C, float eps, void *stream) {
    SQ_REQUIRE(x && y, "sq_pixelnorm_fwd_f32: null tensor pointer");
    SQ_REQUIRE(npix > 0 && C > 0 && C % 4 == 0, "sq_pixelnorm_fwd_f32: C=%d must be a multiple of 4", C);
    SQ_REQUIRE_ALIGNED(x); SQ_REQUIRE_ALIGNED(y);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (C <= 4) hipLaunchKernelGGL(pixelnorm_f32_kernel<1>, dim3(grid_for(npix)), dim3(256), 0, st, x, y, npix, C, eps);
    else if (C <= 8) hipLaunchKernelGGL(pixelnorm_f32_kernel<2>, dim3(grid_for(npix * 2)), dim3(256), 0, st, x, y, npix, C, eps);
    else if (C <= 16) hipLaunchKernelGGL(pixelnorm_f32_kernel<4>, dim3(grid_for(npix * 4)), dim3(256), 0, st, x, y, npix, C, eps);
    else if (C <= 32) hipLaunchKernelGGL(pixelnorm_f32_kernel<8>, dim3(grid_for(npix * 8)), dim3(256), 0, st, x, y, npix, C, eps);
    else hipLaunchKernelGGL(pixelnorm_f32_kernel<16>, dim3(grid_for(npix * 16)), dim3(256), 0, st, x, y, npix, C, eps);
    return sq_check_launch("sq_pixelnorm_fwd_f32");
}

extern "C" int sq_upsample_nn2x_f32(const float *x, float *y, int N, int H, int W, int C, void *stream) {
    SQ_REQUIRE(x && y, "sq_upsample_nn2x_f32: null tensor pointer");
    SQ_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "sq_upsample_nn2x_f32: C=%d must be a multiple of 4", C);
    SQ_REQUIRE_ALIGNED(x); SQ_REQUIRE_ALIGNED(y);
    const int64_t total = (int64_t)N * (2 * H) * (2 * W) * (C / 4);
    hipLaunchKernelGGL(upsample_nn2x_f32_kernel, dim3(grid_for(total)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const float4 *>(x),
                       reinterpret_cast<float4 *>(y), N, H, W, C / 4);
    return sq_check_launch("sq_upsample_nn2x_f32");
}
