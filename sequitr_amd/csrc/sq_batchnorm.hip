// Batch normalisation between a convolution and its activation (optional `batch_norm`
// of conv_layer, tf.layers.batch_normalization defaults: eps 1e-3, momentum 0.99).  NHWC, per-channel
// statistics over N*H*W.  Streaming kernels, HBM-bound: every pass moves 4 B/element in (+ out).
//
//   stats : mean[c], population var[c]; fp64 accumulation, block partials + fixed-order finish
//   fold  : scale = gamma / sqrtf(var + eps), shift = fmaf(-mean, scale, beta)
//   apply : y = act(fmaf(x, scale[c], shift[c]))
//   bwd   : d = act'(y) * dy;  dbeta = sum d;  dgamma = sum d * xhat;
//           dx = gamma * r * (d - (dbeta + xhat * dgamma) / M),  xhat = (x - mean) * r
#include "sq_common.h"

namespace {

constexpr int BN_THREADS = 256;

struct BnGeom {
    int cg;      // float4 channel groups per pixel (C / 4)
    int rows;    // pixels a block covers per pass (BN_THREADS / cg)
};

__device__ __forceinline__ float bn_dact(float dy, float y, int act) {
    if (act == SQ_ACT_RELU) return y > 0.f ? dy : 0.f;
    if (act == SQ_ACT_LEAKY) return y > 0.f ? dy : 0.2f * dy;
    return dy;
}

// partials layout: [block][2][C] doubles (first moment / second moment, or dbeta / dgamma)
template <bool BWD>
__global__ __launch_bounds__(BN_THREADS) void bn_reduce_kernel(const float4 *__restrict__ x,
                                                               const float4 *__restrict__ dy,
                                                               const float4 *__restrict__ yact, int act,
                                                               const float *__restrict__ mean,
                                                               const float *__restrict__ var, float eps,
                                                               double *__restrict__ partials, int64_t npix, int C,
                                                               BnGeom g) {
    __shared__ double red[2][BN_THREADS][4];
    const int t = threadIdx.x;
    const int c4 = t % g.cg, row = t / g.cg;
    double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
    float mu[4] = {0, 0, 0, 0}, r[4] = {1, 1, 1, 1};
    if (BWD && row < g.rows) {
        for (int j = 0; j < 4; ++j) {
            mu[j] = mean[4 * c4 + j];
            r[j] = 1.0f / sqrtf(var[4 * c4 + j] + eps);
        }
    }
    if (row < g.rows) {
        for (int64_t p = (int64_t)blockIdx.x * g.rows + row; p < npix; p += (int64_t)gridDim.x * g.rows) {
            const float4 v = x[p * g.cg + c4];
            const float xv[4] = {v.x, v.y, v.z, v.w};
            if (BWD) {
                const float4 d4 = dy[p * g.cg + c4];
                float dv[4] = {d4.x, d4.y, d4.z, d4.w};
                if (yact) {
                    const float4 y4 = yact[p * g.cg + c4];
                    const float yv[4] = {y4.x, y4.y, y4.z, y4.w};
                    for (int j = 0; j < 4; ++j) dv[j] = bn_dact(dv[j], yv[j], act);
                }
                for (int j = 0; j < 4; ++j) {
                    s0[j] += (double)dv[j];
                    s1[j] += (double)(dv[j] * ((xv[j] - mu[j]) * r[j]));
                }
            } else {
                for (int j = 0; j < 4; ++j) {
                    s0[j] += (double)xv[j];
                    s1[j] += (double)xv[j] * (double)xv[j];
                }
            }
        }
    }
    for (int j = 0; j < 4; ++j) {
        red[0][t][j] = s0[j];
        red[1][t][j] = s1[j];
    }
    __syncthreads();
    // fixed order: thread (c4, j, which) sums the rows 0..rows-1
    for (int o = t; o < 2 * C; o += BN_THREADS) {
        const int which = o / C, c = o % C;
        double s = 0.0;
        for (int rr = 0; rr < g.rows; ++rr) s += red[which][rr * g.cg + (c >> 2)][c & 3];
        partials[((size_t)blockIdx.x * 2 + which) * C + c] = s;
    }
}

__global__ void bn_stats_finish_kernel(const double *__restrict__ partials, int nblk, int C, int64_t npix,
                                       float *__restrict__ mean, float *__restrict__ var) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, q = 0.0;
    for (int b = 0; b < nblk; ++b) {
        s += partials[((size_t)b * 2 + 0) * C + c];
        q += partials[((size_t)b * 2 + 1) * C + c];
    }
    const double m = s / (double)npix;
    double v = q / (double)npix - m * m;
    mean[c] = (float)m;
    var[c] = (float)(v > 0.0 ? v : 0.0);
}

__global__ void bn_bwd_finish_kernel(const double *__restrict__ partials, int nblk, int C,
                                     const float *__restrict__ var, float eps, float *__restrict__ dgamma,
                                     float *__restrict__ dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, q = 0.0;
    for (int b = 0; b < nblk; ++b) {
        s += partials[((size_t)b * 2 + 0) * C + c];
        q += partials[((size_t)b * 2 + 1) * C + c];
    }
    dbeta[c] = (float)s;
    dgamma[c] = (float)q;
}

__global__ void bn_fold_kernel(const float *__restrict__ gamma, const float *__restrict__ beta,
                               const float *__restrict__ mean, const float *__restrict__ var, float eps,
                               float *__restrict__ scale, float *__restrict__ shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float s = gamma[c] / sqrtf(var[c] + eps);
    scale[c] = s;
    shift[c] = fmaf(-mean[c], s, beta[c]);
}

__global__ void bn_update_moving_kernel(float *__restrict__ mmean, float *__restrict__ mvar,
                                        const float *__restrict__ mean, const float *__restrict__ var,
                                        float momentum, float unbias, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    // tf.layers.batch_normalization: moving -= (moving - batch) * (1 - momentum)
    mmean[c] -= (mmean[c] - mean[c]) * (1.0f - momentum);
    mvar[c] -= (mvar[c] - var[c] * unbias) * (1.0f - momentum);
}

__global__ __launch_bounds__(BN_THREADS) void bn_apply_kernel(const float4 *__restrict__ x,
                                                              const float *__restrict__ scale,
                                                              const float *__restrict__ shift,
                                                              float4 *__restrict__ y, int64_t n4, int cg, int act) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = 4 * (int)(i % cg);
        const float4 v = x[i];
        float4 o;
        o.x = sq_act(fmaf(v.x, scale[c + 0], shift[c + 0]), act);
        o.y = sq_act(fmaf(v.y, scale[c + 1], shift[c + 1]), act);
        o.z = sq_act(fmaf(v.z, scale[c + 2], shift[c + 2]), act);
        o.w = sq_act(fmaf(v.w, scale[c + 3], shift[c + 3]), act);
        y[i] = o;
    }
}

__global__ __launch_bounds__(BN_THREADS) void bn_bwd_apply_kernel(
    const float4 *__restrict__ x, const float4 *__restrict__ dy, const float4 *__restrict__ yact, int act,
    const float *__restrict__ mean, const float *__restrict__ var, const float *__restrict__ gamma, float eps,
    const float *__restrict__ dgamma, const float *__restrict__ dbeta, float4 *__restrict__ dx, int64_t n4, int cg,
    float inv_m) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = 4 * (int)(i % cg);
        const float4 v = x[i], d4 = dy[i];
        const float xv[4] = {v.x, v.y, v.z, v.w};
        float dv[4] = {d4.x, d4.y, d4.z, d4.w};
        if (yact) {
            const float4 y4 = yact[i];
            const float yv[4] = {y4.x, y4.y, y4.z, y4.w};
            for (int j = 0; j < 4; ++j) dv[j] = bn_dact(dv[j], yv[j], act);
        }
        float o[4];
        for (int j = 0; j < 4; ++j) {
            const float r = 1.0f / sqrtf(var[c + j] + eps);
            const float xh = (xv[j] - mean[c + j]) * r;
            o[j] = gamma[c + j] * r * (dv[j] - (dbeta[c + j] + xh * dgamma[c + j]) * inv_m);
        }
        dx[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

inline bool bn_geom(int C, BnGeom *g) {
    if (C < 4 || (C & 3) || C / 4 > BN_THREADS) return false;
    g->cg = C / 4;
    g->rows = BN_THREADS / g->cg;
    return true;
}

inline int bn_grid(int64_t npix, const BnGeom &g) {
    int64_t nb = (npix + g.rows - 1) / g.rows;
    return (int)(nb < 1 ? 1 : (nb > 1024 ? 1024 : nb));
}

inline int stream_grid(int64_t n) {
    int64_t nb = (n + BN_THREADS - 1) / BN_THREADS;
    return (int)(nb < 1 ? 1 : (nb > 4096 ? 4096 : nb));
}

}  // namespace

#define SQ_ST(s) ((hipStream_t)(s))

extern "C" int64_t sq_bn_workspace_f32(int64_t npix, int C) {
    BnGeom g;
    if (npix <= 0 || !bn_geom(C, &g)) return -1;
    return (int64_t)bn_grid(npix, g) * 2 * C * (int64_t)sizeof(double);
}

extern "C" int sq_bn_stats_f32(const float *x, float *mean, float *var, void *workspace, int64_t npix, int C,
                               void *stream) {
    BnGeom g;
    SQ_REQUIRE(x && mean && var && workspace && npix > 0, "sq_bn_stats_f32: bad arguments");
    SQ_REQUIRE(bn_geom(C, &g), "sq_bn_stats_f32: C=%d must be a multiple of 4, <= 1024", C);
    SQ_REQUIRE_ALIGNED(x);
    SQ_REQUIRE((((uintptr_t)workspace) & 7u) == 0, "sq_bn_stats_f32: workspace must be 8-byte aligned");
    const int nblk = bn_grid(npix, g);
    hipLaunchKernelGGL(bn_reduce_kernel<false>, dim3(nblk), dim3(BN_THREADS), 0, SQ_ST(stream),
                       reinterpret_cast<const float4 *>(x), (const float4 *)nullptr, (const float4 *)nullptr, 0,
                       (const float *)nullptr, (const float *)nullptr, 0.f, reinterpret_cast<double *>(workspace), npix,
                       C, g);
    int rc = sq_check_launch("sq_bn_stats_f32(reduce)");
    if (rc) return rc;
    hipLaunchKernelGGL(bn_stats_finish_kernel, dim3((C + 63) / 64), dim3(64), 0, SQ_ST(stream),
                       reinterpret_cast<const double *>(workspace), nblk, C, npix, mean, var);
    return sq_check_launch("sq_bn_stats_f32");
}

extern "C" int sq_bn_fold_f32(const float *gamma, const float *beta, const float *mean, const float *var, float eps,
                              float *scale, float *shift, int C, void *stream) {
    SQ_REQUIRE(gamma && beta && mean && var && scale && shift && C > 0, "sq_bn_fold_f32: bad arguments");
    hipLaunchKernelGGL(bn_fold_kernel, dim3((C + 63) / 64), dim3(64), 0, SQ_ST(stream), gamma, beta, mean, var, eps,
                       scale, shift, C);
    return sq_check_launch("sq_bn_fold_f32");
}

extern "C" int sq_bn_update_moving_f32(float *moving_mean, float *moving_var, const float *mean, const float *var,
                                       float momentum, int64_t npix, int C, void *stream) {
    SQ_REQUIRE(moving_mean && moving_var && mean && var && C > 0 && npix > 0, "sq_bn_update_moving_f32: bad arguments");
    const float unbias = npix > 1 ? (float)((double)npix / (double)(npix - 1)) : 1.0f;
    hipLaunchKernelGGL(bn_update_moving_kernel, dim3((C + 63) / 64), dim3(64), 0, SQ_ST(stream), moving_mean,
                       moving_var, mean, var, momentum, unbias, C);
    return sq_check_launch("sq_bn_update_moving_f32");
}

extern "C" int sq_bn_apply_f32(const float *x, const float *scale, const float *shift, float *y, int64_t npix, int C,
                               int act, void *stream) {
    SQ_REQUIRE(x && scale && shift && y && npix > 0, "sq_bn_apply_f32: bad arguments");
    SQ_REQUIRE(C >= 4 && (C & 3) == 0, "sq_bn_apply_f32: C=%d must be a multiple of 4", C);
    SQ_REQUIRE(act >= SQ_ACT_NONE && act <= SQ_ACT_LEAKY, "sq_bn_apply_f32: bad activation %d", act);
    SQ_REQUIRE_ALIGNED(x);
    SQ_REQUIRE_ALIGNED(y);
    const int64_t n4 = npix * (C / 4);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(stream_grid(n4)), dim3(BN_THREADS), 0, SQ_ST(stream),
                       reinterpret_cast<const float4 *>(x), scale, shift, reinterpret_cast<float4 *>(y), n4, C / 4, act);
    return sq_check_launch("sq_bn_apply_f32");
}

extern "C" int sq_bn_bwd_f32(const float *x, const float *dy, const float *y_act, int act, const float *mean,
                             const float *var, const float *gamma, float eps, float *dx, float *dgamma, float *dbeta,
                             void *workspace, int64_t npix, int C, void *stream) {
    BnGeom g;
    SQ_REQUIRE(x && dy && mean && var && gamma && dx && dgamma && dbeta && workspace && npix > 0,
               "sq_bn_bwd_f32: bad arguments");
    SQ_REQUIRE(bn_geom(C, &g), "sq_bn_bwd_f32: C=%d must be a multiple of 4, <= 1024", C);
    SQ_REQUIRE(act >= SQ_ACT_NONE && act <= SQ_ACT_LEAKY, "sq_bn_bwd_f32: bad activation %d", act);
    SQ_REQUIRE(act == SQ_ACT_NONE || y_act, "sq_bn_bwd_f32: y_act is needed to differentiate the activation");
    SQ_REQUIRE_ALIGNED(x);
    SQ_REQUIRE_ALIGNED(dy);
    SQ_REQUIRE_ALIGNED(dx);
    if (act == SQ_ACT_NONE) y_act = nullptr;
    const int nblk = bn_grid(npix, g);
    hipLaunchKernelGGL(bn_reduce_kernel<true>, dim3(nblk), dim3(BN_THREADS), 0, SQ_ST(stream),
                       reinterpret_cast<const float4 *>(x), reinterpret_cast<const float4 *>(dy),
                       reinterpret_cast<const float4 *>(y_act), act, mean, var, eps,
                       reinterpret_cast<double *>(workspace), npix, C, g);
    int rc = sq_check_launch("sq_bn_bwd_f32(reduce)");
    if (rc) return rc;
    hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3((C + 63) / 64), dim3(64), 0, SQ_ST(stream),
                       reinterpret_cast<const double *>(workspace), nblk, C, var, eps, dgamma, dbeta);
    rc = sq_check_launch("sq_bn_bwd_f32(finish)");
    if (rc) return rc;
    const int64_t n4 = npix * (C / 4);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(stream_grid(n4)), dim3(BN_THREADS), 0, SQ_ST(stream),
                       reinterpret_cast<const float4 *>(x), reinterpret_cast<const float4 *>(dy),
                       reinterpret_cast<const float4 *>(y_act), act, mean, var, gamma, eps, dgamma, dbeta,
                       reinterpret_cast<float4 *>(dx), n4, C / 4, (float)(1.0 / (double)npix));
    return sq_check_launch("sq_bn_bwd_f32");
}
