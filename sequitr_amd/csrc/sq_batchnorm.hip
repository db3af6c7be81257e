// Batch normalisation between a convolution and its activation (optional `batch_norm`
// of conv_layer, tf.layers.batch_normalization defaults: eps 1e-3, momentum 0.99).  NHWC, per-channel
// statistics over N*H*W.  Streaming kernels, HBM-bound: every pass moves 4 B/element in (+ out).
//
//   stats : mean[c], population var[c]; fp64 accumulation, block partials + fixed-order finish
//   fold  : scale = gamma / sqrtf(var + eps), shift = fmaf(-mean, scale, beta)
//   apply : y = act(fmaf(x, scale[c], shift[c]))
//   bwd   : d = act'(y) * dy;  dbeta = sum d;  dgamma = sum d * xhat;
//           dx = gamma * r * (d - (dbeta + xhat * dgamma) / M),  xhat = (x - mean) * r
//
// The same kernels on bf16 tensors (round 3: `batch_norm` in the bf16 training graph, BASELINE configs 3-4): elements are
// read as bf16 and widened, every statistic / parameter / intermediate stays f32 (sums f64), results are rounded to bf16
// (RNE) once where they are stored -- the rounding points oracle/bf16_ref.py emulates.
#include "sq_common.h"

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BN_THREADS = 256;

// four consecutive channels of one pixel in HBM: float4 or four bf16
template <typename T> struct Quad;
template <> struct Quad<float> {
    typedef float4 V;
    static __device__ __forceinline__ void load(const V *p, int64_t i, float (&o)[4]) {
        const float4 v = p[i];
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
    static __device__ __forceinline__ void store(V *p, int64_t i, const float (&o)[4]) { p[i] = make_float4(o[0], o[1], o[2], o[3]); }
};
template <> struct Quad<__bf16> {
    typedef bf16x4 V;
    static __device__ __forceinline__ void load(const V *p, int64_t i, float (&o)[4]) {
        const bf16x4 v = p[i];
        o[0] = (float)v[0]; o[1] = (float)v[1]; o[2] = (float)v[2]; o[3] = (float)v[3];
    }
    static __device__ __forceinline__ void store(V *p, int64_t i, const float (&o)[4]) {
        p[i] = (bf16x4){(__bf16)o[0], (__bf16)o[1], (__bf16)o[2], (__bf16)o[3]};
    }
};

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
template <bool BWD, typename T>
__global__ __launch_bounds__(BN_THREADS) void bn_reduce_kernel(const typename Quad<T>::V *__restrict__ x,
                                                               const typename Quad<T>::V *__restrict__ dy,
                                                               const typename Quad<T>::V *__restrict__ yact, int act,
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
            float xv[4];
            Quad<T>::load(x, p * g.cg + c4, xv);
            if (BWD) {
                float dv[4];
                Quad<T>::load(dy, p * g.cg + c4, dv);
                if (yact) {
                    float yv[4];
                    Quad<T>::load(yact, p * g.cg + c4, yv);
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

template <typename T>
__global__ __launch_bounds__(BN_THREADS) void bn_apply_kernel(const typename Quad<T>::V *__restrict__ x,
                                                              const float *__restrict__ scale,
                                                              const float *__restrict__ shift,
                                                              typename Quad<T>::V *__restrict__ y, int64_t n4, int cg, int act) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = 4 * (int)(i % cg);
        float v[4], o[4];
        Quad<T>::load(x, i, v);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = sq_act(fmaf(v[j], scale[c + j], shift[c + j]), act);
        Quad<T>::store(y, i, o);
    }
}

template <typename T>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_apply_kernel(
    const typename Quad<T>::V *__restrict__ x, const typename Quad<T>::V *__restrict__ dy,
    const typename Quad<T>::V *__restrict__ yact, int act,
    const float *__restrict__ mean, const float *__restrict__ var, const float *__restrict__ gamma, float eps,
    const float *__restrict__ dgamma, const float *__restrict__ dbeta, typename Quad<T>::V *__restrict__ dx, int64_t n4, int cg,
    float inv_m) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = 4 * (int)(i % cg);
        float xv[4], dv[4];
        Quad<T>::load(x, i, xv);
        Quad<T>::load(dy, i, dv);
        if (yact) {
            float yv[4];
            Quad<T>::load(yact, i, yv);
            for (int j = 0; j < 4; ++j) dv[j] = bn_dact(dv[j], yv[j], act);
        }
        float o[4];
        for (int j = 0; j < 4; ++j) {
            const float r = 1.0f / sqrtf(var[c + j] + eps);
            const float xh = (xv[j] - mean[c + j]) * r;
            o[j] = gamma[c + j] * r * (dv[j] - (dbeta[c + j] + xh * dgamma[c + j]) * inv_m);
        }
        Quad<T>::store(dx, i, o);
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

template <typename T>
static int bn_stats_impl(const void *x, float *mean, float *var, void *workspace, int64_t npix, int C, void *stream,
                         const char *what) {
    typedef typename Quad<T>::V V;
    BnGeom g;
    SQ_REQUIRE(x && mean && var && workspace && npix > 0, "%s: bad arguments", what);
    SQ_REQUIRE(bn_geom(C, &g), "%s: C=%d must be a multiple of 4, <= 1024", what, C);
    SQ_REQUIRE((((uintptr_t)x) & (sizeof(V) - 1)) == 0, "%s: x is not aligned to %d bytes", what, (int)sizeof(V));
    SQ_REQUIRE((((uintptr_t)workspace) & 7u) == 0, "%s: workspace must be 8-byte aligned", what);
    const int nblk = bn_grid(npix, g);
    hipLaunchKernelGGL((bn_reduce_kernel<false, T>), dim3(nblk), dim3(BN_THREADS), 0, SQ_ST(stream),
                       reinterpret_cast<const V *>(x), (const V *)nullptr, (const V *)nullptr, 0,
                       (const float *)nullptr, (const float *)nullptr, 0.f, reinterpret_cast<double *>(workspace), npix,
                       C, g);
    int rc = sq_check_launch(what);
    if (rc) return rc;
    hipLaunchKernelGGL(bn_stats_finish_kernel, dim3((C + 63) / 64), dim3(64), 0, SQ_ST(stream),
                       reinterpret_cast<const double *>(workspace), nblk, C, npix, mean, var);
    return sq_check_launch(what);
}

extern "C" int sq_bn_stats_f32(const float *x, float *mean, float *var, void *workspace, int64_t npix, int C,
                               void *stream) {
    return bn_stats_impl<float>(x, mean, var, workspace, npix, C, stream, "sq_bn_stats_f32");
}
extern "C" int sq_bn_stats_bf16(const void *x, float *mean, float *var, void *workspace, int64_t npix, int C,
                                void *stream) {
    return bn_stats_impl<__bf16>(x, mean, var, workspace, npix, C, stream, "sq_bn_stats_bf16");
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

template <typename T>
static int bn_apply_impl(const void *x, const float *scale, const float *shift, void *y, int64_t npix, int C, int act,
                         void *stream, const char *what) {
    typedef typename Quad<T>::V V;
    SQ_REQUIRE(x && scale && shift && y && npix > 0, "%s: bad arguments", what);
    SQ_REQUIRE(C >= 4 && (C & 3) == 0, "%s: C=%d must be a multiple of 4", what, C);
    SQ_REQUIRE(act >= SQ_ACT_NONE && act <= SQ_ACT_LEAKY, "%s: bad activation %d", what, act);
    SQ_REQUIRE(((((uintptr_t)x) | ((uintptr_t)y)) & (sizeof(V) - 1)) == 0, "%s: tensors not aligned to %d bytes", what, (int)sizeof(V));
    const int64_t n4 = npix * (C / 4);
    hipLaunchKernelGGL((bn_apply_kernel<T>), dim3(stream_grid(n4)), dim3(BN_THREADS), 0, SQ_ST(stream),
                       reinterpret_cast<const V *>(x), scale, shift, reinterpret_cast<V *>(y), n4, C / 4, act);
    return sq_check_launch(what);
}

extern "C" int sq_bn_apply_f32(const float *x, const float *scale, const float *shift, float *y, int64_t npix, int C,
                               int act, void *stream) {
    return bn_apply_impl<float>(x, scale, shift, y, npix, C, act, stream, "sq_bn_apply_f32");
}
extern "C" int sq_bn_apply_bf16(const void *x, const float *scale, const float *shift, void *y, int64_t npix, int C,
                                int act, void *stream) {
    return bn_apply_impl<__bf16>(x, scale, shift, y, npix, C, act, stream, "sq_bn_apply_bf16");
}

template <typename T>
static int bn_bwd_impl(const void *x, const void *dy, const void *y_act, int act, const float *mean, const float *var,
                       const float *gamma, float eps, void *dx, float *dgamma, float *dbeta, void *workspace, int64_t npix,
                       int C, void *stream, const char *what) {
    typedef typename Quad<T>::V V;
    BnGeom g;
    SQ_REQUIRE(x && dy && mean && var && gamma && dx && dgamma && dbeta && workspace && npix > 0, "%s: bad arguments", what);
    SQ_REQUIRE(bn_geom(C, &g), "%s: C=%d must be a multiple of 4, <= 1024", what, C);
    SQ_REQUIRE(act >= SQ_ACT_NONE && act <= SQ_ACT_LEAKY, "%s: bad activation %d", what, act);
    SQ_REQUIRE(act == SQ_ACT_NONE || y_act, "%s: y_act is needed to differentiate the activation", what);
    SQ_REQUIRE(((((uintptr_t)x) | ((uintptr_t)dy) | ((uintptr_t)dx)) & (sizeof(V) - 1)) == 0,
               "%s: tensors not aligned to %d bytes", what, (int)sizeof(V));
    if (act == SQ_ACT_NONE) y_act = nullptr;
    const int nblk = bn_grid(npix, g);
    hipLaunchKernelGGL((bn_reduce_kernel<true, T>), dim3(nblk), dim3(BN_THREADS), 0, SQ_ST(stream),
                       reinterpret_cast<const V *>(x), reinterpret_cast<const V *>(dy), reinterpret_cast<const V *>(y_act), act,
                       mean, var, eps, reinterpret_cast<double *>(workspace), npix, C, g);
    int rc = sq_check_launch(what);
    if (rc) return rc;
    hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3((C + 63) / 64), dim3(64), 0, SQ_ST(stream),
                       reinterpret_cast<const double *>(workspace), nblk, C, var, eps, dgamma, dbeta);
    rc = sq_check_launch(what);
    if (rc) return rc;
    const int64_t n4 = npix * (C / 4);
    hipLaunchKernelGGL((bn_bwd_apply_kernel<T>), dim3(stream_grid(n4)), dim3(BN_THREADS), 0, SQ_ST(stream),
                       reinterpret_cast<const V *>(x), reinterpret_cast<const V *>(dy), reinterpret_cast<const V *>(y_act), act,
                       mean, var, gamma, eps, dgamma, dbeta, reinterpret_cast<V *>(dx), n4, C / 4,
                       (float)(1.0 / (double)npix));
    return sq_check_launch(what);
}

extern "C" int sq_bn_bwd_f32(const float *x, const float *dy, const float *y_act, int act, const float *mean,
                             const float *var, const float *gamma, float eps, float *dx, float *dgamma, float *dbeta,
                             void *workspace, int64_t npix, int C, void *stream) {
    return bn_bwd_impl<float>(x, dy, y_act, act, mean, var, gamma, eps, dx, dgamma, dbeta, workspace, npix, C, stream,
                              "sq_bn_bwd_f32");
}
extern "C" int sq_bn_bwd_bf16(const void *x, const void *dy, const void *y_act, int act, const float *mean,
                              const float *var, const float *gamma, float eps, void *dx, float *dgamma, float *dbeta,
                              void *workspace, int64_t npix, int C, void *stream) {
    return bn_bwd_impl<__bf16>(x, dy, y_act, act, mean, var, gamma, eps, dx, dgamma, dbeta, workspace, npix, C, stream,
                               "sq_bn_bwd_bf16");
}
