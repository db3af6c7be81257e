// Backward / training-side HBM-bound kernels of the sequitr hot path (gfx950): activation and
// pooling gradients, bridge gradients, the space-to-depth view that turns the 2x2/s2
// transpose-conv backward into 1x1 convolutions, the filter transform for dgrad, the to_image
// head backward, dropout, and the Adam update over the flat parameter buffer.
#include "sq_common.h"

namespace {

inline unsigned grid_for(int64_t items) {
    int64_t b = (items + 255) / 256;
    if (b > 2048) b = 2048;
    return (unsigned)(b < 1 ? 1 : b);
}
#define SQ_GRID_STRIDE(i, n) \
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (n); i += (int64_t)gridDim.x * 256)

// dgrad filter: wt[ky][kx][co][ci] = w[K-1-ky][K-1-kx][ci][co]  (HWIO with the roles of I and O swapped)
__global__ __launch_bounds__(256) void weight_transform_kernel(const float *__restrict__ w, float *__restrict__ wt,
                                                                int K, int Cin, int Cout) {
    const int total = K * K * Cin * Cout;
    SQ_GRID_STRIDE(i, total) {
        const int ci = (int)(i % Cin), co = (int)((i / Cin) % Cout), tap = (int)(i / ((int64_t)Cin * Cout));
        const int ky = tap / K, kx = tap % K;
        wt[i] = w[(((K - 1 - ky) * K + (K - 1 - kx)) * Cin + ci) * Cout + co];
    }
}

__global__ __launch_bounds__(256) void act_bwd_kernel(const float4 *__restrict__ dy, const float4 *__restrict__ y,
                                                       float4 *__restrict__ dx, int64_t n4, int act) {
    const float slope = act == SQ_ACT_LEAKY ? 0.2f : (act == SQ_ACT_RELU ? 0.0f : 1.0f);
    SQ_GRID_STRIDE(i, n4) {
        const float4 g = dy[i], v = y[i];
        dx[i] = make_float4(v.x > 0.f ? g.x : g.x * slope, v.y > 0.f ? g.y : g.y * slope,
                            v.z > 0.f ? g.z : g.z * slope, v.w > 0.f ? g.w : g.w * slope);
    }
}

// max-pool backward: the gradient goes to the FIRST maximum in raster order (TF MaxPoolGrad)
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float4 *__restrict__ x, const float4 *__restrict__ dy,
                                                           float4 *__restrict__ dx, int N, int H, int W, int C4) {
    const int Ho = H >> 1, Wo = W >> 1;
    const int64_t total = (int64_t)N * Ho * Wo * C4;
    SQ_GRID_STRIDE(i, total) {
        const int c = (int)(i % C4);
        int64_t t = i / C4;
        const int xo = (int)(t % Wo);
        t /= Wo;
        const int yo = (int)(t % Ho);
        const int n = (int)(t / Ho);
        const int64_t b00 = (((int64_t)n * H + 2 * yo) * W + 2 * xo) * C4 + c;
        const int64_t b01 = b00 + C4, b10 = b00 + (int64_t)W * C4, b11 = b10 + C4;
        const float4 a = x[b00], b = x[b01], d = x[b10], e = x[b11], g = dy[i];
        float4 ra, rb, rd, re;
#define SQ_POOL_BWD(f)                                                     \
        {                                                                  \
            float m = a.f; int k = 0;                                      \
            if (b.f > m) { m = b.f; k = 1; }                               \
            if (d.f > m) { m = d.f; k = 2; }                               \
            if (e.f > m) { m = e.f; k = 3; }                               \
            ra.f = k == 0 ? g.f : 0.f; rb.f = k == 1 ? g.f : 0.f;          \
            rd.f = k == 2 ? g.f : 0.f; re.f = k == 3 ? g.f : 0.f;          \
        }
        SQ_POOL_BWD(x) SQ_POOL_BWD(y) SQ_POOL_BWD(z) SQ_POOL_BWD(w)
#undef SQ_POOL_BWD
        dx[b00] = ra; dx[b01] = rb; dx[b10] = rd; dx[b11] = re;
    }
}

// avg-pool backward (scale 0.25) and nearest-2x up-sampling forward share the broadcast; the
// up-sampling backward (sum of the 2x2 patch) is avg-pool forward x 4 -> `scale` parameter.
// V = float4 when C % 4 == 0, else float (2-channel images of to_image / from_image).
__device__ __forceinline__ float4 vscale(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float vscale(float a, float s) { return a * s; }
__device__ __forceinline__ float4 vsum4(float4 a, float4 b, float4 d, float4 e) {
    return make_float4((a.x + b.x) + (d.x + e.x), (a.y + b.y) + (d.y + e.y), (a.z + b.z) + (d.z + e.z),
                       (a.w + b.w) + (d.w + e.w));
}
__device__ __forceinline__ float vsum4(float a, float b, float d, float e) { return (a + b) + (d + e); }

template <typename V>
__global__ __launch_bounds__(256) void broadcast2x2_kernel(const V *__restrict__ dy, V *__restrict__ dx,
                                                            int N, int H, int W, int C4, float scale) {
    const int64_t total = (int64_t)N * H * W * C4;           // H, W = the LARGE side
    SQ_GRID_STRIDE(i, total) {
        const int c = (int)(i % C4);
        int64_t t = i / C4;
        const int xx = (int)(t % W);
        t /= W;
        const int yy = (int)(t % H);
        const int n = (int)(t / H);
        dx[i] = vscale(dy[(((int64_t)n * (H >> 1) + (yy >> 1)) * (W >> 1) + (xx >> 1)) * C4 + c], scale);
    }
}

// scale * nearest-neighbour 2x up-sampling, then the backward of the activation whose output `gate` (large side) is:
// the gradient of avg-pool(act(conv)) w.r.t. the conv's pre-activation in one pass (discriminator blocks, gan.py:171-192)
__global__ __launch_bounds__(256) void broadcast2x2_gate_kernel(const float4 *__restrict__ dy, const float4 *__restrict__ gate,
                                                                float4 *__restrict__ dx, int N, int H, int W, int C4,
                                                                float scale, float slope) {
    const int64_t total = (int64_t)N * H * W * C4;
    SQ_GRID_STRIDE(i, total) {
        const int c = (int)(i % C4);
        int64_t t = i / C4;
        const int xx = (int)(t % W);
        t /= W;
        const int yy = (int)(t % H);
        const int n = (int)(t / H);
        const float4 g = vscale(dy[(((int64_t)n * (H >> 1) + (yy >> 1)) * (W >> 1) + (xx >> 1)) * C4 + c], scale);
        const float4 v = gate[i];
        dx[i] = make_float4(v.x > 0.f ? g.x : g.x * slope, v.y > 0.f ? g.y : g.y * slope,
                            v.z > 0.f ? g.z : g.z * slope, v.w > 0.f ? g.w : g.w * slope);
    }
}

template <typename V>
__global__ __launch_bounds__(256) void sumpool2x2_kernel(const V *__restrict__ x, V *__restrict__ y,
                                                          int N, int H, int W, int C4, float scale) {
    const int Ho = H >> 1, Wo = W >> 1;
    const int64_t total = (int64_t)N * Ho * Wo * C4;
    SQ_GRID_STRIDE(i, total) {
        const int c = (int)(i % C4);
        int64_t t = i / C4;
        const int xo = (int)(t % Wo);
        t /= Wo;
        const int yo = (int)(t % Ho);
        const int n = (int)(t / Ho);
        const int64_t b = (((int64_t)n * H + 2 * yo) * W + 2 * xo) * C4 + c;
        y[i] = vscale(vsum4(x[b], x[b + C4], x[b + (int64_t)W * C4], x[b + (int64_t)W * C4 + C4]), scale);
    }
}

__global__ __launch_bounds__(256) void bridge_bwd_kernel(const float4 *__restrict__ dy, const float4 *__restrict__ a,
                                                          const float4 *__restrict__ b, float4 *__restrict__ da,
                                                          float4 *__restrict__ db, int64_t n4, int op) {
    SQ_GRID_STRIDE(i, n4) {
        const float4 g = dy[i];
        if (op == SQ_BRIDGE_MUL) {
            const float4 u = a[i], v = b[i];
            da[i] = make_float4(g.x * v.x, g.y * v.y, g.z * v.z, g.w * v.w);
            db[i] = make_float4(g.x * u.x, g.y * u.y, g.z * u.z, g.w * u.w);
        } else {
            da[i] = g;
            db[i] = op == SQ_BRIDGE_SUB ? make_float4(-g.x, -g.y, -g.z, -g.w) : g;
        }
    }
}

// g[n,i,j,(2a+b)*C + c] = dy[n,2i+a,2j+b,c]
__global__ __launch_bounds__(256) void space_to_depth2_kernel(const float4 *__restrict__ dy, float4 *__restrict__ g,
                                                               int N, int H, int W, int C4) {
    const int64_t total = (int64_t)N * H * W * 4 * C4;       // H, W = the SMALL side
    SQ_GRID_STRIDE(i, total) {
        const int c = (int)(i % C4);
        int64_t t = i / C4;
        const int ab = (int)(t & 3);
        t >>= 2;
        const int j = (int)(t % W);
        t /= W;
        const int ii = (int)(t % H);
        const int n = (int)(t / H);
        g[i] = dy[(((int64_t)n * 2 * H + 2 * ii + (ab >> 1)) * (2 * W) + 2 * j + (ab & 1)) * C4 + c];
    }
}

// zero insertion for the 3x3/s2 transpose conv (`up_kernel` = (3,3)): u[n,2i+1,2j+1,:] = x[n,i,j,:], 0 elsewhere.
// A SAME 3x3 convolution of u with the rotated, transposed kernel is TF's conv2d_transpose(k=3, s=2, SAME).
__global__ __launch_bounds__(256) void zero_insert2x_kernel(const float4 *__restrict__ x, float4 *__restrict__ u,
                                                             int N, int H, int W, int C4) {
    const int64_t total = (int64_t)N * 2 * H * 2 * W * C4;    // H, W = the SMALL side
    SQ_GRID_STRIDE(i, total) {
        const int c = (int)(i % C4);
        int64_t t = i / C4;
        const int xx = (int)(t % (2 * W));
        t /= 2 * W;
        const int yy = (int)(t % (2 * H));
        const int n = (int)(t / (2 * H));
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((xx & 1) && (yy & 1)) v = x[(((int64_t)n * H + (yy >> 1)) * W + (xx >> 1)) * C4 + c];
        u[i] = v;
    }
}

// its adjoint: dx[n,i,j,:] = du[n,2i+1,2j+1,:]
__global__ __launch_bounds__(256) void gather_odd2x_kernel(const float4 *__restrict__ du, float4 *__restrict__ dx,
                                                            int N, int H, int W, int C4) {
    const int64_t total = (int64_t)N * H * W * C4;
    SQ_GRID_STRIDE(i, total) {
        const int c = (int)(i % C4);
        int64_t t = i / C4;
        const int j = (int)(t % W);
        t /= W;
        const int ii = (int)(t % H);
        const int n = (int)(t / H);
        dx[i] = du[(((int64_t)n * 2 * H + 2 * ii + 1) * (2 * W) + 2 * j + 1) * C4 + c];
    }
}

// to_image head backward (1x1, Cout <= 4, Cin in {8,16,32}): dx[p,c] = sum_o dz[p,o] w[c,o]; dW / db
// as block partials [gridDim.x][CIN*COUT + COUT] (wave shuffle tree, then waves 0..3 in order),
// finished in fixed block order by head_finish_kernel.
__device__ __forceinline__ float wave_sum(float v) {
    v += __shfl_xor(v, 32); v += __shfl_xor(v, 16); v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4);  v += __shfl_xor(v, 2);  v += __shfl_xor(v, 1);
    return v;
}

template <int CIN, int COUT>
__global__ __launch_bounds__(256) void head_bwd_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                        const float *__restrict__ dz, float *__restrict__ dx,
                                                        float *__restrict__ partials, int64_t npix) {
    constexpr int NVAL = CIN * COUT + COUT;
    __shared__ float red[4][NVAL];
    float gw[CIN][COUT], gb[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) gb[o] = 0.f;
#pragma unroll
    for (int c = 0; c < CIN; ++c)
#pragma unroll
        for (int o = 0; o < COUT; ++o) gw[c][o] = 0.f;
    // wave-uniform trip count so the shuffles below see full waves
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t base = (int64_t)blockIdx.x * 256; base < npix; base += stride) {
        const int64_t p = base + threadIdx.x;
        if (p < npix) {
            float g[COUT];
#pragma unroll
            for (int o = 0; o < COUT; ++o) { g[o] = dz[p * COUT + o]; gb[o] += g[o]; }
#pragma unroll
            for (int c4 = 0; c4 < CIN / 4; ++c4) {
                const float4 v = *reinterpret_cast<const float4 *>(x + p * CIN + c4 * 4);
                const float xv[4] = {v.x, v.y, v.z, v.w};
                float r[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float s = 0.f;
#pragma unroll
                    for (int o = 0; o < COUT; ++o) {
                        s = __builtin_fmaf(g[o], w[(c4 * 4 + j) * COUT + o], s);
                        gw[c4 * 4 + j][o] = __builtin_fmaf(xv[j], g[o], gw[c4 * 4 + j][o]);
                    }
                    r[j] = s;
                }
                if (dx) *reinterpret_cast<float4 *>(dx + p * CIN + c4 * 4) = make_float4(r[0], r[1], r[2], r[3]);
            }
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < CIN; ++c)
#pragma unroll
        for (int o = 0; o < COUT; ++o) {
            const float v = wave_sum(gw[c][o]);
            if (lane == 0) red[wv][c * COUT + o] = v;
        }
#pragma unroll
    for (int o = 0; o < COUT; ++o) {
        const float v = wave_sum(gb[o]);
        if (lane == 0) red[wv][CIN * COUT + o] = v;
    }
    __syncthreads();
    if ((int)threadIdx.x < NVAL)
        partials[(size_t)blockIdx.x * NVAL + threadIdx.x] =
            ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

__global__ __launch_bounds__(256) void head_finish_kernel(const float *__restrict__ partials, float *__restrict__ dw,
                                                           float *__restrict__ db, int nblk, int nw, int nb, int G) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int i = t / G, g = t % G;
    if (i >= nw + nb) return;
    const float s = sq_group_reduce(partials + i, (size_t)(nw + nb), nblk, g, G);
    if (g != 0) return;
    if (i < nw) dw[i] = s;
    else if (db) db[i - nw] = s;
}

// counter-based dropout mask: sq_dropout_keep4 (sq_common.h), one hash per quad of elements

// 4 elements per thread (16 B of data, 4 B of mask); n % 4 == 0 is required by the entry points
__global__ __launch_bounds__(256) void dropout_fwd_kernel(const float4 *__restrict__ x, float4 *__restrict__ y,
                                                           uchar4 *__restrict__ mask, int64_t n4, float rate,
                                                           unsigned seed, int mask_given, const int *__restrict__ step) {
    const SqDropKey key = sq_dropout_key(seed, step);
    const unsigned thr = sq_dropout_thr16(rate);
    const float inv = 1.0f / (1.0f - rate);
    SQ_GRID_STRIDE(i, n4) {
        uchar4 k;
        if (mask_given) k = mask[i];
        else {
            const unsigned b = sq_dropout_keep4(key, (unsigned)i, thr);
            k = make_uchar4(b & 1u, (b >> 1) & 1u, (b >> 2) & 1u, (b >> 3) & 1u);
            mask[i] = k;
        }
        const float4 v = x[i];
        y[i] = make_float4(k.x ? v.x * inv : 0.f, k.y ? v.y * inv : 0.f, k.z ? v.z * inv : 0.f, k.w ? v.w * inv : 0.f);
    }
}

__global__ __launch_bounds__(256) void dropout_bwd_kernel(const float4 *__restrict__ dy, const uchar4 *__restrict__ mask,
                                                           float4 *__restrict__ dx, int64_t n4, float rate) {
    const float inv = 1.0f / (1.0f - rate);
    SQ_GRID_STRIDE(i, n4) {
        const uchar4 k = mask[i];
        const float4 v = dy[i];
        dx[i] = make_float4(k.x ? v.x * inv : 0.f, k.y ? v.y * inv : 0.f, k.z ? v.z * inv : 0.f, k.w ? v.w * inv : 0.f);
    }
}

// Adam (tf.train.AdamOptimizer form): lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMA; p -= lr_t*m/(sqrt(v)+eps)
// graph-safe step bookkeeping: the step counter lives in device memory, so a captured hipGraph
// replays with the right bias correction.  state[0] = step (int32), state[1] = lr_t (float bits).
// warmup > 0: the learning rate ramps linearly, lr * min(1, t / warmup), evaluated on the device from the same counter
// (a replayed graph walks the schedule by itself; the host never rewrites a kernel argument).
__global__ void adam_prepare_kernel(int *__restrict__ state, float lr, float b1, float b2, int warmup) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const int t = state[0] + 1;
        state[0] = t;
        const double ramp = (warmup > 0 && t < warmup) ? (double)t / (double)warmup : 1.0;
        const double lr_t = (double)lr * ramp * sqrt(1.0 - pow((double)b2, (double)t)) / (1.0 - pow((double)b1, (double)t));
        reinterpret_cast<float *>(state)[1] = (float)lr_t;
    }
}

__device__ __forceinline__ void adam_update(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                            float *__restrict__ v, int64_t i, float lr_t, float b1, float b2, float eps,
                                            float gscale) {
    const float gi = g[i] * gscale;
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = p[i] - lr_t * mi / (__builtin_sqrtf(vi) + eps);
}

__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, const float *__restrict__ g,
                                                    float *__restrict__ m, float *__restrict__ v, int64_t n,
                                                    float lr_t_host, const int *__restrict__ state, float b1, float b2,
                                                    float eps, float gscale) {
    const float lr_t = state ? reinterpret_cast<const float *>(state)[1] : lr_t_host;
    SQ_GRID_STRIDE(i, n) adam_update(p, g, m, v, i, lr_t, b1, b2, eps, gscale);
}

// the same update over a LIST of tensors in one launch (the GAN's per-variable optimiser: ~30 tensors per solver, most of
// them a few thousand elements, a few of them millions): table = n_entries x {p, g, m, v, count, first chunk} as 64-bit
// words in device memory; the work is cut into chunks of ADAM_CHUNK elements numbered across the entries, one block
// per chunk, the block finds its entry by bisection on the first-chunk column
constexpr int ADAM_CHUNK = 2048;
__global__ __launch_bounds__(256) void adam_multi_kernel(const int64_t *__restrict__ table, int n_entries,
                                                          const int *__restrict__ state, float b1, float b2, float eps,
                                                          float gscale) {
    int lo = 0, hi = n_entries - 1;                            // last entry whose first chunk <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[(size_t)mid * 6 + 5] <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const int64_t *e = table + (size_t)lo * 6;
    float *p = reinterpret_cast<float *>(e[0]);
    const float *g = reinterpret_cast<const float *>(e[1]);
    float *m = reinterpret_cast<float *>(e[2]), *v = reinterpret_cast<float *>(e[3]);
    const int64_t n = e[4], base = ((int64_t)blockIdx.x - e[5]) * ADAM_CHUNK;
    const float lr_t = reinterpret_cast<const float *>(state)[1];
#pragma unroll
    for (int k = 0; k < ADAM_CHUNK / 256; ++k) {
        const int64_t i = base + k * 256 + threadIdx.x;
        if (i < n) adam_update(p, g, m, v, i, lr_t, b1, b2, eps, gscale);
    }
}

}  // namespace

#define SQ_ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" int sq_conv_weight_transform_f32(const float *w, float *wt, int K, int Cin, int Cout, void *stream) {
    SQ_REQUIRE(w && wt && K > 0 && Cin > 0 && Cout > 0, "sq_conv_weight_transform_f32: bad arguments");
    hipLaunchKernelGGL(weight_transform_kernel, dim3(grid_for((int64_t)K * K * Cin * Cout)), dim3(256), 0,
                       SQ_ST(stream), w, wt, K, Cin, Cout);
    return sq_check_launch("sq_conv_weight_transform_f32");
}

extern "C" int sq_act_bwd_f32(const float *dy, const float *y, float *dx, int64_t n, int act, void *stream) {
    SQ_REQUIRE(dy && y && dx && n > 0 && n % 4 == 0, "sq_act_bwd_f32: bad arguments (n %% 4 == 0)");
    SQ_REQUIRE(act >= SQ_ACT_NONE && act <= SQ_ACT_LEAKY, "sq_act_bwd_f32: bad activation %d", act);
    SQ_REQUIRE_ALIGNED(dy); SQ_REQUIRE_ALIGNED(y); SQ_REQUIRE_ALIGNED(dx);
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n / 4)), dim3(256), 0, SQ_ST(stream),
                       reinterpret_cast<const float4 *>(dy), reinterpret_cast<const float4 *>(y),
                       reinterpret_cast<float4 *>(dx), n / 4, act);
    return sq_check_launch("sq_act_bwd_f32");
}

extern "C" int sq_maxpool2x2_bwd_f32(const float *x, const float *dy, float *dx, int N, int H, int W, int C,
                                     void *stream) {
    SQ_REQUIRE(x && dy && dx, "sq_maxpool2x2_bwd_f32: null tensor pointer");
    SQ_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && H % 2 == 0 && W % 2 == 0 && C % 4 == 0,
               "sq_maxpool2x2_bwd_f32: need even H,W and C %% 4 == 0");
    SQ_REQUIRE_ALIGNED(x); SQ_REQUIRE_ALIGNED(dy); SQ_REQUIRE_ALIGNED(dx);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for((int64_t)N * (H / 2) * (W / 2) * (C / 4))), dim3(256), 0,
                       SQ_ST(stream), reinterpret_cast<const float4 *>(x), reinterpret_cast<const float4 *>(dy),
                       reinterpret_cast<float4 *>(dx), N, H, W, C / 4);
    return sq_check_launch("sq_maxpool2x2_bwd_f32");
}

extern "C" int sq_broadcast2x2_f32(const float *src, float *dst, int N, int H, int W, int C, float scale,
                                   void *stream) {
    SQ_REQUIRE(src && dst, "sq_broadcast2x2_f32: null tensor pointer");
    SQ_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && H % 2 == 0 && W % 2 == 0,
               "sq_broadcast2x2_f32: need even H,W (large side)");
    if (C % 4 == 0 && SQ_ALIGNED16(src) && SQ_ALIGNED16(dst))
        hipLaunchKernelGGL(broadcast2x2_kernel<float4>, dim3(grid_for((int64_t)N * H * W * (C / 4))), dim3(256), 0,
                           SQ_ST(stream), reinterpret_cast<const float4 *>(src), reinterpret_cast<float4 *>(dst), N, H,
                           W, C / 4, scale);
    else
        hipLaunchKernelGGL(broadcast2x2_kernel<float>, dim3(grid_for((int64_t)N * H * W * C)), dim3(256), 0,
                           SQ_ST(stream), src, dst, N, H, W, C, scale);
    return sq_check_launch("sq_broadcast2x2_f32");
}

extern "C" int sq_broadcast2x2_act_bwd_f32(const float *src, const float *gate, float *dst, int N, int H, int W, int C,
                                           float scale, int act, void *stream) {
    SQ_REQUIRE(src && gate && dst, "sq_broadcast2x2_act_bwd_f32: null tensor pointer");
    SQ_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && H % 2 == 0 && W % 2 == 0 && C % 4 == 0,
               "sq_broadcast2x2_act_bwd_f32: need even H,W (large side) and C %% 4 == 0");
    SQ_REQUIRE(act >= SQ_ACT_NONE && act <= SQ_ACT_LEAKY, "sq_broadcast2x2_act_bwd_f32: bad activation %d", act);
    SQ_REQUIRE_ALIGNED(src); SQ_REQUIRE_ALIGNED(gate); SQ_REQUIRE_ALIGNED(dst);
    const float slope = act == SQ_ACT_LEAKY ? 0.2f : (act == SQ_ACT_RELU ? 0.0f : 1.0f);
    hipLaunchKernelGGL(broadcast2x2_gate_kernel, dim3(grid_for((int64_t)N * H * W * (C / 4))), dim3(256), 0, SQ_ST(stream),
                       reinterpret_cast<const float4 *>(src), reinterpret_cast<const float4 *>(gate),
                       reinterpret_cast<float4 *>(dst), N, H, W, C / 4, scale, slope);
    return sq_check_launch("sq_broadcast2x2_act_bwd_f32");
}

extern "C" int sq_sumpool2x2_f32(const float *x, float *y, int N, int H, int W, int C, float scale, void *stream) {
    SQ_REQUIRE(x && y, "sq_sumpool2x2_f32: null tensor pointer");
    SQ_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && H % 2 == 0 && W % 2 == 0, "sq_sumpool2x2_f32: need even H,W");
    if (C % 4 == 0 && SQ_ALIGNED16(x) && SQ_ALIGNED16(y))
        hipLaunchKernelGGL(sumpool2x2_kernel<float4>, dim3(grid_for((int64_t)N * (H / 2) * (W / 2) * (C / 4))),
                           dim3(256), 0, SQ_ST(stream), reinterpret_cast<const float4 *>(x),
                           reinterpret_cast<float4 *>(y), N, H, W, C / 4, scale);
    else
        hipLaunchKernelGGL(sumpool2x2_kernel<float>, dim3(grid_for((int64_t)N * (H / 2) * (W / 2) * C)), dim3(256), 0,
                           SQ_ST(stream), x, y, N, H, W, C, scale);
    return sq_check_launch("sq_sumpool2x2_f32");
}

extern "C" int sq_bridge_bwd_f32(const float *dy, const float *a, const float *b, float *da, float *db, int64_t n,
                                 int bridge, void *stream) {
    SQ_REQUIRE(dy && da && db && n > 0 && n % 4 == 0, "sq_bridge_bwd_f32: bad arguments (n %% 4 == 0)");
    SQ_REQUIRE(bridge >= SQ_BRIDGE_ADD && bridge <= SQ_BRIDGE_SUB, "sq_bridge_bwd_f32: bad bridge %d", bridge);
    SQ_REQUIRE(bridge != SQ_BRIDGE_MUL || (a && b), "sq_bridge_bwd_f32: eltwise_mul needs both forward operands");
    SQ_REQUIRE_ALIGNED(dy); SQ_REQUIRE_ALIGNED(da); SQ_REQUIRE_ALIGNED(db);
    hipLaunchKernelGGL(bridge_bwd_kernel, dim3(grid_for(n / 4)), dim3(256), 0, SQ_ST(stream),
                       reinterpret_cast<const float4 *>(dy), reinterpret_cast<const float4 *>(a),
                       reinterpret_cast<const float4 *>(b), reinterpret_cast<float4 *>(da),
                       reinterpret_cast<float4 *>(db), n / 4, bridge);
    return sq_check_launch("sq_bridge_bwd_f32");
}

extern "C" int sq_space_to_depth2_f32(const float *dy, float *g, int N, int H, int W, int C, void *stream) {
    SQ_REQUIRE(dy && g, "sq_space_to_depth2_f32: null tensor pointer");
    SQ_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "sq_space_to_depth2_f32: C %% 4 == 0");
    SQ_REQUIRE_ALIGNED(dy); SQ_REQUIRE_ALIGNED(g);
    hipLaunchKernelGGL(space_to_depth2_kernel, dim3(grid_for((int64_t)N * H * W * C)), dim3(256), 0, SQ_ST(stream),
                       reinterpret_cast<const float4 *>(dy), reinterpret_cast<float4 *>(g), N, H, W, C / 4);
    return sq_check_launch("sq_space_to_depth2_f32");
}

extern "C" int sq_zero_insert2x_f32(const float *x, float *u, int N, int H, int W, int C, void *stream) {
    SQ_REQUIRE(x && u, "sq_zero_insert2x_f32: null tensor pointer");
    SQ_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "sq_zero_insert2x_f32: C %% 4 == 0");
    SQ_REQUIRE_ALIGNED(x); SQ_REQUIRE_ALIGNED(u);
    hipLaunchKernelGGL(zero_insert2x_kernel, dim3(grid_for((int64_t)N * H * W * C)), dim3(256), 0, SQ_ST(stream),
                       reinterpret_cast<const float4 *>(x), reinterpret_cast<float4 *>(u), N, H, W, C / 4);
    return sq_check_launch("sq_zero_insert2x_f32");
}

extern "C" int sq_gather_odd2x_f32(const float *du, float *dx, int N, int H, int W, int C, void *stream) {
    SQ_REQUIRE(du && dx, "sq_gather_odd2x_f32: null tensor pointer");
    SQ_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "sq_gather_odd2x_f32: C %% 4 == 0");
    SQ_REQUIRE_ALIGNED(du); SQ_REQUIRE_ALIGNED(dx);
    hipLaunchKernelGGL(gather_odd2x_kernel, dim3(grid_for((int64_t)N * H * W * C / 4)), dim3(256), 0, SQ_ST(stream),
                       reinterpret_cast<const float4 *>(du), reinterpret_cast<float4 *>(dx), N, H, W, C / 4);
    return sq_check_launch("sq_gather_odd2x_f32");
}

static inline int head_blocks(int64_t npix) {
    int64_t b = (npix + 255) / 256;
    return (int)(b > 512 ? 512 : (b < 1 ? 1 : b));
}

extern "C" int64_t sq_conv1x1_small_bwd_workspace_f32(int64_t npix, int Cin, int Cout) {
    if (npix <= 0 || Cin <= 0 || Cout <= 0) return -1;
    return (int64_t)head_blocks(npix) * (Cin * Cout + Cout) * 4;
}

template <int CIN>
static int head_launch(int Cout, int nb, hipStream_t st, const float *x, const float *w, const float *dz, float *dx,
                       float *ws, int64_t npix) {
    switch (Cout) {
    case 1: hipLaunchKernelGGL((head_bwd_kernel<CIN, 1>), dim3(nb), dim3(256), 0, st, x, w, dz, dx, ws, npix); return 0;
    case 2: hipLaunchKernelGGL((head_bwd_kernel<CIN, 2>), dim3(nb), dim3(256), 0, st, x, w, dz, dx, ws, npix); return 0;
    case 3: hipLaunchKernelGGL((head_bwd_kernel<CIN, 3>), dim3(nb), dim3(256), 0, st, x, w, dz, dx, ws, npix); return 0;
    case 4: hipLaunchKernelGGL((head_bwd_kernel<CIN, 4>), dim3(nb), dim3(256), 0, st, x, w, dz, dx, ws, npix); return 0;
    }
    return -1;
}

extern "C" int sq_conv1x1_small_bwd_f32(const float *x, const float *w, const float *dz, float *dx, float *dw,
                                        float *db, float *workspace, int64_t npix, int Cin, int Cout,
                                        void *stream) {
    SQ_REQUIRE(x && w && dz && dw && workspace, "sq_conv1x1_small_bwd_f32: null pointer");
    SQ_REQUIRE(npix > 0 && (Cin == 8 || Cin == 16 || Cin == 32) && Cout >= 1 && Cout <= 4,
               "sq_conv1x1_small_bwd_f32: Cin=%d (8|16|32), Cout=%d (1..4)", Cin, Cout);
    SQ_REQUIRE_ALIGNED(x);
    if (dx) SQ_REQUIRE_ALIGNED(dx);
    const int nb = head_blocks(npix);
    hipStream_t st = SQ_ST(stream);
    if (Cin == 8) head_launch<8>(Cout, nb, st, x, w, dz, dx, workspace, npix);
    else if (Cin == 16) head_launch<16>(Cout, nb, st, x, w, dz, dx, workspace, npix);
    else head_launch<32>(Cout, nb, st, x, w, dz, dx, workspace, npix);
    int rc = sq_check_launch("sq_conv1x1_small_bwd_f32");
    if (rc) return rc;
    const int nw = Cin * Cout;
    { const int G = sq_group_size(nb); hipLaunchKernelGGL(head_finish_kernel, dim3(((nw + Cout) * G + 255) / 256), dim3(256), 0, st, workspace, dw, db, nb, nw, Cout, G); }
    return sq_check_launch("sq_conv1x1_small_bwd_f32(finish)");
}

extern "C" int sq_dropout_fwd_f32(const float *x, float *y, uint8_t *mask, int64_t n, float rate, uint32_t seed,
                                  int mask_given, const int32_t *step_dev, void *stream) {
    SQ_REQUIRE(x && y && mask && n > 0 && n % 4 == 0, "sq_dropout_fwd_f32: bad arguments (n %% 4 == 0)");
    SQ_REQUIRE(rate >= 0.f && rate < 1.f, "sq_dropout_fwd_f32: rate must be in [0,1)");
    SQ_REQUIRE_ALIGNED(x); SQ_REQUIRE_ALIGNED(y);
    hipLaunchKernelGGL(dropout_fwd_kernel, dim3(grid_for(n / 4)), dim3(256), 0, SQ_ST(stream),
                       reinterpret_cast<const float4 *>(x), reinterpret_cast<float4 *>(y),
                       reinterpret_cast<uchar4 *>(mask), n / 4, rate, seed, mask_given, step_dev);
    return sq_check_launch("sq_dropout_fwd_f32");
}

extern "C" int sq_dropout_bwd_f32(const float *dy, const uint8_t *mask, float *dx, int64_t n, float rate, void *stream) {
    SQ_REQUIRE(dy && mask && dx && n > 0 && n % 4 == 0, "sq_dropout_bwd_f32: bad arguments (n %% 4 == 0)");
    SQ_REQUIRE(rate >= 0.f && rate < 1.f, "sq_dropout_bwd_f32: rate must be in [0,1)");
    SQ_REQUIRE_ALIGNED(dy); SQ_REQUIRE_ALIGNED(dx);
    hipLaunchKernelGGL(dropout_bwd_kernel, dim3(grid_for(n / 4)), dim3(256), 0, SQ_ST(stream),
                       reinterpret_cast<const float4 *>(dy), reinterpret_cast<const uchar4 *>(mask),
                       reinterpret_cast<float4 *>(dx), n / 4, rate);
    return sq_check_launch("sq_dropout_bwd_f32");
}

extern "C" int sq_adam_step_f32(float *p, const float *g, float *m, float *v, int64_t n, float lr, float beta1,
                                float beta2, float eps, int step, float grad_scale, void *stream) {
    SQ_REQUIRE(p && g && m && v && n > 0 && step >= 1, "sq_adam_step_f32: bad arguments");
    const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, step)) / (1.0 - pow((double)beta1, step));
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, SQ_ST(stream), p, g, m, v, n, (float)lr_t,
                       (const int *)nullptr, beta1, beta2, eps, grad_scale);
    return sq_check_launch("sq_adam_step_f32");
}

// y += alpha * x over a flat buffer: gradient accumulation over the micro-batches of one optimiser step
// (UNetTrainer.step_accumulate); float4 body + scalar tail, no alignment demand beyond 4 bytes.
__global__ __launch_bounds__(256) void axpy_kernel(float *__restrict__ y, const float *__restrict__ x, float alpha,
                                                   int64_t n) {
    const int64_t n4 = ((((uintptr_t)y | (uintptr_t)x) & 15u) == 0) ? n / 4 : 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 a = ((float4 *)y)[i];
        const float4 b = ((const float4 *)x)[i];
        a.x = fmaf(alpha, b.x, a.x), a.y = fmaf(alpha, b.y, a.y), a.z = fmaf(alpha, b.z, a.z), a.w = fmaf(alpha, b.w, a.w);
        ((float4 *)y)[i] = a;
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        y[i] = fmaf(alpha, x[i], y[i]);
}

extern "C" int sq_axpy_f32(float *y, const float *x, float alpha, int64_t n, void *stream) {
    SQ_REQUIRE(y && x && n > 0, "sq_axpy_f32: bad arguments");
    hipLaunchKernelGGL(axpy_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, SQ_ST(stream), y, x, alpha, n);
    return sq_check_launch("sq_axpy_f32");
}

// the two halves of sq_adam_step_dev_f32 for optimisers that update many tensors per step (the GAN's per-variable
// slots): ONE advance per minimize(), then one apply per tensor
extern "C" int sq_adam_advance_dev(int32_t *state, float lr, float beta1, float beta2, void *stream) {
    SQ_REQUIRE(state, "sq_adam_advance_dev: null state");
    hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(64), 0, SQ_ST(stream), state, lr, beta1, beta2, 0);
    return sq_check_launch("sq_adam_advance_dev");
}

extern "C" int sq_adam_advance_warmup_dev(int32_t *state, float lr, float beta1, float beta2, int warmup_steps,
                                          void *stream) {
    SQ_REQUIRE(state && warmup_steps >= 0, "sq_adam_advance_warmup_dev: bad arguments");
    hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(64), 0, SQ_ST(stream), state, lr, beta1, beta2, warmup_steps);
    return sq_check_launch("sq_adam_advance_warmup_dev");
}

extern "C" int sq_adam_apply_dev_f32(float *p, const float *g, float *m, float *v, int64_t n, float beta1, float beta2,
                                     float eps, const int32_t *state, float grad_scale, void *stream) {
    SQ_REQUIRE(p && g && m && v && state && n > 0, "sq_adam_apply_dev_f32: bad arguments");
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, SQ_ST(stream), p, g, m, v, n, 0.f, state, beta1,
                       beta2, eps, grad_scale);
    return sq_check_launch("sq_adam_apply_dev_f32");
}

extern "C" int sq_adam_multi_chunk(void) { return ADAM_CHUNK; }

// one launch for a list of tensors: table (device memory) = n_entries x {p, g, m, v, element count, first chunk}, six 64-bit
// words each; first chunk = sum over the earlier entries of ceil(count / sq_adam_multi_chunk()); total_chunks = that sum
// over all entries
extern "C" int sq_adam_apply_multi_dev_f32(const void *table, int n_entries, int64_t total_chunks, float beta1, float beta2,
                                           float eps, const int32_t *state, float grad_scale, void *stream) {
    SQ_REQUIRE(table && state && n_entries > 0 && total_chunks > 0 && total_chunks < ((int64_t)1 << 31),
               "sq_adam_apply_multi_dev_f32: bad arguments");
    SQ_REQUIRE((((uintptr_t)table) & 7u) == 0, "sq_adam_apply_multi_dev_f32: table must be 8-byte aligned");
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)total_chunks), dim3(256), 0, SQ_ST(stream),
                       reinterpret_cast<const int64_t *>(table), n_entries, state, beta1, beta2, eps, grad_scale);
    return sq_check_launch("sq_adam_apply_multi_dev_f32");
}

extern "C" int sq_adam_step_dev_f32(float *p, const float *g, float *m, float *v, int64_t n, float lr, float beta1,
                                    float beta2, float eps, int32_t *state, float grad_scale, void *stream) {
    SQ_REQUIRE(p && g && m && v && state && n > 0, "sq_adam_step_dev_f32: bad arguments");
    hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(64), 0, SQ_ST(stream), state, lr, beta1, beta2, 0);
    int rc = sq_check_launch("sq_adam_step_dev_f32(prepare)");
    if (rc) return rc;
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, SQ_ST(stream), p, g, m, v, n, 0.f, state, beta1,
                       beta2, eps, grad_scale);
    return sq_check_launch("sq_adam_step_dev_f32");
}
