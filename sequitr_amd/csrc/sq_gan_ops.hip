// GAN-side operators of sequitr/networks/gan.py for gfx950 (all HBM-bound, f32):
//   pixel_norm backward and its second-order backward (the WGAN-GP penalty differentiates the
//   discriminator's input gradient, gan.py:721-729, and from_image applies pixel_norm, gan.py:115-125),
//   nearest-neighbour resize with align_corners (half_size, gan.py:128-131), the fade-in blend
//   (gan.py:687-694), the real/fake interpolation (gan.py:709-714), per-sample squared norms
//   (gan.py:722), minibatch-stdev (gan.py:204-212), an activation forward, and the weight gradient of
//   1x1 convolutions with a handful of channels on one side (to_image / from_image).
#include "sq_common.h"

namespace {

inline unsigned grid_for(int64_t items) {
    int64_t b = (items + 255) / 256;
    if (b > 2048) b = 2048;
    return (unsigned)(b < 1 ? 1 : b);
}
#define SQ_GRID_STRIDE(i, n) \
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (n); i += (int64_t)gridDim.x * 256)

template <int GL>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int m = 1; m < GL; m <<= 1) v += __shfl_xor(v, m);
    return v;
}

// pixel_norm: y = x r, r = rsqrt(mean_c x^2 + eps).  One GL-lane group per pixel, GL = power of two >= C / 4, <= 16
// (as pixelnorm_f32_kernel: the wide, few-channel layers keep every lane busy).
//   MODE 1 (backward):      dx = r g - r^3 s x,                       s = mean_c(g x)
//   MODE 2 (2nd backward):  given v = dL/d(dx):
//        dg = r v - r^3 t x,                                          t = mean_c(v x)
//        dx2 = -r^3 u x + 3 r^5 s t x - r^3 t g - r^3 s v,            u = mean_c(v g)
template <int MODE, int GL>
__global__ __launch_bounds__(256) void pixelnorm_bwd_kernel(const float *__restrict__ x, const float *__restrict__ g,
                                                             const float *__restrict__ v, float *__restrict__ out1,
                                                             float *__restrict__ out2, int64_t npix, int C, float eps,
                                                             float gate_slope) {
    // gate_slope (MODE 1 only): 1 = plain; otherwise x is the output of an activation (leaky 0.2 / ReLU 0) and dx leaves
    // already through that activation's backward, x > 0 ? dx : dx * slope -- the act_bwd pass that would follow
    constexpr int PPW = 64 / GL;
    const int lane = threadIdx.x & 63, l16 = lane & (GL - 1), sub = lane / GL;
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * 256) >> 6;
    const float invC = 1.0f / (float)C;
    for (int64_t pb = wave * PPW; pb < npix; pb += nwaves * PPW) {
        const int64_t p = pb + sub;
        const bool live = p < npix;
        float sxx = 0.f, sgx = 0.f, svx = 0.f, svg = 0.f;
        if (live)
            for (int c = 4 * l16; c < C; c += 4 * GL) {
                const float4 xv = *reinterpret_cast<const float4 *>(x + p * C + c);
                const float4 gv = *reinterpret_cast<const float4 *>(g + p * C + c);
                sxx += xv.x * xv.x + xv.y * xv.y + xv.z * xv.z + xv.w * xv.w;
                sgx += gv.x * xv.x + gv.y * xv.y + gv.z * xv.z + gv.w * xv.w;
                if (MODE == 2) {
                    const float4 vv = *reinterpret_cast<const float4 *>(v + p * C + c);
                    svx += vv.x * xv.x + vv.y * xv.y + vv.z * xv.z + vv.w * xv.w;
                    svg += vv.x * gv.x + vv.y * gv.y + vv.z * gv.z + vv.w * gv.w;
                }
            }
        sxx = group_sum<GL>(sxx); sgx = group_sum<GL>(sgx);
        if (MODE == 2) { svx = group_sum<GL>(svx); svg = group_sum<GL>(svg); }
        const float r = 1.0f / __builtin_sqrtf(sxx * invC + eps);
        const float r3 = r * r * r, s = sgx * invC, t = svx * invC, u = svg * invC;
        if (live)
            for (int c = 4 * l16; c < C; c += 4 * GL) {
                const float4 xv = *reinterpret_cast<const float4 *>(x + p * C + c);
                const float4 gv = *reinterpret_cast<const float4 *>(g + p * C + c);
                if (MODE == 1) {
                    float4 o = make_float4(r * gv.x - r3 * s * xv.x, r * gv.y - r3 * s * xv.y, r * gv.z - r3 * s * xv.z,
                                           r * gv.w - r3 * s * xv.w);
                    if (gate_slope != 1.0f) {
                        o.x = xv.x > 0.f ? o.x : o.x * gate_slope; o.y = xv.y > 0.f ? o.y : o.y * gate_slope;
                        o.z = xv.z > 0.f ? o.z : o.z * gate_slope; o.w = xv.w > 0.f ? o.w : o.w * gate_slope;
                    }
                    *reinterpret_cast<float4 *>(out1 + p * C + c) = o;
                } else {
                    const float4 vv = *reinterpret_cast<const float4 *>(v + p * C + c);
                    *reinterpret_cast<float4 *>(out1 + p * C + c) =
                        make_float4(r * vv.x - r3 * t * xv.x, r * vv.y - r3 * t * xv.y, r * vv.z - r3 * t * xv.z,
                                    r * vv.w - r3 * t * xv.w);
                    const float a = -r3 * u + 3.0f * r3 * r * r * s * t, b = -r3 * t, d = -r3 * s;
                    *reinterpret_cast<float4 *>(out2 + p * C + c) =
                        make_float4(a * xv.x + b * gv.x + d * vv.x, a * xv.y + b * gv.y + d * vv.y,
                                    a * xv.z + b * gv.z + d * vv.z, a * xv.w + b * gv.w + d * vv.w);
                }
            }
    }
}

// tf.image.resize_nearest_neighbor(align_corners=True): src = round(d * (in-1)/(out-1))
__global__ __launch_bounds__(256) void resize_nn_kernel(const float *__restrict__ x, float *__restrict__ y, int N,
                                                         int Hi, int Wi, int Ho, int Wo, int C) {
    const float sy = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f;
    const float sx = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
    const int64_t total = (int64_t)N * Ho * Wo * C;
    SQ_GRID_STRIDE(i, total) {
        const int c = (int)(i % C);
        int64_t t = i / C;
        const int xo = (int)(t % Wo);
        t /= Wo;
        const int yo = (int)(t % Ho);
        const int n = (int)(t / Ho);
        int ys = (int)__builtin_roundf((float)yo * sy), xs = (int)__builtin_roundf((float)xo * sx);
        ys = ys < Hi - 1 ? ys : Hi - 1;
        xs = xs < Wi - 1 ? xs : Wi - 1;
        y[i] = x[(((int64_t)n * Hi + ys) * Wi + xs) * C + c];
    }
}

// y = alpha*a + (1-alpha)*b ; alpha is a scalar, or per-sample (r, gan.py:709-714) when `per` != NULL
__global__ __launch_bounds__(256) void lerp_kernel(const float4 *__restrict__ a, const float4 *__restrict__ b,
                                                    float4 *__restrict__ y, int64_t n4, int64_t per_sample4,
                                                    float alpha, const float *__restrict__ per) {
    SQ_GRID_STRIDE(i, n4) {
        const float al = per ? per[i / per_sample4] : alpha;
        const float be = 1.0f - al;
        const float4 u = a[i], v = b[i];
        y[i] = make_float4(al * u.x + be * v.x, al * u.y + be * v.y, al * u.z + be * v.z, al * u.w + be * v.w);
    }
}

// y = s * x with s scalar or per-sample: the gradients of the blends above
__global__ __launch_bounds__(256) void scale_kernel(const float4 *__restrict__ x, float4 *__restrict__ y, int64_t n4,
                                                     int64_t per_sample4, float s, const float *__restrict__ per,
                                                     int one_minus) {
    SQ_GRID_STRIDE(i, n4) {
        float k = per ? per[i / per_sample4] : s;
        if (one_minus) k = 1.0f - k;
        const float4 u = x[i];
        y[i] = make_float4(k * u.x, k * u.y, k * u.z, k * u.w);
    }
}

__global__ __launch_bounds__(256) void act_fwd_kernel(const float4 *__restrict__ x, float4 *__restrict__ y, int64_t n4,
                                                       int act) {
    SQ_GRID_STRIDE(i, n4) {
        const float4 v = x[i];
        y[i] = make_float4(sq_act(v.x, act), sq_act(v.y, act), sq_act(v.z, act), sq_act(v.w, act));
    }
}

// per-sample dot products out[n] = sum_i a[n,i]*b[n,i]  (b == a: squared norm); one block per
// (sample, slice), fixed-order finish over the slices in dot_finish_kernel
constexpr int DOT_SLICES = 32;
__global__ __launch_bounds__(256) void dot_per_sample_kernel(const float4 *__restrict__ a, const float4 *__restrict__ b,
                                                              float *__restrict__ partials, int64_t per4) {
    __shared__ float red[256];
    const int n = blockIdx.y, sl = blockIdx.x;
    const int64_t lo = per4 * sl / DOT_SLICES, hi = per4 * (sl + 1) / DOT_SLICES;
    float s = 0.f;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
        const float4 u = a[(int64_t)n * per4 + i], v = b[(int64_t)n * per4 + i];
        s += u.x * v.x + u.y * v.y + u.z * v.z + u.w * v.w;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[n * DOT_SLICES + sl] = red[0];
}
__global__ void dot_finish_kernel(const float *__restrict__ partials, float *__restrict__ out, int N) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    for (int k = 0; k < DOT_SLICES; ++k) s += partials[n * DOT_SLICES + k];
    out[n] = s;
}

// minibatch stdev (gan.py:204-212): sqrt( mean_p ( mean_n (x_np - mean_n x_np)^2 ) ), P = per-sample size.
// One thread per position p (two passes over the batch), block partial sums, single-block finish.
__global__ __launch_bounds__(256) void mbstd_kernel(const float *__restrict__ x, float *__restrict__ partials, int N,
                                                     int64_t P) {
    __shared__ float red[256];
    float acc = 0.f;
    SQ_GRID_STRIDE(p, P) {
        float mu = 0.f;
        for (int n = 0; n < N; ++n) mu += x[(int64_t)n * P + p];
        mu /= (float)N;
        float var = 0.f;
        for (int n = 0; n < N; ++n) {
            const float d = x[(int64_t)n * P + p] - mu;
            var = __builtin_fmaf(d, d, var);
        }
        acc += var / (float)N;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0];
}
__global__ void mbstd_finish_kernel(const float *__restrict__ partials, int nb, float invP, float *__restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < nb; ++i) s += partials[i];
        *out = __builtin_sqrtf(s * invP);
    }
}

// M[ca][cb] = sum_p a[p,ca] * b[p,cb]   with CA <= 4 (the image side of to_image / from_image) and
// Cb % 4 == 0.  Thread = (pixel lane, cb quad); block partials [grid][CA][Cb]; fixed-order finish.
template <int CA>
__global__ __launch_bounds__(256) void wgrad1x1_small_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                              float *__restrict__ partials, int64_t npix, int Cb) {
    extern __shared__ float red[];                           // [256][CA*4]
    const int q4 = Cb / 4;                                   // quads per pixel
    const int quads_per_pass = 256 < q4 ? 256 : q4;
    const int pl = 256 / quads_per_pass;                     // pixel lanes per block
    const int tq = threadIdx.x % quads_per_pass, tp = threadIdx.x / quads_per_pass;
    float *out = partials + (size_t)blockIdx.x * CA * Cb;
    for (int qb = 0; qb < q4; qb += quads_per_pass) {
        const int q = qb + tq;
        float acc[CA][4];
#pragma unroll
        for (int c = 0; c < CA; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[c][j] = 0.f;
        if (q < q4 && tp < pl)
            for (int64_t p = (int64_t)blockIdx.x * pl + tp; p < npix; p += (int64_t)gridDim.x * pl) {
                const float4 bv = *reinterpret_cast<const float4 *>(b + p * Cb + q * 4);
#pragma unroll
                for (int c = 0; c < CA; ++c) {
                    const float av = a[p * CA + c];
                    acc[c][0] = __builtin_fmaf(av, bv.x, acc[c][0]);
                    acc[c][1] = __builtin_fmaf(av, bv.y, acc[c][1]);
                    acc[c][2] = __builtin_fmaf(av, bv.z, acc[c][2]);
                    acc[c][3] = __builtin_fmaf(av, bv.w, acc[c][3]);
                }
            }
#pragma unroll
        for (int c = 0; c < CA; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) red[(c * 4 + j) * 256 + threadIdx.x] = acc[c][j];
        // fold the pixel lanes: a fixed tree where pl is a power of two (thread = tp * quads_per_pass + tq); one thread adding
        // all pl lanes of its quad was up to 2048 serial LDS reads per block (most of this kernel's time at Cb = 8)
        if ((pl & (pl - 1)) == 0) {
            for (int st = pl >> 1; st > 0; st >>= 1) {
                __syncthreads();
                if (tp < st) {
#pragma unroll
                    for (int e = 0; e < CA * 4; ++e) red[e * 256 + threadIdx.x] += red[e * 256 + threadIdx.x + st * quads_per_pass];
                }
            }
            __syncthreads();
            if (tp == 0 && q < q4) {
#pragma unroll
                for (int c = 0; c < CA; ++c)
#pragma unroll
                    for (int j = 0; j < 4; ++j) out[c * Cb + q * 4 + j] = red[(c * 4 + j) * 256 + tq];
            }
        } else {                                                // quad counts that do not divide 256: the serial fold
            __syncthreads();
            if (tp == 0 && q < q4) {
#pragma unroll
                for (int c = 0; c < CA; ++c)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float s = 0.f;
                        for (int k = 0; k < pl; ++k) s += red[(c * 4 + j) * 256 + k * quads_per_pass + tq];
                        out[c * Cb + q * 4 + j] = s;
                    }
            }
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void wgrad1x1_small_finish_kernel(const float *__restrict__ partials,
                                                                     float *__restrict__ m, int nblk, int total, int G) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int i = t / G, g = t % G;
    if (i >= total) return;
    const float s = sq_group_reduce(partials + i, (size_t)total, nblk, g, G);
    if (g == 0) m[i] = s;
}

inline int small_blocks(int64_t npix) {
    int64_t b = (npix + 255) / 256;
    return (int)(b > 256 ? 256 : (b < 1 ? 1 : b));
}


// Mosaic of small images: the GAN's 4x4 / 8x8 levels fill 1/16 resp. 1/4 of the conv kernels' 16x16
// pixel tile.  Packing the batch into ONE image of R x Cc cells of pitch (H+1, W+1) -- a zero row and
// column after every image stand in for the SAME padding -- lets the same kernels run ~4-8x fewer,
// full tiles.  Zero taps contribute fmaf(w, 0, acc) = acc, so every output keeps its exact chain.
__global__ __launch_bounds__(256) void mosaic_pack_kernel(const float4 *__restrict__ x, float4 *__restrict__ m, int N,
                                                          int H, int W, int C4, int R, int Cc) {
    const int MW = Cc * (W + 1);
    const int64_t total = (int64_t)R * (H + 1) * MW * C4;
    SQ_GRID_STRIDE(i, total) {
        const int c = (int)(i % C4);
        int64_t t = i / C4;
        const int mx = (int)(t % MW), my = (int)(t / MW);
        const int cc = mx / (W + 1), xx = mx % (W + 1), rr = my / (H + 1), yy = my % (H + 1);
        const int n = rr * Cc + cc;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (xx < W && yy < H && n < N) v = x[(((int64_t)n * H + yy) * W + xx) * C4 + c];
        m[i] = v;
    }
}

__global__ __launch_bounds__(256) void mosaic_unpack_kernel(const float4 *__restrict__ m, float4 *__restrict__ y, int N,
                                                            int H, int W, int C4, int R, int Cc) {
    const int MW = Cc * (W + 1);
    const int64_t total = (int64_t)N * H * W * C4;
    SQ_GRID_STRIDE(i, total) {
        const int c = (int)(i % C4);
        int64_t t = i / C4;
        const int xx = (int)(t % W);
        t /= W;
        const int yy = (int)(t % H);
        const int n = (int)(t / H);
        const int rr = n / Cc, cc = n % Cc;
        y[i] = m[(((int64_t)rr * (H + 1) + yy) * MW + cc * (W + 1) + xx) * C4 + c];
    }
}

}  // namespace

#define SQ_ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" int sq_pixelnorm_bwd_f32(const float *x, const float *dy, float *dx, int64_t npix, int C, float eps,
                                    void *stream) {
    SQ_REQUIRE(x && dy && dx && npix > 0 && C > 0 && C % 4 == 0, "sq_pixelnorm_bwd_f32: bad arguments (C %% 4 == 0)");
    SQ_REQUIRE_ALIGNED(x); SQ_REQUIRE_ALIGNED(dy); SQ_REQUIRE_ALIGNED(dx);
    if (C <= 4) hipLaunchKernelGGL((pixelnorm_bwd_kernel<1, 1>), dim3(grid_for(npix)), dim3(256), 0, SQ_ST(stream), x, dy, nullptr, dx, nullptr, npix, C, eps, 1.0f);
    else if (C <= 8) hipLaunchKernelGGL((pixelnorm_bwd_kernel<1, 2>), dim3(grid_for(npix * 2)), dim3(256), 0, SQ_ST(stream), x, dy, nullptr, dx, nullptr, npix, C, eps, 1.0f);
    else if (C <= 16) hipLaunchKernelGGL((pixelnorm_bwd_kernel<1, 4>), dim3(grid_for(npix * 4)), dim3(256), 0, SQ_ST(stream), x, dy, nullptr, dx, nullptr, npix, C, eps, 1.0f);
    else if (C <= 32) hipLaunchKernelGGL((pixelnorm_bwd_kernel<1, 8>), dim3(grid_for(npix * 8)), dim3(256), 0, SQ_ST(stream), x, dy, nullptr, dx, nullptr, npix, C, eps, 1.0f);
    else hipLaunchKernelGGL((pixelnorm_bwd_kernel<1, 16>), dim3(grid_for(npix * 16)), dim3(256), 0, SQ_ST(stream), x, dy, nullptr, dx, nullptr, npix, C, eps, 1.0f);
    return sq_check_launch("sq_pixelnorm_bwd_f32");
}

// pixel_norm backward followed by the backward of the activation whose output x is (weighted_conv2d: conv -> leaky ->
// pixel_norm, gan.py:90-98): dx = act'(x) * pixelnorm_bwd(x, dy), one pass instead of two
extern "C" int sq_pixelnorm_bwd_act_f32(const float *x, const float *dy, float *dx, int64_t npix, int C, float eps, int act,
                                        void *stream) {
    SQ_REQUIRE(x && dy && dx && npix > 0 && C > 0 && C % 4 == 0, "sq_pixelnorm_bwd_act_f32: bad arguments (C %% 4 == 0)");
    SQ_REQUIRE(act >= SQ_ACT_NONE && act <= SQ_ACT_LEAKY, "sq_pixelnorm_bwd_act_f32: bad activation %d", act);
    SQ_REQUIRE_ALIGNED(x); SQ_REQUIRE_ALIGNED(dy); SQ_REQUIRE_ALIGNED(dx);
    const float slope = act == SQ_ACT_LEAKY ? 0.2f : (act == SQ_ACT_RELU ? 0.0f : 1.0f);
    if (C <= 4) hipLaunchKernelGGL((pixelnorm_bwd_kernel<1, 1>), dim3(grid_for(npix)), dim3(256), 0, SQ_ST(stream), x, dy, nullptr, dx, nullptr, npix, C, eps, slope);
    else if (C <= 8) hipLaunchKernelGGL((pixelnorm_bwd_kernel<1, 2>), dim3(grid_for(npix * 2)), dim3(256), 0, SQ_ST(stream), x, dy, nullptr, dx, nullptr, npix, C, eps, slope);
    else if (C <= 16) hipLaunchKernelGGL((pixelnorm_bwd_kernel<1, 4>), dim3(grid_for(npix * 4)), dim3(256), 0, SQ_ST(stream), x, dy, nullptr, dx, nullptr, npix, C, eps, slope);
    else if (C <= 32) hipLaunchKernelGGL((pixelnorm_bwd_kernel<1, 8>), dim3(grid_for(npix * 8)), dim3(256), 0, SQ_ST(stream), x, dy, nullptr, dx, nullptr, npix, C, eps, slope);
    else hipLaunchKernelGGL((pixelnorm_bwd_kernel<1, 16>), dim3(grid_for(npix * 16)), dim3(256), 0, SQ_ST(stream), x, dy, nullptr, dx, nullptr, npix, C, eps, slope);
    return sq_check_launch("sq_pixelnorm_bwd_act_f32");
}

extern "C" int sq_pixelnorm_bwd2_f32(const float *x, const float *g, const float *v, float *dg, float *dx2,
                                     int64_t npix, int C, float eps, void *stream) {
    SQ_REQUIRE(x && g && v && dg && dx2 && npix > 0 && C > 0 && C % 4 == 0,
               "sq_pixelnorm_bwd2_f32: bad arguments (C %% 4 == 0)");
    SQ_REQUIRE_ALIGNED(x); SQ_REQUIRE_ALIGNED(g); SQ_REQUIRE_ALIGNED(v); SQ_REQUIRE_ALIGNED(dg); SQ_REQUIRE_ALIGNED(dx2);
    if (C <= 4) hipLaunchKernelGGL((pixelnorm_bwd_kernel<2, 1>), dim3(grid_for(npix)), dim3(256), 0, SQ_ST(stream), x, g, v, dg, dx2, npix, C, eps, 1.0f);
    else if (C <= 8) hipLaunchKernelGGL((pixelnorm_bwd_kernel<2, 2>), dim3(grid_for(npix * 2)), dim3(256), 0, SQ_ST(stream), x, g, v, dg, dx2, npix, C, eps, 1.0f);
    else if (C <= 16) hipLaunchKernelGGL((pixelnorm_bwd_kernel<2, 4>), dim3(grid_for(npix * 4)), dim3(256), 0, SQ_ST(stream), x, g, v, dg, dx2, npix, C, eps, 1.0f);
    else if (C <= 32) hipLaunchKernelGGL((pixelnorm_bwd_kernel<2, 8>), dim3(grid_for(npix * 8)), dim3(256), 0, SQ_ST(stream), x, g, v, dg, dx2, npix, C, eps, 1.0f);
    else hipLaunchKernelGGL((pixelnorm_bwd_kernel<2, 16>), dim3(grid_for(npix * 16)), dim3(256), 0, SQ_ST(stream), x, g, v, dg, dx2, npix, C, eps, 1.0f);
    return sq_check_launch("sq_pixelnorm_bwd2_f32");
}

extern "C" int sq_resize_nearest_f32(const float *x, float *y, int N, int Hi, int Wi, int Ho, int Wo, int C,
                                     void *stream) {
    SQ_REQUIRE(x && y && N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0, "sq_resize_nearest_f32: bad arguments");
    hipLaunchKernelGGL(resize_nn_kernel, dim3(grid_for((int64_t)N * Ho * Wo * C)), dim3(256), 0, SQ_ST(stream), x, y, N,
                       Hi, Wi, Ho, Wo, C);
    return sq_check_launch("sq_resize_nearest_f32");
}

extern "C" int sq_lerp_f32(const float *a, const float *b, float *y, int64_t n, int64_t per_sample, float alpha,
                           const float *alpha_per_sample, void *stream) {
    SQ_REQUIRE(a && b && y && n > 0 && n % 4 == 0 && per_sample > 0 && per_sample % 4 == 0 && n % per_sample == 0,
               "sq_lerp_f32: sizes must be multiples of 4 and n a multiple of per_sample");
    SQ_REQUIRE_ALIGNED(a); SQ_REQUIRE_ALIGNED(b); SQ_REQUIRE_ALIGNED(y);
    hipLaunchKernelGGL(lerp_kernel, dim3(grid_for(n / 4)), dim3(256), 0, SQ_ST(stream),
                       reinterpret_cast<const float4 *>(a), reinterpret_cast<const float4 *>(b),
                       reinterpret_cast<float4 *>(y), n / 4, per_sample / 4, alpha, alpha_per_sample);
    return sq_check_launch("sq_lerp_f32");
}

extern "C" int sq_scale_f32(const float *x, float *y, int64_t n, int64_t per_sample, float s,
                            const float *s_per_sample, int one_minus, void *stream) {
    SQ_REQUIRE(x && y && n > 0 && n % 4 == 0 && per_sample > 0 && per_sample % 4 == 0 && n % per_sample == 0,
               "sq_scale_f32: sizes must be multiples of 4 and n a multiple of per_sample");
    SQ_REQUIRE_ALIGNED(x); SQ_REQUIRE_ALIGNED(y);
    hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n / 4)), dim3(256), 0, SQ_ST(stream),
                       reinterpret_cast<const float4 *>(x), reinterpret_cast<float4 *>(y), n / 4, per_sample / 4, s,
                       s_per_sample, one_minus);
    return sq_check_launch("sq_scale_f32");
}

extern "C" int sq_act_fwd_f32(const float *x, float *y, int64_t n, int act, void *stream) {
    SQ_REQUIRE(x && y && n > 0 && n % 4 == 0, "sq_act_fwd_f32: bad arguments (n %% 4 == 0)");
    SQ_REQUIRE(act >= SQ_ACT_NONE && act <= SQ_ACT_LEAKY, "sq_act_fwd_f32: bad activation %d", act);
    SQ_REQUIRE_ALIGNED(x); SQ_REQUIRE_ALIGNED(y);
    hipLaunchKernelGGL(act_fwd_kernel, dim3(grid_for(n / 4)), dim3(256), 0, SQ_ST(stream),
                       reinterpret_cast<const float4 *>(x), reinterpret_cast<float4 *>(y), n / 4, act);
    return sq_check_launch("sq_act_fwd_f32");
}

extern "C" int64_t sq_dot_per_sample_workspace_f32(int N) { return N > 0 ? (int64_t)N * DOT_SLICES * 4 : -1; }

extern "C" int sq_dot_per_sample_f32(const float *a, const float *b, float *out, float *workspace, int N,
                                     int64_t per_sample, void *stream) {
    SQ_REQUIRE(a && b && out && workspace && N > 0 && N <= 65535 && per_sample > 0 && per_sample % 4 == 0,
               "sq_dot_per_sample_f32: bad arguments (per_sample %% 4 == 0)");
    SQ_REQUIRE_ALIGNED(a); SQ_REQUIRE_ALIGNED(b);
    hipLaunchKernelGGL(dot_per_sample_kernel, dim3(DOT_SLICES, N), dim3(256), 0, SQ_ST(stream),
                       reinterpret_cast<const float4 *>(a), reinterpret_cast<const float4 *>(b), workspace,
                       per_sample / 4);
    int rc = sq_check_launch("sq_dot_per_sample_f32");
    if (rc) return rc;
    hipLaunchKernelGGL(dot_finish_kernel, dim3((N + 63) / 64), dim3(64), 0, SQ_ST(stream), workspace, out, N);
    return sq_check_launch("sq_dot_per_sample_f32(finish)");
}

extern "C" int sq_mbstd_fwd_f32(const float *x, float *out, float *workspace, int N, int64_t per_sample,
                                void *stream) {
    SQ_REQUIRE(x && out && workspace && N > 0 && per_sample > 0, "sq_mbstd_fwd_f32: bad arguments");
    int nb = (int)((per_sample + 255) / 256);
    if (nb > 256) nb = 256;                                   // workspace: 256 floats
    hipLaunchKernelGGL(mbstd_kernel, dim3(nb), dim3(256), 0, SQ_ST(stream), x, workspace, N, per_sample);
    int rc = sq_check_launch("sq_mbstd_fwd_f32");
    if (rc) return rc;
    hipLaunchKernelGGL(mbstd_finish_kernel, dim3(1), dim3(64), 0, SQ_ST(stream), workspace, nb, 1.0f / (float)per_sample,
                       out);
    return sq_check_launch("sq_mbstd_fwd_f32(finish)");
}

extern "C" int64_t sq_wgrad1x1_small_workspace_f32(int64_t npix, int Ca, int Cb) {
    if (npix <= 0 || Ca < 1 || Ca > 7 || Cb <= 0 || Cb % 4) return -1;
    return (int64_t)small_blocks(npix) * Ca * Cb * 4;
}

extern "C" int sq_wgrad1x1_small_f32(const float *a, const float *b, float *m, float *workspace, int64_t npix,
                                     int Ca, int Cb, void *stream) {
    SQ_REQUIRE(a && b && m && workspace, "sq_wgrad1x1_small_f32: null pointer");
    SQ_REQUIRE(npix > 0 && Ca >= 1 && Ca <= 7 && Cb > 0 && Cb % 4 == 0,
               "sq_wgrad1x1_small_f32: Ca=%d (1..7), Cb=%d (multiple of 4)", Ca, Cb);
    SQ_REQUIRE_ALIGNED(b);
    const int nb = small_blocks(npix);
    hipStream_t st = SQ_ST(stream);
    const size_t lds = 256 * Ca * 4 * sizeof(float);
    switch (Ca) {
    case 1: hipLaunchKernelGGL(wgrad1x1_small_kernel<1>, dim3(nb), dim3(256), lds, st, a, b, workspace, npix, Cb); break;
    case 2: hipLaunchKernelGGL(wgrad1x1_small_kernel<2>, dim3(nb), dim3(256), lds, st, a, b, workspace, npix, Cb); break;
    case 3: hipLaunchKernelGGL(wgrad1x1_small_kernel<3>, dim3(nb), dim3(256), lds, st, a, b, workspace, npix, Cb); break;
    case 4: hipLaunchKernelGGL(wgrad1x1_small_kernel<4>, dim3(nb), dim3(256), lds, st, a, b, workspace, npix, Cb); break;
    case 5: hipLaunchKernelGGL(wgrad1x1_small_kernel<5>, dim3(nb), dim3(256), lds, st, a, b, workspace, npix, Cb); break;
    case 6: hipLaunchKernelGGL(wgrad1x1_small_kernel<6>, dim3(nb), dim3(256), lds, st, a, b, workspace, npix, Cb); break;
    default: hipLaunchKernelGGL(wgrad1x1_small_kernel<7>, dim3(nb), dim3(256), lds, st, a, b, workspace, npix, Cb); break;
    }
    int rc = sq_check_launch("sq_wgrad1x1_small_f32");
    if (rc) return rc;
    const int total = Ca * Cb;
    const int G = sq_group_size(nb);
    hipLaunchKernelGGL(wgrad1x1_small_finish_kernel, dim3((total * G + 255) / 256), dim3(256), 0, st, workspace, m, nb, total, G);
    return sq_check_launch("sq_wgrad1x1_small_f32(finish)");
}

extern "C" int sq_mosaic_pack_f32(const float *x, float *m, int N, int H, int W, int C, int R, int Cc, void *stream) {
    SQ_REQUIRE(x && m && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && R > 0 && Cc > 0 && (int64_t)R * Cc >= N,
               "sq_mosaic_pack_f32: need C %% 4 == 0 and R*Cc >= N (N=%d R=%d Cc=%d C=%d)", N, R, Cc, C);
    SQ_REQUIRE_ALIGNED(x); SQ_REQUIRE_ALIGNED(m);
    hipLaunchKernelGGL(mosaic_pack_kernel, dim3(grid_for((int64_t)R * (H + 1) * Cc * (W + 1) * (C / 4))), dim3(256), 0,
                       SQ_ST(stream), reinterpret_cast<const float4 *>(x), reinterpret_cast<float4 *>(m), N, H, W, C / 4,
                       R, Cc);
    return sq_check_launch("sq_mosaic_pack_f32");
}

extern "C" int sq_mosaic_unpack_f32(const float *m, float *y, int N, int H, int W, int C, int R, int Cc, void *stream) {
    SQ_REQUIRE(m && y && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && R > 0 && Cc > 0 && (int64_t)R * Cc >= N,
               "sq_mosaic_unpack_f32: need C %% 4 == 0 and R*Cc >= N (N=%d R=%d Cc=%d C=%d)", N, R, Cc, C);
    SQ_REQUIRE_ALIGNED(m); SQ_REQUIRE_ALIGNED(y);
    hipLaunchKernelGGL(mosaic_unpack_kernel, dim3(grid_for((int64_t)N * H * W * (C / 4))), dim3(256), 0, SQ_ST(stream),
                       reinterpret_cast<const float4 *>(m), reinterpret_cast<float4 *>(y), N, H, W, C / 4, R, Cc);
    return sq_check_launch("sq_mosaic_unpack_f32");
}

// ---------------------------------------------------------------------------------------------------------------
// minibatch-stdev feature of the discriminator (gan.py:204-212) up to second order, and the WGAN-GP loss algebra
// (gan.py:709-729): tiny tensors (32 x 16 x 512 values; N scalars) that ran as ~200 framework element-wise / reduce
// launches per iteration.  Two launches per op: 64 blocks per minibatch group reduce their share of the positions
// (fixed order), then every block of the second launch folds the 64 partials itself, in index order, and applies
// the element-wise part to its share -- no atomics, run-to-run identical.
//   x (G*n, P) -- G groups of n samples (the stacked D(Gz) | D(X) pass has G = 2), P = 4*4*C values per sample.
//   forward : s_g = sqrt( (1/P) sum_j (1/n) sum_i (x_ij - mu_j)^2 ),  y (G*n, cells) = s_g   (cells = 16: the (N,4,4,1) map)
//   backward: ds_g = sum of dy over the group;  dx_ij = ds_g * c * (x_ij - mu_j) / s_g,  c = 1 / (P n)
//   backward of the backward, cotangent V of dx:  A = sum_ij V_ij (x_ij - mu_j)
//             d(dy) = c A / s_g  (every element of the group)
//             d x_kl = ds_g c / s_g * ( (V_kl - Vbar_l) - A c (x_kl - mu_l) / s_g^2 )
namespace {

constexpr int MB_T = 1024;     // the loss kernels: one workgroup
constexpr int MB_B = 64;       // blocks per group of the statistic kernels
constexpr int MB_BT = 256;

__device__ __forceinline__ float mb_block_sum(float v, float *red) {
    const int t = threadIdx.x, T = blockDim.x;
    red[t] = v;
    __syncthreads();
    for (int k = T / 2; k > 0; k >>= 1) {
        if (t < k) red[t] += red[t + k];
        __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
}

// stage 1, grid (MB_B, groups): block b sums its positions' batch variances (and, with v, its share of
// A = sum_ij V_ij (x_ij - mu_j)) -> partials[g][b] = {var sum, A}
// one element of a feature tensor as f32.  bf16: the element's DWORD is loaded and the right half taken -- two neighbouring lanes
// share a dword, which the memory pipe serves like the f32 kernel's loads; 2-byte loads ran the statistic kernel at half the
// f32 kernel's speed (17.3 vs 7.7 us) for half the bytes.  Same values, same order of the sums as the f32 kernels.
__device__ __forceinline__ float mb_ld(const float *base, int64_t idx) { return base[idx]; }
__device__ __forceinline__ float mb_ld(const __bf16 *base, int64_t idx) {
    const unsigned w = *reinterpret_cast<const unsigned *>(base + (idx & ~(int64_t)1));   // tensors are 4-byte aligned, P is even
    return __builtin_bit_cast(float, (idx & 1) ? (w & 0xFFFF0000u) : (w << 16));
}

// TX: storage type of the feature tensors x / v / dx / dx2 (float, or __bf16 under the GAN's bf16 storage: the arithmetic is
// f32 either way and a bf16 result is the f32 value rounded once -- what a separate cast kernel did); dy / ddy stay f32
template <typename TX>
__global__ __launch_bounds__(MB_BT) void mbstd_stats_kernel(const TX *__restrict__ x, const TX *__restrict__ v,
                                                            float *__restrict__ partials, int n, int64_t P) {
    __shared__ float red[MB_BT];
    const int g = blockIdx.y;
    const TX *xg = x + (int64_t)g * n * P;
    const TX *vg = v ? v + (int64_t)g * n * P : nullptr;
    float acc = 0.f, aa = 0.f;
    for (int64_t p = (int64_t)blockIdx.x * MB_BT + threadIdx.x; p < P; p += (int64_t)MB_B * MB_BT) {
        // eight samples' loads in flight at a time (same order of the sums): one load per trip was a chain of 2 n round trips
        float mu = 0.f;
        for (int i0 = 0; i0 < n; i0 += 8) {
            float xv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) xv[k] = i0 + k < n ? mb_ld(xg, (int64_t)(i0 + k) * P + p) : 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) mu += xv[k];
        }
        mu /= (float)n;
        float var = 0.f;
        for (int i0 = 0; i0 < n; i0 += 8) {
            float xv[8], vv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                xv[k] = i0 + k < n ? mb_ld(xg, (int64_t)(i0 + k) * P + p) : mu;
                vv[k] = (vg && i0 + k < n) ? mb_ld(vg, (int64_t)(i0 + k) * P + p) : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (i0 + k < n) {
                    const float d = xv[k] - mu;
                    var = __builtin_fmaf(d, d, var);
                    if (vg) aa = __builtin_fmaf(vv[k], d, aa);
                }
            }
        }
        acc += var / (float)n;
    }
    const float sv = mb_block_sum(acc, red), sa = mb_block_sum(aa, red);
    if (threadIdx.x == 0) {
        partials[((int64_t)g * MB_B + blockIdx.x) * 2] = sv;
        partials[((int64_t)g * MB_B + blockIdx.x) * 2 + 1] = sa;
    }
}

// every block of stage 2 folds the MB_B partials itself, in index order: the same s (and A) in every block
__device__ __forceinline__ void mb_fold(const float *partials, int g, int64_t P, float *s, float *A) {
    float sv = 0.f, sa = 0.f;
    for (int b = 0; b < MB_B; ++b) {
        sv += partials[((int64_t)g * MB_B + b) * 2];
        sa += partials[((int64_t)g * MB_B + b) * 2 + 1];
    }
    *s = __builtin_sqrtf(sv / (float)P);
    *A = sa;
}

__device__ __forceinline__ float mb_group_dy_sum(const float *dyg, int count, float *red) {
    float a = 0.f;
    for (int e = threadIdx.x; e < count; e += MB_BT) a += dyg[e];
    return mb_block_sum(a, red);
}

__global__ __launch_bounds__(MB_BT) void mbstd_map_fill_kernel(const float *__restrict__ partials, float *__restrict__ y, int n,
                                                               int64_t P, int cells) {
    const int g = blockIdx.x;
    float s, A;
    mb_fold(partials, g, P, &s, &A);
    for (int e = threadIdx.x; e < n * cells; e += MB_BT) y[(int64_t)g * n * cells + e] = s;
}

template <typename TX>
__global__ __launch_bounds__(MB_BT) void mbstd_map_bwd_kernel(const TX *__restrict__ x, const float *__restrict__ dy,
                                                              const float *__restrict__ partials, TX *__restrict__ dx, int n,
                                                              int64_t P, int cells) {
    __shared__ float red[MB_BT];
    const int g = blockIdx.y;
    const TX *xg = x + (int64_t)g * n * P;
    float s, A;
    mb_fold(partials, g, P, &s, &A);
    const float ds = mb_group_dy_sum(dy + (int64_t)g * n * cells, n * cells, red);
    const float k = ds / ((float)P * (float)n) / s;
    for (int64_t p = (int64_t)blockIdx.x * MB_BT + threadIdx.x; p < P; p += (int64_t)MB_B * MB_BT) {
        float mu = 0.f;
        for (int i = 0; i < n; ++i) mu += mb_ld(xg, (int64_t)i * P + p);
        mu /= (float)n;
        for (int i = 0; i < n; ++i) dx[((int64_t)g * n + i) * P + p] = (TX)(k * (mb_ld(xg, (int64_t)i * P + p) - mu));
    }
}

template <typename TX>
__global__ __launch_bounds__(MB_BT) void mbstd_map_bwd2_kernel(const TX *__restrict__ x, const float *__restrict__ dy,
                                                               const TX *__restrict__ v, const float *__restrict__ partials,
                                                               float *__restrict__ ddy, TX *__restrict__ dx2, int n, int64_t P,
                                                               int cells) {
    __shared__ float red[MB_BT];
    const int g = blockIdx.y;
    const TX *xg = x + (int64_t)g * n * P, *vg = v + (int64_t)g * n * P;
    float s, A;
    mb_fold(partials, g, P, &s, &A);
    const float ds = mb_group_dy_sum(dy + (int64_t)g * n * cells, n * cells, red);
    const float c = 1.0f / ((float)P * (float)n);
    if (blockIdx.x == 0) {
        const float gdy = c * A / s;
        for (int e = threadIdx.x; e < n * cells; e += MB_BT) ddy[(int64_t)g * n * cells + e] = gdy;
    }
    const float k1 = ds * c / s, k2 = A * c / (s * s);
    for (int64_t p = (int64_t)blockIdx.x * MB_BT + threadIdx.x; p < P; p += (int64_t)MB_B * MB_BT) {
        float mu = 0.f, vb = 0.f;
        for (int i = 0; i < n; ++i) { mu += mb_ld(xg, (int64_t)i * P + p); vb += mb_ld(vg, (int64_t)i * P + p); }
        mu /= (float)n;
        vb /= (float)n;
        for (int i = 0; i < n; ++i)
            dx2[((int64_t)g * n + i) * P + p] = (TX)(k1 * ((mb_ld(vg, (int64_t)i * P + p) - vb) - k2 * (mb_ld(xg, (int64_t)i * P + p) - mu)));
    }
}

// WGAN-GP losses of one level (gan.py:715-729): gn2 = squared norm of d D(mix) / d mix per sample.
//   pen_i = 10 max(sqrt(gn2_i) - 1, 0)^2, eps_i = 0.001 Dx_i^2, d_loss = mean(-Dx + Dz + pen + eps), g_loss = mean(-Dz)
// out[0] = d_loss, out[1] = g_loss.  Dx / gn2 may be NULL (generator step: g_loss only).  Single workgroup.
__global__ __launch_bounds__(MB_T) void wgan_losses_fwd_kernel(const float *__restrict__ Dz, const float *__restrict__ Dx,
                                                               const float *__restrict__ gn2, float *__restrict__ out, int N) {
    __shared__ float red[MB_T];
    float d = 0.f, gl = 0.f;
    for (int i = threadIdx.x; i < N; i += MB_T) {
        const float z = Dz[i];
        gl += -z;
        if (Dx) {
            const float xv = Dx[i], nrm = __builtin_sqrtf(gn2[i]), ex = nrm - 1.0f > 0.0f ? nrm - 1.0f : 0.0f;
            d += ((-xv + z) + 10.0f * (ex * ex)) + 0.001f * (xv * xv);
        }
    }
    const float sd = mb_block_sum(d, red), sg = mb_block_sum(gl, red);
    if (threadIdx.x == 0) {
        out[0] = sd / (float)N;
        out[1] = sg / (float)N;
    }
}

// gd, gg: upstream gradients of d_loss / g_loss (device scalars; NULL = 0)
__global__ __launch_bounds__(MB_T) void wgan_losses_bwd_kernel(const float *__restrict__ Dz, const float *__restrict__ Dx,
                                                               const float *__restrict__ gn2, const float *__restrict__ gd,
                                                               const float *__restrict__ gg, float *__restrict__ dDz,
                                                               float *__restrict__ dDx, float *__restrict__ dgn2, int N) {
    const float ud = gd ? gd[0] / (float)N : 0.f, ug = gg ? gg[0] / (float)N : 0.f;
    for (int i = threadIdx.x; i < N; i += MB_T) {
        dDz[i] = (Dx ? ud : 0.f) - ug;
        if (Dx) {
            const float xv = Dx[i], nrm = __builtin_sqrtf(gn2[i]), ex = nrm - 1.0f > 0.0f ? nrm - 1.0f : 0.0f;
            dDx[i] = ud * (-1.0f + 0.002f * xv);
            dgn2[i] = ex > 0.0f ? ud * 10.0f * ex / nrm : 0.f;   // d/d(gn2) of 10 (sqrt(gn2) - 1)^2 = 10 (sqrt - 1) / sqrt
        }
    }
}

}  // namespace

extern "C" int64_t sq_mbstd_map_workspace(int groups) { return groups > 0 ? (int64_t)groups * MB_B * 2 * 4 : -1; }

namespace {
template <typename TX>
int mbstd_map_fwd_t(const TX *x, float *y, float *workspace, int groups, int n, int64_t per_sample, int cells, void *stream,
                    const char *what) {
    hipLaunchKernelGGL(mbstd_stats_kernel<TX>, dim3(MB_B, groups), dim3(MB_BT), 0, SQ_ST(stream), x, (const TX *)nullptr, workspace, n,
                       per_sample);
    int rc = sq_check_launch(what);
    if (rc) return rc;
    hipLaunchKernelGGL(mbstd_map_fill_kernel, dim3(groups), dim3(MB_BT), 0, SQ_ST(stream), workspace, y, n, per_sample, cells);
    return sq_check_launch(what);
}
template <typename TX>
int mbstd_map_bwd_t(const TX *x, const float *dy, TX *dx, float *workspace, int groups, int n, int64_t per_sample, int cells,
                    void *stream, const char *what) {
    hipLaunchKernelGGL(mbstd_stats_kernel<TX>, dim3(MB_B, groups), dim3(MB_BT), 0, SQ_ST(stream), x, (const TX *)nullptr, workspace, n,
                       per_sample);
    int rc = sq_check_launch(what);
    if (rc) return rc;
    hipLaunchKernelGGL(mbstd_map_bwd_kernel<TX>, dim3(MB_B, groups), dim3(MB_BT), 0, SQ_ST(stream), x, dy, workspace, dx, n, per_sample,
                       cells);
    return sq_check_launch(what);
}
template <typename TX>
int mbstd_map_bwd2_t(const TX *x, const float *dy, const TX *v, float *ddy, TX *dx2, float *workspace, int groups, int n,
                     int64_t per_sample, int cells, void *stream, const char *what) {
    hipLaunchKernelGGL(mbstd_stats_kernel<TX>, dim3(MB_B, groups), dim3(MB_BT), 0, SQ_ST(stream), x, v, workspace, n, per_sample);
    int rc = sq_check_launch(what);
    if (rc) return rc;
    hipLaunchKernelGGL(mbstd_map_bwd2_kernel<TX>, dim3(MB_B, groups), dim3(MB_BT), 0, SQ_ST(stream), x, dy, v, workspace, ddy, dx2, n,
                       per_sample, cells);
    return sq_check_launch(what);
}
}  // namespace

extern "C" int sq_mbstd_map_fwd_f32(const float *x, float *y, float *workspace, int groups, int n, int64_t per_sample,
                                    int cells, void *stream) {
    SQ_REQUIRE(x && y && workspace && groups > 0 && n > 0 && per_sample > 0 && cells > 0, "sq_mbstd_map_fwd_f32: bad arguments");
    return mbstd_map_fwd_t<float>(x, y, workspace, groups, n, per_sample, cells, stream, "sq_mbstd_map_fwd_f32");
}
extern "C" int sq_mbstd_map_bwd_f32(const float *x, const float *dy, float *dx, float *workspace, int groups, int n,
                                    int64_t per_sample, int cells, void *stream) {
    SQ_REQUIRE(x && dy && dx && workspace && groups > 0 && n > 0 && per_sample > 0 && cells > 0,
               "sq_mbstd_map_bwd_f32: bad arguments");
    return mbstd_map_bwd_t<float>(x, dy, dx, workspace, groups, n, per_sample, cells, stream, "sq_mbstd_map_bwd_f32");
}
extern "C" int sq_mbstd_map_bwd2_f32(const float *x, const float *dy, const float *v, float *ddy, float *dx2,
                                     float *workspace, int groups, int n, int64_t per_sample, int cells, void *stream) {
    SQ_REQUIRE(x && dy && v && ddy && dx2 && workspace && groups > 0 && n > 0 && per_sample > 0 && cells > 0,
               "sq_mbstd_map_bwd2_f32: bad arguments");
    return mbstd_map_bwd2_t<float>(x, dy, v, ddy, dx2, workspace, groups, n, per_sample, cells, stream, "sq_mbstd_map_bwd2_f32");
}
// the same three on bf16 FEATURE tensors (x, v, dx, dx2 bf16; the map y, dy, ddy f32): the statistic is taken of the values the
// tensor stores, the gradients are the f32 results rounded once -- no cast launches around the statistic
extern "C" int sq_mbstd_map_fwd_bf16(const void *x, float *y, float *workspace, int groups, int n, int64_t per_sample, int cells,
                                     void *stream) {
    SQ_REQUIRE(x && y && workspace && groups > 0 && n > 0 && per_sample > 0 && per_sample % 2 == 0 && cells > 0,
               "sq_mbstd_map_fwd_bf16: bad arguments (an even number of values per sample)");
    return mbstd_map_fwd_t<__bf16>(reinterpret_cast<const __bf16 *>(x), y, workspace, groups, n, per_sample, cells, stream,
                                   "sq_mbstd_map_fwd_bf16");
}
extern "C" int sq_mbstd_map_bwd_bf16(const void *x, const float *dy, void *dx, float *workspace, int groups, int n,
                                     int64_t per_sample, int cells, void *stream) {
    SQ_REQUIRE(x && dy && dx && workspace && groups > 0 && n > 0 && per_sample > 0 && per_sample % 2 == 0 && cells > 0,
               "sq_mbstd_map_bwd_bf16: bad arguments (an even number of values per sample)");
    return mbstd_map_bwd_t<__bf16>(reinterpret_cast<const __bf16 *>(x), dy, reinterpret_cast<__bf16 *>(dx), workspace, groups, n,
                                   per_sample, cells, stream, "sq_mbstd_map_bwd_bf16");
}
extern "C" int sq_mbstd_map_bwd2_bf16(const void *x, const float *dy, const void *v, float *ddy, void *dx2, float *workspace,
                                      int groups, int n, int64_t per_sample, int cells, void *stream) {
    SQ_REQUIRE(x && dy && v && ddy && dx2 && workspace && groups > 0 && n > 0 && per_sample > 0 && per_sample % 2 == 0 && cells > 0,
               "sq_mbstd_map_bwd2_bf16: bad arguments (an even number of values per sample)");
    return mbstd_map_bwd2_t<__bf16>(reinterpret_cast<const __bf16 *>(x), dy, reinterpret_cast<const __bf16 *>(v), ddy,
                                    reinterpret_cast<__bf16 *>(dx2), workspace, groups, n, per_sample, cells, stream,
                                    "sq_mbstd_map_bwd2_bf16");
}
extern "C" int sq_wgan_losses_fwd_f32(const float *Dz, const float *Dx, const float *gn2, float *out2, int N, void *stream) {
    SQ_REQUIRE(Dz && out2 && N > 0 && ((Dx == nullptr) == (gn2 == nullptr)), "sq_wgan_losses_fwd_f32: bad arguments");
    hipLaunchKernelGGL(wgan_losses_fwd_kernel, dim3(1), dim3(MB_T), 0, SQ_ST(stream), Dz, Dx, gn2, out2, N);
    return sq_check_launch("sq_wgan_losses_fwd_f32");
}
extern "C" int sq_wgan_losses_bwd_f32(const float *Dz, const float *Dx, const float *gn2, const float *g_dloss,
                                      const float *g_gloss, float *dDz, float *dDx, float *dgn2, int N, void *stream) {
    SQ_REQUIRE(Dz && dDz && N > 0 && ((Dx == nullptr) == (gn2 == nullptr)) && (!Dx || (dDx && dgn2)),
               "sq_wgan_losses_bwd_f32: bad arguments");
    hipLaunchKernelGGL(wgan_losses_bwd_kernel, dim3(1), dim3(MB_T), 0, SQ_ST(stream), Dz, Dx, gn2, g_dloss, g_gloss, dDz, dDx,
                       dgn2, N);
    return sq_check_launch("sq_wgan_losses_bwd_f32");
}
