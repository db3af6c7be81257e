// GAN-side operators of sequitr/networks/gan.py on bf16 feature tensors, gfx950 (all HBM-bound):
// the storage form of BASELINE config 5 -- activations and activation gradients live in HBM as bf16, every kernel
// computes in f32 and rounds once per stored value (RNE), parameters / images / losses stay f32.
//   pixel_norm forward, backward (optionally through the leaky-ReLU in front of it) and second-order backward
//   (gan.py:49-51; the WGAN-GP penalty differentiates the discriminator's input gradient, gan.py:721-729),
//   2x2 sum / average pooling and its adjoint, nearest-neighbour 2x up-sampling (gan.py:133-136, 189-192),
//   a stand-alone activation, and the three image-side 1x1 convolutions (to_image / from_image, gan.py:102-125):
//   few f32 channels on one side, a bf16 feature tensor on the other.
// 8 channels (one 16-byte piece) per lane everywhere: every feature tensor of the GAN has C % 8 == 0.
// Where a fused form replaces two stored passes (gate variants) it keeps BOTH roundings, so fused == unfused, bit for bit.
#include "sq_common.h"
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

inline unsigned grid_for(int64_t items) {
    int64_t b = (items + 255) / 256;
    if (b > 4096) b = 4096;
    return (unsigned)(b < 1 ? 1 : b);
}
#define SQ_GRID_STRIDE(i, n) \
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (n); i += (int64_t)gridDim.x * 256)

struct F8 {
    float v[8];
};
__device__ __forceinline__ F8 ld8(const __bf16 *p) {
    const bf16x8 h = *reinterpret_cast<const bf16x8 *>(p);
    F8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r.v[j] = (float)h[j];
    return r;
}
__device__ __forceinline__ void st8(__bf16 *p, const F8 &a) {
    bf16x8 h;
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = (__bf16)a.v[j];
    *reinterpret_cast<bf16x8 *>(p) = h;
}
__device__ __forceinline__ float dot8(const F8 &a, const F8 &b) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s = __builtin_fmaf(a.v[j], b.v[j], s);
    return s;
}
// the value a stored bf16 gradient takes through the backward of the activation whose output is g
__device__ __forceinline__ float gate_bf16(float t, float g, float slope) {
    return g > 0.f ? t : (float)(__bf16)(t * slope);
}

template <int GL>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int m = 1; m < GL; m <<= 1) v += __shfl_xor(v, m);
    return v;
}

// pixel_norm: y = x r, r = 1 / sqrt(mean_c x^2 + eps).  One GL-lane group per pixel, 8 channels per lane and pass.
//   MODE 0 (forward):       out1 = x r
//   MODE 1 (backward):      out1 = r g - r^3 s x,                      s = mean_c(g x)     [gate_slope != 1: x is the
//                           output of an activation, out1 leaves through its backward as a second stored rounding]
//   MODE 2 (2nd backward):  v = dL/d(dx):  out1 = dg = r v - r^3 t x,  t = mean_c(v x)
//                           out2 = dx2 = (-r^3 u + 3 r^5 s t) x - r^3 t g - r^3 s v,       u = mean_c(v g)
// NCH > 0: C == 8 * GL * NCH -- a lane's NCH chunks of every operand are loaded ONCE, all requests in flight together, and both passes
// (the channel sums, then the outputs) run on the registers; NCH == 0: any C, the operands are read again for the second pass.
// Same per-lane order of additions either way: same bits.
template <int MODE, int GL, int NCH = 0>
__global__ __launch_bounds__(256) void pixelnorm_bf16_kernel(const __bf16 *__restrict__ x, const __bf16 *__restrict__ g,
                                                              const __bf16 *__restrict__ v, __bf16 *__restrict__ out1,
                                                              __bf16 *__restrict__ out2, int64_t npix, int C, float eps,
                                                              float gate_slope) {
    constexpr int PPW = 64 / GL;
    const int lane = threadIdx.x & 63, lg = lane & (GL - 1), sub = lane / GL;
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * 256) >> 6;
    const float invC = 1.0f / (float)C;
    for (int64_t pb = wave * PPW; pb < npix; pb += nwaves * PPW) {
        const int64_t p = pb + sub;
        const bool live = p < npix;
        float sxx = 0.f, sgx = 0.f, svx = 0.f, svg = 0.f;
        if constexpr (NCH > 0) {
            F8 xr[NCH], gr[MODE >= 1 ? NCH : 1], vr[MODE == 2 ? NCH : 1];
            if (live) {
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    const int c = 8 * lg + 8 * GL * k;
                    xr[k] = ld8(x + p * C + c);
                    if (MODE >= 1) gr[k] = ld8(g + p * C + c);
                    if (MODE == 2) vr[k] = ld8(v + p * C + c);
                }
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    sxx += dot8(xr[k], xr[k]);
                    if (MODE >= 1) sgx += dot8(gr[k], xr[k]);
                    if (MODE == 2) { svx += dot8(vr[k], xr[k]); svg += dot8(vr[k], gr[k]); }
                }
            }
            sxx = group_sum<GL>(sxx);
            if (MODE >= 1) sgx = group_sum<GL>(sgx);
            if (MODE == 2) { svx = group_sum<GL>(svx); svg = group_sum<GL>(svg); }
            const float r = 1.0f / __builtin_sqrtf(sxx * invC + eps);
            const float r3 = r * r * r, s = sgx * invC, t = svx * invC, u = svg * invC;
            if (live) {
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    const int c = 8 * lg + 8 * GL * k;
                    const F8 xv = xr[k];
                    F8 o;
                    if (MODE == 0) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) o.v[j] = xv.v[j] * r;
                        st8(out1 + p * C + c, o);
                    } else if (MODE == 1) {
                        const F8 gv = gr[k];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            o.v[j] = r * gv.v[j] - r3 * s * xv.v[j];
                            if (gate_slope != 1.0f) o.v[j] = gate_bf16((float)(__bf16)o.v[j], xv.v[j], gate_slope);
                        }
                        st8(out1 + p * C + c, o);
                    } else {
                        const F8 gv = gr[k], vv = vr[k];
                        const float a = -r3 * u + 3.0f * r3 * r * r * s * t, b = -r3 * t, d = -r3 * s;
                        F8 o2;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            o.v[j] = r * vv.v[j] - r3 * t * xv.v[j];
                            o2.v[j] = a * xv.v[j] + b * gv.v[j] + d * vv.v[j];
                        }
                        st8(out1 + p * C + c, o);
                        st8(out2 + p * C + c, o2);
                    }
                }
            }
            continue;
        }
        if (live)
            for (int c = 8 * lg; c < C; c += 8 * GL) {
                const F8 xv = ld8(x + p * C + c);
                sxx += dot8(xv, xv);
                if (MODE >= 1) {
                    const F8 gv = ld8(g + p * C + c);
                    sgx += dot8(gv, xv);
                    if (MODE == 2) {
                        const F8 vv = ld8(v + p * C + c);
                        svx += dot8(vv, xv);
                        svg += dot8(vv, gv);
                    }
                }
            }
        sxx = group_sum<GL>(sxx);
        if (MODE >= 1) sgx = group_sum<GL>(sgx);
        if (MODE == 2) { svx = group_sum<GL>(svx); svg = group_sum<GL>(svg); }
        const float r = 1.0f / __builtin_sqrtf(sxx * invC + eps);
        const float r3 = r * r * r, s = sgx * invC, t = svx * invC, u = svg * invC;
        if (live)
            for (int c = 8 * lg; c < C; c += 8 * GL) {
                const F8 xv = ld8(x + p * C + c);
                F8 o;
                if (MODE == 0) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) o.v[j] = xv.v[j] * r;
                    st8(out1 + p * C + c, o);
                } else if (MODE == 1) {
                    const F8 gv = ld8(g + p * C + c);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        o.v[j] = r * gv.v[j] - r3 * s * xv.v[j];
                        if (gate_slope != 1.0f) o.v[j] = gate_bf16((float)(__bf16)o.v[j], xv.v[j], gate_slope);
                    }
                    st8(out1 + p * C + c, o);
                } else {
                    const F8 gv = ld8(g + p * C + c), vv = ld8(v + p * C + c);
                    const float a = -r3 * u + 3.0f * r3 * r * r * s * t, b = -r3 * t, d = -r3 * s;
                    F8 o2;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        o.v[j] = r * vv.v[j] - r3 * t * xv.v[j];
                        o2.v[j] = a * xv.v[j] + b * gv.v[j] + d * vv.v[j];
                    }
                    st8(out1 + p * C + c, o);
                    st8(out2 + p * C + c, o2);
                }
            }
    }
}

template <int MODE>
int launch_pixelnorm(const void *x, const void *g, const void *v, void *o1, void *o2, int64_t npix, int C, float eps,
                     float slope, hipStream_t st) {
    const __bf16 *xb = reinterpret_cast<const __bf16 *>(x), *gb = reinterpret_cast<const __bf16 *>(g),
                 *vb = reinterpret_cast<const __bf16 *>(v);
    __bf16 *a = reinterpret_cast<__bf16 *>(o1), *b = reinterpret_cast<__bf16 *>(o2);
#define SQ_PN(GL_, NCH_) hipLaunchKernelGGL((pixelnorm_bf16_kernel<MODE, GL_, NCH_>), dim3(grid_for(npix * GL_)), dim3(256), 0, st, xb, gb, vb, a, b, npix, C, eps, slope)
    static const bool regs = [] { const char *e = getenv("SQ_PIXELNORM_REGS"); return !(e && e[0] == '0'); }();   // A/B switch
    if (regs && C == 8) SQ_PN(1, 1);
    else if (regs && C == 16) SQ_PN(2, 1);
    else if (regs && C == 32) SQ_PN(4, 1);
    else if (regs && C == 64) SQ_PN(8, 1);
    else if (regs && C == 128) SQ_PN(16, 1);
    else if (regs && C == 256) SQ_PN(16, 2);
    else if (regs && C == 512) SQ_PN(16, 4);
    else if (C <= 8) SQ_PN(1, 0);
    else if (C <= 16) SQ_PN(2, 0);
    else if (C <= 32) SQ_PN(4, 0);
    else if (C <= 64) SQ_PN(8, 0);
    else SQ_PN(16, 0);
#undef SQ_PN
    return 0;
}

// scale * ((a + b) + (c + d)) over every 2x2 patch (average pool: scale 0.25 -- the oracle's order), x (N,H,W,C) -> y (N,H/2,W/2,C)
__global__ __launch_bounds__(256) void sumpool2x2_bf16_kernel(const __bf16 *__restrict__ x, __bf16 *__restrict__ y, int N,
                                                               int Ho, int Wo, int C8, float scale) {
    const int64_t total = (int64_t)N * Ho * Wo * C8;
    const int64_t row = (int64_t)2 * Wo * C8 * 8;               // elements per input row
    SQ_GRID_STRIDE(i, total) {
        const int c = (int)(i % C8);
        int64_t t = i / C8;
        const int ox = (int)(t % Wo);
        t /= Wo;                                                // = n * Ho + oy
        const __bf16 *p = x + (2 * t) * row + ((int64_t)2 * ox * C8 + c) * 8;
        const F8 a = ld8(p), b = ld8(p + C8 * 8), cc = ld8(p + row), d = ld8(p + row + C8 * 8);
        F8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o.v[j] = ((a.v[j] + b.v[j]) + (cc.v[j] + d.v[j])) * scale;
        st8(y + i * 8, o);
    }
}

// scale * src copied to each pixel of its 2x2 patch: src (N,h,w,C) -> dst (N,2h,2w,C).  GATE: dst additionally leaves
// through the backward of the activation whose output is `gate` (N,2h,2w,C): t = bf16(scale * src), dst = gate > 0 ? t : bf16(t * slope)
template <bool GATE>
__global__ __launch_bounds__(256) void broadcast2x2_bf16_kernel(const __bf16 *__restrict__ src, const __bf16 *__restrict__ gate,
                                                                 __bf16 *__restrict__ dst, int N, int h, int w, int C8,
                                                                 float scale, float slope) {
    const int64_t total = (int64_t)N * h * w * C8;
    const int64_t row = (int64_t)2 * w * C8 * 8;
    SQ_GRID_STRIDE(i, total) {
        const int c = (int)(i % C8);
        int64_t t = i / C8;
        const int sx = (int)(t % w);
        t /= w;                                                 // = n * h + sy
        F8 s = ld8(src + i * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) s.v[j] = (float)(__bf16)(s.v[j] * scale);
        const int64_t o = (2 * t) * row + ((int64_t)2 * sx * C8 + c) * 8;
        const int64_t offs[4] = {o, o + C8 * 8, o + row, o + row + C8 * 8};
        if (GATE) {
            F8 g[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) g[q] = ld8(gate + offs[q]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                F8 r;
#pragma unroll
                for (int j = 0; j < 8; ++j) r.v[j] = gate_bf16(s.v[j], g[q].v[j], slope);
                st8(dst + offs[q], r);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) st8(dst + offs[q], s);
        }
    }
}

__global__ __launch_bounds__(256) void act_fwd_bf16_kernel(const __bf16 *__restrict__ x, __bf16 *__restrict__ y, int64_t n8,
                                                            int act) {
    SQ_GRID_STRIDE(i, n8) {
        F8 a = ld8(x + i * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) a.v[j] = sq_act(a.v[j], act);
        st8(y + i * 8, a);
    }
}

// from_image / the dgrad of to_image: y[p][c] = act(sum_a x[p][a] * (w[a][c] * wscale) + bias[c]), x f32 with CA <= 4 channels,
// y bf16 with C % 8 == 0 channels.  Thread = (pixel, 8-channel group); fmaf chain in channel order from 0, as the f32 direct kernel.
template <int CA>
__global__ __launch_bounds__(256) void conv1x1_smallin_bf16_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                                    const float *__restrict__ bias, __bf16 *__restrict__ y,
                                                                    int64_t npix, int C8, float wscale, int act) {
    const int64_t total = npix * C8;
    SQ_GRID_STRIDE(i, total) {
        const int c = (int)(i % C8) * 8;
        const int64_t p = i / C8;
        float xv[CA];
#pragma unroll
        for (int a = 0; a < CA; ++a) xv[a] = x[p * CA + a];
        F8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float acc = 0.f;
#pragma unroll
            for (int a = 0; a < CA; ++a) acc = __builtin_fmaf(w[a * (C8 * 8) + c + j] * wscale, xv[a], acc);
            o.v[j] = sq_act(acc + (bias ? bias[c + j] : 0.f), act);
        }
        st8(y + i * 8, o);
    }
}

// to_image / the dgrad of from_image: y[p][o] = act(sum_c x[p][c] * (w[c][o] * wscale) + bias[o]), x bf16 (C % 8 == 0), y f32 with
// CO <= 4 channels.  Thread per pixel, fmaf chain in channel order from 0 (conv1x1_small_f32_kernel's).
template <int CO>
__global__ __launch_bounds__(256) void conv1x1_smallout_bf16_kernel(const __bf16 *__restrict__ x, const float *__restrict__ w,
                                                                     const float *__restrict__ bias, float *__restrict__ y,
                                                                     int64_t npix, int C, float wscale, int act) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= npix) return;
    float acc[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) acc[o] = 0.f;
    for (int c = 0; c < C; c += 8) {
        const F8 xv = ld8(x + p * C + c);
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int o = 0; o < CO; ++o) acc[o] = __builtin_fmaf(w[(c + j) * CO + o] * wscale, xv.v[j], acc[o]);
    }
#pragma unroll
    for (int o = 0; o < CO; ++o) y[p * CO + o] = sq_act(acc[o] + (bias ? bias[o] : 0.f), act);
}

// M[a][c] = sum_p A[p][a] * B[p][c]: A f32 with CA <= 4 channels (the image side), B bf16 with C % 8 == 0 -- the weight gradient of
// both image-side convolutions and (A = ones) the bias gradient of from_image.  Thread = (pixel lane, 8-channel group), block
// partials [grid][CA][C] (+ [CA] with want_asum: the per-channel sums of A, the bias gradient of to_image, from the same pass),
// fixed-order finish (sq_group_reduce): no atomics, run-to-run identical.
template <int CA>
__global__ __launch_bounds__(256) void wgrad1x1_small_bf16_kernel(const float *__restrict__ a, const __bf16 *__restrict__ b,
                                                                   float *__restrict__ partials, int64_t npix, int C,
                                                                   int want_asum) {
    extern __shared__ float red[];                              // [256][CA * 8]
    const int c8n = C / 8;
    const int gpp = 256 < c8n ? 256 : c8n;                      // channel groups per pass
    const int pl = 256 / gpp;                                   // pixel lanes per block
    const int tg = threadIdx.x % gpp, tp = threadIdx.x / gpp;
    float *out = partials + (size_t)blockIdx.x * (CA * C + (want_asum ? CA : 0));
    float as[CA];
#pragma unroll
    for (int c = 0; c < CA; ++c) as[c] = 0.f;
    for (int gb = 0; gb < c8n; gb += gpp) {
        const int gq = gb + tg;
        const bool sums = want_asum && gq == 0;                // the threads of channel group 0 also add up A
        float acc[CA][8];
#pragma unroll
        for (int c = 0; c < CA; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[c][j] = 0.f;
        if (gq < c8n && tp < pl) {
            // four pixels per trip, their loads issued before the first is used: with one 16-byte load in flight per
            // thread the 256 x 256 levels ran at 0.55 TB/s (a per-thread chain of dependent round trips)
            const int64_t step = (int64_t)gridDim.x * pl;
            int64_t p = (int64_t)blockIdx.x * pl + tp;
            for (; p + 3 * step < npix; p += 4 * step) {
                bf16x8 bq[4];
                float aq[4][CA];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    bq[u] = *reinterpret_cast<const bf16x8 *>(b + (p + u * step) * C + gq * 8);
#pragma unroll
                    for (int c = 0; c < CA; ++c) aq[u][c] = a ? a[(p + u * step) * CA + c] : 1.0f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int c = 0; c < CA; ++c) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[c][j] = __builtin_fmaf(aq[u][c], (float)bq[u][j], acc[c][j]);
                        if (sums) as[c] += aq[u][c];
                    }
            }
            for (; p < npix; p += step) {
                const F8 bv = ld8(b + p * C + gq * 8);
#pragma unroll
                for (int c = 0; c < CA; ++c) {
                    const float av = a ? a[p * CA + c] : 1.0f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[c][j] = __builtin_fmaf(av, bv.v[j], acc[c][j]);
                    if (sums) as[c] += av;
                }
            }
        }
#pragma unroll
        for (int c = 0; c < CA; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) red[(c * 8 + j) * 256 + threadIdx.x] = acc[c][j];
        // fold the pixel lanes with a fixed tree (pl is a power of two; thread = tp * gpp + tg): one thread adding all pl lanes
        // of its channel group was 4096 serial LDS reads per block at C = 8 -- most of this kernel's time
        for (int st = pl >> 1; st > 0; st >>= 1) {
            __syncthreads();
            if (tp < st) {
#pragma unroll
                for (int e = 0; e < CA * 8; ++e) red[e * 256 + threadIdx.x] += red[e * 256 + threadIdx.x + st * gpp];
            }
        }
        __syncthreads();
        if (tp == 0 && gq < c8n) {
#pragma unroll
            for (int c = 0; c < CA; ++c)
#pragma unroll
                for (int j = 0; j < 8; ++j) out[c * C + gq * 8 + j] = red[(c * 8 + j) * 256 + tg];
        }
        __syncthreads();
    }
    if (want_asum) {                                            // pixel lane tp of channel group 0 holds its share of the sums
        if (tg == 0 && tp < pl)
#pragma unroll
            for (int c = 0; c < CA; ++c) red[tp * CA + c] = as[c];
        __syncthreads();
        if ((int)threadIdx.x < CA) {
            float s2 = 0.f;
            for (int k = 0; k < pl; ++k) s2 += red[k * CA + threadIdx.x];
            out[CA * C + threadIdx.x] = s2;
        }
    }
}
__global__ __launch_bounds__(256) void wgrad1x1_small_bf16_finish_kernel(const float *__restrict__ partials, float *__restrict__ m,
                                                                          float *__restrict__ asum, int nblk, int nm, int total,
                                                                          int G, float scale) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int i = t / G, g = t % G;
    if (i >= total) return;
    const float s = sq_group_reduce(partials + i, (size_t)total, nblk, g, G);
    if (g != 0) return;
    if (i < nm) m[i] = scale == 1.0f ? s : s * scale;
    else asum[i - nm] = s;
}

inline int small_blocks(int64_t npix) {
    int64_t b = (npix + 255) / 256;
    return (int)(b > 1024 ? 1024 : (b < 1 ? 1 : b));
}

// The discriminator's output block (gan.py:213-226): concat([conv (N,P,C) features, minibatch-stdev map (N,P,1)], -1) flattened to
// (N, P (C+1)) float32 rows for the dense layer -- from the bf16 conv output and the f32 map in ONE pass (it was a cast, a
// concatenation and, in every backward pass, two slice copies, a cast and the zero fills of the slices' gradients)
__global__ __launch_bounds__(256) void head_concat_fwd_kernel(const __bf16 *__restrict__ conv, const float *__restrict__ mb,
                                                               float *__restrict__ flat, int64_t npix, int C) {
    const int64_t total = npix * (C + 1);
    SQ_GRID_STRIDE(i, total) {
        const int64_t pix = i / (C + 1);
        const int c = (int)(i - pix * (C + 1));
        flat[i] = c < C ? (float)conv[pix * C + c] : mb[pix];
    }
}
// its adjoint: dconv (N,P,C) = bf16(dflat[..., :C]), dmb (N,P) = dflat[..., C]
__global__ __launch_bounds__(256) void head_concat_bwd_kernel(const float *__restrict__ dflat, __bf16 *__restrict__ dconv,
                                                               float *__restrict__ dmb, int64_t npix, int C) {
    const int64_t total = npix * (C + 1);
    SQ_GRID_STRIDE(i, total) {
        const int64_t pix = i / (C + 1);
        const int c = (int)(i - pix * (C + 1));
        const float v = dflat[i];
        if (c < C) dconv[pix * C + c] = (__bf16)v;
        else dmb[pix] = v;
    }
}

}  // namespace

#define SQ_ST(s) reinterpret_cast<hipStream_t>(s)
#define BF(p) reinterpret_cast<const __bf16 *>(p)
#define BFM(p) reinterpret_cast<__bf16 *>(p)

extern "C" int sq_pixelnorm_fwd_bf16(const void *x, void *y, int64_t npix, int C, float eps, void *stream) {
    SQ_REQUIRE(x && y && npix > 0 && C > 0 && C % 8 == 0, "sq_pixelnorm_fwd_bf16: bad arguments (C %% 8 == 0)");
    SQ_REQUIRE_ALIGNED(x); SQ_REQUIRE_ALIGNED(y);
    launch_pixelnorm<0>(x, nullptr, nullptr, y, nullptr, npix, C, eps, 1.0f, SQ_ST(stream));
    return sq_check_launch("sq_pixelnorm_fwd_bf16");
}

// act: SQ_ACT_NONE = plain backward; RELU / LEAKY: x is that activation's output and dx leaves through its backward too
extern "C" int sq_pixelnorm_bwd_bf16(const void *x, const void *dy, void *dx, int64_t npix, int C, float eps, int act,
                                     void *stream) {
    SQ_REQUIRE(x && dy && dx && npix > 0 && C > 0 && C % 8 == 0, "sq_pixelnorm_bwd_bf16: bad arguments (C %% 8 == 0)");
    SQ_REQUIRE(act >= SQ_ACT_NONE && act <= SQ_ACT_LEAKY, "sq_pixelnorm_bwd_bf16: bad activation %d", act);
    SQ_REQUIRE_ALIGNED(x); SQ_REQUIRE_ALIGNED(dy); SQ_REQUIRE_ALIGNED(dx);
    const float slope = act == SQ_ACT_LEAKY ? 0.2f : (act == SQ_ACT_RELU ? 0.0f : 1.0f);
    launch_pixelnorm<1>(x, dy, nullptr, dx, nullptr, npix, C, eps, slope, SQ_ST(stream));
    return sq_check_launch("sq_pixelnorm_bwd_bf16");
}

extern "C" int sq_pixelnorm_bwd2_bf16(const void *x, const void *g, const void *v, void *dg, void *dx2, int64_t npix, int C,
                                      float eps, void *stream) {
    SQ_REQUIRE(x && g && v && dg && dx2 && npix > 0 && C > 0 && C % 8 == 0, "sq_pixelnorm_bwd2_bf16: bad arguments (C %% 8 == 0)");
    SQ_REQUIRE_ALIGNED(x); SQ_REQUIRE_ALIGNED(g); SQ_REQUIRE_ALIGNED(v); SQ_REQUIRE_ALIGNED(dg); SQ_REQUIRE_ALIGNED(dx2);
    launch_pixelnorm<2>(x, g, v, dg, dx2, npix, C, eps, 1.0f, SQ_ST(stream));
    return sq_check_launch("sq_pixelnorm_bwd2_bf16");
}

// x (N,H,W,C) -> y (N,H/2,W/2,C), H and W even
extern "C" int sq_sumpool2x2_bf16(const void *x, void *y, int N, int H, int W, int C, float scale, void *stream) {
    SQ_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && C % 8 == 0,
               "sq_sumpool2x2_bf16: bad arguments (even H, W; C %% 8 == 0)");
    SQ_REQUIRE_ALIGNED(x); SQ_REQUIRE_ALIGNED(y);
    hipLaunchKernelGGL(sumpool2x2_bf16_kernel, dim3(grid_for((int64_t)N * (H / 2) * (W / 2) * (C / 8))), dim3(256), 0,
                       SQ_ST(stream), BF(x), BFM(y), N, H / 2, W / 2, C / 8, scale);
    return sq_check_launch("sq_sumpool2x2_bf16");
}

// src (N,H/2,W/2,C) -> dst (N,H,W,C); H, W are the DESTINATION size (as sq_broadcast2x2_f32)
extern "C" int sq_broadcast2x2_bf16(const void *src, void *dst, int N, int H, int W, int C, float scale, void *stream) {
    SQ_REQUIRE(src && dst && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && C % 8 == 0,
               "sq_broadcast2x2_bf16: bad arguments (even H, W; C %% 8 == 0)");
    SQ_REQUIRE_ALIGNED(src); SQ_REQUIRE_ALIGNED(dst);
    hipLaunchKernelGGL(broadcast2x2_bf16_kernel<false>, dim3(grid_for((int64_t)N * (H / 2) * (W / 2) * (C / 8))), dim3(256), 0,
                       SQ_ST(stream), BF(src), (const __bf16 *)nullptr, BFM(dst), N, H / 2, W / 2, C / 8, scale, 1.0f);
    return sq_check_launch("sq_broadcast2x2_bf16");
}

extern "C" int sq_broadcast2x2_act_bwd_bf16(const void *src, const void *gate, void *dst, int N, int H, int W, int C,
                                            float scale, int act, void *stream) {
    SQ_REQUIRE(src && gate && dst && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && C % 8 == 0,
               "sq_broadcast2x2_act_bwd_bf16: bad arguments (even H, W; C %% 8 == 0)");
    SQ_REQUIRE(act == SQ_ACT_RELU || act == SQ_ACT_LEAKY, "sq_broadcast2x2_act_bwd_bf16: activation %d has no gate", act);
    SQ_REQUIRE_ALIGNED(src); SQ_REQUIRE_ALIGNED(gate); SQ_REQUIRE_ALIGNED(dst);
    hipLaunchKernelGGL(broadcast2x2_bf16_kernel<true>, dim3(grid_for((int64_t)N * (H / 2) * (W / 2) * (C / 8))), dim3(256), 0,
                       SQ_ST(stream), BF(src), BF(gate), BFM(dst), N, H / 2, W / 2, C / 8, scale,
                       act == SQ_ACT_LEAKY ? 0.2f : 0.0f);
    return sq_check_launch("sq_broadcast2x2_act_bwd_bf16");
}

extern "C" int sq_act_fwd_bf16(const void *x, void *y, int64_t n, int act, void *stream) {
    SQ_REQUIRE(x && y && n > 0 && n % 8 == 0, "sq_act_fwd_bf16: bad arguments (n %% 8 == 0)");
    SQ_REQUIRE(act >= SQ_ACT_NONE && act <= SQ_ACT_LEAKY, "sq_act_fwd_bf16: bad activation %d", act);
    SQ_REQUIRE_ALIGNED(x); SQ_REQUIRE_ALIGNED(y);
    hipLaunchKernelGGL(act_fwd_bf16_kernel, dim3(grid_for(n / 8)), dim3(256), 0, SQ_ST(stream), BF(x), BFM(y), n / 8, act);
    return sq_check_launch("sq_act_fwd_bf16");
}

// x f32 (npix, Ca), w f32 (Ca, C) row-major, bias f32 (C) or NULL -> y bf16 (npix, C).  Ca 1..4, C % 8 == 0.
extern "C" int sq_conv1x1_smallin_fwd_bf16(const float *x, const float *w, const float *bias, void *y, int64_t npix, int Ca,
                                           int C, float wscale, int act, void *stream) {
    SQ_REQUIRE(x && w && y && npix > 0 && Ca >= 1 && Ca <= 4 && C > 0 && C % 8 == 0,
               "sq_conv1x1_smallin_fwd_bf16: Ca=%d (1..4), C=%d (multiple of 8)", Ca, C);
    SQ_REQUIRE(act >= SQ_ACT_NONE && act <= SQ_ACT_LEAKY, "sq_conv1x1_smallin_fwd_bf16: bad activation %d", act);
    SQ_REQUIRE_ALIGNED(y);
    const unsigned nb = grid_for(npix * (C / 8));
    hipStream_t st = SQ_ST(stream);
    switch (Ca) {
    case 1: hipLaunchKernelGGL(conv1x1_smallin_bf16_kernel<1>, dim3(nb), dim3(256), 0, st, x, w, bias, BFM(y), npix, C / 8, wscale, act); break;
    case 2: hipLaunchKernelGGL(conv1x1_smallin_bf16_kernel<2>, dim3(nb), dim3(256), 0, st, x, w, bias, BFM(y), npix, C / 8, wscale, act); break;
    case 3: hipLaunchKernelGGL(conv1x1_smallin_bf16_kernel<3>, dim3(nb), dim3(256), 0, st, x, w, bias, BFM(y), npix, C / 8, wscale, act); break;
    default: hipLaunchKernelGGL(conv1x1_smallin_bf16_kernel<4>, dim3(nb), dim3(256), 0, st, x, w, bias, BFM(y), npix, C / 8, wscale, act); break;
    }
    return sq_check_launch("sq_conv1x1_smallin_fwd_bf16");
}

// x bf16 (npix, C), w f32 (C, Co) row-major, bias f32 (Co) or NULL -> y f32 (npix, Co).  C % 8 == 0, Co 1..4.
extern "C" int sq_conv1x1_smallout_fwd_bf16(const void *x, const float *w, const float *bias, float *y, int64_t npix, int C,
                                            int Co, float wscale, int act, void *stream) {
    SQ_REQUIRE(x && w && y && npix > 0 && Co >= 1 && Co <= 4 && C > 0 && C % 8 == 0,
               "sq_conv1x1_smallout_fwd_bf16: C=%d (multiple of 8), Co=%d (1..4)", C, Co);
    SQ_REQUIRE(act >= SQ_ACT_NONE && act <= SQ_ACT_LEAKY, "sq_conv1x1_smallout_fwd_bf16: bad activation %d", act);
    SQ_REQUIRE_ALIGNED(x);
    const unsigned nb = (unsigned)((npix + 255) / 256);
    hipStream_t st = SQ_ST(stream);
    switch (Co) {
    case 1: hipLaunchKernelGGL(conv1x1_smallout_bf16_kernel<1>, dim3(nb), dim3(256), 0, st, BF(x), w, bias, y, npix, C, wscale, act); break;
    case 2: hipLaunchKernelGGL(conv1x1_smallout_bf16_kernel<2>, dim3(nb), dim3(256), 0, st, BF(x), w, bias, y, npix, C, wscale, act); break;
    case 3: hipLaunchKernelGGL(conv1x1_smallout_bf16_kernel<3>, dim3(nb), dim3(256), 0, st, BF(x), w, bias, y, npix, C, wscale, act); break;
    default: hipLaunchKernelGGL(conv1x1_smallout_bf16_kernel<4>, dim3(nb), dim3(256), 0, st, BF(x), w, bias, y, npix, C, wscale, act); break;
    }
    return sq_check_launch("sq_conv1x1_smallout_fwd_bf16");
}

extern "C" int64_t sq_wgrad1x1_small_workspace_bf16(int64_t npix, int Ca, int C) {
    if (npix <= 0 || Ca < 1 || Ca > 4 || C <= 0 || C % 8) return -1;
    return (int64_t)small_blocks(npix) * (Ca * C + Ca) * 4;
}

// m f32 (Ca, C) = scale * sum_p a[p][:]^T b[p][:]; a f32 (npix, Ca) or NULL (= ones: the per-channel sums of b, Ca must be 1),
// b bf16 (npix, C).  asum (Ca) f32 or NULL: the per-channel sums of a over the pixels (not scaled), from the same pass.
// workspace: sq_wgrad1x1_small_workspace_bf16 bytes.
extern "C" int sq_wgrad1x1_small_bf16(const float *a, const void *b, float *m, float *asum, float *workspace, int64_t npix,
                                      int Ca, int C, float scale, void *stream) {
    SQ_REQUIRE(b && m && workspace, "sq_wgrad1x1_small_bf16: null pointer");
    SQ_REQUIRE(npix > 0 && Ca >= 1 && Ca <= 4 && C > 0 && C % 8 == 0 && (a || Ca == 1) && (a || !asum),
               "sq_wgrad1x1_small_bf16: Ca=%d (1..4; 1 when a is NULL), C=%d (multiple of 8)", Ca, C);
    const int wa = asum ? 1 : 0;
    SQ_REQUIRE_ALIGNED(b);
    const int nb = small_blocks(npix);
    hipStream_t st = SQ_ST(stream);
    const size_t lds = 256 * Ca * 8 * sizeof(float);
    switch (Ca) {
    case 1: hipLaunchKernelGGL(wgrad1x1_small_bf16_kernel<1>, dim3(nb), dim3(256), lds, st, a, BF(b), workspace, npix, C, wa); break;
    case 2: hipLaunchKernelGGL(wgrad1x1_small_bf16_kernel<2>, dim3(nb), dim3(256), lds, st, a, BF(b), workspace, npix, C, wa); break;
    case 3: hipLaunchKernelGGL(wgrad1x1_small_bf16_kernel<3>, dim3(nb), dim3(256), lds, st, a, BF(b), workspace, npix, C, wa); break;
    default: hipLaunchKernelGGL(wgrad1x1_small_bf16_kernel<4>, dim3(nb), dim3(256), lds, st, a, BF(b), workspace, npix, C, wa); break;
    }
    int rc = sq_check_launch("sq_wgrad1x1_small_bf16");
    if (rc) return rc;
    const int nm = Ca * C, total = nm + (wa ? Ca : 0);
    const int G = sq_group_size(nb);
    hipLaunchKernelGGL(wgrad1x1_small_bf16_finish_kernel, dim3((total * G + 255) / 256), dim3(256), 0, st, workspace, m, asum, nb, nm,
                       total, G, scale);
    return sq_check_launch("sq_wgrad1x1_small_bf16(finish)");
}

// concat([float32(conv), mb[..., None]], -1) over the last axis: conv bf16 (npix, C), mb f32 (npix) -> flat f32 (npix, C + 1)
extern "C" int sq_head_concat_fwd_bf16(const void *conv, const float *mb, float *flat, int64_t npix, int C, void *stream) {
    SQ_REQUIRE(conv && mb && flat && npix > 0 && C > 0, "sq_head_concat_fwd_bf16: bad arguments");
    hipLaunchKernelGGL(head_concat_fwd_kernel, dim3(grid_for(npix * (C + 1))), dim3(256), 0, SQ_ST(stream), BF(conv), mb, flat, npix, C);
    return sq_check_launch("sq_head_concat_fwd_bf16");
}
// the adjoint split: dflat f32 (npix, C + 1) -> dconv bf16 (npix, C) (rounded to nearest even) and dmb f32 (npix)
extern "C" int sq_head_concat_bwd_bf16(const float *dflat, void *dconv, float *dmb, int64_t npix, int C, void *stream) {
    SQ_REQUIRE(dflat && dconv && dmb && npix > 0 && C > 0, "sq_head_concat_bwd_bf16: bad arguments");
    hipLaunchKernelGGL(head_concat_bwd_kernel, dim3(grid_for(npix * (C + 1))), dim3(256), 0, SQ_ST(stream), dflat,
                       reinterpret_cast<__bf16 *>(dconv), dmb, npix, C);
    return sq_check_launch("sq_head_concat_bwd_bf16");
}
