// f32 NHWC convolutions for gfx950 (MI355X): the conv_layer / conv_layer_1x1 hooks of
// sequitr/networks/unet.py:326-333 and weighted_conv2d of sequitr/networks/gan.py:61-99.
//
// Main kernel: implicit GEMM on v_mfma_f32_16x16x4_f32 (exact f32, bit-for-bit a
// k-ordered fmaf chain).  Roles are swapped relative to the textbook GEMM so the
// accumulator layout gives every lane 4 CONSECUTIVE output channels of one pixel:
//     A[i][k] = W'[k][cout i]      (16 output channels x 4 reduction slots)
//     B[k][j] = X [pixel j][k]     (4 reduction slots x 16 pixels of one tile row)
//     D[i][j] : lane l holds couts 4*(l>>4)+{0..3} of pixel (l&15)  -> one 16-B store.
// Reduction order (the numerics contract, include/sequitr_hip.h): 16-channel chunk,
// tap (raster), channel -- one accumulator per output, never split.
//
// Block = 256 threads = 4 waves, output tile 16x16 pixels x BN channels; wave w owns
// tile rows 4w..4w+3.  Per chunk the (16+K-1)^2 halo (KC channels, pixel stride KC+2
// floats => conflict-free ds_read_b32 for the B operand) and the K*K*KC x BN weight
// slab (row stride BN or BN+16 => conflict-free A operand) are staged in LDS.
#include <stdlib.h>
#include "sq_common.h"

namespace {

constexpr int TH = 16, TW = 16;

template <int BN, int KS, int KC>
struct ConvCfg {
    static constexpr int HALO_W = TW + KS - 1;
    static constexpr int HALO_H = TH + KS - 1;
    static constexpr int HP = HALO_W * HALO_H;          // halo pixels
    static constexpr int PS = KC + 2;                   // pixel stride in floats
    static constexpr int BNS = (BN % 32 == 0) ? BN + 16 : BN;
    static constexpr int WROWS = KS * KS * KC;
    static constexpr int XS_FLOATS = HP * PS;
    static constexpr int LDS_BYTES = (XS_FLOATS + WROWS * BNS) * 4;
    static_assert((XS_FLOATS * 4) % 16 == 0, "weight slab must start 16-B aligned");
};

template <int BN, int KS, int KC>
__global__ __launch_bounds__(256) void conv_mfma_f32_kernel(
    const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
    float *__restrict__ y, int N, int H, int W, int Cin, int Cout, float wscale, int act,
    int tiles_x, int tiles_y, unsigned nblk_sp) {
    using C = ConvCfg<BN, KS, KC>;
    constexpr int NR = BN / 16;
    constexpr int PAD = KS / 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *xs = smem;
    float *ws = smem + C::XS_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, kk = lane >> 4;
    const unsigned sp = sq_xcd_remap(blockIdx.x, nblk_sp);
    const int tx = sp % tiles_x, ty = (sp / tiles_x) % tiles_y, n = sp / (tiles_x * tiles_y);
    const int x0 = tx * TW, y0 = ty * TH, n0 = blockIdx.y * BN;

    f32x4 acc[4][NR];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int nb = 0; nb < NR; ++nb) acc[r][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const float *xb_lds = xs + ((4 * wv) * C::HALO_W + li) * C::PS + kk;
    const float *wa_lds = ws + kk * C::BNS + li;

    for (int cc = 0; cc < Cin; cc += KC) {
        // ---- stage the halo chunk: HP pixels x KC channels -------------------------
        constexpr int QPP = KC / 4;
        for (int idx = tid; idx < C::HP * QPP; idx += 256) {
            const int pix = idx / QPP, q = idx % QPP;
            const int py = pix / C::HALO_W, px = pix % C::HALO_W;
            const int gy = y0 - PAD + py, gx = x0 - PAD + px;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy >= 0 && gy < H && gx >= 0 && gx < W)
                v = *reinterpret_cast<const float4 *>(x + ((size_t)(n * H + gy) * W + gx) * Cin + cc + q * 4);
            float *d = xs + pix * C::PS + q * 4;
            *reinterpret_cast<float2 *>(d) = make_float2(v.x, v.y);
            *reinterpret_cast<float2 *>(d + 2) = make_float2(v.z, v.w);
        }
        // ---- stage the weight slab: (tap, c) rows x BN couts, scaled ----------------
        constexpr int Q4 = BN / 4;
        for (int idx = tid; idx < C::WROWS * Q4; idx += 256) {
            const int r = idx / Q4, q4 = idx % Q4;
            const int tap = r / KC, c = r % KC;
            const int co = n0 + q4 * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (co < Cout)
                v = *reinterpret_cast<const float4 *>(w + ((size_t)(tap * Cin + cc + c)) * Cout + co);
            v.x *= wscale; v.y *= wscale; v.z *= wscale; v.w *= wscale;
            *reinterpret_cast<float4 *>(ws + r * C::BNS + q4 * 4) = v;
        }
        __syncthreads();

        // ---- K*K taps x KC/4 MFMA steps ---------------------------------------------
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
#pragma unroll
                for (int s = 0; s < KC / 4; ++s) {
                    float a[NR], b[4];
#pragma unroll
                    for (int nb = 0; nb < NR; ++nb)
                        a[nb] = wa_lds[((ky * KS + kx) * KC + s * 4) * C::BNS + nb * 16];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        b[r] = xb_lds[((r + ky) * C::HALO_W + kx) * C::PS + s * 4];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int nb = 0; nb < NR; ++nb)
                            acc[r][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[nb], b[r], acc[r][nb], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // ---- epilogue: + bias, activation, one 16-B store per (row, cout block) ---------
    const int gx = x0 + li;
#pragma unroll
    for (int nb = 0; nb < NR; ++nb) {
        const int co = n0 + nb * 16 + 4 * kk;
        if (co >= Cout) continue;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias) bv = *reinterpret_cast<const float4 *>(bias + co);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gy = y0 + 4 * wv + r;
            if (gy < H && gx < W) {
                float4 o;
                o.x = sq_act(acc[r][nb][0] + (bias ? bv.x : 0.f), act);
                o.y = sq_act(acc[r][nb][1] + (bias ? bv.y : 0.f), act);
                o.z = sq_act(acc[r][nb][2] + (bias ? bv.z : 0.f), act);
                o.w = sq_act(acc[r][nb][3] + (bias ? bv.w : 0.f), act);
                *reinterpret_cast<float4 *>(y + ((size_t)(n * H + gy) * W + gx) * Cout + co) = o;
            }
        }
    }
}

// Direct VALU convolution for Cin in {1,2}: a 9- or 18-deep reduction cannot fill an
// MFMA tile.  Thread per pixel, 16 output channels per thread (blockIdx.y = cout group).
template <int CIN, int KS>
__global__ __launch_bounds__(256) void conv_direct_f32_kernel(
    const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
    float *__restrict__ y, int N, int H, int W, int Cout, float wscale, int act,
    int tiles_x, int tiles_y) {
    constexpr int HALO_W = TW + KS - 1, HALO_H = TH + KS - 1, HP = HALO_W * HALO_H, PAD = KS / 2;
    __shared__ float xs[HP * CIN];
    __shared__ float ws[KS * KS * CIN * 16];
    const int tid = threadIdx.x;
    const int sp = blockIdx.x;
    const int tx = sp % tiles_x, ty = (sp / tiles_x) % tiles_y, n = sp / (tiles_x * tiles_y);
    const int x0 = tx * TW, y0 = ty * TH, n0 = blockIdx.y * 16;
    for (int idx = tid; idx < HP * CIN; idx += 256) {
        const int pix = idx / CIN, c = idx % CIN;
        const int gy = y0 - PAD + pix / HALO_W, gx = x0 - PAD + pix % HALO_W;
        float v = 0.f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = x[((size_t)(n * H + gy) * W + gx) * CIN + c];
        xs[idx] = v;
    }
    for (int idx = tid; idx < KS * KS * CIN * 16; idx += 256) {
        const int r = idx / 16, o = idx % 16;
        ws[idx] = (n0 + o < Cout) ? w[(size_t)r * Cout + n0 + o] * wscale : 0.f;
    }
    __syncthreads();
    const int py = tid >> 4, px = tid & 15;
    float acc[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) acc[o] = 0.f;
    // more than 9 filter rows: keep the tap loops rolled, or the compiler hoists every (wave-uniform) weight read
    // into registers and spills
    constexpr int TAP_UNROLL = KS * KS * CIN <= 4 ? KS : 1;      // (the 9-row single-channel filter too: 155 -> 38 VGPRs)
#pragma unroll TAP_UNROLL
    for (int ky = 0; ky < KS; ++ky)
#pragma unroll TAP_UNROLL
        for (int kx = 0; kx < KS; ++kx)
#pragma unroll
            for (int c = 0; c < CIN; ++c) {
                const float xv = xs[((py + ky) * HALO_W + px + kx) * CIN + c];
                const float *wr = ws + ((ky * KS + kx) * CIN + c) * 16;
#pragma unroll
                for (int o = 0; o < 16; ++o) acc[o] = __builtin_fmaf(wr[o], xv, acc[o]);
            }
    const int gy = y0 + py, gx = x0 + px;
    if (gy < H && gx < W) {
        float *yo = y + ((size_t)(n * H + gy) * W + gx) * Cout + n0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (n0 + q * 4 >= Cout) break;
            float4 o;
            o.x = sq_act(acc[q * 4 + 0] + (bias ? bias[n0 + q * 4 + 0] : 0.f), act);
            o.y = sq_act(acc[q * 4 + 1] + (bias ? bias[n0 + q * 4 + 1] : 0.f), act);
            o.z = sq_act(acc[q * 4 + 2] + (bias ? bias[n0 + q * 4 + 2] : 0.f), act);
            o.w = sq_act(acc[q * 4 + 3] + (bias ? bias[n0 + q * 4 + 3] : 0.f), act);
            *reinterpret_cast<float4 *>(yo + q * 4) = o;
        }
    }
}

// 1x1 convolution to a handful of channels (to_image heads, Cout <= 4), thread per
// pixel, optionally fused with the prediction argmax (ties -> lowest index).
template <int COUT>
__global__ __launch_bounds__(256) void conv1x1_small_f32_kernel(
    const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
    float *__restrict__ logits, uint8_t *__restrict__ mask, int64_t npix, int Cin,
    float wscale, int act) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= npix) return;
    float acc[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) acc[o] = 0.f;
    const float4 *xp = reinterpret_cast<const float4 *>(x + p * Cin);
    for (int c4 = 0; c4 < Cin / 4; ++c4) {
        const float4 v = xp[c4];
        const float xv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int o = 0; o < COUT; ++o)
                acc[o] = __builtin_fmaf(w[(c4 * 4 + j) * COUT + o] * wscale, xv[j], acc[o]);
    }
    int best = 0;
#pragma unroll
    for (int o = 0; o < COUT; ++o) {
        acc[o] = sq_act(acc[o] + (bias ? bias[o] : 0.f), act);
        logits[p * COUT + o] = acc[o];
    }
    if (mask) {
        float bestv = acc[0];
#pragma unroll
        for (int o = 1; o < COUT; ++o)
            if (acc[o] > bestv) { bestv = acc[o]; best = o; }
        mask[p] = (uint8_t)best;
    }
}

template <int BN, int KS, int KC>
int launch_mfma(const float *x, const float *w, const float *bias, float *y, int N, int H, int W,
                int Cin, int Cout, float wscale, int act, hipStream_t st) {
    using C = ConvCfg<BN, KS, KC>;
    static bool attr_set = false;
    auto kern = conv_mfma_f32_kernel<BN, KS, KC>;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES) != hipSuccess) {
            sq_set_error("conv_mfma_f32: cannot reserve %d bytes of LDS", C::LDS_BYTES);
            return SQ_ELAUNCH;
        }
        attr_set = true;
    }
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const unsigned nblk = (unsigned)(tiles_x * tiles_y) * (unsigned)N;
    dim3 grid(nblk, (Cout + BN - 1) / BN);
    hipLaunchKernelGGL(kern, grid, dim3(256), C::LDS_BYTES, st, x, w, bias, y, N, H, W, Cin, Cout,
                       wscale, act, tiles_x, tiles_y, nblk);
    return sq_check_launch("sq_conv2d_nhwc_fwd_f32");
}

template <int KS, int KC>
int dispatch_bn(const float *x, const float *w, const float *bias, float *y, int N, int H, int W,
                int Cin, int Cout, float wscale, int act, hipStream_t st) {
    if (Cout >= 64) return launch_mfma<64, KS, KC>(x, w, bias, y, N, H, W, Cin, Cout, wscale, act, st);
    if (Cout > 16) return launch_mfma<32, KS, KC>(x, w, bias, y, N, H, W, Cin, Cout, wscale, act, st);
    return launch_mfma<16, KS, KC>(x, w, bias, y, N, H, W, Cin, Cout, wscale, act, st);
}

template <int CIN, int KS>
int launch_direct(const float *x, const float *w, const float *bias, float *y, int N, int H, int W,
                  int Cout, float wscale, int act, hipStream_t st) {
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    dim3 grid((unsigned)(tiles_x * tiles_y) * (unsigned)N, (Cout + 15) / 16);
    hipLaunchKernelGGL((conv_direct_f32_kernel<CIN, KS>), grid, dim3(256), 0, st, x, w, bias, y, N, H,
                       W, Cout, wscale, act, tiles_x, tiles_y);
    return sq_check_launch("sq_conv2d_nhwc_fwd_f32(direct)");
}

int launch_small(const float *x, const float *w, const float *bias, float *logits, uint8_t *mask,
                 int64_t npix, int Cin, int Cout, float wscale, int act, hipStream_t st,
                 const char *what) {
    const unsigned nb = (unsigned)((npix + 255) / 256);
    switch (Cout) {
    case 1: hipLaunchKernelGGL(conv1x1_small_f32_kernel<1>, dim3(nb), dim3(256), 0, st, x, w, bias, logits, mask, npix, Cin, wscale, act); break;
    case 2: hipLaunchKernelGGL(conv1x1_small_f32_kernel<2>, dim3(nb), dim3(256), 0, st, x, w, bias, logits, mask, npix, Cin, wscale, act); break;
    case 3: hipLaunchKernelGGL(conv1x1_small_f32_kernel<3>, dim3(nb), dim3(256), 0, st, x, w, bias, logits, mask, npix, Cin, wscale, act); break;
    case 4: hipLaunchKernelGGL(conv1x1_small_f32_kernel<4>, dim3(nb), dim3(256), 0, st, x, w, bias, logits, mask, npix, Cin, wscale, act); break;
    case 5: hipLaunchKernelGGL(conv1x1_small_f32_kernel<5>, dim3(nb), dim3(256), 0, st, x, w, bias, logits, mask, npix, Cin, wscale, act); break;
    case 6: hipLaunchKernelGGL(conv1x1_small_f32_kernel<6>, dim3(nb), dim3(256), 0, st, x, w, bias, logits, mask, npix, Cin, wscale, act); break;
    case 7: hipLaunchKernelGGL(conv1x1_small_f32_kernel<7>, dim3(nb), dim3(256), 0, st, x, w, bias, logits, mask, npix, Cin, wscale, act); break;
    default: sq_set_error("%s: Cout=%d unsupported (1..7)", what, Cout); return SQ_EINVAL;
    }
    return sq_check_launch(what);
}

// SQ_CONV_IMPL=1 selects the simple one-tile-per-block kernel of this file (kept for in-process
// A/B timing); the default (2) is the pipelined persistent kernel of sq_conv_f32_v2.hip.  Both
// produce identical bits.  Read once; not a correctness knob.
int conv_impl() {
    static int impl = 0;
    if (!impl) {
        const char *e = getenv("SQ_CONV_IMPL");
        impl = (e && e[0] == '1') ? 1 : 2;
    }
    return impl;
}

}  // namespace

int sq_conv_mfma_v2(const float *x, const float *w, const float *bias, float *y, int N, int H, int W,
                    int Cin, int Cout, int K, float wscale, int act, hipStream_t st);

extern "C" int sq_conv2d_nhwc_fwd_f32(const float *x, const float *w, const float *bias, float *y,
                                      int N, int H, int W, int Cin, int Cout, int K, float wscale,
                                      int act, void *stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    SQ_REQUIRE(x && w && y, "sq_conv2d_nhwc_fwd_f32: null tensor pointer");
    SQ_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "sq_conv2d_nhwc_fwd_f32: bad shape");
    SQ_REQUIRE(K == 1 || K == 3, "sq_conv2d_nhwc_fwd_f32: K=%d unsupported (1 or 3)", K);
    SQ_REQUIRE(act >= SQ_ACT_NONE && act <= SQ_ACT_LEAKY, "sq_conv2d_nhwc_fwd_f32: bad activation %d", act);
    SQ_REQUIRE_ALIGNED(x); SQ_REQUIRE_ALIGNED(w); SQ_REQUIRE_ALIGNED(y);
    if (bias) SQ_REQUIRE_ALIGNED(bias);
    if (K == 1 && Cout <= 7 && Cin % 4 == 0)                  // class heads (the reference allows up to 5 classes)
        return launch_small(x, w, bias, y, nullptr, (int64_t)N * H * W, Cin, Cout, wscale, act, st,
                            "sq_conv2d_nhwc_fwd_f32(1x1)");
    SQ_REQUIRE(Cout % 4 == 0, "sq_conv2d_nhwc_fwd_f32: Cout=%d must be a multiple of 4", Cout);
    if (Cin == 1)
        return K == 3 ? launch_direct<1, 3>(x, w, bias, y, N, H, W, Cout, wscale, act, st)
                      : launch_direct<1, 1>(x, w, bias, y, N, H, W, Cout, wscale, act, st);
    if (Cin == 2)
        return K == 3 ? launch_direct<2, 3>(x, w, bias, y, N, H, W, Cout, wscale, act, st)
                      : launch_direct<2, 1>(x, w, bias, y, N, H, W, Cout, wscale, act, st);
    if (Cin >= 3 && Cin <= 7) {                                 // multi-channel tiles (K = 3); dgrad of a 3..7-class head (K = 1)
#define SQ_DIRECT(C)                                                                                \
    case C:                                                                                         \
        return K == 3 ? launch_direct<C, 3>(x, w, bias, y, N, H, W, Cout, wscale, act, st)          \
                      : launch_direct<C, 1>(x, w, bias, y, N, H, W, Cout, wscale, act, st);
        switch (Cin) { SQ_DIRECT(3) SQ_DIRECT(4) SQ_DIRECT(5) SQ_DIRECT(6) SQ_DIRECT(7) }
#undef SQ_DIRECT
    }
    // the pipelined kernel addresses tensors through 32-bit buffer offsets: < 2 GiB each
    const bool fits32 = (size_t)N * H * W * (size_t)(Cin > Cout ? Cin : Cout) * 4 < ((size_t)1 << 31);
    if ((Cin % 16 == 0 || Cin == 8) && conv_impl() == 2 && fits32)
        return sq_conv_mfma_v2(x, w, bias, y, N, H, W, Cin, Cout, K, wscale, act, st);
    if (Cin % 16 == 0)
        return K == 3 ? dispatch_bn<3, 16>(x, w, bias, y, N, H, W, Cin, Cout, wscale, act, st)
                      : dispatch_bn<1, 16>(x, w, bias, y, N, H, W, Cin, Cout, wscale, act, st);
    if (Cin == 8)
        return K == 3 ? dispatch_bn<3, 8>(x, w, bias, y, N, H, W, Cin, Cout, wscale, act, st)
                      : dispatch_bn<1, 8>(x, w, bias, y, N, H, W, Cin, Cout, wscale, act, st);
    sq_set_error("sq_conv2d_nhwc_fwd_f32: Cin=%d unsupported (1..8 or a multiple of 16)", Cin);
    return SQ_EINVAL;
}

extern "C" int sq_conv1x1_argmax_fwd_f32(const float *x, const float *w, const float *bias,
                                         float *logits, uint8_t *mask, int N, int H, int W, int Cin,
                                         int Cout, void *stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    SQ_REQUIRE(x && w && logits, "sq_conv1x1_argmax_fwd_f32: null tensor pointer");
    SQ_REQUIRE(N > 0 && H > 0 && W > 0, "sq_conv1x1_argmax_fwd_f32: bad shape");
    SQ_REQUIRE(Cin % 4 == 0 && Cin > 0, "sq_conv1x1_argmax_fwd_f32: Cin=%d must be a multiple of 4", Cin);
    SQ_REQUIRE_ALIGNED(x);
    return launch_small(x, w, bias, logits, mask, (int64_t)N * H * W, Cin, Cout, 1.0f, SQ_ACT_NONE, st,
                        "sq_conv1x1_argmax_fwd_f32");
}
