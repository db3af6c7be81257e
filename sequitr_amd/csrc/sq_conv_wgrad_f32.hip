// Weight gradient of the KxK SAME convolution (conv_layer / weighted_conv2d backward) on
// v_mfma_f32_16x16x4_f32, gfx950.
//
//   dW[tap][ci][co] = sum over pixels p of  X[p + tap][ci] * dY[p][co]
//   db[co]          = sum over pixels p of  dY[p][co]
//
// GEMM view per (16-channel ci chunk, BN-channel co chunk) pair: D[i = ci][j = co], reduction
// over PIXELS (4 per MFMA: A[i][k] = X[pixel k][ci i], B[k][j] = dY[pixel k][co j]), one
// accumulator block per tap.  Blocks are persistent over 16x16 pixel tiles and keep their
// K*K*NR accumulator blocks in registers across all their tiles; the next tile's X halo and
// dY tile are prefetched into registers during the MFMA phase (as in sq_conv_f32_v2.hip).
// The reduction over the 4.2M pixels of a level-0 batch is split across blocks and finished
// by a second kernel that adds the block partials IN A FIXED ORDER: no float atomics, results
// are run-to-run reproducible (MI355X_MICROARCH.md "Global float atomics").
#include "sq_common.h"
#include <stdlib.h>

namespace {

constexpr int TH = 16, TW = 16;

template <int BN, int KS, int KC>
struct WCfg {
    static constexpr int HALO_W = TW + KS - 1;
    static constexpr int HP = HALO_W * (TH + KS - 1);
    static constexpr int PSX = KC;                               // kk*KC + ci: conflict-free for KC = 16
    static constexpr int PSY = (BN % 32 == 0) ? BN + 16 : BN;    // kk*PSY + co: conflict-free
    static constexpr int XS_FLOATS = HP * PSX;
    static constexpr int YS_FLOATS = TH * TW * PSY;
    static constexpr int NTAP = KS * KS;
    static constexpr int NR = BN / 16;
    static constexpr int ROWS = NTAP * 16 + 1;                   // +1: the bias-gradient row
    static constexpr int RED_FLOATS = ROWS * BN;                 // cross-wave reduction image
    static constexpr int LDS_FLOATS = (XS_FLOATS + YS_FLOATS) > RED_FLOATS ? (XS_FLOATS + YS_FLOATS) : RED_FLOATS;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
    static constexpr int QPP = KC / 4;
    static constexpr int XITEMS = HP * QPP;
    static constexpr int XSLOTS = (XITEMS + 255) / 256;
    static constexpr int YITEMS = TH * TW * (BN / 4);
    static constexpr int YSLOTS = YITEMS / 256;
    static_assert(YITEMS % 256 == 0, "dY tile must divide evenly over the block");
    static_assert((XS_FLOATS * 4) % 16 == 0, "dY image must start 16-B aligned");
};

// partials layout: [gridDim.x][npairs][ROWS][BN]
template <int BN, int KS, int KC>
__global__ __launch_bounds__(256, 2) void conv_wgrad_f32_kernel(
    const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ partials,
    int N, int H, int W, int Cin, int Cout, int tiles_x, int tiles_y, int ntiles, int tiles_per_block) {
    using C = WCfg<BN, KS, KC>;
    constexpr int NR = C::NR, PAD = KS / 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *xs = smem;
    float *ys = smem + C::XS_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, kk = lane >> 4;
    const int nco = (Cout + BN - 1) / BN;
    const int ci0 = (blockIdx.y / nco) * KC, co0 = (blockIdx.y % nco) * BN;
    const int t_begin = blockIdx.x * tiles_per_block;
    const int t_end = min(t_begin + tiles_per_block, ntiles);

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(x), 0, (int)((size_t)N * H * W * Cin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(dy), 0, (int)((size_t)N * H * W * Cout * 4), 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    // the per-slot index decode is redone per tile (a few integer ops against 288+ MFMAs) instead of living in
    // ~40 registers next to the accumulators: the <32,3,16> instance spilled 59 VGPRs with the tables
    float4 xr[C::XSLOTS], yr[C::YSLOTS];
    auto issue = [&](int tile) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
        const int x0 = tx * TW, y0 = ty * TH;
        const int xbase = (((n * H + y0 - PAD) * W + x0 - PAD) * Cin) * 4;
        const int ybase = (((n * H + y0) * W + x0) * Cout) * 4;
#pragma unroll
        for (int sl = 0; sl < C::XSLOTS; ++sl) {
            const int idx = tid + sl * 256;
            const int pix = idx / C::QPP, q = idx % C::QPP;
            const int py = pix / C::HALO_W, px = pix % C::HALO_W;
            const bool inb = idx < C::XITEMS && (unsigned)(y0 - PAD + py) < (unsigned)H &&
                             (unsigned)(x0 - PAD + px) < (unsigned)W;
            const unsigned off = inb ? (unsigned)(xbase + ((py * W + px) * Cin + ci0 + q * 4) * 4) : OOB;
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, off, 0, 0);
            xr[sl] = *reinterpret_cast<const float4 *>(&v);
        }
#pragma unroll
        for (int sl = 0; sl < C::YSLOTS; ++sl) {
            const int idx = tid + sl * 256;
            const int pix = idx / (BN / 4), q = idx % (BN / 4);
            const int py = pix / TW, px = pix % TW;
            const bool inb = (y0 + py) < H && (x0 + px) < W && co0 + q * 4 < Cout;
            const unsigned off = inb ? (unsigned)(ybase + ((py * W + px) * Cout + co0 + q * 4) * 4) : OOB;
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(yrsrc, off, 0, 0);
            yr[sl] = *reinterpret_cast<const float4 *>(&v);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int sl = 0; sl < C::XSLOTS; ++sl) {
            const int idx = tid + sl * 256;
            if (idx < C::XITEMS)
                *reinterpret_cast<float4 *>(xs + (idx / C::QPP) * C::PSX + (idx % C::QPP) * 4) = xr[sl];
        }
#pragma unroll
        for (int sl = 0; sl < C::YSLOTS; ++sl) {
            const int idx = tid + sl * 256;
            *reinterpret_cast<float4 *>(ys + (idx / (BN / 4)) * C::PSY + (idx % (BN / 4)) * 4) = yr[sl];
        }
    };

    f32x4 acc[C::NTAP][NR];
    float bsum[NR];
#pragma unroll
    for (int t = 0; t < C::NTAP; ++t)
#pragma unroll
        for (int nb = 0; nb < NR; ++nb) acc[t][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nb = 0; nb < NR; ++nb) bsum[nb] = 0.f;

    // A: X[pixel kk of the group][ci li]; B: dY[pixel kk][co li]
    const float *xa_lds = xs + ((4 * wv) * C::HALO_W + kk) * C::PSX + li;
    const float *yb_lds = ys + ((4 * wv) * TW + kk) * C::PSY + li;

    auto load_frag = [&](int ks, float (&a)[C::NTAP], float (&b)[NR]) {
        const int r = ks >> 2, g = ks & 3;                     // tile row within the wave, 4-pixel group
#pragma unroll
        for (int t = 0; t < C::NTAP; ++t)
            a[t] = xa_lds[((r + t / KS) * C::HALO_W + 4 * g + t % KS) * C::PSX];
#pragma unroll
        for (int nb = 0; nb < NR; ++nb) b[nb] = yb_lds[(r * TW + 4 * g) * C::PSY + nb * 16];
    };

    if (t_begin < t_end) {
        issue(t_begin);
        commit();
    }
    __syncthreads();
    for (int tile = t_begin; tile < t_end; ++tile) {
        const bool has_next = tile + 1 < t_end;
        if (has_next) issue(tile + 1);
        {
            float a0[C::NTAP], b0[NR], a1[C::NTAP], b1[NR];
            load_frag(0, a0, b0);
#pragma unroll
            for (int ks = 0; ks < 16; ks += 2) {
                load_frag(ks + 1, a1, b1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nb = 0; nb < NR; ++nb) {
                    bsum[nb] += b0[nb];
#pragma unroll
                    for (int t = 0; t < C::NTAP; ++t)
                        acc[t][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[t], b0[nb], acc[t][nb], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (ks + 2 < 16) load_frag(ks + 2, a0, b0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nb = 0; nb < NR; ++nb) {
                    bsum[nb] += b1[nb];
#pragma unroll
                    for (int t = 0; t < C::NTAP; ++t)
                        acc[t][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[t], b1[nb], acc[t][nb], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        if (has_next) {
            commit();
            __syncthreads();
        }
    }

    // ---- cross-wave reduction in a fixed order (wave 0, 1, 2, 3), then one partial per block ----
    // D layout: lane holds rows (ci) 4*kk+{0..3}, column (co) li of every [tap][nb] block.
    float *red = smem;
#pragma unroll
    for (int nb = 0; nb < NR; ++nb) {       // fold the 4 pixel slots (kk) of the bias sums
        bsum[nb] += __shfl_xor(bsum[nb], 16);
        bsum[nb] += __shfl_xor(bsum[nb], 32);
    }
    for (int w = 0; w < 4; ++w) {
        if (wv == w) {
#pragma unroll
            for (int t = 0; t < C::NTAP; ++t)
#pragma unroll
                for (int nb = 0; nb < NR; ++nb)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float *d = red + (t * 16 + 4 * kk + j) * BN + nb * 16 + li;
                        *d = (w == 0) ? acc[t][nb][j] : *d + acc[t][nb][j];
                    }
            if (kk == 0) {
#pragma unroll
                for (int nb = 0; nb < NR; ++nb) {
                    float *d = red + (C::NTAP * 16) * BN + nb * 16 + li;
                    *d = (w == 0) ? bsum[nb] : *d + bsum[nb];
                }
            }
        }
        __syncthreads();
    }
    float *out = partials + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * C::RED_FLOATS;
    for (int i = tid; i < C::RED_FLOATS; i += 256) out[i] = red[i];
}

// factor the finish kernel applies to dW (not db): set by sq_conv2d_nhwc_wgrad_scaled_f32 around its dispatch
thread_local float t_dw_scale_f32 = 1.0f;

// second stage: dW[tap][ci][co] = sum_b partials[b][pair][tap*16 + ci%16][co%BN], b ascending
template <int BN, int KS, int KC>
__global__ __launch_bounds__(256) void conv_wgrad_finish_kernel(const float *__restrict__ partials,
                                                                 float *__restrict__ dw, float *__restrict__ db,
                                                                 int nblk, int Cin, int Cout, int G, float dw_scale) {
    using C = WCfg<BN, KS, KC>;
    const int nco = (Cout + BN - 1) / BN, npairs = (Cin / KC) * nco;
    const int total = C::NTAP * Cin * Cout;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int i = t / G, g = t % G;
    const size_t stride = (size_t)npairs * C::RED_FLOATS;
    if (i < total) {
        const int co = i % Cout, ci = (i / Cout) % Cin, tap = i / (Cout * Cin);
        const int pair = (ci / KC) * nco + co / BN;
        const size_t off = (size_t)pair * C::RED_FLOATS + (tap * 16 + ci % KC) * BN + co % BN;
        const float s = sq_group_reduce(partials + off, stride, nblk, g, G);
        if (g == 0) dw[i] = dw_scale == 1.0f ? s : s * dw_scale;
    } else if (i < total + Cout) {
        const int co = i - total;
        const size_t off = (size_t)(co / BN) * C::RED_FLOATS + (C::NTAP * 16) * BN + co % BN;
        const float s = sq_group_reduce(partials + off, stride, nblk, g, G);
        if (g == 0 && db) db[co] = s;
    }
}

template <int BN, int KS, int KC>
int64_t ws_floats(int N, int H, int W, int Cin, int Cout, int *gx_out, int *tpb_out) {
    using C = WCfg<BN, KS, KC>;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int ntiles = tiles_x * tiles_y * N;
    const int npairs = (Cin / KC) * ((Cout + BN - 1) / BN);
    int want = (512 + npairs - 1) / npairs;                    // ~2 resident blocks per CU overall
    if (want < 1) want = 1;
    int tpb = (ntiles + want - 1) / want;
    if (tpb < 1) tpb = 1;
    const int gx = (ntiles + tpb - 1) / tpb;
    if (gx_out) *gx_out = gx;
    if (tpb_out) *tpb_out = tpb;
    return (int64_t)gx * npairs * C::RED_FLOATS;
}

template <int BN, int KS, int KC>
int launch(const float *x, const float *dy, float *dw, float *db, float *ws, int N, int H, int W, int Cin,
           int Cout, hipStream_t st) {
    using C = WCfg<BN, KS, KC>;
    static bool attr_set = false;
    auto kern = conv_wgrad_f32_kernel<BN, KS, KC>;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                C::LDS_BYTES) != hipSuccess) {
            sq_set_error("conv_wgrad_f32: cannot reserve %d bytes of LDS", C::LDS_BYTES);
            return SQ_ELAUNCH;
        }
        attr_set = true;
    }
    int gx, tpb;
    ws_floats<BN, KS, KC>(N, H, W, Cin, Cout, &gx, &tpb);
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int npairs = (Cin / KC) * ((Cout + BN - 1) / BN);
    hipLaunchKernelGGL(kern, dim3(gx, npairs), dim3(256), C::LDS_BYTES, st, x, dy, ws, N, H, W, Cin, Cout,
                       tiles_x, tiles_y, tiles_x * tiles_y * N, tpb);
    int rc = sq_check_launch("sq_conv2d_nhwc_wgrad_f32");
    if (rc) return rc;
    const int G = sq_group_size(gx);
    const int64_t total = ((int64_t)KS * KS * Cin * Cout + Cout) * G;
    hipLaunchKernelGGL((conv_wgrad_finish_kernel<BN, KS, KC>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, ws,
                       dw, db, gx, Cin, Cout, G, t_dw_scale_f32);
    return sq_check_launch("sq_conv2d_nhwc_wgrad_f32(finish)");
}

#define SQ_WGRAD_DISPATCH(FN, ...)                                                        \
    do {                                                                                  \
        if (Cin % 16 == 0) {                                                              \
            if (K == 3) { if (Cout > 16) return FN<32, 3, 16>(__VA_ARGS__); return FN<16, 3, 16>(__VA_ARGS__); } \
            if (Cout > 16) return FN<32, 1, 16>(__VA_ARGS__);                             \
            return FN<16, 1, 16>(__VA_ARGS__);                                            \
        }                                                                                 \
        if (K == 3) { if (Cout > 16) return FN<32, 3, 8>(__VA_ARGS__); return FN<16, 3, 8>(__VA_ARGS__); } \
        if (Cout > 16) return FN<32, 1, 8>(__VA_ARGS__);                                  \
        return FN<16, 1, 8>(__VA_ARGS__);                                                 \
    } while (0)

int64_t ws_dispatch(int N, int H, int W, int Cin, int Cout, int K) {
    SQ_WGRAD_DISPATCH(ws_floats, N, H, W, Cin, Cout, nullptr, nullptr);
}
int launch_dispatch(const float *x, const float *dy, float *dw, float *db, float *ws, int N, int H, int W,
                    int Cin, int Cout, int K, hipStream_t st) {
    SQ_WGRAD_DISPATCH(launch, x, dy, dw, db, ws, N, H, W, Cin, Cout, st);
}

// ---- first layer (Cin = 1..7, 3x3): the 9 taps ride the 16 MFMA rows, one accumulator per input channel ------
//   A[i = tap][k = pixel] = X[pixel + tap][c] (rows 9..15 zero), B[k][j = co] = dY[pixel][co]
// partials: [gridDim.x][9*CIN + 1][Cout]  (row tap*CIN + c as in dW, then the bias row); blockIdx.y = 16-channel co group.
template <typename TY, int CIN>
__global__ __launch_bounds__(256) void conv_wgrad_cin1_f32_kernel(
    const float *__restrict__ x, const TY *__restrict__ dy, float *__restrict__ partials, int N, int H, int W,
    int Cout, int tiles_x, int tiles_y, int ntiles, int tiles_per_block) {
    constexpr int HW = TW + 2, ROWS = 9 * CIN + 1;
    __shared__ float xs[HW * HW * CIN + 8];
    __shared__ __attribute__((aligned(16))) float ys[TH * TW * 16];
    __shared__ float red[4][ROWS * 16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, kk = lane >> 4;
    const int co0 = blockIdx.y * 16;
    const int t_begin = blockIdx.x * tiles_per_block, t_end = min(t_begin + tiles_per_block, ntiles);
    const int ky = li / 3, kx = li % 3;
    const bool live_row = li < 9;
    f32x4 acc[CIN];
#pragma unroll
    for (int c = 0; c < CIN; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    // register double buffer: the next tile's loads are in flight while this one is multiplied (the kernel used to
    // load, wait, compute, and hid the HBM latency only through resident blocks)
    constexpr int XSL = (HW * HW * CIN + 255) / 256;            // halo items (one float) per thread
    constexpr int YV = sizeof(TY) == 4 ? 4 : 2;                 // 16-byte pieces of dY per thread: 256 px x 16 ch
    float xr[XSL];
    uint4 yr[YV];
    auto fetch = [&](int tile) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
        const int x0 = tx * TW, y0 = ty * TH;
#pragma unroll
        for (int sl = 0; sl < XSL; ++sl) {
            const int idx = tid + sl * 256;
            const int pix = idx / CIN, c = idx % CIN;
            const int gy = y0 - 1 + pix / HW, gx = x0 - 1 + pix % HW;
            xr[sl] = (idx < HW * HW * CIN && gy >= 0 && gy < H && gx >= 0 && gx < W)
                         ? x[(((size_t)n * H + gy) * W + gx) * CIN + c] : 0.f;
        }
#pragma unroll
        for (int v = 0; v < YV; ++v) {
            constexpr int PER = 16 / (16 / (int)sizeof(TY));   // 16-byte pieces per pixel: 4 (f32), 2 (bf16)
            constexpr int EL = 16 / (int)sizeof(TY);           // channels per piece
            const int idx = tid + v * 256, pix = idx / PER, q = idx % PER;
            const int gy = y0 + pix / TW, gx = x0 + pix % TW;
            yr[v] = make_uint4(0, 0, 0, 0);
            if (gy < H && gx < W && co0 + q * EL < Cout)        // Cout % 4 == 0; a partial last group (bf16: Cout % 8 == 4)
                yr[v] = (co0 + q * EL + EL <= Cout)             // is read 8 bytes wide
                            ? *reinterpret_cast<const uint4 *>(dy + (((size_t)n * H + gy) * W + gx) * Cout + co0 + q * EL)
                            : make_uint4(reinterpret_cast<const uint2 *>(dy + (((size_t)n * H + gy) * W + gx) * Cout + co0 + q * EL)->x,
                                         reinterpret_cast<const uint2 *>(dy + (((size_t)n * H + gy) * W + gx) * Cout + co0 + q * EL)->y, 0, 0);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int sl = 0; sl < XSL; ++sl)
            if (tid + sl * 256 < HW * HW * CIN) xs[tid + sl * 256] = xr[sl];
#pragma unroll
        for (int v = 0; v < YV; ++v) {
            const int idx = tid + v * 256;
            if constexpr (sizeof(TY) == 4) {
                *reinterpret_cast<uint4 *>(ys + (idx >> 2) * 16 + (idx & 3) * 4) = yr[v];
            } else {
                typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
                const bf16x8_t h = __builtin_bit_cast(bf16x8_t, yr[v]);
                float *dst = ys + (idx >> 1) * 16 + (idx & 1) * 8;
                *reinterpret_cast<float4 *>(dst) = make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
                *reinterpret_cast<float4 *>(dst + 4) = make_float4((float)h[4], (float)h[5], (float)h[6], (float)h[7]);
            }
        }
    };
    if (t_begin < t_end) fetch(t_begin);
    for (int tile = t_begin; tile < t_end; ++tile) {
        commit();
        __syncthreads();
        if (tile + 1 < t_end) fetch(tile + 1);
#pragma unroll 4
        for (int ks = 0; ks < 16; ++ks) {
            const int r = 4 * wv + (ks >> 2), g = ks & 3;
            const float b = ys[(r * TW + 4 * g + kk) * 16 + li];
            bsum += b;
            const float *xp = xs + ((r + ky) * HW + 4 * g + kk + kx) * CIN;
#pragma unroll
            for (int c = 0; c < CIN; ++c) {
                const float a = live_row ? xp[c] : 0.f;
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    bsum += __shfl_xor(bsum, 16);
    bsum += __shfl_xor(bsum, 32);
    // D: rows (taps) 4*kk + j, column (co) li
#pragma unroll
    for (int c = 0; c < CIN; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (4 * kk + j < 9) red[wv][((4 * kk + j) * CIN + c) * 16 + li] = acc[c][j];
    if (kk == 0) red[wv][9 * CIN * 16 + li] = bsum;
    __syncthreads();
    for (int t = tid; t < ROWS * 16; t += 256) {
        const int row = t / 16, c = t % 16;
        if (co0 + c < Cout)
            partials[((size_t)blockIdx.x * ROWS + row) * Cout + co0 + c] =
                ((red[0][t] + red[1][t]) + red[2][t]) + red[3][t];
    }
}

// rows = 9*Cin + 1
__global__ __launch_bounds__(256) void conv_wgrad_cin1_finish_kernel(const float *__restrict__ partials,
                                                                      float *__restrict__ dw, float *__restrict__ db,
                                                                      int nblk, int Cout, int G, int rows) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int i = t / G, g = t % G;
    if (i >= rows * Cout) return;
    const float s = sq_group_reduce(partials + i, (size_t)rows * Cout, nblk, g, G);
    if (g != 0) return;
    if (i < (rows - 1) * Cout) dw[i] = s;
    else if (db) db[i - (rows - 1) * Cout] = s;
}

int cin1_grid(int N, int H, int W, int *tpb_out) {
    const int ntiles = ((W + TW - 1) / TW) * ((H + TH - 1) / TH) * N;
    // the kernel is single-buffered (load, barrier, 16 MFMAs, barrier): it hides latency only through resident blocks,
    // so the grid fills the block slots of every CU (2048 blocks: -0.5 % on the bf16 step vs 1024, within noise of 4096)
    static const int target = [] { const char *e = getenv("SQ_CIN1_BLOCKS"); return e ? atoi(e) : 2048; }();
    int tpb = (ntiles + target - 1) / target;
    if (tpb < 1) tpb = 1;
    if (tpb_out) *tpb_out = tpb;
    return (ntiles + tpb - 1) / tpb;
}

bool shape_ok(int N, int H, int W, int Cin, int Cout, int K) {
    if (Cin >= 1 && Cin <= 7 && K == 3 && N > 0 && H > 0 && W > 0 && Cout > 0 && Cout % 4 == 0) return true;
    return N > 0 && H > 0 && W > 0 && (K == 1 || K == 3) && (Cin % 16 == 0 || Cin == 8) && Cin > 0 &&
           Cout > 0 && Cout % 4 == 0 && (size_t)N * H * W * (size_t)(Cin > Cout ? Cin : Cout) * 4 < ((size_t)1 << 31);
}

template <typename TY>
int launch_cin_small(const float *x, const TY *dy, float *dw, float *db, float *workspace, int N, int H, int W, int Cin,
                     int Cout, hipStream_t st, const char *who) {
    int tpb;
    const int gx = cin1_grid(N, H, W, &tpb);
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const dim3 grid(gx, (Cout + 15) / 16);
#define SQ_CIN_SMALL(C)                                                                                              \
    case C:                                                                                                          \
        hipLaunchKernelGGL((conv_wgrad_cin1_f32_kernel<TY, C>), grid, dim3(256), 0, st, x, dy, workspace, N, H, W,   \
                           Cout, tiles_x, tiles_y, tiles_x * tiles_y * N, tpb);                                      \
        break;
    switch (Cin) {
        SQ_CIN_SMALL(1) SQ_CIN_SMALL(2) SQ_CIN_SMALL(3) SQ_CIN_SMALL(4) SQ_CIN_SMALL(5) SQ_CIN_SMALL(6) SQ_CIN_SMALL(7)
    default: sq_set_error("%s: Cin=%d unsupported (1..7)", who, Cin); return SQ_EINVAL;
    }
#undef SQ_CIN_SMALL
    int rc = sq_check_launch(who);
    if (rc) return rc;
    const int G = sq_group_size(gx), rows = 9 * Cin + 1;
    hipLaunchKernelGGL(conv_wgrad_cin1_finish_kernel, dim3((rows * Cout * G + 255) / 256), dim3(256), 0, st, workspace, dw,
                       db, gx, Cout, G, rows);
    return sq_check_launch(who);
}

}  // namespace

extern "C" int64_t sq_conv2d_nhwc_wgrad_workspace_f32(int N, int H, int W, int Cin, int Cout, int K) {
    if (!shape_ok(N, H, W, Cin, Cout, K)) return -1;
    if (Cin <= 7) return (int64_t)cin1_grid(N, H, W, nullptr) * (9 * Cin + 1) * Cout * 4;
    return ws_dispatch(N, H, W, Cin, Cout, K) * 4;
}

extern "C" int sq_conv2d_nhwc_wgrad_f32(const float *x, const float *dy, float *dw, float *db,
                                        float *workspace, int N, int H, int W, int Cin, int Cout, int K,
                                        void *stream) {
    SQ_REQUIRE(x && dy && dw && workspace, "sq_conv2d_nhwc_wgrad_f32: null pointer");
    SQ_REQUIRE(shape_ok(N, H, W, Cin, Cout, K),
               "sq_conv2d_nhwc_wgrad_f32: unsupported shape N=%d H=%d W=%d Cin=%d Cout=%d K=%d "
               "(Cin 1..7 with K 3, 8 or %%16; Cout %%4; K 1|3; tensors < 2 GiB)", N, H, W, Cin, Cout, K);
    SQ_REQUIRE_ALIGNED(x); SQ_REQUIRE_ALIGNED(dy); SQ_REQUIRE_ALIGNED(workspace);
    if (Cin <= 7)
        return launch_cin_small<float>(x, dy, dw, db, workspace, N, H, W, Cin, Cout, reinterpret_cast<hipStream_t>(stream),
                                       "sq_conv2d_nhwc_wgrad_f32(small Cin)");
    return launch_dispatch(x, dy, dw, db, workspace, N, H, W, Cin, Cout, K, reinterpret_cast<hipStream_t>(stream));
}

// dW multiplied by dw_scale in the finish kernel (MFMA kernels: Cin 8 or a multiple of 16; the small-Cin kernel has no
// scaled form): see sq_conv2d_nhwc_wgrad_scaled_mixed_f32
extern "C" int sq_conv2d_nhwc_wgrad_scaled_f32(const float *x, const float *dy, float *dw, float *db, float *workspace,
                                               int N, int H, int W, int Cin, int Cout, int K, float dw_scale, void *stream) {
    SQ_REQUIRE(Cin > 7 || dw_scale == 1.0f, "sq_conv2d_nhwc_wgrad_scaled_f32: Cin=%d has no scaled form", Cin);
    t_dw_scale_f32 = dw_scale;
    const int rc = sq_conv2d_nhwc_wgrad_f32(x, dy, dw, db, workspace, N, H, W, Cin, Cout, K, stream);
    t_dw_scale_f32 = 1.0f;
    return rc;
}

// first-layer weight gradient with a bf16 dY (the bf16 training graph): same MFMA-over-taps kernel
extern "C" int64_t sq_conv3x3_first_wgrad_workspace_bf16(int N, int H, int W, int Cin, int Cout) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin < 1 || Cin > 7 || Cout <= 0 || Cout % 4) return -1;
    return (int64_t)cin1_grid(N, H, W, nullptr) * (9 * Cin + 1) * Cout * 4;
}

extern "C" int sq_conv3x3_first_wgrad_bf16(const float *x, const void *dy, float *dw, float *db, float *workspace,
                                           int N, int H, int W, int Cin, int Cout, void *stream) {
    SQ_REQUIRE(x && dy && dw && workspace && N > 0 && H > 0 && W > 0 && Cin >= 1 && Cin <= 7 && Cout > 0 && Cout % 4 == 0,
               "sq_conv3x3_first_wgrad_bf16: bad arguments (Cin 1..7, Cout %% 4 == 0)");
    return launch_cin_small<__bf16>(x, reinterpret_cast<const __bf16 *>(dy), dw, db, workspace, N, H, W, Cin, Cout,
                                    reinterpret_cast<hipStream_t>(stream), "sq_conv3x3_first_wgrad_bf16");
}
