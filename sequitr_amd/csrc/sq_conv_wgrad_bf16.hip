// Weight gradient of the KxK SAME convolution on bf16 activations, gfx950:
//   dW[tap][ci][co] = sum_p X[p + tap][ci] * dY[p][co],  db[co] = sum_p dY[p][co]   (fp32 results)
// on v_mfma_f32_16x16x32_bf16 with the reduction over PIXELS (32 per instruction).  Both operands are
// needed "channel-major over pixels" while NHWC tiles in LDS are pixel-major: ds_read_b64_tr_b16 (the
// gfx950 transposing LDS read) delivers, per 16-lane group, a 4-pixel x 16-channel block with lane i
// receiving channel i of the 4 pixels -- exactly the MFMA fragment, no shuffles, no transposed copy.
// Semantics were confirmed on hardware with tools/probe_tr16.hip.
//
// Block = (16 NI input channels, 16 NO output channels) x a run of 16x16 pixel tiles (persistent,
// accumulators stay in registers across the run); wave w owns tile rows 4w..4w+3 = two 32-pixel k-steps.
// X is re-read once per output-channel block and dY once per input-channel block, so NI = NO = 2 where the
// layer has >= 32 channels halves the traffic this kernel is bound by.
// Pixel <-> k mapping of one k-step (rows r, r+1 of the tile), lane group g, element e:
//     row = r + (g >> 1),  x = 4*(g & 1) + 8*(e >> 2) + (e & 3)
// chosen so the two groups of a 32-lane half read blocks 128 B apart (conflict-free).
// Partials + fixed-order finish kernel as in sq_conv_wgrad_f32.hip (no float atomics).
// TIO = float ("mixed"): X and dY are f32 tensors, rounded to bf16 (RNE) while they are staged into LDS --
// the weight gradient of sq_conv2d_nhwc_fwd_mixed_f32 (f32 graph, bf16 multiply, f32 accumulate).
#include "sq_common.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#ifndef SQ_WGRAD_OCC1_ABOVE
#define SQ_WGRAD_OCC1_ABOVE 18      // accumulator blocks above which a shape is built for one block per CU
#endif
#ifndef SQ_WG_ABLATE
#define SQ_WG_ABLATE 0              // experiments only: 1 = no MFMA (reads kept), 2 = no LDS reads (MFMA kept), 3 = no commit writes
#endif
#ifndef SQ_WGRAD_BUDGET2
#define SQ_WGRAD_BUDGET2 200        // register budget (2 blocks per CU) the prefetch depth is sized for
#endif

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int TH = 16, TW = 16, PSB = 32;          // 16 bf16 channels per pixel row in LDS

template <int KS, int NI, int NO>
struct WB {
    static constexpr int HALO_W = TW + KS - 1;
    static constexpr int HP = HALO_W * (TH + KS - 1);
    static constexpr int NTAP = KS * KS;
    static constexpr int CI = 16 * NI, CO = 16 * NO;          // channels of X / dY one block contracts
    // one 16-channel plane of the tile in LDS; the X planes are 64 B out of phase so that the 16-byte commit writes of
    // one pixel's two planes (consecutive lanes) land in different banks
    static constexpr int XPLANE = HP * PSB + (NI > 1 ? 64 : 0), YPLANE = TH * TW * PSB;
    static constexpr int XS_BYTES = XPLANE * NI;
    static constexpr int YS_BYTES = YPLANE * NO;
    static constexpr int ROWS = NTAP * CI + 1;
    static constexpr int RED_FLOATS = ROWS * CO;
    static constexpr int LDS_BYTES = (XS_BYTES + YS_BYTES) > RED_FLOATS * 4 ? (XS_BYTES + YS_BYTES) : RED_FLOATS * 4;
    static constexpr int XITEMS = HP * 2 * NI, XSLOTS = (XITEMS + 255) / 256;
    static constexpr int YITEMS = TH * TW * 2 * NO, YSLOTS = YITEMS / 256;
};

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char *p0, const unsigned char *p1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p1);
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// NI x NO 16-channel planes per block: every byte of X is read Cout / (16 NO) times and every byte of dY
// Cin / (16 NI) times over the whole launch, so wider blocks cut the (L2 / HBM) traffic that bounds this
// kernel; the accumulators (NTAP x NI x NO MFMA blocks) are what limits NI, NO.
// blocks per CU the register budget is sized for: the 36-accumulator-block shapes take the whole file
#ifndef SQ_WGRAD_OCC3_UPTO
#define SQ_WGRAD_OCC3_UPTO 18       // accumulator blocks up to which a shape is built for THREE blocks per CU (168 VGPRs, a shorter prefetch ring)
#endif
#ifndef SQ_WGRAD_OCC4_16x16
#define SQ_WGRAD_OCC4_16x16 0       // ... and the 3x3 16 x 16-channel shape for FOUR (128 VGPRs; the 1x1 32 x 64 shape spills there)
#endif
template <int KS, int NI, int NO>
constexpr int wgrad_occ() {
    constexpr int blocks = KS * KS * NI * NO;
    return blocks > SQ_WGRAD_OCC1_ABOVE ? 1 : ((SQ_WGRAD_OCC4_16x16 && KS == 3 && NI * NO == 1) ? 4 : (blocks <= SQ_WGRAD_OCC3_UPTO ? 3 : 2));
}

__device__ __forceinline__ uint4 f32x8_to_bf16x8(const uint4 &a, const uint4 &b) {
    const float4 lo = __builtin_bit_cast(float4, a), hi = __builtin_bit_cast(float4, b);
    bf16x8 h;
    h[0] = (__bf16)lo.x; h[1] = (__bf16)lo.y; h[2] = (__bf16)lo.z; h[3] = (__bf16)lo.w;
    h[4] = (__bf16)hi.x; h[5] = (__bf16)hi.y; h[6] = (__bf16)hi.z; h[7] = (__bf16)hi.w;
    return __builtin_bit_cast(uint4, h);
}

// MOS: the image the kernel tiles is a mosaic of mos.n small images (sq_conv_bf16.hip, SqDropEpi::mos_*; the layout of
// sq_mosaic_pack_f32) that is never built: X and dY loads address the compact (n, h, w, C) tensors, separator pixels read 0
struct SqMos {
    int h = 0, w = 0, cc = 0, n = 0;
    unsigned mh = 0, mw = 0;                                    // ceil(2^16 / (h+1)), ceil(2^16 / (w+1))
};

// RAG: Cin and / or Cout is 8 mod 16 (the GAN's 256 x 256 level): the last 16-channel plane of that operand is half
// empty -- its upper 8 channels load as zeros and the finish kernel drops their rows / columns
// The kernel body, shared by the one-layer kernel and the grouped kernel below: (bx, by) = this block's position in the
// layer's own (tile range, channel-block pair) grid of gy pairs.
template <int KS, int NI, int NO, int PF, typename TIO, bool MOS, bool RAG>
__device__ __forceinline__ void conv_wgrad_bf16_body(
    const TIO *__restrict__ x, const TIO *__restrict__ dy, float *__restrict__ partials, int N, int H,
    int W, int Cin, int Cout, int tiles_x, int tiles_y, int ntiles, int tiles_per_block, const SqMos &mos, int bx, int by, int gy, int gxl) {
    using C = WB<KS, NI, NO>;
    constexpr int PAD = KS / 2;
    constexpr int ES = (int)sizeof(TIO), XV = ES == 4 ? 2 : 1;  // 16-byte loads per 8-channel LDS item
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *xs = smem, *ys = smem + C::XS_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, kg = lane >> 4, q = li >> 2, p = li & 3;
    const int nco = (Cout + C::CO - 1) / C::CO;
    const int ci0 = (by / nco) * C::CI, co0 = (by % nco) * C::CO;
    // tiles_per_block < 0: block bx of the layer's gxl tile-range blocks takes tiles bx, bx + gxl, ... -- the tiles in flight
    // at any moment are a contiguous run of the image (DRAM pages, shared halos), as in conv_mfma_bf16_kernel
    const bool il = tiles_per_block < 0;
    const int ts = il ? gxl : 1;
    const int t_begin = il ? bx : bx * tiles_per_block;
    const int cnt = il ? (ntiles - bx + ts - 1) / ts : min(tiles_per_block, ntiles - t_begin);

    const size_t io_pixels = MOS ? (size_t)mos.n * mos.h * mos.w : (size_t)N * H * W;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<TIO *>(x), 0, (int)(io_pixels * Cin * ES), 0x00020000);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<TIO *>(dy), 0, (int)(io_pixels * Cout * ES), 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    auto mos_pixel = [&](int gy, int gx) {                      // compact pixel of mosaic pixel (gy, gx), -1: separator / outside
        if ((unsigned)gy >= (unsigned)H || (unsigned)gx >= (unsigned)W) return -1;
        const int cc = (int)(((unsigned)gx * mos.mw) >> 16), xx = gx - cc * (mos.w + 1);
        const int rr = (int)(((unsigned)gy * mos.mh) >> 16), yy = gy - rr * (mos.h + 1);
        const int im = rr * mos.cc + cc;
        return (xx < mos.w && yy < mos.h && im < mos.n) ? (im * mos.h + yy) * mos.w + xx : -1;
    };

    // 16-byte items: item = (pixel, plane, half); a pixel's 32 NI bytes are contiguous in HBM.  The index
    // decode is redone per tile (a handful of integer ops) rather than kept in registers: the accumulators
    // need them.
    // PF register sets: the loads of tile t+PF are issued while tile t is computed, i.e. PF-1 whole
    // iterations before they are committed to LDS -- one iteration is far shorter than an HBM round trip
    uint4 xr[PF][C::XSLOTS][XV], yr[PF][C::YSLOTS][XV];
    // halo geometry of this thread's X items, fixed for the whole run (the divisions by the halo width are not
    // redone per tile): position in the halo (py | px << 16; py = 0x7FFF, never in range, for a slot past the tile) and the byte offset
    // relative to the halo's first pixel
    int xpos[C::XSLOTS], xrel[C::XSLOTS];
#pragma unroll
    for (int sl = 0; sl < C::XSLOTS; ++sl) {
        const int idx = tid + sl * 256, pix = idx / (2 * NI), rem = idx % (2 * NI);
        const int py = pix / C::HALO_W, px = pix % C::HALO_W;
        xpos[sl] = idx < C::XITEMS ? (py | (px << 16)) : 0x7FFF;
        xrel[sl] = ((py * W + px) * Cin + rem * 8) * ES;
    }
    auto issue = [&](int tile, uint4 (&xr)[C::XSLOTS][XV], uint4 (&yr)[C::YSLOTS][XV]) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
        const int x0 = tx * TW, y0 = ty * TH;
        const int xbase = (((n * H + y0 - PAD) * W + x0 - PAD) * Cin + ci0) * ES;
        const int ybase = (((n * H + y0) * W + x0) * Cout + co0) * ES;
#pragma unroll
        for (int sl = 0; sl < C::XSLOTS; ++sl) {
            const int py = xpos[sl] & 0xFFFF, px = xpos[sl] >> 16;
            bool inb;
            unsigned off;
            if constexpr (MOS) {
                const int pc = py != 0x7FFF ? mos_pixel(y0 - PAD + py, x0 - PAD + px) : -1;
                inb = pc >= 0;
                off = (unsigned)((pc * Cin + ci0 + ((tid + sl * 256) % (2 * NI)) * 8) * ES);
            } else {
                inb = (unsigned)(y0 - PAD + py) < (unsigned)H && (unsigned)(x0 - PAD + px) < (unsigned)W;
                if constexpr (RAG) inb = inb && ci0 + ((tid + sl * 256) % (2 * NI)) * 8 < Cin;
                off = inb ? (unsigned)(xbase + xrel[sl]) : OOB;
            }
#pragma unroll
            for (int h = 0; h < XV; ++h) {
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, inb ? off + 16 * h : OOB, 0, 0);
                xr[sl][h] = *reinterpret_cast<const uint4 *>(&v);
            }
        }
#pragma unroll
        for (int sl = 0; sl < C::YSLOTS; ++sl) {
            const int idx = tid + sl * 256, pix = idx / (2 * NO), rem = idx % (2 * NO);
            const int py = pix / TW, px = pix % TW;
            bool inb;
            unsigned off;
            if constexpr (MOS) {
                const int pc = mos_pixel(y0 + py, x0 + px);
                inb = pc >= 0;
                off = (unsigned)((pc * Cout + co0 + rem * 8) * ES);
            } else {
                inb = (y0 + py) < H && (x0 + px) < W;
                if constexpr (RAG) inb = inb && co0 + rem * 8 < Cout;
                off = inb ? (unsigned)(ybase + ((py * W + px) * Cout + rem * 8) * ES) : OOB;
            }
#pragma unroll
            for (int h = 0; h < XV; ++h) {
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(yrsrc, inb ? off + 16 * h : OOB, 0, 0);
                yr[sl][h] = *reinterpret_cast<const uint4 *>(&v);
            }
        }
    };
    auto item = [&](const uint4 (&r)[XV]) {
        if constexpr (XV == 2) return f32x8_to_bf16x8(r[0], r[XV - 1]);
        else return r[0];
    };
    auto commit = [&](const uint4 (&xr)[C::XSLOTS][XV], const uint4 (&yr)[C::YSLOTS][XV]) {
#pragma unroll
        for (int sl = 0; sl < C::XSLOTS; ++sl) {
            const int idx = tid + sl * 256, pix = idx / (2 * NI), rem = idx % (2 * NI);
#if SQ_WG_ABLATE == 3
            if (idx < C::XITEMS && xr[sl][0].x == 0x12345678u)
#else
            if (idx < C::XITEMS)
#endif
                *reinterpret_cast<uint4 *>(xs + (rem >> 1) * C::XPLANE + pix * PSB + (rem & 1) * 16) = item(xr[sl]);
        }
#pragma unroll
        for (int sl = 0; sl < C::YSLOTS; ++sl) {
            const int idx = tid + sl * 256, pix = idx / (2 * NO), rem = idx % (2 * NO);
            *reinterpret_cast<uint4 *>(ys + (rem >> 1) * C::YPLANE + pix * PSB + (rem & 1) * 16) = item(yr[sl]);
        }
    };

    f32x4 acc[C::NTAP][NI][NO];
#pragma unroll
    for (int t = 0; t < C::NTAP; ++t)
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int o = 0; o < NO; ++o) acc[t][i][o] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 bacc[NO];                                             // every row = sum over pixels of dY[.][co = li]
#pragma unroll
    for (int o = 0; o < NO; ++o) bacc[o] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

    // this lane's tr-read address inside a k-step: pixel (row kg>>1, x 4*(kg&1) + q), 8-byte piece p;
    // the second read of a fragment is 8 pixels further along the row
    const int lane_px = (kg >> 1) * TW + 4 * (kg & 1) + q;                 // in the dY tile
    const int lane_hx = (kg >> 1) * C::HALO_W + 4 * (kg & 1) + q;          // in the X halo
    const unsigned char *yb = ys + ((4 * wv) * TW + lane_px) * PSB + p * 8;
    const unsigned char *xa = xs + ((4 * wv) * C::HALO_W + lane_hx) * PSB + p * 8;

    // an out-of-range tile index loads nothing (every lane's offset is out of bounds) and is never committed
    auto issue_if = [&](int k, uint4 (&xr_)[C::XSLOTS][XV], uint4 (&yr_)[C::YSLOTS][XV]) {
        if (k < cnt) issue(t_begin + k * ts, xr_, yr_);
    };
#pragma unroll
    for (int u = 0; u < PF; ++u) issue_if(u, xr[u], yr[u]);
    if (cnt > 0) commit(xr[0], yr[0]);
    __syncthreads();
    for (int base = 0; base < cnt; base += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int k = base + u;
            if (k >= cnt) break;
            issue_if(k + PF, xr[u], yr[u]);                    // set u was committed before this tile's compute
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const unsigned char *yk = yb + (2 * ks) * TW * PSB;
                bf16x8 b[NO];
#pragma unroll
                for (int o = 0; o < NO; ++o) {
#if SQ_WG_ABLATE == 2
                    b[o] = ones;
#else
                    b[o] = tr_frag(yk + o * C::YPLANE, yk + o * C::YPLANE + 8 * PSB);
#endif
                    // db on the matrix pipe: a row of ones times the dY fragment (16 conversions + adds on the VALU before)
                    bacc[o] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, b[o], bacc[o], 0, 0, 0);
                }
#pragma unroll
                for (int t = 0; t < C::NTAP; ++t) {
                    const unsigned char *xk = xa + ((2 * ks + t / KS) * C::HALO_W + t % KS) * PSB;
#pragma unroll
                    for (int i = 0; i < NI; ++i) {
#if SQ_WG_ABLATE == 2
                        const bf16x8 a = ones;
                        (void)xk;
#else
                        const bf16x8 a = tr_frag(xk + i * C::XPLANE, xk + i * C::XPLANE + 8 * PSB);
#endif
#pragma unroll
                        for (int o = 0; o < NO; ++o) {
#if SQ_WG_ABLATE == 1
                            acc[t][i][o][0] += (float)a[0] + (float)a[7] + (float)b[o][3];
#else
                            acc[t][i][o] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[o], acc[t][i][o], 0, 0, 0);
#endif
                        }
                    }
                }
            }
            __syncthreads();
            if (k + 1 < cnt) {
                // (a second LDS buffer with one barrier per tile was measured: slower on the 16-channel and 1x1 shapes,
                // within 3 % on the deep 3x3 ones -- the barriers are not what this loop waits for)
                commit(xr[(u + 1) % PF], yr[(u + 1) % PF]);
                __syncthreads();
            }
        }
    }

    // cross-wave reduction in wave order, then one partial per block: red[(tap*CI + ci)*CO + co], bias row last
    float *red = reinterpret_cast<float *>(smem);
    float bsum[NO];
#pragma unroll
    for (int o = 0; o < NO; ++o) bsum[o] = bacc[o][0];          // row 4 * kg: identical in every row
    for (int w = 0; w < 4; ++w) {
        if (wv == w) {
#pragma unroll
            for (int t = 0; t < C::NTAP; ++t)
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int o = 0; o < NO; ++o)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float *d = red + (t * C::CI + i * 16 + 4 * kg + j) * C::CO + o * 16 + li;
                            *d = (w == 0) ? acc[t][i][o][j] : *d + acc[t][i][o][j];
                        }
            if (kg == 0) {
#pragma unroll
                for (int o = 0; o < NO; ++o) {
                    float *d = red + (C::NTAP * C::CI) * C::CO + o * 16 + li;
                    *d = (w == 0) ? bsum[o] : *d + bsum[o];
                }
            }
        }
        __syncthreads();
    }
    float *out = partials + ((size_t)bx * gy + by) * C::RED_FLOATS;
    for (int i = tid; i < C::RED_FLOATS; i += 256) out[i] = red[i];
}

template <int KS, int NI, int NO, int PF, typename TIO, bool MOS = false, bool RAG = false>
__global__ __launch_bounds__(256, (wgrad_occ<KS, NI, NO>())) void conv_wgrad_bf16_kernel(
    const TIO *__restrict__ x, const TIO *__restrict__ dy, float *__restrict__ partials, int N, int H,
    int W, int Cin, int Cout, int tiles_x, int tiles_y, int ntiles, int tiles_per_block, SqMos mos) {
    conv_wgrad_bf16_body<KS, NI, NO, PF, TIO, MOS, RAG>(x, dy, partials, N, H, W, Cin, Cout, tiles_x, tiles_y, ntiles,
                                                        tiles_per_block, mos, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y,
                                                        (int)gridDim.x);
}

// Several layers' weight gradients in ONE launch: the deep layers of a training step are 13 launches of ~37 us each for
// ~10 us of matrix work -- ramp-up (the first tiles' round trips) and the cross-wave reduction at the end run with the chip
// otherwise idle.  Queued behind each other in one grid, one layer's ramp and tail overlap the next layer's tiles.  Every
// block finds its layer by its index (blocks of layer e: first[e] .. first[e+1]-1, pair-major) and runs the body above with
// that layer's arguments: same partials, same finish order, same bits as the one-layer launches.
constexpr int GROUP_MAX = 16;
struct SqWgradGroup {
    int n, pair_major;
    int first[GROUP_MAX + 1];
    const void *x[GROUP_MAX], *dy[GROUP_MAX];
    float *partials[GROUP_MAX];
    int N[GROUP_MAX], H[GROUP_MAX], W[GROUP_MAX], Cin[GROUP_MAX], Cout[GROUP_MAX], tiles_x[GROUP_MAX], tiles_y[GROUP_MAX],
        tpb[GROUP_MAX], gx[GROUP_MAX];
    SqMos mos[GROUP_MAX];                                       // MOS groups: N, H, W are the mosaic's (1, R (h+1), Cc (w+1))
};
template <int KS, int NI, int NO, int PF, typename TIO, bool MOS = false, bool RAG = false>
__global__ __launch_bounds__(256, (wgrad_occ<KS, NI, NO>())) void conv_wgrad_bf16_group_kernel(const SqWgradGroup g) {
    using C = WB<KS, NI, NO>;
    int e = 0;
    while (e + 1 < g.n && (int)blockIdx.x >= g.first[e + 1]) ++e;
    const int local = (int)blockIdx.x - g.first[e], gx = g.gx[e];
    const int gy = ((g.Cin[e] + C::CI - 1) / C::CI) * ((g.Cout[e] + C::CO - 1) / C::CO);
    // block -> (tile range bx, channel-block pair by).  Workgroups go round-robin over the 8 XCDs; with a multiple of 8 tile ranges
    // the pairs of one range are made 8 ids apart: the same XCD (they share that range's X and dY tiles through its L2) and
    // neighbours in its dispatch order (they start together and walk the range in step)
    int bx, by;
    if (g.pair_major && (gx & 7) == 0) {
        const int per = 8 * gy, r = local % per;
        bx = (local / per) * 8 + (r & 7);
        by = r >> 3;
    } else {
        bx = local % gx;
        by = local / gx;
    }
    conv_wgrad_bf16_body<KS, NI, NO, PF, TIO, MOS, RAG>(
        reinterpret_cast<const TIO *>(g.x[e]), reinterpret_cast<const TIO *>(g.dy[e]), g.partials[e], g.N[e], g.H[e], g.W[e],
        g.Cin[e], g.Cout[e], g.tiles_x[e], g.tiles_y[e], g.tiles_x[e] * g.tiles_y[e] * g.N[e], g.tpb[e], g.mos[e], bx, by, gy, gx);
}

// factor the finish kernel applies to dW (not db): set by the *_scaled_* entry points around their dispatch, 1 otherwise
thread_local float t_dw_scale = 1.0f;
thread_local SqMos t_mos;                                       // set by sq_conv2d_nhwc_wgrad_mixed_mosaic_f32 around its dispatch
// > 0: the launch is the 1x1 wgrad of a 2x2/s2 transpose conv in space-to-depth form (Cout = 4 * t_convT_cout); the
// finish kernel then writes the transpose conv's own parameter layouts (sq_convT2x2s2_wgrad_bf16)
thread_local int t_convT_cout = 0;

// Finish: dW / db = sum over the K-split blocks of their partials, in a fixed order.  Threads walk the PARTIAL layout
// ([pair][row][co], the order the blocks wrote): 256 / G consecutive elements x G lanes-groups per block, group g adds
// blocks g, g + G, ... in order, then a fixed LDS tree folds the groups -- every load is a full run of consecutive
// floats (the earlier version walked dW order with the group along the lanes: 4 useful bytes per 128-byte line).
template <int KS, int NI, int NO>
__device__ __forceinline__ void conv_wgrad_bf16_finish_body(const float *__restrict__ partials, float *__restrict__ dw,
                                                            float *__restrict__ db, int nblk, int Cin, int Cout, int G,
                                                            float dw_scale, int ct, int bx, float (*red)[256], int acc = 0) {
    // acc: bit 0 -- dW is added to what dw holds (a further contribution to a parameter's gradient), bit 1 -- the same for db
    using C = WB<KS, NI, NO>;
    const int nco = (Cout + C::CO - 1) / C::CO, npairs = ((Cin + C::CI - 1) / C::CI) * nco;
    const int total = npairs * C::RED_FLOATS;
    const int OUT = 256 / G;
    const int ol = threadIdx.x % OUT, g = threadIdx.x / OUT;
    const int j = bx * OUT + ol;
    const size_t stride = (size_t)npairs * C::RED_FLOATS;
    auto column = [&](int jj) {                                 // this group's share of element jj, in block order
        float s = 0.f;
        const float *p = partials + jj;
        int b = g;
        for (; b + 3 * G < nblk; b += 4 * G) {                  // four loads in flight
            const float v0 = p[(size_t)b * stride], v1 = p[(size_t)(b + G) * stride], v2 = p[(size_t)(b + 2 * G) * stride],
                        v3 = p[(size_t)(b + 3 * G) * stride];
            s = (((s + v0) + v1) + v2) + v3;
        }
        for (; b < nblk; b += G) s += p[(size_t)b * stride];
        return s;
    };
    const bool live = j < total;
    const int pair = live ? j / C::RED_FLOATS : 0, r = live ? j % C::RED_FLOATS : 0, row = r / C::CO, col = r % C::CO;
    const int co = (pair % nco) * C::CO + col;
    const bool bias_row = row == C::NTAP * C::CI;
    // ragged channel counts (8 mod 16): rows / columns past the tensor are zeros of the padding, not gradients
    const bool inside = co < Cout && (bias_row || (pair / nco) * C::CI + row % C::CI < Cin);
    // transpose conv (ct > 0): db[c] = ((q0 + q1) + q2) + q3 over the sub-pixel columns q * ct + c; the lanes of
    // column c (q = 0) reduce the other three columns as well
    const bool fold = ct > 0 && live && bias_row && pair / nco == 0 && co < ct;
    red[0][threadIdx.x] = live ? column(j) : 0.f;
    if (ct > 0) {
#pragma unroll
        for (int q = 1; q < 4; ++q) {
            const int cq = co + q * ct;
            red[q][threadIdx.x] = fold ? column((cq / C::CO) * C::RED_FLOATS + (C::NTAP * C::CI) * C::CO + cq % C::CO) : 0.f;
        }
    }
    __syncthreads();
    for (int m = G >> 1; m > 0; m >>= 1) {
        if (g < m) {
            red[0][threadIdx.x] += red[0][threadIdx.x + m * OUT];
            if (ct > 0) {
#pragma unroll
                for (int q = 1; q < 4; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + m * OUT];
            }
        }
        __syncthreads();
    }
    if (g != 0 || !live || !inside) return;
    const float s = red[0][threadIdx.x];
    if (!bias_row) {
        const int tap = row / C::CI, ci = (pair / nco) * C::CI + row % C::CI;
        // ct: dW of the transpose conv, (2,2,ct,Cin), from column q * ct + c of its space-to-depth form
        const size_t o = ct ? (size_t)co * Cin + ci : ((size_t)tap * Cin + ci) * Cout + co;
        const float v = dw_scale == 1.0f ? s : s * dw_scale;    // the equalised-LR factor of gan.py:79, one f32 multiply
        dw[o] = (acc & 1) ? dw[o] + v : v;
    } else if (db && pair / nco == 0) {                         // the bias row is taken from the ci-block 0 pairs
        if (!ct) db[co] = (acc & 2) ? db[co] + s : s;
        else if (fold) {
            const float v = ((s + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
            db[co] = (acc & 2) ? db[co] + v : v;
        }
    }
}

template <int KS, int NI, int NO>
__global__ __launch_bounds__(256) void conv_wgrad_bf16_finish_kernel(const float *__restrict__ partials,
                                                                      float *__restrict__ dw, float *__restrict__ db,
                                                                      int nblk, int Cin, int Cout, int G, float dw_scale,
                                                                      int ct) {
    __shared__ float red[4][256];
    conv_wgrad_bf16_finish_body<KS, NI, NO>(partials, dw, db, nblk, Cin, Cout, G, dw_scale, ct, (int)blockIdx.x, red);
}

// the finish passes of a group of layers in one launch (blocks of layer e: first[e] .. first[e+1]-1)
struct SqWgradFinishGroup {
    int n;
    int first[GROUP_MAX + 1];
    const float *partials[GROUP_MAX];
    float *dw[GROUP_MAX], *db[GROUP_MAX];
    int nblk[GROUP_MAX], Cin[GROUP_MAX], Cout[GROUP_MAX], G[GROUP_MAX], ct[GROUP_MAX], acc[GROUP_MAX];
    float dw_scale[GROUP_MAX];
};
template <int KS, int NI, int NO>
__global__ __launch_bounds__(256) void conv_wgrad_bf16_finish_group_kernel(const SqWgradFinishGroup g) {
    __shared__ float red[4][256];
    int e = 0;
    while (e + 1 < g.n && (int)blockIdx.x >= g.first[e + 1]) ++e;
    conv_wgrad_bf16_finish_body<KS, NI, NO>(g.partials[e], g.dw[e], g.db[e], g.nblk[e], g.Cin[e], g.Cout[e], g.G[e], g.dw_scale[e],
                                            g.ct[e], (int)blockIdx.x - g.first[e], red, g.acc[e]);
}

inline bool wgrad_interleave() {                                // SQ_WGRAD_INTERLEAVE=0: contiguous tile runs per block (A/B switch)
    static const bool v = [] { const char *e = getenv("SQ_WGRAD_INTERLEAVE"); return !(e && e[0] == '0'); }();
    return v;
}

template <int KS, int NI, int NO>
void plan(int N, int H, int W, int Cin, int Cout, int *gx, int *tpb, int64_t *ws_floats) {
    using C = WB<KS, NI, NO>;
    const int ntiles = ((W + TW - 1) / TW) * ((H + TH - 1) / TH) * N;
    const int npairs = ((Cin + C::CI - 1) / C::CI) * ((Cout + C::CO - 1) / C::CO);
    // (512: targets that leave a layer with a number of tile ranges that is not a multiple of 8 -- 384, 768 -- put the channel-block
    // pairs of one range on different XCDs, their shared operands are then fetched once per pair: 2.81 -> 3.2 ms per training step)
    int want = (512 + npairs - 1) / npairs;
    if (want < 1) want = 1;
    int t = (ntiles + want - 1) / want;
    if (t < 1) t = 1;
    *tpb = t;
    *gx = (ntiles + t - 1) / t;
    *ws_floats = (int64_t)(*gx) * npairs * C::RED_FLOATS;
}

template <int KS, int NI, int NO>
int finish(float *ws, float *dw, float *db, int gx, int Cin, int Cout, hipStream_t st) {
    using C = WB<KS, NI, NO>;
    const int npairs = ((Cin + C::CI - 1) / C::CI) * ((Cout + C::CO - 1) / C::CO);
    int G = sq_group_size(gx);
    if (G > 16) G = 16;                                         // >= 16 consecutive floats (64 B) per load of a group
    const int64_t total = (int64_t)npairs * C::RED_FLOATS;
    const int OUT = 256 / G;
    hipLaunchKernelGGL((conv_wgrad_bf16_finish_kernel<KS, NI, NO>), dim3((unsigned)((total + OUT - 1) / OUT)), dim3(256), 0, st,
                       ws, dw, db, gx, Cin, Cout, G, t_dw_scale, KS == 1 ? t_convT_cout : 0);
    return sq_check_launch("sq_conv2d_nhwc_wgrad_bf16(finish)");
}

// the mosaic form (f32 tensors, 3x3): same plan, same finish; X / dY are the compact small-image tensors
template <int KS, int NI, int NO, int PF, typename TIO>
int launch_mos(const TIO *x, const TIO *dy, float *dw, float *db, float *ws, int N, int H, int W, int Cin, int Cout,
               hipStream_t st) {
    using C = WB<KS, NI, NO>;
    static bool attr_set = false;
    auto kern = conv_wgrad_bf16_kernel<KS, NI, NO, PF, TIO, true>;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                C::LDS_BYTES) != hipSuccess) {
            sq_set_error("conv_wgrad_bf16: cannot reserve %d bytes of LDS", C::LDS_BYTES);
            return SQ_ELAUNCH;
        }
        attr_set = true;
    }
    int gx, tpb;
    int64_t wsf;
    plan<KS, NI, NO>(N, H, W, Cin, Cout, &gx, &tpb, &wsf);
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int npairs = (Cin / C::CI) * (Cout / C::CO);
    hipLaunchKernelGGL(kern, dim3(gx, npairs), dim3(256), C::LDS_BYTES, st, x, dy, ws, N, H, W, Cin, Cout, tiles_x, tiles_y,
                       tiles_x * tiles_y * N, wgrad_interleave() ? -tpb : tpb, t_mos);
    int rc = sq_check_launch("sq_conv2d_nhwc_wgrad_mixed_mosaic_f32");
    if (rc) return rc;
    return finish<KS, NI, NO>(ws, dw, db, gx, Cin, Cout, st);
}

template <int KS, int NI, int NO, typename TIO>
int launch(const TIO *x, const TIO *dy, float *dw, float *db, float *ws, int N, int H, int W, int Cin,
           int Cout, hipStream_t st) {
    using C = WB<KS, NI, NO>;
    static bool attr_set = false;
    // prefetch depth: as deep as the accumulators leave registers for (f32 tensors: twice the registers per set)
    constexpr int acc_regs = C::NTAP * NI * NO * 4, set_regs = (C::XSLOTS + C::YSLOTS) * 4 * (int)(sizeof(TIO) / 2);
    constexpr int budget = (wgrad_occ<KS, NI, NO>() == 1 ? 300 : (wgrad_occ<KS, NI, NO>() == 4 ? 100 : (wgrad_occ<KS, NI, NO>() == 3 ? 140 : SQ_WGRAD_BUDGET2))) - acc_regs - 40;
    constexpr int PF = budget / set_regs >= 4 ? 4 : (budget / set_regs >= 3 ? 3 : (budget / set_regs >= 2 ? 2 : 1));
    if constexpr (KS == 3) {
        if (t_mos.h) return launch_mos<KS, NI, NO, PF, TIO>(x, dy, dw, db, ws, N, H, W, Cin, Cout, st);
    }
    auto kern = conv_wgrad_bf16_kernel<KS, NI, NO, PF, TIO>;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                C::LDS_BYTES) != hipSuccess) {
            sq_set_error("conv_wgrad_bf16: cannot reserve %d bytes of LDS", C::LDS_BYTES);
            return SQ_ELAUNCH;
        }
        attr_set = true;
    }
    int gx, tpb;
    int64_t wsf;
    plan<KS, NI, NO>(N, H, W, Cin, Cout, &gx, &tpb, &wsf);
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int npairs = (Cin / C::CI) * (Cout / C::CO);
    hipLaunchKernelGGL(kern, dim3(gx, npairs), dim3(256), C::LDS_BYTES, st, x, dy, ws, N, H, W, Cin, Cout, tiles_x,
                       tiles_y, tiles_x * tiles_y * N, wgrad_interleave() ? -tpb : tpb, SqMos{});
    int rc = sq_check_launch("sq_conv2d_nhwc_wgrad_bf16");
    if (rc) return rc;
    return finish<KS, NI, NO>(ws, dw, db, gx, Cin, Cout, st);
}


// ragged channel counts (Cin or Cout = 8 mod 16): 16 x 16 channel blocks, the half-empty plane masked in the loader
template <int KS, typename TIO>
int launch_rag(const TIO *x, const TIO *dy, float *dw, float *db, float *ws, int N, int H, int W, int Cin, int Cout,
               hipStream_t st) {
    using C = WB<KS, 1, 1>;
    static bool attr_set = false;
    constexpr int PF = sizeof(TIO) == 4 ? 2 : 4;
    auto kern = conv_wgrad_bf16_kernel<KS, 1, 1, PF, TIO, false, true>;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                C::LDS_BYTES) != hipSuccess) {
            sq_set_error("conv_wgrad_bf16: cannot reserve %d bytes of LDS", C::LDS_BYTES);
            return SQ_ELAUNCH;
        }
        attr_set = true;
    }
    int gx, tpb;
    int64_t wsf;
    plan<KS, 1, 1>(N, H, W, Cin, Cout, &gx, &tpb, &wsf);
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int npairs = ((Cin + 15) / 16) * ((Cout + 15) / 16);
    hipLaunchKernelGGL(kern, dim3(gx, npairs), dim3(256), C::LDS_BYTES, st, x, dy, ws, N, H, W, Cin, Cout, tiles_x, tiles_y,
                       tiles_x * tiles_y * N, wgrad_interleave() ? -tpb : tpb, SqMos{});
    int rc = sq_check_launch("sq_conv2d_nhwc_wgrad_bf16(ragged)");
    if (rc) return rc;
    return finish<KS, 1, 1>(ws, dw, db, gx, Cin, Cout, st);
}

// SQ_WGRAD_BF16_NARROW=1: 16 x 16 channel blocks everywhere (A/B switch for the wider blocks);
// SQ_WGRAD_BF16_MAX="ni,no": upper bound on the block shape (tuning experiments)
inline bool narrow_blocks() {
    static const bool v = [] { const char *e = getenv("SQ_WGRAD_BF16_NARROW"); return e && e[0] == '1'; }();
    return v;
}
inline void max_shape(int *ni, int *no) {
    static int mi = 0, mo = 0;
    if (!mi) {
        mi = 2, mo = 4;
        const char *e = getenv("SQ_WGRAD_BF16_MAX");
        if (e && e[0] >= '1' && e[0] <= '2' && e[1] == ',' && (e[2] == '1' || e[2] == '2' || e[2] == '4')) {
            mi = e[0] - '0';
            mo = e[2] - '0';
        }
    }
    *ni = mi;
    *no = mo;
}

// SQ_WGRAD_BF16_K3="ni,no": force the 3x3 block shape where the channel counts allow it (tuning experiments)
inline int k3_shape() {
    static const int v = [] { const char *e = getenv("SQ_WGRAD_BF16_K3"); return (e && e[0] && e[1] == ',' && e[2]) ? (e[0] - '0') * 10 + (e[2] - '0') : 0; }();
    return v;
}

// channel-block shape per layer, from the measured sweep (tools/wgrad_bf16_bench.py under rocprofv3): every
// launch is bound by the re-read traffic X * Cout/(16 NO) + dY * Cin/(16 NI) at ~4-5 TB/s; 3x3: 16 x 32 channels where Cout
// allows (with the layers sharing grouped launches: -1.0 % on the training step and -1.2 % on the GAN iteration against
// 32 x 16, three same-box repeats each; alone the two measured the same), else 32 x 16
// (32 x 32 needs 36 accumulator blocks = the whole register file at one block per CU, and is slower);
// 1x1 (transpose-conv backward): 32 x 64
#define SQ_WGRAD_BF16_DISPATCH(FN, ...)                                                              \
    do {                                                                                             \
        const bool wide = !narrow_blocks();                                                          \
        int mi_, mo_;                                                                                \
        max_shape(&mi_, &mo_);                                                                       \
        const bool i2 = wide && mi_ >= 2 && Cin % 32 == 0, o2 = wide && mo_ >= 2 && Cout % 32 == 0,  \
                   o4 = wide && mo_ >= 4 && Cout % 64 == 0;                                          \
        if (K == 3) {                                                                                \
            const int k3_ = k3_shape();                                                              \
            if (k3_ == 14 && Cout % 64 == 0) return FN<3, 1, 4>(__VA_ARGS__);                        \
            if (k3_ == 22 && Cin % 32 == 0 && Cout % 32 == 0) return FN<3, 2, 2>(__VA_ARGS__);       \
            if (k3_ == 12 && Cout % 32 == 0) return FN<3, 1, 2>(__VA_ARGS__);                        \
            if (o2) return FN<3, 1, 2>(__VA_ARGS__);                                                 \
            if (i2) return FN<3, 2, 1>(__VA_ARGS__);                                                 \
            return FN<3, 1, 1>(__VA_ARGS__);                                                         \
        }                                                                                            \
        if (i2 && o4) return FN<1, 2, 4>(__VA_ARGS__);                                               \
        if (i2 && o2) return FN<1, 2, 2>(__VA_ARGS__);                                               \
        if (i2) return FN<1, 2, 1>(__VA_ARGS__);                                                     \
        if (o2) return FN<1, 1, 2>(__VA_ARGS__);                                                     \
        return FN<1, 1, 1>(__VA_ARGS__);                                                             \
    } while (0)

int64_t plan_floats(int N, int H, int W, int Cin, int Cout, int K) {
    int gx, tpb;
    int64_t wsf = 0;
#define SQ_PLAN_CALL(KS_, NI_, NO_) (plan<KS_, NI_, NO_>(N, H, W, Cin, Cout, &gx, &tpb, &wsf), wsf)
    if (Cin % 16 || Cout % 16) return K == 3 ? SQ_PLAN_CALL(3, 1, 1) : SQ_PLAN_CALL(1, 1, 1);
    const bool wide = !narrow_blocks();
    int mi_, mo_;
    max_shape(&mi_, &mo_);
    const bool i2 = wide && mi_ >= 2 && Cin % 32 == 0, o2 = wide && mo_ >= 2 && Cout % 32 == 0,
               o4 = wide && mo_ >= 4 && Cout % 64 == 0;
    if (K == 3) {
        const int k3_ = k3_shape();
        if (k3_ == 14 && Cout % 64 == 0) return SQ_PLAN_CALL(3, 1, 4);
        if (k3_ == 22 && Cin % 32 == 0 && Cout % 32 == 0) return SQ_PLAN_CALL(3, 2, 2);
        if (k3_ == 12 && Cout % 32 == 0) return SQ_PLAN_CALL(3, 1, 2);
        if (o2) return SQ_PLAN_CALL(3, 1, 2);
        if (i2) return SQ_PLAN_CALL(3, 2, 1);
        return SQ_PLAN_CALL(3, 1, 1);
    }
    if (i2 && o4) return SQ_PLAN_CALL(1, 2, 4);
    if (i2 && o2) return SQ_PLAN_CALL(1, 2, 2);
    if (i2) return SQ_PLAN_CALL(1, 2, 1);
    if (o2) return SQ_PLAN_CALL(1, 1, 2);
    return SQ_PLAN_CALL(1, 1, 1);
#undef SQ_PLAN_CALL
}

template <int KS, int NI, int NO>
int launch_b(const __bf16 *x, const __bf16 *dy, float *dw, float *db, float *ws, int N, int H, int W, int Cin, int Cout,
             hipStream_t st) {
    return launch<KS, NI, NO, __bf16>(x, dy, dw, db, ws, N, H, W, Cin, Cout, st);
}
template <int KS, int NI, int NO>
int launch_f(const float *x, const float *dy, float *dw, float *db, float *ws, int N, int H, int W, int Cin, int Cout,
             hipStream_t st) {
    return launch<KS, NI, NO, float>(x, dy, dw, db, ws, N, H, W, Cin, Cout, st);
}

int launch_any(const __bf16 *x, const __bf16 *dy, float *dw, float *db, float *ws, int N, int H, int W, int Cin,
               int Cout, int K, hipStream_t st) {
    if (Cin % 16 || Cout % 16)
        return K == 3 ? launch_rag<3, __bf16>(x, dy, dw, db, ws, N, H, W, Cin, Cout, st)
                      : launch_rag<1, __bf16>(x, dy, dw, db, ws, N, H, W, Cin, Cout, st);
    SQ_WGRAD_BF16_DISPATCH(launch_b, x, dy, dw, db, ws, N, H, W, Cin, Cout, st);
}
int launch_any_mixed(const float *x, const float *dy, float *dw, float *db, float *ws, int N, int H, int W, int Cin,
                     int Cout, int K, hipStream_t st) {
    SQ_WGRAD_BF16_DISPATCH(launch_f, x, dy, dw, db, ws, N, H, W, Cin, Cout, st);
}

// block shape (ni, no) of a layer: the choice SQ_WGRAD_BF16_DISPATCH makes
inline void shape_for(int K, int Cin, int Cout, int *ni, int *no) {
    const bool wide = !narrow_blocks();
    int mi_, mo_;
    max_shape(&mi_, &mo_);
    const bool i2 = wide && mi_ >= 2 && Cin % 32 == 0, o2 = wide && mo_ >= 2 && Cout % 32 == 0,
               o4 = wide && mo_ >= 4 && Cout % 64 == 0;
    *ni = 1, *no = 1;
    if (K == 3) {
        const int k3_ = k3_shape();
        if (k3_ == 14 && Cout % 64 == 0) { *no = 4; return; }
        if (k3_ == 22 && Cin % 32 == 0 && Cout % 32 == 0) { *ni = 2, *no = 2; return; }
        if (k3_ == 12 && Cout % 32 == 0) { *no = 2; return; }
        if (o2) { *no = 2; return; }
        if (i2) { *ni = 2; return; }
        return;
    }
    if (i2 && o4) { *ni = 2, *no = 4; return; }
    if (i2 && o2) { *ni = 2, *no = 2; return; }
    if (i2) { *ni = 2; return; }
    if (o2) { *no = 2; return; }
}

template <int KS, int NI, int NO>
constexpr int pf_bf16() {
    using C = WB<KS, NI, NO>;
    constexpr int acc_regs = C::NTAP * NI * NO * 4, set_regs = (C::XSLOTS + C::YSLOTS) * 4;
    constexpr int budget = (wgrad_occ<KS, NI, NO>() == 1 ? 300 : (wgrad_occ<KS, NI, NO>() == 4 ? 100 : (wgrad_occ<KS, NI, NO>() == 3 ? 140 : SQ_WGRAD_BUDGET2))) - acc_regs - 40;
    return budget / set_regs >= 4 ? 4 : (budget / set_regs >= 3 ? 3 : (budget / set_regs >= 2 ? 2 : 1));
}

// one grouped main launch + one grouped finish launch for items [0, n) (n <= GROUP_MAX), all of block shape <KS, NI, NO>
// geometry the kernel tiles: the tensor itself, or the mosaic of its small images
inline void item_geom(const sq_wgrad_item &it, int *N, int *H, int *W) {
    if (it.mosaic_R > 0) { *N = 1; *H = it.mosaic_R * (it.H + 1); *W = it.mosaic_Cc * (it.W + 1); }
    else { *N = it.N; *H = it.H; *W = it.W; }
}

template <int KS, int NI, int NO, bool MOS = false, bool RAG = false>
int launch_group(const sq_wgrad_item *const *items, int n, float *ws, hipStream_t st) {
    using C = WB<KS, NI, NO>;
    constexpr int PF = pf_bf16<KS, NI, NO>();
    static bool attr_set = false;
    auto kern = conv_wgrad_bf16_group_kernel<KS, NI, NO, PF, __bf16, MOS, RAG>;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                C::LDS_BYTES) != hipSuccess) {
            sq_set_error("conv_wgrad_bf16(group): cannot reserve %d bytes of LDS", C::LDS_BYTES);
            return SQ_ELAUNCH;
        }
        attr_set = true;
    }
    SqWgradGroup g;
    SqWgradFinishGroup f;
    g.n = f.n = n;
    static const int pair_major = [] { const char *e = getenv("SQ_WGRAD_PAIR_MAJOR"); return e ? atoi(e) : 1; }();
    g.pair_major = pair_major;
    int blocks = 0, fblocks = 0;
    float *wp = ws;
    // Tiles per block: a layer launched alone is cut into ~512 blocks to fill the chip, and every block pays the cross-wave
    // reduction and an 18 KB partial at its end -- the "fixed cost" of these launches.  Sharing the grid, the layers fill the
    // chip together, so each is cut into a quarter as many, four times longer blocks (measured on one box, 13 + 4 layers of the
    // training step: own plans 3.10 ms, / 2: 3.00, / 4: 2.95, / 8: 3.03; an equal-work split into ~1500 blocks that gave the
    // four transpose-conv layers 1500 blocks instead of 512: 3.29).  SQ_WGRAD_GROUP_SHRINK overrides the divisor.
    static const int shrink_env = [] { const char *e = getenv("SQ_WGRAD_GROUP_SHRINK"); return e ? atoi(e) : 0; }();
    const int shrink = shrink_env > 0 ? shrink_env : (n < 4 ? n : 4);
    for (int e = 0; e < n; ++e) {
        const sq_wgrad_item &it = *items[e];
        int gx, tpb, gN, gH, gW;
        int64_t wsf;
        item_geom(it, &gN, &gH, &gW);
        plan<KS, NI, NO>(gN, gH, gW, it.Cin, it.Cout, &gx, &tpb, &wsf);
        const int npairs = ((it.Cin + C::CI - 1) / C::CI) * ((it.Cout + C::CO - 1) / C::CO);
        static const int dbg = [] { const char *e = getenv("SQ_WGRAD_GROUP_DEBUG"); return e ? atoi(e) : 0; }();
        if (shrink > 1) {
            const int ntl = ((gW + TW - 1) / TW) * ((gH + TH - 1) / TH) * gN;
            int g2 = gx / shrink;
            if (g2 < 1) g2 = 1;
            tpb = (ntl + g2 - 1) / g2;
            gx = (ntl + tpb - 1) / tpb;
        }
        if (dbg) fprintf(stderr, "group<%d,%d,%d,%d,%d> item %d/%d: N=%d H=%d W=%d Cin=%d Cout=%d npairs=%d tpb=%d gx=%d acc=%d\n", KS, NI, NO,
                         (int)MOS, (int)RAG, e, n, gN, gH, gW, it.Cin, it.Cout, npairs, tpb, gx, it.accumulate);
        g.first[e] = blocks;
        g.x[e] = it.x; g.dy[e] = it.dy; g.partials[e] = wp;
        g.N[e] = gN; g.H[e] = gH; g.W[e] = gW; g.Cin[e] = it.Cin; g.Cout[e] = it.Cout;
        g.tiles_x[e] = (gW + TW - 1) / TW; g.tiles_y[e] = (gH + TH - 1) / TH;
        g.mos[e] = SqMos{};
        if (MOS) {
            g.mos[e].h = it.H; g.mos[e].w = it.W; g.mos[e].cc = it.mosaic_Cc; g.mos[e].n = it.N;
            g.mos[e].mh = (65536u + (unsigned)it.H) / (unsigned)(it.H + 1);
            g.mos[e].mw = (65536u + (unsigned)it.W) / (unsigned)(it.W + 1);
        }
        g.tpb[e] = wgrad_interleave() ? -tpb : tpb; g.gx[e] = gx;
        blocks += gx * npairs;
        int G = sq_group_size(gx);
        if (G > 16) G = 16;
        const int64_t total = (int64_t)npairs * C::RED_FLOATS;
        const int OUT = 256 / G;
        f.first[e] = fblocks;
        f.partials[e] = wp; f.dw[e] = it.dw; f.db[e] = it.db;
        f.nblk[e] = gx; f.Cin[e] = it.Cin; f.Cout[e] = it.Cout; f.G[e] = G; f.dw_scale[e] = it.dw_scale;
        f.ct[e] = KS == 1 ? it.convT_cout : 0;
        f.acc[e] = it.accumulate;
        fblocks += (int)((total + OUT - 1) / OUT);
        wp += wsf;
    }
    g.first[n] = blocks;
    f.first[n] = fblocks;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), C::LDS_BYTES, st, g);
    int rc = sq_check_launch("sq_conv2d_nhwc_wgrad_group_bf16");
    if (rc) return rc;
    hipLaunchKernelGGL((conv_wgrad_bf16_finish_group_kernel<KS, NI, NO>), dim3(fblocks), dim3(256), 0, st, f);
    return sq_check_launch("sq_conv2d_nhwc_wgrad_group_bf16(finish)");
}

int launch_group_any(int K, int ni, int no, int kind, const sq_wgrad_item *const *items, int n, float *ws, hipStream_t st) {
    if (kind == 2)                                              // ragged channel counts: 16 x 16 channel blocks
        return K == 3 ? launch_group<3, 1, 1, false, true>(items, n, ws, st) : launch_group<1, 1, 1, false, true>(items, n, ws, st);
    if (kind == 1) {                                            // small-image mosaics (3x3)
#define SQ_GM(I_, O_) if (ni == I_ && no == O_) return launch_group<3, I_, O_, true>(items, n, ws, st)
        SQ_GM(1, 4); SQ_GM(2, 2); SQ_GM(1, 2); SQ_GM(2, 1); SQ_GM(1, 1);
#undef SQ_GM
    }
#define SQ_G(K_, I_, O_) if (K == K_ && ni == I_ && no == O_) return launch_group<K_, I_, O_>(items, n, ws, st)
    SQ_G(3, 1, 4); SQ_G(3, 2, 2); SQ_G(3, 1, 2); SQ_G(3, 2, 1); SQ_G(3, 1, 1);
    SQ_G(1, 2, 4); SQ_G(1, 2, 2); SQ_G(1, 2, 1); SQ_G(1, 1, 2); SQ_G(1, 1, 1);
#undef SQ_G
    sq_set_error("sq_conv2d_nhwc_wgrad_group_bf16: no kernel for K=%d blocks %dx%d", K, ni, no);
    return SQ_EINVAL;
}

bool ok_shape(int N, int H, int W, int Cin, int Cout, int K, int elem_bytes = 2) {
    // bf16 tensors: channel counts that are 8 mod 16 run the ragged form; the mixed (f32 tensor) entries keep % 16
    const int m = elem_bytes == 2 ? 8 : 16;
    return N > 0 && H > 0 && W > 0 && (K == 1 || K == 3) && Cin > 0 && Cin % m == 0 && Cout > 0 && Cout % m == 0 &&
           (size_t)N * H * W * (size_t)(Cin > Cout ? Cin : Cout) * elem_bytes < ((size_t)1 << 31);
}

}  // namespace

extern "C" int64_t sq_conv2d_nhwc_wgrad_workspace_bf16(int N, int H, int W, int Cin, int Cout, int K) {
    if (!ok_shape(N, H, W, Cin, Cout, K)) return -1;
    return plan_floats(N, H, W, Cin, Cout, K) * 4;
}

// dW (K,K,Cin,Cout) f32 and db (Cout) f32 from bf16 X (N,H,W,Cin) and bf16 dY (N,H,W,Cout).
extern "C" int sq_conv2d_nhwc_wgrad_bf16(const void *x, const void *dy, float *dw, float *db, float *workspace,
                                         int N, int H, int W, int Cin, int Cout, int K, void *stream) {
    SQ_REQUIRE(x && dy && dw && workspace, "sq_conv2d_nhwc_wgrad_bf16: null pointer");
    SQ_REQUIRE(ok_shape(N, H, W, Cin, Cout, K),
               "sq_conv2d_nhwc_wgrad_bf16: unsupported shape Cin=%d Cout=%d K=%d (both %% 8, K 1|3, < 2 GiB)", Cin, Cout, K);
    SQ_REQUIRE(!(t_mos.h || t_convT_cout) || (Cin % 16 == 0 && Cout % 16 == 0),
               "sq_conv2d_nhwc_wgrad_bf16: the mosaic / transpose-conv forms need channel counts that are multiples of 16");
    SQ_REQUIRE_ALIGNED(x); SQ_REQUIRE_ALIGNED(dy); SQ_REQUIRE_ALIGNED(workspace);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const __bf16 *xb = reinterpret_cast<const __bf16 *>(x), *yb = reinterpret_cast<const __bf16 *>(dy);
    return launch_any(xb, yb, dw, db, workspace, N, H, W, Cin, Cout, K, st);
}

// "mixed" weight gradient: f32 X and dY, rounded to bf16 on the way into LDS, f32 accumulation -- the wgrad of
// sq_conv2d_nhwc_fwd_mixed_f32.  Same block shapes, workspace and fixed-order finish as the bf16 entry.
extern "C" int64_t sq_conv2d_nhwc_wgrad_workspace_mixed_f32(int N, int H, int W, int Cin, int Cout, int K) {
    if (!ok_shape(N, H, W, Cin, Cout, K, 4)) return -1;
    return plan_floats(N, H, W, Cin, Cout, K) * 4;
}

extern "C" int sq_conv2d_nhwc_wgrad_mixed_f32(const float *x, const float *dy, float *dw, float *db, float *workspace,
                                              int N, int H, int W, int Cin, int Cout, int K, void *stream) {
    SQ_REQUIRE(x && dy && dw && workspace, "sq_conv2d_nhwc_wgrad_mixed_f32: null pointer");
    SQ_REQUIRE(ok_shape(N, H, W, Cin, Cout, K, 4),
               "sq_conv2d_nhwc_wgrad_mixed_f32: unsupported shape Cin=%d Cout=%d K=%d (both %% 16, K 1|3, < 2 GiB)", Cin,
               Cout, K);
    SQ_REQUIRE_ALIGNED(x); SQ_REQUIRE_ALIGNED(dy); SQ_REQUIRE_ALIGNED(workspace);
    return launch_any_mixed(x, dy, dw, db, workspace, N, H, W, Cin, Cout, K, reinterpret_cast<hipStream_t>(stream));
}

// parameter gradients of conv_transpose_layer (2x2, stride 2; unet.py:312-318) from its input x (N,H,W,Cin) and the
// gradient of its output in space-to-depth form g (N,H,W,4*Cout): dW (2,2,Cout,Cin) f32 and db (Cout) f32 or NULL.
// The 1x1 weight-gradient kernel of sq_conv2d_nhwc_wgrad_bf16 with a finish kernel that writes these layouts (the
// permutation and the 4-way bias sum were two framework kernels per level).  Workspace: that of the 1x1 wgrad Cin -> 4*Cout.
extern "C" int sq_convT2x2s2_wgrad_bf16(const void *x, const void *g, float *dw, float *db, float *workspace, int N, int H,
                                        int W, int Cin, int Cout, void *stream) {
    SQ_REQUIRE(Cout > 0 && Cout % 4 == 0, "sq_convT2x2s2_wgrad_bf16: Cout=%d", Cout);
    t_convT_cout = Cout;
    const int rc = sq_conv2d_nhwc_wgrad_bf16(x, g, dw, db, workspace, N, H, W, Cin, 4 * Cout, 1, stream);
    t_convT_cout = 0;
    return rc;
}

// the same on a batch of small images (Nimg, h, w, C) taken as ONE mosaic image of R x Cc cells (sq_mosaic_pack_f32's layout,
// 3x3 only) without building the mosaics of X and dY: same sums as pack -> wgrad (separator pixels contribute zeros).
// Workspace: sq_conv2d_nhwc_wgrad_workspace_bf16(1, R*(h+1), Cc*(w+1), Cin, Cout, 3).
extern "C" int sq_conv2d_nhwc_wgrad_mixed_mosaic_f32(const float *x, const float *dy, float *dw, float *db, float *workspace,
                                                     int Nimg, int h, int w, int Cin, int Cout, int R, int Cc, float dw_scale,
                                                     void *stream) {
    SQ_REQUIRE(Nimg > 0 && h > 0 && w > 0 && h <= 8 && w <= 8 && R > 0 && Cc > 0 && (int64_t)R * Cc >= Nimg,
               "sq_conv2d_nhwc_wgrad_mixed_mosaic_f32: images up to 8 x 8, R * Cc >= Nimg (Nimg=%d R=%d Cc=%d)", Nimg, R, Cc);
    const int H = R * (h + 1), W = Cc * (w + 1);
    SQ_REQUIRE(H < (1 << 13) && W < (1 << 13), "sq_conv2d_nhwc_wgrad_mixed_mosaic_f32: mosaic < 8192");
    t_mos.h = h; t_mos.w = w; t_mos.cc = Cc; t_mos.n = Nimg;
    t_mos.mh = (65536u + (unsigned)h) / (unsigned)(h + 1);
    t_mos.mw = (65536u + (unsigned)w) / (unsigned)(w + 1);
    t_dw_scale = dw_scale;
    const int rc = sq_conv2d_nhwc_wgrad_mixed_f32(x, dy, dw, db, workspace, 1, H, W, Cin, Cout, 3, stream);
    t_dw_scale = 1.0f;
    t_mos = SqMos{};
    return rc;
}

// dW additionally multiplied by dw_scale in the finish kernel (db is not): the gradient of a weighted_conv2d kernel is
// wscale * (raw dW), gan.py:75-79 -- one f32 multiply per element, exactly what a separate scalar-multiply pass gives.
extern "C" int sq_conv2d_nhwc_wgrad_scaled_mixed_f32(const float *x, const float *dy, float *dw, float *db, float *workspace,
                                                     int N, int H, int W, int Cin, int Cout, int K, float dw_scale,
                                                     void *stream) {
    t_dw_scale = dw_scale;
    const int rc = sq_conv2d_nhwc_wgrad_mixed_f32(x, dy, dw, db, workspace, N, H, W, Cin, Cout, K, stream);
    t_dw_scale = 1.0f;
    return rc;
}

// the two GAN forms above on bf16 tensors (bf16 storage, config 5): dW times dw_scale in the finish kernel ...
extern "C" int sq_conv2d_nhwc_wgrad_scaled_bf16(const void *x, const void *dy, float *dw, float *db, float *workspace, int N,
                                                int H, int W, int Cin, int Cout, int K, float dw_scale, void *stream) {
    t_dw_scale = dw_scale;
    const int rc = sq_conv2d_nhwc_wgrad_bf16(x, dy, dw, db, workspace, N, H, W, Cin, Cout, K, stream);
    t_dw_scale = 1.0f;
    return rc;
}

// ... and the batch of small images addressed as one mosaic (3x3; Cin, Cout % 16 == 0).
// Workspace: sq_conv2d_nhwc_wgrad_workspace_bf16(1, R*(h+1), Cc*(w+1), Cin, Cout, 3).
extern "C" int sq_conv2d_nhwc_wgrad_mosaic_bf16(const void *x, const void *dy, float *dw, float *db, float *workspace, int Nimg,
                                                int h, int w, int Cin, int Cout, int R, int Cc, float dw_scale, void *stream) {
    SQ_REQUIRE(Nimg > 0 && h > 0 && w > 0 && h <= 8 && w <= 8 && R > 0 && Cc > 0 && (int64_t)R * Cc >= Nimg,
               "sq_conv2d_nhwc_wgrad_mosaic_bf16: images up to 8 x 8, R * Cc >= Nimg (Nimg=%d R=%d Cc=%d)", Nimg, R, Cc);
    const int H = R * (h + 1), W = Cc * (w + 1);
    SQ_REQUIRE(H < (1 << 13) && W < (1 << 13), "sq_conv2d_nhwc_wgrad_mosaic_bf16: mosaic < 8192");
    t_mos.h = h; t_mos.w = w; t_mos.cc = Cc; t_mos.n = Nimg;
    t_mos.mh = (65536u + (unsigned)h) / (unsigned)(h + 1);
    t_mos.mw = (65536u + (unsigned)w) / (unsigned)(w + 1);
    t_dw_scale = dw_scale;
    const int rc = sq_conv2d_nhwc_wgrad_bf16(x, dy, dw, db, workspace, 1, H, W, Cin, Cout, 3, stream);
    t_dw_scale = 1.0f;
    t_mos = SqMos{};
    return rc;
}

// ---- several layers in one launch (sq_wgrad_item, include/sequitr_hip.h) ----------------------------------------------------
static int item_kind(const sq_wgrad_item &it) { return it.mosaic_R > 0 ? 1 : ((it.Cin % 16 || it.Cout % 16) ? 2 : 0); }
static void item_shape(const sq_wgrad_item &it, int *ni, int *no) {
    if (item_kind(it) == 2) { *ni = *no = 1; return; }
    shape_for(it.K, it.Cin, it.Cout, ni, no);
}
static int64_t item_floats(const sq_wgrad_item &it) {
    int N, H, W;
    item_geom(it, &N, &H, &W);
    return plan_floats(N, H, W, it.Cin, it.Cout, it.K);
}
static bool group_item_ok(const sq_wgrad_item &it) {
    int N, H, W;
    item_geom(it, &N, &H, &W);
    if (!(it.x && it.dy && it.dw && ok_shape(N, H, W, it.Cin, it.Cout, it.K))) return false;
    if (it.mosaic_R > 0)                                        // images up to 8 x 8, 3x3, full 16-channel blocks
        return it.K == 3 && it.H <= 8 && it.W <= 8 && it.mosaic_Cc > 0 && (int64_t)it.mosaic_R * it.mosaic_Cc >= it.N &&
               H < (1 << 13) && W < (1 << 13) && it.Cin % 16 == 0 && it.Cout % 16 == 0 && it.convT_cout == 0;
    if (it.Cin % 16 || it.Cout % 16) return it.convT_cout == 0;
    return it.convT_cout == 0 || (it.K == 1 && it.convT_cout * 4 == it.Cout);
}

extern "C" int64_t sq_conv2d_nhwc_wgrad_group_workspace_bf16(const sq_wgrad_item *items, int n) {
    if (!items || n <= 0) return -1;
    int64_t total = 0;
    for (int i = 0; i < n; ++i) {
        if (!group_item_ok(items[i])) return -1;
        total += item_floats(items[i]) * 4;
    }
    return total;
}

// dW / db of n layers (bf16 X and dY, channel counts multiples of 16) with one main launch and one finish launch per block
// shape among them: each layer's result is bit-identical to sq_conv2d_nhwc_wgrad_scaled_bf16 / sq_convT2x2s2_wgrad_bf16 on it.
// workspace: sq_conv2d_nhwc_wgrad_group_workspace_bf16(items, n) bytes.
extern "C" int sq_conv2d_nhwc_wgrad_group_bf16(const sq_wgrad_item *items, int n, float *workspace, void *stream) {
    SQ_REQUIRE(items && n > 0 && workspace, "sq_conv2d_nhwc_wgrad_group_bf16: null pointer / no items");
    SQ_REQUIRE_ALIGNED(workspace);
    for (int i = 0; i < n; ++i) {
        SQ_REQUIRE(group_item_ok(items[i]), "sq_conv2d_nhwc_wgrad_group_bf16: item %d: Cin=%d Cout=%d K=%d (both %% 8, K 1|3, < 2 GiB; mosaic / convT: %% 16)", i,
                   items[i].Cin, items[i].Cout, items[i].K);
        SQ_REQUIRE_ALIGNED(items[i].x); SQ_REQUIRE_ALIGNED(items[i].dy);
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // buckets by (K, ni, no), GROUP_MAX items per launch.  Launches of one stream run in order, so every bucket lays its
    // partials out from the start of the workspace (the previous bucket's finish pass has read its own by then).
    std::vector<char> done(n, 0);
    for (int i = 0; i < n; ++i) {
        if (done[i]) continue;
        int ni, no;
        item_shape(items[i], &ni, &no);
        const int kind = item_kind(items[i]);
        const sq_wgrad_item *bucket[GROUP_MAX];
        int nb = 0;
        for (int j = i; j < n && nb < GROUP_MAX; ++j) {
            if (done[j]) continue;
            int nj, oj;
            item_shape(items[j], &nj, &oj);
            if (items[j].K != items[i].K || nj != ni || oj != no || item_kind(items[j]) != kind) continue;
            // two contributions to one gradient never share a launch (their finish blocks would race): the later one waits
            // for the next round of this shape -- rounds run in item order, so "write, then accumulate" stays in order
            bool clash = false;
            for (int b2 = 0; b2 < nb && !clash; ++b2)
                clash = bucket[b2]->dw == items[j].dw || (items[j].db && bucket[b2]->db == items[j].db);
            for (int j2 = i; j2 < j && !clash; ++j2)              // ... nor overtake an earlier one that is still waiting
                clash = !done[j2] && (items[j2].dw == items[j].dw || (items[j].db && items[j2].db == items[j].db));
            if (clash) continue;
            bucket[nb++] = &items[j];
            done[j] = 1;
        }
        const int rc = launch_group_any(items[i].K, ni, no, kind, bucket, nb, workspace, st);
        if (rc) return rc;
    }
    return SQ_OK;
}
