// The level-0 block of the U-Net in f32 (16 -> 16 channels, 3x3, 512^2 tiles: sequitr/networks/unet.py:238-243,
// 252-253, 299-322) -- the same arithmetic and the same fmaf chains as conv_mfma_f32_v2_kernel<16, 3, 16, MODE>
// (bit-identical results), rebuilt round the one thing the counters say these launches are bound by: at 16 output
// channels a wave issues only 4 MFMAs per operand fragment, and every OTHER instruction a SIMD issues (VALU, SALU,
// LDS, VMEM -- about 2 cycles each, measured over the six forms of the v2 kernel: profiles/r04_l0_issue_model.txt)
// is time its matrix pipe stands still.  The v2 forms issue 4.3-5.1 such instructions per MFMA; this kernel ~1.5:
//   * the 36 weight fragments of the 16 x 16 x 3 x 3 filter live in REGISTERS for the whole launch (no weight slab in
//     LDS, no A reads);
//   * the halo image is stored channel-TRANSPOSED ([pixel][kk][s] = channel 4 s + kk, 24 floats per pixel): one
//     ds_read_b128 hands a lane its B operands of all four channel steps of a tap, conflict-free (6 li + kk covers the
//     16 slots of a bank row for each of the instruction's 16-lane groups), and a halo row read for tap row ky of
//     output row r is the same fragment as tap row ky - 1 of output row r + 1: the wave walks its 6 halo rows ONCE
//     (18 ds_read_b128 per tile instead of 180 ds_read_b32), each output row still seeing its taps in raster order;
//   * interior tiles (88 % at 512^2) issue their halo loads with the tile offset in an SGPR and a loop-invariant
//     per-lane offset: no bounds arithmetic; the tile coordinates advance by carries, not by division;
//   * the epilogue is compiled per kind (store / store + 2x2 max-pool / 1x1 head + argmax) with ReLU as v_max;
//   * the 1x1 head runs from registers: the accumulators' [channel quad][pixel] layout is transposed across the
//     wave's four 16-lane rows with v_permlane16_swap / v_permlane32_swap (16 instructions, no LDS image, no wave
//     barrier) so that every lane owns one pixel with its 16 channels, the head is 16 v_mfma_f32_4x4x1 (one channel
//     each, 2 passes) instead of 16 v_mfma_f32_16x16x4 (8 passes, 14 of 16 rows zero), and all 64 lanes store their
//     pixel's two logits as one 8-byte store;
//   * FIRST (conv1 of down0 on the fly): the 20 x 20 input patch is double-buffered in LDS and fetched two tiles
//     ahead, so a tile costs two block barriers instead of three;
//   * UP (transpose conv + bridge on the fly): the low-resolution patch is fetched two tiles ahead and committed at the
//     top of the iteration: two barriers instead of three.
// Shapes it takes: Cin = Cout = 16, K = 3, H and W multiples of 16, ReLU, head_c = 2 -- everything else stays with
// the generic kernel (sq_conv_f32_v2.hip), which is also the A/B reference (SQ_CONV_L0=0).
#include <stdlib.h>
#include "sq_common.h"
#include "sq_conv_epi.h"

#ifndef SQ_L0_SETPRIO
#define SQ_L0_SETPRIO 1              // wave priority low while feeding the matrix pipe, high while staging / storing (A/B switch)
#endif
#ifndef SQ_L0_OCC
#define SQ_L0_OCC 4                  // resident blocks per CU of the plain / FIRST forms (A/B switch)
#endif
#ifndef SQ_L0_UP_OCC
#define SQ_L0_UP_OCC 2               // resident blocks per CU of the UP form: 190 registers unspilled; at 3 (168) the prefetch registers spill: 625 vs 486 us
#endif

namespace {

constexpr int TH = 16, TW = 16, HWD = 18, HP = HWD * HWD;
constexpr int PS = 24;                          // floats per halo pixel: [kk][s] + 8 of padding (conflict-free b128)
constexpr int XS_FLOATS = HP * PS;              // 7776 floats = 31104 B
constexpr int XITEMS = HP * 4, XSLOTS = 6;      // 1296 float4 of a halo over 256 threads
constexpr int IN_W = 20, IN_FLOATS = IN_W * IN_W;
constexpr int UP_W = 10, UP_CIN = 32, UP_PS = 40, UP_FLOATS = UP_W * UP_W * UP_PS;   // [pixel][16-channel half][kk][s] + 8 of padding
constexpr unsigned OOB = 0x80000000u;

enum { EPI_STORE = 0, EPI_POOL = 1, EPI_HEAD = 2 };
enum { M_PLAIN = 0, M_FIRST = 1, M_UP = 2 };

struct L0Args {
    const float *x, *w, *bias;
    float *y;
    int N, H, W;
    int tiles_x, tiles_y, ntiles;
    int gx, gy, gn;                             // the grid stride G = gn * tiles_x * tiles_y + gy * tiles_x + gx
    SqConvEpi epi;
};

struct Pos { int tx, ty, n; };

typedef unsigned u32x4 __attribute__((__vector_size__(4 * sizeof(unsigned))));
typedef unsigned u32x2 __attribute__((__vector_size__(2 * sizeof(unsigned))));

// E[g][j] (register j of lane row g) -> T[g][s] = E[s][g]: the 4 x 4 transpose between "register" and "16-lane row"
__device__ __forceinline__ void row_transpose(float &a0, float &a1, float &a2, float &a3) {
    const auto p = __builtin_amdgcn_permlane16_swap(__float_as_uint(a0), __float_as_uint(a1), false, false);
    const auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(a2), __float_as_uint(a3), false, false);
    const auto u = __builtin_amdgcn_permlane32_swap(p[0], q[0], false, false);
    const auto v = __builtin_amdgcn_permlane32_swap(p[1], q[1], false, false);
    a0 = __uint_as_float(u[0]); a2 = __uint_as_float(u[1]);
    a1 = __uint_as_float(v[0]); a3 = __uint_as_float(v[1]);
}

__device__ __forceinline__ float relu(float v) { return __builtin_fmaxf(v, 0.0f); }   // NaN -> 0, -0 -> +0 like (v > 0 ? v : 0)

template <int MODE, int EPI, int BRIDGE, int NB>            // NB: output channels / 16 (2: the 16 -> 32 first conv of level 1)
__global__ __launch_bounds__(256, (MODE == M_UP ? SQ_L0_UP_OCC : (NB == 2 ? 2 : SQ_L0_OCC))) void conv_l0_kernel(const L0Args a) {
    static_assert(NB == 1 || (NB == 2 && MODE == M_PLAIN && EPI == EPI_STORE), "32 output channels: the plain form only");
    constexpr int CO = 16 * NB;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *xs = smem;
    float *xin = smem + XS_FLOATS;              // FIRST: two 20 x 20 patches
    float *xl = smem + XS_FLOATS;               // UP: the 10 x 10 x 32 low-resolution patch, channel-transposed like the halo image

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, kk = lane >> 4;
    const int H = a.H, W = a.W;
    const int G = (int)gridDim.x;
    const int vb = (int)sq_xcd_remap(blockIdx.x, gridDim.x);
    if (vb >= a.ntiles) return;
    const int t_count = (a.ntiles - vb + G - 1) / G;

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.x), 0, (int)((size_t)a.N * H * W * (MODE == M_FIRST ? 1 : 16) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(
        a.y, 0, a.y ? (int)((size_t)a.N * H * W * CO * 4) : 0, 0x00020000);

    // ---- the filter, once per launch: A[m = output channel li][k = channel 4 s + kk] of tap t ---------------------
    float af[9][4][NB];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) af[t][s][nb] = a.w[(t * 16 + 4 * s + kk) * CO + nb * 16 + li];
    float4 bv[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
        bv[nb] = a.bias ? *reinterpret_cast<const float4 *>(a.bias + nb * 16 + 4 * kk) : make_float4(0.f, 0.f, 0.f, 0.f);

    auto decode = [&](int tile) {
        Pos p;
        p.tx = tile % a.tiles_x;
        p.ty = (tile / a.tiles_x) % a.tiles_y;
        p.n = tile / (a.tiles_x * a.tiles_y);
        return p;
    };
    auto advance = [&](Pos p) {                 // tile + G, by carries
        p.tx += a.gx;
        if (p.tx >= a.tiles_x) { p.tx -= a.tiles_x; p.ty += 1; }
        p.ty += a.gy;
        if (p.ty >= a.tiles_y) { p.ty -= a.tiles_y; p.n += 1; }
        p.n += a.gn;
        return p;
    };
    auto interior = [&](const Pos &p) {
        return p.tx > 0 && p.tx < a.tiles_x - 1 && p.ty > 0 && p.ty < a.tiles_y - 1;
    };

    // ---- halo loads (PLAIN): lane -> (pixel, channel quad) in memory order, 6 x 16 B per thread --------------------
    float4 xr[XSLOTS];
    int xrel[XSLOTS], xpp[XSLOTS];
    if constexpr (MODE == M_PLAIN) {
#pragma unroll
        for (int sl = 0; sl < XSLOTS; ++sl) {
            const int idx = tid + sl * 256, pix = idx >> 2, q = idx & 3;
            const int py = pix / HWD, px = pix % HWD;
            xrel[sl] = idx < XITEMS ? ((py * W + px) * 16 + q * 4) * 4 : (int)OOB;
            xpp[sl] = (py << 8) | px;
        }
    }
    auto issue_halo = [&](const Pos &p) {
        const int x0 = p.tx * TW - 1, y0 = p.ty * TH - 1;
        const int base = (((p.n * H + y0) * W + x0) * 16) * 4;      // "negative" on the top / left border: the checked path
        if (interior(p)) {
#pragma unroll
            for (int sl = 0; sl < XSLOTS; ++sl) {
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, xrel[sl], base, 0);
                xr[sl] = *reinterpret_cast<const float4 *>(&v);
            }
        } else {
#pragma unroll
            for (int sl = 0; sl < XSLOTS; ++sl) {
                const int py = xpp[sl] >> 8, px = xpp[sl] & 255;
                const bool inb = (unsigned)(y0 + py) < (unsigned)H && (unsigned)(x0 + px) < (unsigned)W && xrel[sl] != (int)OOB;
                const unsigned off = inb ? (unsigned)(base + xrel[sl]) : OOB;
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, off, 0, 0);
                xr[sl] = *reinterpret_cast<const float4 *>(&v);
            }
        }
    };
    float *cw = xs + (tid >> 2) * PS + (tid & 3);                   // channel 4 q + j of the pixel -> [kk = j][s = q]
    auto commit_halo = [&]() {
#pragma unroll
        for (int sl = 0; sl < XSLOTS; ++sl) {
            if (sl < XSLOTS - 1 || tid < XITEMS - (XSLOTS - 1) * 256) {
                float *d = cw + sl * 64 * PS;
                d[0] = xr[sl].x; d[4] = xr[sl].y; d[8] = xr[sl].z; d[12] = xr[sl].w;
            }
        }
    };

    // ---- FIRST: conv1 (3x3, 1 -> 16, bias, ReLU) of the 18 x 18 halo on the matrix cores, straight into the halo
    // image (the chain: acc = 0, 9 taps in raster order; taps 9..11 of the third step carry zero weights) -----------
    float a1[3] = {0.f, 0.f, 0.f};
    int toff[3] = {0, 0, 0};
    float4 b1v = make_float4(0.f, 0.f, 0.f, 0.f);
    float inr[2] = {0.f, 0.f};
    int fsrc[6] = {0, 0, 0, 0, 0, 0}, fpix[6] = {0, 0, 0, 0, 0, 0};
    int irel[2] = {0, 0};
    if constexpr (MODE == M_FIRST) {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int tap = 4 * s + kk;
            a1[s] = tap < 9 ? a.epi.first_w[tap * 16 + li] : 0.f;
            const int tc = tap < 9 ? tap : 8;
            toff[s] = (tc / 3) * IN_W + tc % 3;
        }
        b1v = *reinterpret_cast<const float4 *>(a.epi.first_b + 4 * kk);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int pix = (wv + 4 * i) * 16 + li, pc = pix < HP ? pix : HP - 1;
            fsrc[i] = (pc / HWD) * IN_W + pc % HWD;
            fpix[i] = pix;
        }
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            const int idx = tid + sl * 256;
            irel[sl] = idx < IN_FLOATS ? ((idx / IN_W) * W + idx % IN_W) * 4 : (int)OOB;
        }
    }
    auto issue_patch = [&](const Pos &p) {                          // 20 x 20 single-channel patch (halo of the halo)
        const int x0 = p.tx * TW - 2, y0 = p.ty * TH - 2;
        const int base = ((p.n * H + y0) * W + x0) * 4;
        if (interior(p)) {
#pragma unroll
            for (int sl = 0; sl < 2; ++sl)
                inr[sl] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, irel[sl], base, 0));
        } else {
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                const int idx = tid + sl * 256, py = idx / IN_W, px = idx % IN_W;
                const bool inb = idx < IN_FLOATS && (unsigned)(y0 + py) < (unsigned)H && (unsigned)(x0 + px) < (unsigned)W;
                const unsigned off = inb ? (unsigned)(base + irel[sl]) : OOB;
                inr[sl] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, off, 0, 0));
            }
        }
    };
    auto commit_patch = [&](int buf) {
        float *d = xin + buf * IN_FLOATS;
        d[tid] = inr[0];
        if (tid < IN_FLOATS - 256) d[tid + 256] = inr[1];
    };
    auto first_conv = [&](const Pos &p, int buf) {
        const float *src = xin + buf * IN_FLOATS;
        const bool inner = interior(p);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (wv + 4 * i < (HP + 15) / 16) {                      // 21 column blocks: wave 0 takes six, the others five
                f32x4 c1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 3; ++s)
                    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], src[fsrc[i] + toff[s]], c1, 0, 0, 0);
                float v0 = relu(c1[0] + b1v.x), v1 = relu(c1[1] + b1v.y), v2 = relu(c1[2] + b1v.z), v3 = relu(c1[3] + b1v.w);
                if (!inner) {                                        // halo pixels outside the image are conv2's ZERO PADDING
                    const int pc = fpix[i] < HP ? fpix[i] : HP - 1;
                    const int gy = p.ty * TH - 1 + pc / HWD, gx = p.tx * TW - 1 + pc % HWD;
                    const bool inside = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
                    v0 = inside ? v0 : 0.f; v1 = inside ? v1 : 0.f; v2 = inside ? v2 : 0.f; v3 = inside ? v3 : 0.f;
                }
                row_transpose(v0, v1, v2, v3);                      // lane row kk now holds channels kk, 4 + kk, 8 + kk, 12 + kk
                if (fpix[i] < HP) *reinterpret_cast<float4 *>(xs + fpix[i] * PS + 4 * kk) = make_float4(v0, v1, v2, v3);
            }
        }
    };

    // ---- UP: merged = bridge(convT2x2s2(up_x) + bias, skip) of the 18 x 18 halo, wave w = parity class w -----------
    float aw[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float4 upb = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 lr[4], sr[6];
    int srel[6] = {0, 0, 0, 0, 0, 0}, lrel[4] = {0, 0, 0, 0};
    unsigned upk[6] = {0, 0, 0, 0, 0, 0};     // per class block: LDS float offsets of the low-res operand (low 16 bits) and of the halo pixel (high 16, 0xFFFF = none)
    const int Hl = H >> 1, Wl = W >> 1;
    const __amdgpu_buffer_rsrc_t lrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.epi.up_x), 0, MODE == M_UP ? (int)((size_t)a.N * Hl * Wl * UP_CIN * 4) : 0, 0x00020000);
    if constexpr (MODE == M_UP) {
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) aw[s8] = a.epi.up_w[(wv * 16 + li) * UP_CIN + 4 * s8 + kk];
        if (a.epi.up_b) upb = *reinterpret_cast<const float4 *>(a.epi.up_b + 4 * kk);
        const int oy = ((wv >> 1) + 1) & 1, ox = ((wv & 1) + 1) & 1;   // halo row hy has parity (hy + 1) & 1
#pragma unroll
        for (int blk = 0; blk < 6; ++blk) {
            const int j = blk * 16 + li, jc = j < 81 ? j : 80;
            const int hy = 2 * (jc / 9) + oy, hx = 2 * (jc % 9) + ox;
            srel[blk] = j < 81 ? ((hy * W + hx) * 16 + 4 * kk) * 4 : (int)OOB;
            const unsigned usrc = (((hy + 1) >> 1) * UP_W + ((hx + 1) >> 1)) * UP_PS + 4 * kk;
            const unsigned udst = j < 81 ? (hy * HWD + hx) * PS + 4 * kk : 0xFFFFu;
            upk[blk] = usrc | (udst << 16);
        }
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
            const int idx = tid + sl * 256, lp = idx >> 3, q = idx & 7;
            lrel[sl] = idx < UP_W * UP_W * 8 ? (((lp / UP_W) * Wl + lp % UP_W) * UP_CIN + q * 4) * 4 : (int)OOB;
        }
    }
    auto issue_skip = [&](const Pos &p) {       // the skip halo in the transpose conv's own fragment layout
        const int x0 = p.tx * TW - 1, y0 = p.ty * TH - 1;
        const int base = (((p.n * H + y0) * W + x0) * 16) * 4;
        if (interior(p)) {
#pragma unroll
            for (int blk = 0; blk < 6; ++blk) {
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, srel[blk], base, 0);
                sr[blk] = *reinterpret_cast<const float4 *>(&v);
            }
        } else {
            const int oy = ((wv >> 1) + 1) & 1, ox = ((wv & 1) + 1) & 1;
#pragma unroll
            for (int blk = 0; blk < 6; ++blk) {
                const int j = blk * 16 + li, jc = j < 81 ? j : 80;
                const int hy = 2 * (jc / 9) + oy, hx = 2 * (jc % 9) + ox;
                const bool inb = j < 81 && (unsigned)(y0 + hy) < (unsigned)H && (unsigned)(x0 + hx) < (unsigned)W;
                const unsigned off = inb ? (unsigned)(base + srel[blk]) : OOB;
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, off, 0, 0);
                sr[blk] = *reinterpret_cast<const float4 *>(&v);
            }
        }
    };
    auto issue_low = [&](const Pos &p) {        // low-resolution rows 8 ty - 1 .. 8 ty + 8 under the 18 halo rows
        const int ly0 = p.ty * (TH / 2) - 1, lx0 = p.tx * (TW / 2) - 1;
        const int base = (((p.n * Hl + ly0) * Wl + lx0) * UP_CIN) * 4;
        if (interior(p)) {
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(lrsrc, lrel[sl], base, 0);
                lr[sl] = *reinterpret_cast<const float4 *>(&v);
            }
        } else {
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
                const int idx = tid + sl * 256, lp = idx >> 3;
                const int ly = lp / UP_W, lx = lp % UP_W;
                const bool inb = idx < UP_W * UP_W * 8 && (unsigned)(ly0 + ly) < (unsigned)Hl && (unsigned)(lx0 + lx) < (unsigned)Wl;
                const unsigned off = inb ? (unsigned)(base + lrel[sl]) : OOB;
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(lrsrc, off, 0, 0);
                lr[sl] = *reinterpret_cast<const float4 *>(&v);
            }
        }
    };
    float *lw = xl + (tid >> 3) * UP_PS + ((tid & 7) >> 2) * 16 + (tid & 3);    // channel 4 q + j -> [half q / 4][kk = j][s = q % 4]
    auto commit_low = [&]() {
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
            if (sl < 3 || tid < UP_W * UP_W * 8 - 768) {
                float *d = lw + sl * 32 * UP_PS;
                d[0] = lr[sl].x; d[4] = lr[sl].y; d[8] = lr[sl].z; d[12] = lr[sl].w;
            }
        }
    };
    auto up_conv = [&](const Pos &p) {
        const bool inner = interior(p);
#pragma unroll
        for (int bp = 0; bp < 6; bp += 2) {                          // two class blocks at a time: two independent MFMA chains
            float4 bl[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float *src = xl + (upk[bp + t] & 0xFFFFu);
                bl[t][0] = *reinterpret_cast<const float4 *>(src);
                bl[t][1] = *reinterpret_cast<const float4 *>(src + 16);
            }
            f32x4 c1[2];
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const float4 q = bl[t][s8 >> 2];
                    const float b = (s8 & 3) == 0 ? q.x : ((s8 & 3) == 1 ? q.y : ((s8 & 3) == 2 ? q.z : q.w));
                    c1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[s8], b, s8 == 0 ? (f32x4){0.f, 0.f, 0.f, 0.f} : c1[t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int blk = bp + t;
                const float u[4] = {c1[t][0] + upb.x, c1[t][1] + upb.y, c1[t][2] + upb.z, c1[t][3] + upb.w};
                const float sk[4] = {sr[blk].x, sr[blk].y, sr[blk].z, sr[blk].w};
                float m[4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    m[e] = BRIDGE == SQ_BRIDGE_ADD ? u[e] + sk[e]
                         : (BRIDGE == SQ_BRIDGE_MUL ? u[e] * sk[e] : (BRIDGE == SQ_BRIDGE_SUB ? u[e] - sk[e] : u[e]));
                if (!inner) {                                        // outside the image = the conv's zero padding
                    const int j = blk * 16 + li, jc = j < 81 ? j : 80;
                    const int hy = 2 * (jc / 9) + (((wv >> 1) + 1) & 1), hx = 2 * (jc % 9) + (((wv & 1) + 1) & 1);
                    const int gy = p.ty * TH - 1 + hy, gx = p.tx * TW - 1 + hx;
                    const bool inside = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
#pragma unroll
                    for (int e = 0; e < 4; ++e) m[e] = inside ? m[e] : 0.f;
                }
                row_transpose(m[0], m[1], m[2], m[3]);
                if ((upk[blk] >> 16) != 0xFFFFu) *reinterpret_cast<float4 *>(xs + (upk[blk] >> 16)) = make_float4(m[0], m[1], m[2], m[3]);
            }
        }
    };

    // ---- the 3x3 convolution of one tile: the wave walks halo rows h = 4 wv .. 4 wv + 5; row h feeds tap row ky of
    // output row r = h - ky.  Per output row the order is ky, kx, channel: the chain of the oracle ------------------
    f32x4 acc[4][NB];
    const float *xb = xs + ((4 * wv) * HWD + li) * PS + 4 * kk;
    auto mfma_phase = [&]() {
        float4 bq[2][3];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) bq[0][kx] = *reinterpret_cast<const float4 *>(xb + kx * PS);
#if SQ_L0_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll
        for (int h = 0; h < 6; ++h) {
            if (h + 1 < 6) {
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
                    bq[(h + 1) & 1][kx] = *reinterpret_cast<const float4 *>(xb + ((h + 1) * HWD + kx) * PS);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float4 q = bq[h & 1][kx];
                const float b[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
#pragma unroll
                    for (int ky = 2; ky >= 0; --ky) {                // oldest output row first
                        const int r = h - ky;
                        if (r < 0 || r > 3) continue;
                        const bool first = ky == 0 && kx == 0 && s == 0;
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            acc[r][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ky * 3 + kx][s][nb], b[s],
                                                                              first ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[r][nb], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#if SQ_L0_SETPRIO
        __builtin_amdgcn_s_setprio(3);
#endif
    };

    // ---- epilogues ---------------------------------------------------------------------------------------------------
    const int yvoff = (((4 * wv) * W + li) * CO + 4 * kk) * 4;
    const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc(
        a.epi.pooled, 0, EPI == EPI_POOL ? (int)((size_t)a.N * Hl * Wl * 16 * 4) : 0, 0x00020000);
    const int pvoff = (((2 * wv) * Wl + (li >> 1)) * 16 + 4 * kk) * 4;
    // head (16 -> 2 channels per pixel) on v_mfma_f32_4x4x1_16b_f32: 16 blocks of (4 outputs x 4 pixels), one channel
    // per instruction = one fmaf per output, 2 passes instead of the 8 of a 16x16x4 whose 14 other rows would be zeros.
    // Block b = lanes 4b .. 4b+3: lane l supplies B = channel c of ITS pixel and A = head_w[c][l % 4] (0 for l % 4 >= 2),
    // and ends with D[output i][its pixel] in register i.  The A values are a 4 x 16 table in LDS, read once per tile.
    float *hws = smem + XS_FLOATS;
    float hb0 = 0.f, hb1 = 0.f;
    const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc(
        a.epi.logits, 0, EPI == EPI_HEAD ? (int)((size_t)a.N * H * W * 2 * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t mrsrc = __builtin_amdgcn_make_buffer_rsrc(
        a.epi.mask, 0, (EPI == EPI_HEAD && a.epi.mask) ? (int)((size_t)a.N * H * W) : 0, 0x00020000);
    const int gvoff = ((4 * wv + kk) * W + li) * 8, mvoff = (4 * wv + kk) * W + li;
    if constexpr (EPI == EPI_HEAD) {
        static_assert(MODE == M_PLAIN, "the head table sits right behind the halo image");
        if (tid < 64) hws[tid] = (tid >> 4) < 2 ? a.epi.head_w[(tid & 15) * 2 + (tid >> 4)] : 0.f;   // [l % 4][channel]; visible after the prologue's barrier
        if (a.epi.head_b) { hb0 = a.epi.head_b[0]; hb1 = a.epi.head_b[1]; }
    }
    auto epilogue = [&](const Pos &p) {
        if constexpr (NB > 1) {                  // 32 output channels: two 16-byte stores per pixel row and lane
            const int tbase = (((p.n * H + p.ty * TH) * W + p.tx * TW) * CO) * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const f32x4 t = (f32x4){relu(acc[r][nb][0] + bv[nb].x), relu(acc[r][nb][1] + bv[nb].y),
                                            relu(acc[r][nb][2] + bv[nb].z), relu(acc[r][nb][3] + bv[nb].w)};
                    __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4 *>(&t), yrsrc,
                                                           yvoff + tbase + r * W * CO * 4 + nb * 64, 0, 0);
                }
            return;
        }
        float o[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            o[r][0] = relu(acc[r][0][0] + bv[0].x); o[r][1] = relu(acc[r][0][1] + bv[0].y);
            o[r][2] = relu(acc[r][0][2] + bv[0].z); o[r][3] = relu(acc[r][0][3] + bv[0].w);
        }
        if constexpr (EPI != EPI_HEAD) {
            const int tbase = (((p.n * H + p.ty * TH) * W + p.tx * TW) * 16) * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const f32x4 t = (f32x4){o[r][0], o[r][1], o[r][2], o[r][3]};
                // the tile offset goes into the VGPR offset, not the SGPR one: a 16-byte store with an SGPR offset whose data
                // registers the next VALU instruction overwrites stored the NEW value now and then (lanes 12-15 of a row, a
                // few pixels in 10^4, nondeterministic) -- hipcc guards that hazard only for the immediate-offset form
                __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4 *>(&t), yrsrc, yvoff + tbase + r * W * 64, 0, 0);
            }
        }
        if constexpr (EPI == EPI_POOL) {        // rows (0,1) and (2,3) in registers, the x neighbour is lane ^ 1
            const int pbase = (((p.n * Hl + p.ty * (TH / 2)) * Wl + p.tx * (TW / 2)) * 16) * 4;
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                f32x4 mp;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = __builtin_fmaxf(o[2 * pr][j], o[2 * pr + 1][j]);
                    const float u = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
                    mp[j] = __builtin_fmaxf(u, v);
                }
                __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4 *>(&mp), prsrc,
                                                       (li & 1) ? (int)OOB : pvoff + pbase + pr * Wl * 64, 0, 0);
            }
        }
        if constexpr (EPI == EPI_HEAD) {
#pragma unroll
            for (int j = 0; j < 4; ++j) row_transpose(o[0][j], o[1][j], o[2][j], o[3][j]);
            // lane row g now holds the 16 channels of pixel (row g, column li): o[r][j] = channel 4 r + j
            float4 hq[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) hq[r] = *reinterpret_cast<const float4 *>(hws + (lane & 3) * 16 + 4 * r);
            f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float hv[4] = {hq[r].x, hq[r].y, hq[r].z, hq[r].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) z = __builtin_amdgcn_mfma_f32_4x4x1f32(hv[j], o[r][j], z, 0, 0, 0);
            }
            const float z0 = z[0], z1 = z[1];
            const float v0 = a.epi.head_b ? z0 + hb0 : z0, v1 = a.epi.head_b ? z1 + hb1 : z1;
            const int pbase = (p.n * H + p.ty * TH) * W + p.tx * TW;
            const float lg[2] = {v0, v1};
            __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const u32x2 *>(lg), grsrc, gvoff, pbase * 8, 0);
            __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(v1 > v0 ? 1 : 0), mrsrc, mvoff, pbase, 0);
        }
    };

    // ---- the tile loop -----------------------------------------------------------------------------------------------
    Pos cur = decode(vb), nxt = cur, nn = cur;
    if constexpr (MODE == M_PLAIN) {
        issue_halo(cur);
        __builtin_amdgcn_s_waitcnt(0x0F70);                         // vmcnt(0)
        commit_halo();
        __syncthreads();
        for (int it = 0; it < t_count; ++it) {
            const bool has_next = it + 1 < t_count;
            if (has_next) { nxt = advance(cur); issue_halo(nxt); }
            mfma_phase();
            __builtin_amdgcn_s_waitcnt(0x0F70);                     // the prefetch has landed; stated outside the branch (v2 kernel, main loop)
            if (has_next) {
                __syncthreads();                                    // every wave is done reading this tile's halo image
                commit_halo();
            }
            epilogue(cur);
            if (has_next) __syncthreads();
            cur = nxt;
        }
    }
    if constexpr (MODE == M_FIRST) {
        issue_patch(cur);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        commit_patch(0);
        __syncthreads();
        first_conv(cur, 0);
        if (t_count > 1) {
            nxt = advance(cur);
            issue_patch(nxt);
            __builtin_amdgcn_s_waitcnt(0x0F70);
            commit_patch(1);
        }
        __syncthreads();
        for (int it = 0; it < t_count; ++it) {
            const bool has_next = it + 1 < t_count, has_next2 = it + 2 < t_count;
            if (has_next2) { nn = advance(nxt); issue_patch(nn); }   // two tiles ahead
            mfma_phase();
            __builtin_amdgcn_s_waitcnt(0x0F70);
            if (has_next) {
                __syncthreads();                                    // halo image free; patch (it + 1) & 1 complete
                first_conv(nxt, (it + 1) & 1);
            }
            if (has_next2) commit_patch(it & 1);                    // last read by first_conv of THIS tile, one barrier ago
            epilogue(cur);
            if (has_next) __syncthreads();
            cur = nxt;
            nxt = nn;
        }
    }
    if constexpr (MODE == M_UP) {
        issue_low(cur);
        issue_skip(cur);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        commit_low();
        if (t_count > 1) { nxt = advance(cur); issue_low(nxt); }
        __syncthreads();
        up_conv(cur);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        for (int it = 0; it < t_count; ++it) {
            const bool has_next = it + 1 < t_count, has_next2 = it + 2 < t_count;
            if (has_next) { commit_low(); issue_skip(nxt); }         // the patch fetched an iteration ago; its last readers passed a barrier
            if (has_next2) { nn = advance(nxt); issue_low(nn); }
            mfma_phase();
            __builtin_amdgcn_s_waitcnt(0x0F70);
            if (has_next) {
                __syncthreads();                                    // halo image free, low-resolution patch complete
                up_conv(nxt);
            }
            epilogue(cur);
            if (has_next) __syncthreads();
            cur = nxt;
            nxt = nn;
        }
    }
}

inline bool l0_enabled() {                                          // SQ_CONV_L0=0: A/B switch back to the generic kernel
    const char *e = getenv("SQ_CONV_L0");                           // read per launch: the parity tests flip it in-process
    return !(e && e[0] == '0');
}

template <int MODE, int EPI, int BRIDGE = 0, int NB = 1>
int launch_l0(const L0Args &a0, hipStream_t st) {
    static bool attr_set = false;
    auto kern = conv_l0_kernel<MODE, EPI, BRIDGE, NB>;
    constexpr int lds = (XS_FLOATS + (MODE == M_FIRST ? 2 * IN_FLOATS : (MODE == M_UP ? UP_FLOATS : 0)) + (EPI == EPI_HEAD ? 64 : 0)) * 4;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            sq_set_error("conv_l0: cannot reserve %d bytes of LDS", lds);
            return SQ_ELAUNCH;
        }
        attr_set = true;
    }
    L0Args a = a0;
    a.tiles_x = a.W / TW;
    a.tiles_y = a.H / TH;
    a.ntiles = a.tiles_x * a.tiles_y * a.N;
    const int want = 256 * (MODE == M_UP ? SQ_L0_UP_OCC : (NB == 2 ? 2 : SQ_L0_OCC));
    const int G = a.ntiles < want ? a.ntiles : want;
    const int per_image = a.tiles_x * a.tiles_y;
    a.gn = G / per_image;
    a.gy = (G % per_image) / a.tiles_x;
    a.gx = (G % per_image) % a.tiles_x;
    hipLaunchKernelGGL(kern, dim3(G), dim3(256), lds, st, a);
    return sq_check_launch("conv_l0");
}

}  // namespace

int sq_conv_l0_launch(int mode, const float *x, const float *w, const float *bias, float *y, int N, int H, int W,
                      int cout, int act, const SqConvEpi &epi, hipStream_t st) {
    if (cout != 16 && !(cout == 32 && mode == M_PLAIN && !epi.head_w && !epi.pooled)) return SQ_L0_NOT_MINE;
    if (!l0_enabled() || act != SQ_ACT_RELU || H % 16 != 0 || W % 16 != 0 || epi.x2) return SQ_L0_NOT_MINE;
    if (epi.head_w && epi.head_c != 2) return SQ_L0_NOT_MINE;
    if (epi.head_w && epi.pooled) return SQ_L0_NOT_MINE;
    L0Args a = {};
    a.x = x; a.w = w; a.bias = bias; a.y = y; a.N = N; a.H = H; a.W = W; a.epi = epi;
    const int e = epi.head_w ? EPI_HEAD : (epi.pooled ? EPI_POOL : EPI_STORE);
    if (mode == M_PLAIN) {
        if (e == EPI_HEAD) return launch_l0<M_PLAIN, EPI_HEAD>(a, st);
        if (e == EPI_POOL) return launch_l0<M_PLAIN, EPI_POOL>(a, st);
        if (cout == 32) return launch_l0<M_PLAIN, EPI_STORE, 0, 2>(a, st);
        return launch_l0<M_PLAIN, EPI_STORE>(a, st);
    }
    if (mode == M_FIRST) {
        if (e == EPI_HEAD) return SQ_L0_NOT_MINE;
        if (e == EPI_POOL) return launch_l0<M_FIRST, EPI_POOL>(a, st);
        return launch_l0<M_FIRST, EPI_STORE>(a, st);
    }
    if (mode == M_UP) {
        if (e != EPI_STORE) return SQ_L0_NOT_MINE;
        switch (epi.up_bridge) {
            case SQ_BRIDGE_ADD: return launch_l0<M_UP, EPI_STORE, SQ_BRIDGE_ADD>(a, st);
            case SQ_BRIDGE_MUL: return launch_l0<M_UP, EPI_STORE, SQ_BRIDGE_MUL>(a, st);
            case SQ_BRIDGE_SUB: return launch_l0<M_UP, EPI_STORE, SQ_BRIDGE_SUB>(a, st);
            default: return launch_l0<M_UP, EPI_STORE, SQ_BRIDGE_NONE>(a, st);
        }
    }
    return SQ_L0_NOT_MINE;
}
