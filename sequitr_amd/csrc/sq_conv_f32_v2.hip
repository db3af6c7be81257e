// Pipelined f32 implicit-GEMM convolution for gfx950 -- same arithmetic and reduction order
// as conv_mfma_f32_kernel in sq_conv_f32.hip (bit-identical results), restructured so the
// matrix pipe never waits on HBM/L2:
//   * persistent blocks: each block walks a contiguous run of 16x16 tiles; for Cin == 16
//     (one chunk) the weight slab is staged ONCE per block instead of once per tile;
//   * T14 split staging: the global loads of work item i+1 (next 16-channel chunk, or the
//     next tile's halo) are issued into registers BEFORE the MFMA phase of item i and are
//     written to LDS after it, so their latency hides under ~4.6-18k cycles of MFMA;
//   * operand fragments are double-buffered in registers: the ds_reads of MFMA step s+1 are
//     issued before the MFMAs of step s (the compiler otherwise sinks them to just-in-time).
// Fused variants (all bit-identical to the unfused sequence of kernels):
//   * epilogue also writes the 2x2 max-pooled activation (conv_block + max_pool_layer,
//     sequitr/networks/unet.py:241-243): saves re-reading the full-resolution tensor;
//   * epilogue applies the 1x1 to_image head + argmax (unet.py:252-253) and does not store
//     the 16-channel activation at all: saves a 512 MiB write + read per 32-tile batch;
//   * FIRST: the 1 -> 16 channel first convolution of down0 (unet.py:238, conv_block conv1)
//     is evaluated on the fly for the 18x18 halo of each tile (VALU, hidden under other waves'
//     MFMAs) and never touches HBM: the level-0 block reads 4 B/pixel instead of 64.
//   * UP: conv_transpose_layer + bridge of up0 (unet.py:312-319) are evaluated on the fly for the halo of the
//     following conv: the block loads the SKIP halo (the usual prefetch) plus a 10x10x32 low-resolution patch,
//     wave w computes parity class w of the 2x2/s2 transpose conv on the matrix cores (A = its 16x32 slice of
//     the kernel, in registers) and merges it into the halo image in LDS.  The up-scaled / merged level-0
//     tensor (537 MB per 32-tile batch) is never written or re-read and the HBM-bound convT launch disappears.
#include <stdlib.h>
#include <type_traits>
#include "sq_common.h"
#include "sq_conv_epi.h"

#ifndef SQ_V2_FAST32
#define SQ_V2_FAST32 1           // the ReLU / full-tile epilogue copy also in the <32,3,32> form (1 spilled register; A/B switch)
#endif
#ifndef SQ_V2_INNER
#define SQ_V2_INNER 1            // 64-channel blocks: SGPR tile offsets for interior halos and for the weight slab (A/B switch)
#endif
#ifndef SQ_TILE_INTERLEAVE
#define SQ_TILE_INTERLEAVE 1    // block b takes tiles b, b+G, b+2G, ...; 0: a contiguous run per block (A/B switch, HISTORY.md 4a)
#endif
#ifndef SQ_STORE_PERMUTE
#define SQ_STORE_PERMUTE 0      // 1: lane-contiguous stores through ds_bpermute (A/B switch: no measurable difference, HISTORY.md 4a)
#endif


namespace {

constexpr int TH = 16, TW = 16;

template <int BN, int KS, int KC>
struct Cfg2 {
    static constexpr int HALO_W = TW + KS - 1;
    static constexpr int HALO_H = TH + KS - 1;
    static constexpr int HP = HALO_W * HALO_H;
    static constexpr int PS = KC + 2;                       // pixel stride (floats): conflict-free B reads
    // weight row stride: conflict-free A reads.  KC <= 16: rows of BN % 32 == 0 floats are padded by 16, so consecutive
    // rows (= lane groups kk, kk + 1 of one ds_read_b32) alternate bank halves.  KC == 32 (a 32-channel stage, see
    // load_frag) has no room for the padding: odd rows are stored with their two 16-float halves swapped (col ^ 16).
    static constexpr int WSWZ = (KC == 32 && BN == 32) ? 16 : 0;
    static constexpr int BNS = WSWZ ? BN : ((BN % 32 == 0) ? BN + 16 : BN);
    static constexpr int WROWS = KS * KS * KC;
    static constexpr int XS_FLOATS = HP * PS;
    static constexpr int WS_FLOATS = WROWS * BNS;
    static constexpr int IN_W = TW + 4;                     // FIRST: 20x20 single-channel input patch
    static constexpr int IN_FLOATS = IN_W * IN_W;           // 400 (16-B multiple)
    static constexpr int HEAD_PS = 18;                      // head scratch: [16 pixels][16 ch + 2]: conflict-free reads
    static constexpr int HEAD_FLOATS = 4 * 16 * HEAD_PS;    // one 16x16 image per wave
    static constexpr int UP_CIN = 32;                       // UP: channels of the low-resolution input
    static constexpr int UP_W = TW / 2 + 2;                 // UP: 10x10 low-resolution patch under the 18x18 halo
    static constexpr int UP_PS = UP_CIN + 2;                // its pixel stride: conflict-free B reads
    static constexpr int UP_FLOATS = UP_W * UP_W * UP_PS;   // 3400 floats
    static constexpr int LDS_BYTES = (XS_FLOATS + WS_FLOATS) * 4;
    static constexpr int LDS_BYTES_FIRST = (XS_FLOATS + WS_FLOATS + IN_FLOATS) * 4;
    static constexpr int LDS_BYTES_UP = (XS_FLOATS + WS_FLOATS + UP_FLOATS) * 4;
    static constexpr int QPP = KC / 4;                      // float4 per halo pixel
    static constexpr int XITEMS = HP * QPP;
    static constexpr int XSLOTS = (XITEMS + 255) / 256;
    static constexpr int WITEMS = WROWS * (BN / 4);
    static constexpr int WSLOTS = (WITEMS + 255) / 256;
    static constexpr int NSTEP = KS * KS * (KC / 4);
    static constexpr int OCC = (BN >= 64 || KC == 32) ? 2 : (BN >= 32 ? 3 : 4);  // blocks per CU (LDS-limited)
    static_assert(KC <= 16 || (KC == 32 && BN == 32), "the 32-channel stage exists for the 32 -> 32 layers (LDS: 2 blocks per CU)");
    static_assert((XS_FLOATS * 4) % 16 == 0, "weight slab must start 16-B aligned");
    static_assert((WS_FLOATS * 4) % 16 == 0, "input patch must start 16-B aligned");
};

template <int BN, int KS, int KC, int MODE>            // MODE 0 plain, 1 FIRST, 2 UP
__global__ __launch_bounds__(256, (MODE == 2 ? 3 : Cfg2<BN, KS, KC>::OCC)) void conv_mfma_f32_v2_kernel(
    const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
    float *__restrict__ y, int N, int H, int W, int Cin, int Cout, float wscale, int act,
    int tiles_x, int tiles_y, int ntiles, int tiles_per_block, SqConvEpi epi) {
    using C = Cfg2<BN, KS, KC>;
    constexpr bool FIRST = MODE == 1, UP = MODE == 2;
    constexpr int NR = BN / 16;
    constexpr int PAD = KS / 2;
    static_assert(MODE == 0 || (BN == 16 && KS == 3 && KC == 16), "FIRST / UP are level-0 blocks (16 channels)");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *xs = smem;
    float *ws = smem + C::XS_FLOATS;
    float *xin = ws + C::WS_FLOATS;                         // FIRST only
    float *xl = ws + C::WS_FLOATS;                          // UP only: low-resolution patch [100 px][34]
    float *head_scratch = ws + C::WS_FLOATS + (FIRST ? C::IN_FLOATS : (UP ? C::UP_FLOATS : 0));   // only when epi.head_w

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    __builtin_assume(tid < 256);                               // lets the `idx < ITEMS` guards of the full staging slots fold away
    const int li = lane & 15, kk = lane >> 4;
    const int n0 = blockIdx.y * BN;
    const int vb = (int)sq_xcd_remap(blockIdx.x, gridDim.x);
#if SQ_TILE_INTERLEAVE
    // block b takes tiles b, b + G, b + 2G, ...: the G tiles in flight at any moment are a contiguous run of the
    // image (whole rows of DRAM pages, shared halos resident in the same L2) instead of G scattered tile rows.
    // Measured: nothing on a single plain layer, -0.4 % on the whole inference step (the UP launch, whose halos the
    // L2 did not absorb: 1079 MB read against 805 MB of input).
    const int tstride = (int)gridDim.x;
    const int t_begin = vb;
    if (t_begin >= ntiles) return;
    const int t_count = (ntiles - vb + tstride - 1) / tstride;
#else
    const int tstride = 1;
    const int t_begin = vb * tiles_per_block;
    const int t_end = min(t_begin + tiles_per_block, ntiles);
    if (t_begin >= t_end) return;
    const int t_count = t_end - t_begin;
#endif
    const int nchunk = FIRST ? 1 : Cin / KC;
    const int nitems = t_count * nchunk;
    const bool restage_w = nchunk > 1;

    float4 xr[C::XSLOTS];
    float4 wr[C::WSLOTS];
    float inr[2];                                           // FIRST: 400 input pixels over 256 threads
    float4 lr[4];                                           // UP: 100 px x 8 float4 of the low-res patch over 256 threads
    float4 sr[6];                                           // UP: the skip halo in the transpose conv's own fragment layout

    // Buffer resources: out-of-range offsets read as 0 / drop the store, so image borders,
    // ragged tiles and the "idx >= items" tail need no branches (hipcc otherwise wraps every
    // predicated load in its own s_cbranch_execz block and the loads stop overlapping).
    // All descriptor inputs are kernel arguments => provably wave-uniform (no waterfall loop).
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(x), 0, (int)((size_t)N * H * W * (FIRST ? 1 : (epi.x2 ? Cin / 2 : Cin)) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(w), 0, KS * KS * (FIRST ? 16 : Cin) * Cout * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(
        y, 0, y ? (int)((size_t)N * H * W * Cout * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t lrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(epi.up_x), 0, UP ? (int)((size_t)N * (H / 2) * (W / 2) * C::UP_CIN * 4) : 0, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;   // tensors are < 2 GiB (checked on the host)
    const int CinW = FIRST ? 16 : Cin;      // input channels of the MFMA convolution
    // two-source input (concat bridge): both tensors carry Cs = Cin / 2 channels; a 16-channel chunk lies wholly in one
    const int Cs = epi.x2 ? Cin / 2 : Cin;
    const __amdgpu_buffer_rsrc_t x2rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(epi.x2), 0, epi.x2 ? (int)((size_t)N * H * W * Cs * 4) : 0, 0x00020000);

    // per-thread, tile-independent part of the halo addresses
    int xrel[C::XSLOTS], xpy[C::XSLOTS], xpx[C::XSLOTS];
#pragma unroll
    for (int sl = 0; sl < C::XSLOTS; ++sl) {
        const int idx = tid + sl * 256;
        const int pix = idx / C::QPP, q = idx % C::QPP;
        xpy[sl] = pix / C::HALO_W;
        xpx[sl] = pix % C::HALO_W;
        xrel[sl] = idx < C::XITEMS ? ((xpy[sl] * W + xpx[sl]) * Cs + q * 4) * 4 : (int)OOB;
    }
    int wrel[C::WSLOTS];
#pragma unroll
    for (int sl = 0; sl < C::WSLOTS; ++sl) {
        const int idx = tid + sl * 256;
        const int r = idx / (BN / 4), q4 = idx % (BN / 4);
        const int tap = r / KC, c = r % KC;
        const int co = n0 + q4 * 4;
        wrel[sl] = (idx < C::WITEMS && co < Cout) ? ((tap * CinW + c) * Cout + co) * 4 : (int)OOB;
    }

    // ---- issue the global loads of one work item (tile, chunk) into registers -------------
    // live == false: every lane's offset is out of range, nothing is fetched (the item past the last one -- issue()
    // and commit() run unconditionally every iteration, see the main loop)
    auto issue = [&](int tx, int ty, int n, int cc, bool want_w, bool live) {
        if constexpr (FIRST) {
            // 20x20 single-channel patch around the tile (halo of the halo)
            const int x0 = tx * TW - 2, y0 = ty * TH - 2;
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                const int idx = tid + sl * 256;
                const int py = idx / C::IN_W, px = idx % C::IN_W;
                const bool inb = live && idx < C::IN_FLOATS && (unsigned)(y0 + py) < (unsigned)H &&
                                 (unsigned)(x0 + px) < (unsigned)W;
                const unsigned off = inb ? (unsigned)((((n * H + y0 + py) * W) + x0 + px) * 4) : OOB;
                inr[sl] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, off, 0, 0));
            }
        } else {
            const int x0 = tx * TW - PAD, y0 = ty * TH - PAD;
            const bool second = cc >= Cs;                                  // wave-uniform: which tensor holds this chunk
            const int base = (((n * H + y0) * W + x0) * Cs + (second ? cc - Cs : cc)) * 4;   // may be "negative": wraps back
            if constexpr (!UP) {
                const __amdgpu_buffer_rsrc_t rs = second ? x2rsrc : xrsrc;      // wave-uniform select (4 s_cselect)
                // a halo that lies wholly inside the image: the tile offset rides in the SGPR offset, the lane's part is
                // loop-invariant -- no bounds arithmetic (64-channel blocks: 57 VALU instructions per item otherwise)
                const bool inner = SQ_V2_INNER && (BN >= 32 || KC == 32) && live && x0 >= 0 && y0 >= 0 && x0 + C::HALO_W <= W && y0 + C::HALO_H <= H;
                if (inner) {
#pragma unroll
                    for (int sl = 0; sl < C::XSLOTS; ++sl) {
                        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, xrel[sl], base, 0);
                        xr[sl] = *reinterpret_cast<const float4 *>(&v);
                    }
                } else
#pragma unroll
                for (int sl = 0; sl < C::XSLOTS; ++sl) {
                    int hpy = xpy[sl], hpx = xpx[sl];
                    if constexpr (KC == 32) {                       // 11 slots: the halo coordinates are recomputed on the border tiles
                        int tv = tid;                               // instead of living in 22 registers (opaque: not hoisted back out)
                        asm volatile("" : "+v"(tv));
                        const int hp = (tv + sl * 256) / C::QPP;
                        hpy = hp / C::HALO_W;
                        hpx = hp - hpy * C::HALO_W;
                    }
                    const bool inb = live && (unsigned)(y0 + hpy) < (unsigned)H && (unsigned)(x0 + hpx) < (unsigned)W &&
                                     xrel[sl] != (int)OOB;
                    const unsigned off = inb ? (unsigned)(base + xrel[sl]) : OOB;
                    const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
                    xr[sl] = *reinterpret_cast<const float4 *>(&v);
                }
            } else {
                // UP: the skip halo is fetched in the fragment layout of the transpose conv it will be merged with
                // (wave = parity class, 6 blocks of 16 class pixels, lane (li, kk) = channels 4kk..4kk+3 of pixel li):
                // the bridge then runs in registers and the merged value is written to LDS once.  The in-place merge
                // (commit the skip halo, read it back, write the product) ran at 92 % LDS bank conflicts
                // (profiles/r01u_pmc_sq_summary.json): 544 -> 505 us per launch.
                const int oy = ((wv >> 1) + 1) & 1, ox = ((wv & 1) + 1) & 1;
#pragma unroll
                for (int blk = 0; blk < 6; ++blk) {
                    const int j = blk * 16 + li;
                    const int hy = 2 * (j / 9) + oy, hx = 2 * (j % 9) + ox;
                    const bool inb = live && j < 81 && (unsigned)(y0 + hy) < (unsigned)H && (unsigned)(x0 + hx) < (unsigned)W;
                    const unsigned off = inb ? (unsigned)(base + ((hy * W + hx) * Cin + 4 * kk) * 4) : OOB;
                    const auto v = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, off, 0, 0);
                    sr[blk] = *reinterpret_cast<const float4 *>(&v);
                }
            }
        }
        if constexpr (UP) {
            // low-resolution patch: rows 8 ty - 1 .. 8 ty + 8 (the 18 halo rows start at the odd row 16 ty - 1)
            const int ly0 = ty * (TH / 2) - 1, lx0 = tx * (TW / 2) - 1, Hl = H >> 1, Wl = W >> 1;
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
                const int idx = tid + sl * 256, lp = idx >> 3, q = idx & 7;
                const int ly = lp / C::UP_W, lx = lp % C::UP_W;
                const bool inb = live && idx < C::UP_W * C::UP_W * 8 && (unsigned)(ly0 + ly) < (unsigned)Hl &&
                                 (unsigned)(lx0 + lx) < (unsigned)Wl;
                const unsigned off = inb ? (unsigned)((((n * Hl + ly0 + ly) * Wl + lx0 + lx) * C::UP_CIN + q * 4) * 4) : OOB;
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(lrsrc, off, 0, 0);
                lr[sl] = *reinterpret_cast<const float4 *>(&v);
            }
        }
        if (want_w) {
            const int wbase = cc * Cout * 4;
            if (SQ_V2_INNER && (BN >= 32 || KC == 32) && live) {   // out-of-range slots keep their out-of-range VGPR offset
#pragma unroll
                for (int sl = 0; sl < C::WSLOTS; ++sl) {
                    // KC == 32: slot sl is tap sl (256 threads = 32 rows x 8 quads): one lane offset + a scalar per slot
                    const auto v = KC == 32 ? __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wrel[0], wbase + sl * CinW * Cout * 4, 0)
                                            : __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wrel[sl], wbase, 0);
                    wr[sl] = *reinterpret_cast<const float4 *>(&v);
                }
            } else
#pragma unroll
            for (int sl = 0; sl < C::WSLOTS; ++sl) {
                const unsigned off = (live && wrel[sl] != (int)OOB) ? (unsigned)(wbase + wrel[sl]) : OOB;
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, off, 0, 0);
                wr[sl] = *reinterpret_cast<const float4 *>(&v);
            }
        }
    };
    // ---- registers -> LDS (same index map) ----------------------------------------------------
    auto commit = [&](bool want_w) {
        if constexpr (FIRST) {
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                const int idx = tid + sl * 256;
                if (idx < C::IN_FLOATS) xin[idx] = inr[sl];
            }
        } else if constexpr (!UP) {
#pragma unroll
            for (int sl = 0; sl < C::XSLOTS; ++sl) {
                const int idx = tid + sl * 256;
                if (idx < C::XITEMS) {
                    const int pix = idx / C::QPP, q = idx % C::QPP;
                    float *d = xs + pix * C::PS + q * 4;
                    *reinterpret_cast<float2 *>(d) = make_float2(xr[sl].x, xr[sl].y);
                    *reinterpret_cast<float2 *>(d + 2) = make_float2(xr[sl].z, xr[sl].w);
                }
            }
        }
        if constexpr (UP) {
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
                const int idx = tid + sl * 256, lp = idx >> 3, q = idx & 7;
                if (idx < C::UP_W * C::UP_W * 8) {
                    float *d = xl + lp * C::UP_PS + q * 4;
                    *reinterpret_cast<float2 *>(d) = make_float2(lr[sl].x, lr[sl].y);
                    *reinterpret_cast<float2 *>(d + 2) = make_float2(lr[sl].z, lr[sl].w);
                }
            }
        }
        if (want_w) {
            // x * 1.0f == x: the U-Net's layers skip the scaling (as a branch round the loop: inside it hipcc multiplies
            // and then selects, six instructions per slot)
            auto put = [&](auto scaledc) {
#pragma unroll
                for (int sl = 0; sl < C::WSLOTS; ++sl) {
                    const int idx = tid + sl * 256;
                    if (idx < C::WITEMS) {
                        const int r = idx / (BN / 4), q4 = idx % (BN / 4);
                        float4 v = wr[sl];
                        if constexpr (decltype(scaledc)::value) { v.x *= wscale; v.y *= wscale; v.z *= wscale; v.w *= wscale; }
                        *reinterpret_cast<float4 *>(ws + r * C::BNS + ((q4 * 4) ^ ((r & 1) ? C::WSWZ : 0))) = v;
                    }
                }
            };
            if (wscale == 1.0f) put(std::integral_constant<bool, false>{});
            else put(std::integral_constant<bool, true>{});
        }
    };
    // ---- FIRST: conv1 (3x3, 1 -> 16, bias, ReLU) of the 18x18 halo, straight into the halo image,
    // also on the matrix cores: the 9 taps are the reduction (3 MFMA steps of 4, taps 9..11 carry zero
    // weights, so the chain is exactly "acc = 0; 9 taps in raster order"), 16 output channels are the
    // rows and 16 halo pixels the columns.  21 column blocks cover the 324 halo pixels (6/5/5/5 per
    // wave).  Halo pixels outside the image are conv2's ZERO PADDING, not conv1 evaluated out there.
    float a1[3] = {0.f, 0.f, 0.f};
    int toff[3] = {0, 0, 0};
    float4 b1v = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (FIRST) {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int tap = 4 * s + kk;
            a1[s] = tap < 9 ? epi.first_w[tap * 16 + li] : 0.f;
            const int tc = tap < 9 ? tap : 8;
            toff[s] = (tc / 3) * C::IN_W + tc % 3;
        }
        b1v = *reinterpret_cast<const float4 *>(epi.first_b + 4 * kk);
    }
    auto first_conv = [&](int tx, int ty) {
        for (int blk = wv; blk < (C::HP + 15) / 16; blk += 4) {
            const int pix = blk * 16 + li;
            const int pc = pix < C::HP ? pix : C::HP - 1;
            const int py = pc / C::HALO_W, px = pc % C::HALO_W;
            const float *src = xin + py * C::IN_W + px;
            f32x4 c1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 3; ++s) c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], src[toff[s]], c1, 0, 0, 0);
            const int gy = ty * TH - 1 + py, gx = tx * TW - 1 + px;
            const bool inside = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            float v0 = c1[0] + b1v.x, v1 = c1[1] + b1v.y, v2 = c1[2] + b1v.z, v3 = c1[3] + b1v.w;
            v0 = (inside && v0 > 0.f) ? v0 : 0.f;
            v1 = (inside && v1 > 0.f) ? v1 : 0.f;
            v2 = (inside && v2 > 0.f) ? v2 : 0.f;
            v3 = (inside && v3 > 0.f) ? v3 : 0.f;
            if (pix < C::HP) {
                float *d = xs + pix * C::PS + 4 * kk;
                *reinterpret_cast<float2 *>(d) = make_float2(v0, v1);
                *reinterpret_cast<float2 *>(d + 2) = make_float2(v2, v3);
            }
        }
    };

    // ---- UP: merged = bridge(convT2x2s2(up_x) + bias, skip) for the 18x18 halo, written straight into the halo image
    // (the skip operand is already in this lane's registers, sr[blk]).
    // Wave w owns parity class (a, b) = (w >> 1, w & 1) of the transpose conv: its 81 halo pixels in 6 column
    // blocks of 16; per block 8 MFMA steps over the 32 input channels (one fmaf chain c = 0..31 per output,
    // then + bias, then the bridge: exactly sq_convT2x2s2_nhwc_fwd_f32 / the oracle).  A = the class's 16x32
    // kernel slice, held in registers for the whole launch.
    float aw[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float4 upb = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (UP) {
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) aw[s8] = epi.up_w[(wv * 16 + li) * C::UP_CIN + 4 * s8 + kk];
        if (epi.up_b) upb = *reinterpret_cast<const float4 *>(epi.up_b + 4 * kk);
    }
    auto up_conv = [&](int tx, int ty) {
        const int pa = wv >> 1, pb = wv & 1;
        // halo row hy has parity (hy + 1) & 1 (global row 16 ty - 1 + hy): class rows are hy = 2 jy + ((pa + 1) & 1)
        const int oy = (pa + 1) & 1, ox = (pb + 1) & 1;
#pragma unroll
        for (int blk = 0; blk < 6; ++blk) {
            const int j = blk * 16 + li, jc = j < 81 ? j : 80;
            const int hy = 2 * (jc / 9) + oy, hx = 2 * (jc % 9) + ox;
            const float *src = xl + (((hy + 1) >> 1) * C::UP_W + ((hx + 1) >> 1)) * C::UP_PS + kk;
            f32x4 c1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[s8], src[4 * s8], c1, 0, 0, 0);
            const int gy = ty * TH - 1 + hy, gx = tx * TW - 1 + hx;
            const bool inside = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            if (j < 81) {
                float *d = xs + (hy * C::HALO_W + hx) * C::PS + 4 * kk;
                const float u[4] = {c1[0] + upb.x, c1[1] + upb.y, c1[2] + upb.z, c1[3] + upb.w};
                const float sk[4] = {sr[blk].x, sr[blk].y, sr[blk].z, sr[blk].w};
                float m[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = epi.up_bridge == SQ_BRIDGE_ADD ? u[e] + sk[e]
                                  : (epi.up_bridge == SQ_BRIDGE_MUL ? u[e] * sk[e]
                                  : (epi.up_bridge == SQ_BRIDGE_SUB ? u[e] - sk[e] : u[e]));
                    m[e] = inside ? v : 0.f;                // outside the image = the conv's zero padding
                }
                *reinterpret_cast<float2 *>(d) = make_float2(m[0], m[1]);
                *reinterpret_cast<float2 *>(d + 2) = make_float2(m[2], m[3]);
            }
        }
    };

    f32x4 acc[4][NR];
    const float *xb_lds = xs + ((4 * wv) * C::HALO_W + li) * C::PS + kk;
    const float *wa_lds = ws + kk * C::BNS + li;

    // step st of an item.  KC <= 16: taps in raster order, 4 channels per step -- the chain of DESIGN section 3.
    // KC == 32: ONE staged item holds two 16-channel chain chunks; the steps still walk chunk 0 (all taps) and then
    // chunk 1 (all taps), so the fmaf chain is unchanged -- only barriers, commits and the weight restaging per MFMA
    // are halved (a 32 -> 32 layer becomes a single-item-per-tile layer whose weights are staged once per block).
    auto load_frag = [&](int st, float (&a)[NR], float (&b)[4]) {
        constexpr int SPC = KS * KS * 4;                          // steps per 16-channel chain chunk
        const int sub = KC > 16 ? st / SPC : 0, sr = KC > 16 ? st % SPC : st;
        constexpr int Q = KC > 16 ? 4 : KC / 4;
        const int tap = sr / Q, s = sr % Q;
        const int ky = tap / KS, kx = tap % KS;
        const int ch = sub * 16 + s * 4;                          // first channel of the step inside the staged item
#pragma unroll
        for (int nb = 0; nb < NR; ++nb)                           // row parity = kk & 1 (tap * KC + ch is even)
            a[nb] = wa_lds[(tap * KC + ch) * C::BNS + ((nb * 16) ^ ((kk & 1) ? C::WSWZ : 0))];
#pragma unroll
        for (int r = 0; r < 4; ++r) b[r] = xb_lds[((r + ky) * C::HALO_W + kx) * C::PS + ch];
    };

    // activation as two selects (no per-element scalar branches in the store tail):
    //   v > 0 ? v : (relu ? +0 : v * slope),  slope = 1 (none) or 0.2 (leaky)
    const float slope = act == SQ_ACT_LEAKY ? 0.2f : 1.0f;
    const bool is_relu = act == SQ_ACT_RELU;
    // FAST: ReLU on a tile that lies wholly inside the tensor -- one v_max per value and lane-part + tile-part store
    // offsets instead of two selects per value and a bounds test per store (about 500 of a 64-channel tile's 800 VALU
    // instructions; profiles/r04_l0_issue_model.txt on what those cost)
    auto actk = [&](float v, auto fastc) {
        if constexpr (decltype(fastc)::value) return __builtin_fmaxf(v, 0.0f);       // NaN -> 0, -0 -> +0, as v > 0 ? v : 0
        else {
            const float neg = is_relu ? 0.0f : v * slope;
            return v > 0.0f ? v : neg;
        }
    };

    // fused head operands, once per block: A[i = head output][k = channel] = head_w[channel][output]
    float ah[4] = {0.f, 0.f, 0.f, 0.f};
    float hb[4] = {0.f, 0.f, 0.f, 0.f};
    if (epi.head_w) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) ah[s4] = li < epi.head_c ? epi.head_w[(4 * s4 + kk) * epi.head_c + li] : 0.f;
        if (epi.head_b) {
#pragma unroll
            for (int q = 0; q < 4; ++q) hb[q] = q < epi.head_c ? epi.head_b[q] : 0.f;
        }
    }

    // SQ_STORE_PERMUTE: lane-contiguous output stores.  The MFMA leaves lane 16*q + p with the channel quad q of pixel
    // p, so a b128 store in that order touches four different 64-byte segments per 4 consecutive lanes (16 bytes of
    // each).  ds_bpermute (cross-lane, no LDS memory) can hand lane 4*p + q that quad instead: every 4 lanes = one
    // contiguous 64-byte segment.  Measured: no difference at any level -- the TA coalesces the instruction's 1 KiB either way.
#if SQ_STORE_PERMUTE
    const int st_src = (((lane & 3) << 4) | (lane >> 2)) << 2;   // byte address of the source lane for ds_bpermute
    const int st_px = lane >> 2, st_q = lane & 3;
#endif
    // the lane's bias quads, fetched once: a load inside the epilogue costs every tile an L2 round trip
    float4 bvr[NR];
#pragma unroll
    for (int nb = 0; nb < NR; ++nb) {
        const int co = n0 + nb * 16 + 4 * kk;
        bvr[nb] = (bias && co < Cout) ? *reinterpret_cast<const float4 *>(bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int fast_lane = (((4 * wv) * W + li) * Cout + 4 * kk) * 4;    // the lane's part of a full tile's output offsets
    auto epilogue = [&](int tx, int ty, int n, auto fastc) {
        constexpr bool FAST = decltype(fastc)::value;
        auto actf = [&](float v) { return actk(v, fastc); };
        const int gx = tx * TW + li;
#pragma unroll
        for (int nb = 0; nb < NR; ++nb) {
            const int co = n0 + nb * 16 + 4 * kk;
            const float4 bv = bvr[nb];
            f32x4 o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gy = ty * TH + 4 * wv + r;
                o[r][0] = actf(bias ? acc[r][nb][0] + bv.x : acc[r][nb][0]);
                o[r][1] = actf(bias ? acc[r][nb][1] + bv.y : acc[r][nb][1]);
                o[r][2] = actf(bias ? acc[r][nb][2] + bv.z : acc[r][nb][2]);
                o[r][3] = actf(bias ? acc[r][nb][3] + bv.w : acc[r][nb][3]);
                if (epi.store_y) {
#if SQ_STORE_PERMUTE
                    const float t0 = __int_as_float(__builtin_amdgcn_ds_bpermute(st_src, __float_as_int(o[r][0])));
                    const float t1 = __int_as_float(__builtin_amdgcn_ds_bpermute(st_src, __float_as_int(o[r][1])));
                    const float t2 = __int_as_float(__builtin_amdgcn_ds_bpermute(st_src, __float_as_int(o[r][2])));
                    const float t3 = __int_as_float(__builtin_amdgcn_ds_bpermute(st_src, __float_as_int(o[r][3])));
                    f32x4 t = (f32x4){t0, t1, t2, t3};
                    const int sgx = tx * TW + st_px, sco = n0 + nb * 16 + 4 * st_q;
                    const bool ok = gy < H && sgx < W && sco < Cout;
                    const unsigned off = ok ? (unsigned)((((n * H + gy) * W + sgx) * Cout + sco) * 4) : OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(
                        *reinterpret_cast<__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned *>(&t),
                        yrsrc, off, 0, 0);
#else
                    unsigned off;
                    if constexpr (FAST) {
                        off = (unsigned)(fast_lane + nb * 64) + (unsigned)((((n * H + ty * TH + r) * W + tx * TW) * Cout + n0) * 4);
                    } else {
                        const bool ok = gy < H && gx < W && co < Cout;
                        off = ok ? (unsigned)((((n * H + gy) * W + gx) * Cout + co) * 4) : OOB;
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(
                        *reinterpret_cast<__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned *>(&o[r]),
                        yrsrc, off, 0, 0);
#endif
                }
                acc[r][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};      // ready for the next tile
            }
            // ---- fused 2x2 max-pool: rows (0,1) and (2,3) in registers, x-neighbour = lane ^ 1 --------
            if (epi.pooled) {
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    float mp[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float a = o[2 * pr][j], b = o[2 * pr + 1][j];
                        const float v = b > a ? b : a;
                        const float u = __shfl_xor(v, 1);
                        mp[j] = u > v ? u : v;
                    }
                    const int py = (ty * TH + 4 * wv) / 2 + pr, px = gx >> 1;
                    if ((li & 1) == 0 && py < (H >> 1) && px < (W >> 1) && co < Cout)
                        *reinterpret_cast<float4 *>(epi.pooled + ((size_t)(n * (H >> 1) + py) * (W >> 1) + px) * Cout + co) =
                            make_float4(mp[0], mp[1], mp[2], mp[3]);
                }
            }
            // ---- fused 1x1 head (Cout == 16 only), also on the matrix cores.  The accumulator layout has
            // the 16 channels of a pixel spread over 4 lanes; the wave transposes its 16-pixel x 16-channel
            // row through a private LDS image (no block barrier: same-wave DS operations are ordered) and
            // runs 4 MFMA steps with A = head_w^T (rows = head outputs), B = [channel][pixel]: one fmaf
            // chain over c = 0..15 per output, exactly as conv1x1_small / the oracle.  The kk == 0 lanes end
            // up with the head outputs of pixel li: + bias, store 16 x head_c contiguous floats, argmax.
            if (epi.head_w) {
                float *hs = head_scratch + wv * (16 * C::HEAD_PS);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float *d = hs + li * C::HEAD_PS + 4 * kk;
                    *reinterpret_cast<float2 *>(d) = make_float2(o[r][0], o[r][1]);
                    *reinterpret_cast<float2 *>(d + 2) = make_float2(o[r][2], o[r][3]);
                    __builtin_amdgcn_wave_barrier();
                    f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4)
                        z = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[s4], hs[li * C::HEAD_PS + 4 * s4 + kk], z, 0, 0, 0);
                    __builtin_amdgcn_wave_barrier();
                    const int gy = ty * TH + 4 * wv + r;
                    if (kk == 0 && gy < H && gx < W) {
                        const size_t p = (size_t)(n * H + gy) * W + gx;
                        float best = 0.f;
                        int bi = 0;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (q < epi.head_c) {
                                const float v = epi.head_b ? z[q] + hb[q] : z[q];
                                epi.logits[p * epi.head_c + q] = v;
                                if (q == 0 || v > best) { best = v; bi = q; }
                            }
                        if (epi.mask) epi.mask[p] = (uint8_t)bi;
                    }
                }
            }
        }
    };

    // tile coordinates advance by carries (three divisions by run-time values per decode were ~80 scalar instructions per
    // item in issue() alone); plain ints, not a struct (sq_conv_bf16.hip's experiment put structs on the stack)
    const int per_image = tiles_x * tiles_y;
    const int s_n = tstride / per_image, s_y = (tstride % per_image) / tiles_x, s_x = (tstride % per_image) % tiles_x;
    int ctx = t_begin % tiles_x, cty = (t_begin / tiles_x) % tiles_y, cn = t_begin / per_image;   // the tile being multiplied
    int ntx = ctx + s_x, nty = cty, nn = cn;                                                      // the next one of the walk
    if (ntx >= tiles_x) { ntx -= tiles_x; nty += 1; }
    nty += s_y;
    if (nty >= tiles_y) { nty -= tiles_y; nn += 1; }
    nn += s_n;
    issue(ctx, cty, cn, 0, true, true);
    __builtin_amdgcn_s_waitcnt(0x0F70);                         // vmcnt(0), stated outside commit()'s branches (see the main loop)
    commit(true);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int nb = 0; nb < NR; ++nb) acc[r][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    if constexpr (FIRST) {
        first_conv(ctx, cty);
        __syncthreads();
    }
    if constexpr (UP) {
        up_conv(ctx, cty);
        __syncthreads();
    }

    int tile = t_begin, chunk = 0;
    for (int it = 0; it < nitems; ++it) {
        // next work item
        int ntile = tile, nchk = chunk + 1;
        if (nchk == nchunk) { nchk = 0; ntile = tile + tstride; }
        const bool has_next = it + 1 < nitems;
        const bool same = ntile == tile;
        const int itx = same ? ctx : ntx, ity = same ? cty : nty, itn = same ? cn : nn;     // the next ITEM's tile
        if (has_next) issue(itx, ity, itn, nchk * KC, restage_w, true);

        // ---- MFMA phase: NSTEP steps; the ds_reads of step s+1 are issued BEFORE the MFMAs
        // of step s (sched_barrier fences stop hipcc sinking them back to just-in-time) ----------
        {
            float a0[NR], b0[4], a1[NR], b1[4];
            load_frag(0, a0, b0);
            // wave priority: LOW while a wave only feeds the matrix pipe, HIGH while it stages / stores -- the
            // other resident blocks' MFMA phases fill the pipe anyway, and a wave that gets through its commit and
            // epilogue quickly is back issuing MFMAs sooner (+1.6 % on the whole step, measured)
            __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int st = 0; st < C::NSTEP; st += 2) {
                if (st + 1 < C::NSTEP) load_frag(st + 1, a1, b1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int nb = 0; nb < NR; ++nb)
                        acc[r][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[nb], b0[r], acc[r][nb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (st + 1 < C::NSTEP) {
                    if (st + 2 < C::NSTEP) load_frag(st + 2, a0, b0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int nb = 0; nb < NR; ++nb)
                            acc[r][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[nb], b1[r], acc[r][nb], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_s_setprio(3);
        }
        // ---- stage the next item, THEN store this tile: the stores drain under the next MFMA
        // phase instead of being waited for together with the prefetch (vmcnt is in-order) -------
        // The prefetch has landed -- stated OUTSIDE the `has_next` branch.  The compiler does not correlate the two
        // `if (has_next)` tests, so it also walks "issued but never committed", keeps the prefetch registers marked as
        // pending loads round the back edge and guards the next issue()'s register writes with s_waitcnt vmcnt(5..0);
        // at run time those wait for the epilogue's STORES (one counter for loads and stores on gfx9): a store round
        // trip per tile, seen as "issue 32 % / epilogue 34 % / MFMA 28 %" in tools/conv_ablation.py timeline.
        __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0); expcnt / lgkmcnt untouched
        if (has_next) {
            __syncthreads();            // every wave is done reading this item's LDS image
            commit(restage_w);
            if constexpr (FIRST) {
                __syncthreads();        // the 20x20 patch is complete
                first_conv(itx, ity);
            }
            if constexpr (UP) {
                __syncthreads();        // skip halo and low-resolution patch are complete
                up_conv(itx, ity);
            }
        }
        if (chunk == nchunk - 1) {
            const bool fast = is_relu && cty * TH + TH <= H && ctx * TW + TW <= W && n0 + BN <= Cout;
            // the second copy of the epilogue costs the <32,3,32> form 15 spilled registers: 64-channel blocks only
            if constexpr ((BN == 64 || (SQ_V2_FAST32 && BN == 32 && KC == 32)) && MODE == 0) {
                if (fast) epilogue(ctx, cty, cn, std::integral_constant<bool, true>{});
                else epilogue(ctx, cty, cn, std::integral_constant<bool, false>{});
            } else {
                (void)fast;
                epilogue(ctx, cty, cn, std::integral_constant<bool, false>{});
            }
        }
        if (has_next) __syncthreads();
        if (ntile != tile) {                                     // the walk moves on by one tile stride
            ctx = ntx; cty = nty; cn = nn;
            ntx += s_x;
            if (ntx >= tiles_x) { ntx -= tiles_x; nty += 1; }
            nty += s_y;
            if (nty >= tiles_y) { nty -= tiles_y; nn += 1; }
            nn += s_n;
        }
        tile = ntile;
        chunk = nchk;
    }
}

template <int BN, int KS, int KC, int MODE>
int launch_v2(const float *x, const float *w, const float *bias, float *y, int N, int H, int W, int Cin,
              int Cout, float wscale, int act, const SqConvEpi &epi, hipStream_t st) {
    using C = Cfg2<BN, KS, KC>;
    static bool attr_set = false;
    auto kern = conv_mfma_f32_v2_kernel<BN, KS, KC, MODE>;
    constexpr int lds_base = MODE == 1 ? C::LDS_BYTES_FIRST : (MODE == 2 ? C::LDS_BYTES_UP : C::LDS_BYTES);
    constexpr int lds_max = lds_base + C::HEAD_FLOATS * 4;
    const int lds = lds_base + (epi.head_w ? C::HEAD_FLOATS * 4 : 0);
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds_max) != hipSuccess) {
            sq_set_error("conv_mfma_f32_v2: cannot reserve %d bytes of LDS", lds_max);
            return SQ_ELAUNCH;
        }
        attr_set = true;
    }
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int ntiles = tiles_x * tiles_y * N;
    const int gy = (Cout + BN - 1) / BN;
    // persistent grid: about OCC resident blocks per CU over both grid dimensions
    int want = (256 * (MODE == 2 ? 3 : C::OCC) + gy - 1) / gy;
    if (want < 1) want = 1;
    int tpb = (ntiles + want - 1) / want;
    if (tpb < 1) tpb = 1;
    const int gx = (ntiles + tpb - 1) / tpb;
    hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(256), lds, st, x, w, bias, y, N, H, W, Cin, Cout,
                       wscale, act, tiles_x, tiles_y, ntiles, tpb, epi);
    return sq_check_launch("sq_conv2d_nhwc_fwd_f32(v2)");
}

inline bool stage32() {                                         // SQ_CONV_STAGE32=0: A/B switch back to 16-channel items
    static const bool v = [] { const char *e = getenv("SQ_CONV_STAGE32"); return !(e && e[0] == '0'); }();
    return v;
}

template <int KS, int KC>
int dispatch_bn(const float *x, const float *w, const float *bias, float *y, int N, int H, int W, int Cin,
                int Cout, float wscale, int act, const SqConvEpi &epi, hipStream_t st) {
    // widest channel block the layer fills the chip with: small images (GAN 4x4..32x32 levels, small
    // batches) have few pixel tiles, so trade operand reuse for blocks until there are ~2 per CU.
    // BN only changes which block computes an output, never its fmaf chain.
    const int ntiles = ((W + TW - 1) / TW) * ((H + TH - 1) / TH) * N;
    int bn = Cout >= 64 ? 64 : (Cout > 16 ? 32 : 16);
    // (32-channel blocks for the wide layers, three blocks per CU: 5.189 vs 5.107 ms per step -- 64 it stays)
    while (bn > 16 && (int64_t)ntiles * ((Cout + bn - 1) / bn) < 2 * 256) bn >>= 1;
    // 32 -> 32 (and wider-input) 3x3 layers on 32-channel blocks: stage 32 input channels per item (same chain)
    if constexpr (KS == 3 && KC == 16) {
        if (bn == 32 && Cin % 32 == 0 && !epi.x2 && stage32())
            return launch_v2<32, 3, 32, false>(x, w, bias, y, N, H, W, Cin, Cout, wscale, act, epi, st);
    }
    if (bn == 64) return launch_v2<64, KS, KC, false>(x, w, bias, y, N, H, W, Cin, Cout, wscale, act, epi, st);
    if (bn == 32) return launch_v2<32, KS, KC, false>(x, w, bias, y, N, H, W, Cin, Cout, wscale, act, epi, st);
    return launch_v2<16, KS, KC, false>(x, w, bias, y, N, H, W, Cin, Cout, wscale, act, epi, st);
}

bool fits32(int N, int H, int W, int C) { return (size_t)N * H * W * (size_t)C * 4 < ((size_t)1 << 31); }

}  // namespace

// internal entry used by sq_conv2d_nhwc_fwd_f32's dispatcher (sq_conv_f32.hip)
int sq_conv_mfma_v2(const float *x, const float *w, const float *bias, float *y, int N, int H, int W,
                    int Cin, int Cout, int K, float wscale, int act, hipStream_t st) {
    SqConvEpi epi = {};
    epi.store_y = 1;
    if (Cin == 16 && (Cout == 16 || Cout == 32) && K == 3 && wscale == 1.0f) {
        const int r = sq_conv_l0_launch(0, x, w, bias, y, N, H, W, Cout, act, epi, st);
        if (r != SQ_L0_NOT_MINE) return r;
    }
    if (Cin % 16 == 0)
        return K == 3 ? dispatch_bn<3, 16>(x, w, bias, y, N, H, W, Cin, Cout, wscale, act, epi, st)
                      : dispatch_bn<1, 16>(x, w, bias, y, N, H, W, Cin, Cout, wscale, act, epi, st);
    return K == 3 ? dispatch_bn<3, 8>(x, w, bias, y, N, H, W, Cin, Cout, wscale, act, epi, st)
                  : dispatch_bn<1, 8>(x, w, bias, y, N, H, W, Cin, Cout, wscale, act, epi, st);
}

// conv_layer on the concat bridge (unet.py:196-197, 321): y = act(conv(concat([xa, xb], -1)) + bias) with the
// concatenated tensor never materialised -- the chunk loop takes its first Cin/2 channels from xa (the up-scaled
// tensor), the rest from xb (the skip tensor).  Same chain as sq_conv2d_nhwc_fwd_f32 on torch.cat([xa, xb], -1).
extern "C" int sq_conv2d_concat_nhwc_fwd_f32(const float *xa, const float *xb, const float *w, const float *bias, float *y,
                                             int N, int H, int W, int Ca, int Cout, int K, int act, void *stream) {
    SQ_REQUIRE(xa && xb && w && y, "sq_conv2d_concat_nhwc_fwd_f32: null tensor pointer");
    SQ_REQUIRE(N > 0 && H > 0 && W > 0 && (K == 1 || K == 3), "sq_conv2d_concat_nhwc_fwd_f32: bad shape / K");
    SQ_REQUIRE(Ca % 16 == 0 && Ca > 0 && Cout % 4 == 0 && Cout > 0,
               "sq_conv2d_concat_nhwc_fwd_f32: channels per source %d (multiple of 16), Cout=%d (multiple of 4)", Ca, Cout);
    SQ_REQUIRE(fits32(N, H, W, 2 * Ca > Cout ? 2 * Ca : Cout), "sq_conv2d_concat_nhwc_fwd_f32: tensors must be < 2 GiB");
    SQ_REQUIRE(act >= SQ_ACT_NONE && act <= SQ_ACT_LEAKY, "sq_conv2d_concat_nhwc_fwd_f32: bad activation %d", act);
    SQ_REQUIRE_ALIGNED(xa); SQ_REQUIRE_ALIGNED(xb); SQ_REQUIRE_ALIGNED(w); SQ_REQUIRE_ALIGNED(y);
    if (bias) SQ_REQUIRE_ALIGNED(bias);
    SqConvEpi epi = {};
    epi.store_y = 1;
    epi.x2 = xb;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    return K == 3 ? dispatch_bn<3, 16>(xa, w, bias, y, N, H, W, 2 * Ca, Cout, 1.0f, act, epi, st)
                  : dispatch_bn<1, 16>(xa, w, bias, y, N, H, W, 2 * Ca, Cout, 1.0f, act, epi, st);
}

// conv_block tail + max_pool_layer (sequitr/networks/unet.py:241-243, 265-277): 3x3 conv + bias + act,
// writes y (N,H,W,Cout) AND pooled (N,H/2,W/2,Cout).
extern "C" int sq_conv3x3_pool_fwd_f32(const float *x, const float *w, const float *bias, float *y, float *pooled,
                                       int N, int H, int W, int Cin, int Cout, int act, void *stream) {
    SQ_REQUIRE(x && w && y && pooled, "sq_conv3x3_pool_fwd_f32: null tensor pointer");
    SQ_REQUIRE(N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "sq_conv3x3_pool_fwd_f32: H, W must be even");
    SQ_REQUIRE(Cin % 16 == 0 && Cin > 0 && Cout % 4 == 0 && Cout > 0,
               "sq_conv3x3_pool_fwd_f32: Cin=%d (multiple of 16), Cout=%d (multiple of 4)", Cin, Cout);
    SQ_REQUIRE(fits32(N, H, W, Cin > Cout ? Cin : Cout), "sq_conv3x3_pool_fwd_f32: tensors must be < 2 GiB");
    SQ_REQUIRE(act >= SQ_ACT_NONE && act <= SQ_ACT_LEAKY, "sq_conv3x3_pool_fwd_f32: bad activation %d", act);
    SQ_REQUIRE_ALIGNED(x); SQ_REQUIRE_ALIGNED(w); SQ_REQUIRE_ALIGNED(y); SQ_REQUIRE_ALIGNED(pooled);
    if (bias) SQ_REQUIRE_ALIGNED(bias);
    SqConvEpi epi = {};
    epi.store_y = 1;
    epi.pooled = pooled;
    if (Cin == 16 && Cout == 16) {
        const int r = sq_conv_l0_launch(0, x, w, bias, y, N, H, W, 16, act, epi, reinterpret_cast<hipStream_t>(stream));
        if (r != SQ_L0_NOT_MINE) return r;
    }
    return dispatch_bn<3, 16>(x, w, bias, y, N, H, W, Cin, Cout, 1.0f, act, epi, reinterpret_cast<hipStream_t>(stream));
}

// last conv_layer of up0 + conv_layer_1x1 head + prediction (unet.py:252-253, 321): the 16-channel
// activation is never stored; logits (N,H,W,head_c) and the uint8 mask come straight from the epilogue.
extern "C" int sq_conv3x3_head_fwd_f32(const float *x, const float *w, const float *bias, const float *head_w,
                                       const float *head_b, float *logits, uint8_t *mask, int N, int H, int W,
                                       int Cin, int head_c, int act, void *stream) {
    SQ_REQUIRE(x && w && head_w && logits, "sq_conv3x3_head_fwd_f32: null tensor pointer");
    SQ_REQUIRE(N > 0 && H > 0 && W > 0 && Cin % 16 == 0 && Cin > 0, "sq_conv3x3_head_fwd_f32: Cin=%d (multiple of 16)", Cin);
    SQ_REQUIRE(head_c >= 1 && head_c <= 4, "sq_conv3x3_head_fwd_f32: head_c=%d (1..4)", head_c);
    SQ_REQUIRE(fits32(N, H, W, Cin > 16 ? Cin : 16), "sq_conv3x3_head_fwd_f32: tensors must be < 2 GiB");
    SQ_REQUIRE(act >= SQ_ACT_NONE && act <= SQ_ACT_LEAKY, "sq_conv3x3_head_fwd_f32: bad activation %d", act);
    SQ_REQUIRE_ALIGNED(x); SQ_REQUIRE_ALIGNED(w);
    if (bias) SQ_REQUIRE_ALIGNED(bias);
    SqConvEpi epi = {};
    epi.store_y = 0;
    epi.head_w = head_w; epi.head_b = head_b; epi.logits = logits; epi.mask = mask; epi.head_c = head_c;
    if (Cin == 16) {
        const int r = sq_conv_l0_launch(0, x, w, bias, nullptr, N, H, W, 16, act, epi, reinterpret_cast<hipStream_t>(stream));
        if (r != SQ_L0_NOT_MINE) return r;
    }
    return launch_v2<16, 3, 16, false>(x, w, bias, nullptr, N, H, W, Cin, 16, 1.0f, act, epi,
                                       reinterpret_cast<hipStream_t>(stream));
}

// conv_block of down0 (unet.py:238, 265-277) for a single-channel input: conv1 (3x3, 1 -> 16, bias, ReLU)
// is evaluated in LDS and only conv2's output y (N,H,W,16) -- and optionally its max-pool -- reach HBM.
extern "C" int sq_conv3x3_first_block_fwd_f32(const float *x, const float *w1, const float *b1, const float *w2,
                                              const float *b2, float *y, float *pooled, int N, int H, int W,
                                              void *stream) {
    SQ_REQUIRE(x && w1 && b1 && w2 && y, "sq_conv3x3_first_block_fwd_f32: null tensor pointer");
    SQ_REQUIRE(N > 0 && H > 0 && W > 0, "sq_conv3x3_first_block_fwd_f32: bad shape");
    SQ_REQUIRE(!pooled || (H % 2 == 0 && W % 2 == 0), "sq_conv3x3_first_block_fwd_f32: pooling needs even H, W");
    SQ_REQUIRE(fits32(N, H, W, 16), "sq_conv3x3_first_block_fwd_f32: tensors must be < 2 GiB");
    SQ_REQUIRE_ALIGNED(w2); SQ_REQUIRE_ALIGNED(y);
    if (b2) SQ_REQUIRE_ALIGNED(b2);
    if (pooled) SQ_REQUIRE_ALIGNED(pooled);
    SqConvEpi epi = {};
    epi.store_y = 1;
    epi.pooled = pooled;
    epi.first_w = w1; epi.first_b = b1;
    {
        const int r = sq_conv_l0_launch(1, x, w2, b2, y, N, H, W, 16, SQ_ACT_RELU, epi, reinterpret_cast<hipStream_t>(stream));
        if (r != SQ_L0_NOT_MINE) return r;
    }
    return launch_v2<16, 3, 16, true>(x, w2, b2, y, N, H, W, 1, 16, 1.0f, SQ_ACT_RELU, epi,
                                      reinterpret_cast<hipStream_t>(stream));
}

// conv_transpose_layer + bridge + first conv_layer of up0 (unet.py:299-322) in one kernel, for the level-0
// shape (transpose conv 32 -> 16 channels, 3x3 conv 16 -> 16): y = act(conv3x3(bridge(convT2x2s2(x_low) + bt,
// skip)) + bias).  x_low (N,H/2,W/2,32); wt (2,2,16,32) TF layout; skip, y (N,H,W,16); bridge = SQ_BRIDGE_*.
// Bit-identical to sq_convT2x2s2_nhwc_fwd_f32 followed by sq_conv2d_nhwc_fwd_f32.
extern "C" int sq_convT_conv3x3_fwd_f32(const float *x_low, const float *wt, const float *bt, const float *skip,
                                        int bridge, const float *w, const float *bias, float *y, int N, int H, int W,
                                        int act, void *stream) {
    SQ_REQUIRE(x_low && wt && skip && w && y, "sq_convT_conv3x3_fwd_f32: null tensor pointer");
    SQ_REQUIRE(N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "sq_convT_conv3x3_fwd_f32: H, W must be even");
    SQ_REQUIRE(bridge >= SQ_BRIDGE_NONE && bridge <= SQ_BRIDGE_SUB, "sq_convT_conv3x3_fwd_f32: bad bridge %d", bridge);
    SQ_REQUIRE(fits32(N, H, W, 16), "sq_convT_conv3x3_fwd_f32: tensors must be < 2 GiB");
    SQ_REQUIRE(act >= SQ_ACT_NONE && act <= SQ_ACT_LEAKY, "sq_convT_conv3x3_fwd_f32: bad activation %d", act);
    SQ_REQUIRE_ALIGNED(x_low); SQ_REQUIRE_ALIGNED(skip); SQ_REQUIRE_ALIGNED(w); SQ_REQUIRE_ALIGNED(y);
    if (bias) SQ_REQUIRE_ALIGNED(bias);
    if (bt) SQ_REQUIRE_ALIGNED(bt);
    SqConvEpi epi = {};
    epi.store_y = 1;
    epi.up_x = x_low; epi.up_w = wt; epi.up_b = bt; epi.up_bridge = bridge;
    {
        const int r = sq_conv_l0_launch(2, skip, w, bias, y, N, H, W, 16, act, epi, reinterpret_cast<hipStream_t>(stream));
        if (r != SQ_L0_NOT_MINE) return r;
    }
    return launch_v2<16, 3, 16, 2>(skip, w, bias, y, N, H, W, 16, 16, 1.0f, act, epi,
                                   reinterpret_cast<hipStream_t>(stream));
}
