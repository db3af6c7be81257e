// conv_transpose_layer + bridge of the decoder's levels 1..3 (sequitr/networks/unet.py:299-322, 336-338) in f32: the
// 2x2 stride-2 transpose convolution as the GEMM  D[rho][p] = chain_c fmaf(Wt[rho][c], x[p][c])  (rho = (2a + b) Cout + o,
// c ascending: the chain of convT2x2_mfma_f32_kernel in sq_convt_loss.hip and of the oracle, bit for bit), + bias, then
// the bridge with the skip tensor.  Rebuilt on what the level-0 kernels of this round showed (sq_conv_f32_l0.hip): what
// a SIMD issues BESIDE its MFMAs is what it loses, and a streaming epilogue needs its loads in flight early.
//   * persistent blocks walk (pixel tile, row tile) work; a block tile is 128 rows x 32 WNI input pixels (WNI = 1, 2, 4;
//     a wave owns 64 rows x 16 WNI pixels): wide tiles issue more MFMAs per operand fragment and win stand-alone, narrow
//     ones keep more blocks per CU and win inside the network's step (the default);
//   * both operands are staged CHANNEL-TRANSPOSED ([row][16-channel half][kk][s] = channel 16 half + 4 s + kk, 40
//     floats per row): one ds_read_b128 is a lane's operand for four k steps, conflict-free (10 li + kk covers the 16
//     slots of a bank row in each of the instruction's lane groups) -- 16 LDS reads per 128 MFMAs instead of 160;
//   * T14 split staging across chunks AND tiles: the global loads of the next 32-channel chunk (or of the next tile's
//     first chunk) are in flight under this chunk's MFMAs;
//   * all sixteen 16-byte bridge operands of a lane are requested before the tile's LAST chunk is multiplied and are in
//     registers when the epilogue starts; the stores drain under the next tile's MFMAs;
//   * output addresses are a per-lane pixel part + a per-lane class part (both 32-bit), one pair of divisions per tile.
// Takes Cin % 32 == 0, Cout % 32 == 0, tensors < 2 GiB; anything else stays with the 64 x 64 kernel (also the A/B
// reference: SQ_CONVT_V2=0).
#include <stdlib.h>
#include "sq_common.h"

#ifndef SQ_CT_WNI
#define SQ_CT_WNI 1                 // default pixel-tile width / 32 (env SQ_CONVT_WNI overrides; 1: 5.133, 2: 5.140, 4: 5.162 ms per inference step)
#endif
#ifndef SQ_CT_FRAG_DBUF
#define SQ_CT_FRAG_DBUF 0
#endif

namespace {

constexpr int BM = 128, KCH = 32;             // the pixel tile is 32 * WNI wide (template): 128 or 64
constexpr int RS = 40;                          // floats per staged row: [half][kk][s] + 8 of padding
constexpr int AS_FLOATS = BM * RS;
constexpr unsigned OOB = 0x80000000u;

typedef unsigned u32x4 __attribute__((__vector_size__(4 * sizeof(unsigned))));

struct CtArgs {
    const float *x, *w, *bias, *skip;
    float *y;
    int P, H, W, Cin, Cout;
    int mtiles, ntiles;                         // row tiles (4 Cout / 128), pixel tiles
};

// WNI: 16-pixel column blocks per wave.  4: 128 x 128 block tiles, 64 x 64 per wave (240 registers, two blocks per CU);
// 2: 128 x 64 tiles, 64 x 32 per wave -- half the accumulators and bridge operands, three blocks per CU: a block's epilogue
// (its share of the bridge + output traffic) then overlaps two other blocks' MFMAs instead of one
template <int BRIDGE, int WNI>
__global__ __launch_bounds__(256, (WNI == 4 ? 2 : (WNI == 2 ? 3 : 4))) void convT2x2_v2_kernel(const CtArgs a) {
    constexpr int BN = 32 * WNI;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *as = smem;
    float *xs = smem + AS_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, kk = lane >> 4;
    const int wm = wv >> 1, wn = wv & 1;        // the wave's 64 x (16 WNI) quadrant
    const int Cin = a.Cin, Cout = a.Cout, H = a.H, W = a.W;
    const int G = (int)gridDim.x;
    const int vb = (int)sq_xcd_remap(blockIdx.x, gridDim.x);
    const int total = a.mtiles * a.ntiles;
    if (vb >= total) return;
    const int t_count = (total - vb + G - 1) / G;
    const int nchunk = Cin / KCH;
    const int nitems = t_count * nchunk;

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.x), 0, (int)((size_t)a.P * Cin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.w), 0, (int)((size_t)4 * Cout * Cin * 4), 0x00020000);
    const int out_bytes = (int)((size_t)a.P * 4 * Cout * 4);
    const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.skip), 0, BRIDGE != SQ_BRIDGE_NONE ? out_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, out_bytes, 0x00020000);

    // staging: 128 rows x 8 float4 of a chunk over 256 threads = 4 per thread and operand; thread -> (row, quad) in
    // memory order
    const int srow = tid >> 3, sq = tid & 7;
    const int grel = (srow * Cin + sq * 4) * 4;                     // + 32 rows per slot
    float *cwa = as + srow * RS + (sq >> 2) * 16 + (sq & 3);        // channel 4 q + j -> [half q / 4][kk = j][s = q % 4]
    float *cwx = xs + srow * RS + (sq >> 2) * 16 + (sq & 3);
    float4 ar[4], br[WNI];
    auto issue = [&](int mt, int nt, int cc) {
        const int abase = (mt * BM * Cin + cc) * 4, xbase = (nt * BN * Cin + cc) * 4;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, grel + it * 32 * Cin * 4, abase, 0);
            ar[it] = *reinterpret_cast<const float4 *>(&v);
        }
#pragma unroll
        for (int it = 0; it < WNI; ++it) {                          // pixels past P read as zeros (buffer range check)
            // the tile offset sits in the VGPR offset: that is the one the range check sees
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, grel + it * 32 * Cin * 4 + xbase, 0, 0);
            br[it] = *reinterpret_cast<const float4 *>(&v);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            float *d = cwa + it * 32 * RS;
            d[0] = ar[it].x; d[4] = ar[it].y; d[8] = ar[it].z; d[12] = ar[it].w;
        }
#pragma unroll
        for (int it = 0; it < WNI; ++it) {
            float *e = cwx + it * 32 * RS;
            e[0] = br[it].x; e[4] = br[it].y; e[8] = br[it].z; e[12] = br[it].w;
        }
    };

    f32x4 acc[4][WNI];
    const float *fa = as + (64 * wm + li) * RS + 4 * kk;
    const float *fb = xs + (16 * WNI * wn + li) * RS + 4 * kk;
    auto load_frags = [&](int half, float4 (&fa4)[4], float4 (&fb4)[WNI]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) fa4[i] = *reinterpret_cast<const float4 *>(fa + i * 16 * RS + half * 16);
#pragma unroll
        for (int i = 0; i < WNI; ++i) fb4[i] = *reinterpret_cast<const float4 *>(fb + i * 16 * RS + half * 16);
    };
    auto mfma_half = [&](const float4 (&fa4)[4], const float4 (&fb4)[WNI], bool first) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const float av = s == 0 ? fa4[mi].x : (s == 1 ? fa4[mi].y : (s == 2 ? fa4[mi].z : fa4[mi].w));
#pragma unroll
                for (int ni = 0; ni < WNI; ++ni) {
                    const float bvv = s == 0 ? fb4[ni].x : (s == 1 ? fb4[ni].y : (s == 2 ? fb4[ni].z : fb4[ni].w));
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                        av, bvv, (first && s == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[mi][ni], 0, 0, 0);
                }
            }
        }
    };

    // ---- epilogue geometry: offset = pixel part [ni] + class part [mi] (bytes) ----------------------------------------
    int pixoff[WNI], clsoff[4];
    float4 bq[4], sk[4][WNI];
    // The row tile of a block never changes (the grid is a multiple of the row tiles per pixel tile), so the class part
    // of the offsets and the bias quads are made ONCE; the pixel part advances from tile to tile by carries (the tile
    // stride in pixels is a constant of the block): per tile this was six divisions and four bias loads for 64-512 MFMAs
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int rho = (vb % a.mtiles) * BM + 64 * wm + 16 * mi + 4 * kk;
        const int ab = rho / Cout, o = rho - ab * Cout;
        clsoff[mi] = (((ab >> 1) * 2 * W + (ab & 1)) * Cout + o) * 4;
        bq[mi] = a.bias ? *reinterpret_cast<const float4 *>(a.bias + o) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int step_px = (G / a.mtiles) * BN;                        // pixels from one tile of the walk to the next
    const int s_j = step_px % W, s_i = (step_px / W) % H, s_n = step_px / (W * H);
    int gp = (vb / a.mtiles) * BN + 16 * WNI * wn + li;             // the lane's first pixel of the current tile
    int gj, gi, gn;
    {
        const unsigned pu = (unsigned)(gp < a.P ? gp : 0), t = pu / (unsigned)W;
        gj = (int)(pu - t * (unsigned)W);
        gn = (int)(t / (unsigned)H);
        gi = (int)(t - (unsigned)gn * (unsigned)H);
    }
    auto geometry = [&]() {                                         // pixel offsets of the current tile
        int jj = gj, ii = gi, n32 = gn;
#pragma unroll
        for (int ni = 0; ni < WNI; ++ni) {
            const int p = gp + 16 * ni;
            pixoff[ni] = p < a.P ? (int)((((n32 * 2 * H + 2 * ii) * (2 * W)) + 2 * jj) * Cout * 4) : (int)OOB;
            if (ni + 1 < WNI) {
                jj += 16;
                while (jj >= W) {
                    jj -= W;
                    if (++ii == H) { ii = 0; ++n32; }
                }
            }
        }
    };
    auto geometry_advance = [&]() {                                 // ... and on to the next tile of the walk
        gp += step_px;
        gj += s_j;
        if (gj >= W) { gj -= W; gi += 1; }
        gi += s_i;
        if (gi >= H) { gi -= H; gn += 1; }
        gn += s_n;
    };
    auto issue_skip = [&]() {
        if constexpr (BRIDGE != SQ_BRIDGE_NONE) {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < WNI; ++ni) {
                    const unsigned off = pixoff[ni] == (int)OOB ? OOB : (unsigned)(pixoff[ni] + clsoff[mi]);
                    const auto v = __builtin_amdgcn_raw_buffer_load_b128(srsrc, off, 0, 0);
                    sk[mi][ni] = *reinterpret_cast<const float4 *>(&v);
                }
        }
    };
    auto epilogue = [&]() {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < WNI; ++ni) {
                f32x4 v = acc[mi][ni];
                if (a.bias) { v[0] += bq[mi].x; v[1] += bq[mi].y; v[2] += bq[mi].z; v[3] += bq[mi].w; }
                if constexpr (BRIDGE == SQ_BRIDGE_ADD) { v[0] += sk[mi][ni].x; v[1] += sk[mi][ni].y; v[2] += sk[mi][ni].z; v[3] += sk[mi][ni].w; }
                if constexpr (BRIDGE == SQ_BRIDGE_MUL) { v[0] *= sk[mi][ni].x; v[1] *= sk[mi][ni].y; v[2] *= sk[mi][ni].z; v[3] *= sk[mi][ni].w; }
                if constexpr (BRIDGE == SQ_BRIDGE_SUB) { v[0] -= sk[mi][ni].x; v[1] -= sk[mi][ni].y; v[2] -= sk[mi][ni].z; v[3] -= sk[mi][ni].w; }
                const unsigned off = pixoff[ni] == (int)OOB ? OOB : (unsigned)(pixoff[ni] + clsoff[mi]);
                __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4 *>(&v), yrsrc, off, 0, 0);
            }
    };

    // ---- the item loop: item = (tile, 32-channel chunk); tiles of one pixel tile are neighbours in the walk ----------
    int tile = vb, chunk = 0;
    issue(tile % a.mtiles, tile / a.mtiles, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);                             // vmcnt(0)
    commit();
    __syncthreads();
    for (int it = 0; it < nitems; ++it) {
        int ntile = tile, nchk = chunk + 1;
        if (nchk == nchunk) { nchk = 0; ntile = tile + G; }
        const bool has_next = it + 1 < nitems;
        const bool last = chunk == nchunk - 1;
        if (last) {                                                 // the bridge operands, in flight under this chunk's MFMAs
            geometry();
            issue_skip();
        }
        if (has_next) issue(ntile % a.mtiles, ntile / a.mtiles, nchk * KCH);
        {
#if SQ_CT_FRAG_DBUF
            float4 fa0[4], fb0[WNI], fa1[4], fb1[WNI];
            load_frags(0, fa0, fb0);
            load_frags(1, fa1, fb1);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_half(fa0, fb0, chunk == 0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_half(fa1, fb1, false);
            __builtin_amdgcn_s_setprio(3);
#else
            // one fragment set (32 registers), the second half's reads behind the first half's MFMAs: with the sixteen
            // bridge operands in flight the double-buffered form spills (267 registers wanted)
            float4 fa0[4], fb0[WNI];
            load_frags(0, fa0, fb0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_half(fa0, fb0, chunk == 0);
            __builtin_amdgcn_sched_barrier(0);
            load_frags(1, fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_half(fa0, fb0, false);
            __builtin_amdgcn_s_setprio(3);
#endif
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);                         // prefetch + bridge operands have landed
        if (has_next) {
            __syncthreads();                                        // every wave is done reading this item's images
            commit();
        }
        if (last) { epilogue(); geometry_advance(); }
        if (has_next) __syncthreads();
        tile = ntile;
        chunk = nchk;
    }
}

template <int BRIDGE, int WNI>
int launch_ct(const CtArgs &a0, hipStream_t st) {
    static bool attr_set = false;
    auto kern = convT2x2_v2_kernel<BRIDGE, WNI>;
    constexpr int BN = 32 * WNI;
    constexpr int lds = (AS_FLOATS + BN * RS) * 4;
    CtArgs a = a0;
    a.ntiles = (a.P + BN - 1) / BN;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            sq_set_error("convT2x2_v2: cannot reserve %d bytes of LDS", lds);
            return SQ_ELAUNCH;
        }
        attr_set = true;
    }
    const int total = a.mtiles * a.ntiles;
    const int want = 256 * (WNI == 4 ? 2 : (WNI == 2 ? 3 : 4));
    int G = total < want ? total : want;
    if (G > a.mtiles) G -= G % a.mtiles;                            // the row tiles of one pixel tile start together
    hipLaunchKernelGGL(kern, dim3(G), dim3(256), lds, st, a);
    return sq_check_launch("sq_convT2x2s2_nhwc_fwd_f32(v2)");
}

}  // namespace

// internal entry used by sq_convT2x2s2_nhwc_fwd_f32 (sq_convt_loss.hip); SQ_NOT_MINE when the shape is not this kernel's
int sq_convT_v2_launch(const float *x, const float *w, const float *bias, const float *skip, float *y, int N, int H,
                       int W, int Cin, int Cout, int bridge, hipStream_t st) {
    const char *e = getenv("SQ_CONVT_V2");
    if (e && e[0] == '0') return SQ_NOT_MINE;
    const size_t P = (size_t)N * H * W;
    if (Cin % 32 != 0 || Cout % 32 != 0) return SQ_NOT_MINE;
    if (P * 4 * Cout * 4 >= ((size_t)1 << 31) || P * Cin * 4 >= ((size_t)1 << 31)) return SQ_NOT_MINE;
    CtArgs a = {};
    a.x = x; a.w = w; a.bias = bias; a.skip = skip; a.y = y;
    a.P = (int)P; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.mtiles = 4 * Cout / BM;
    static const int wni = [] { const char *v = getenv("SQ_CONVT_WNI"); return v ? atoi(v) : SQ_CT_WNI; }();   // A/B switch: 4 or 2
    if (wni == 4) {
        switch (bridge) {
            case SQ_BRIDGE_ADD: return launch_ct<SQ_BRIDGE_ADD, 4>(a, st);
            case SQ_BRIDGE_MUL: return launch_ct<SQ_BRIDGE_MUL, 4>(a, st);
            case SQ_BRIDGE_SUB: return launch_ct<SQ_BRIDGE_SUB, 4>(a, st);
            default: return launch_ct<SQ_BRIDGE_NONE, 4>(a, st);
        }
    }
    if (wni == 1) {
        switch (bridge) {
            case SQ_BRIDGE_ADD: return launch_ct<SQ_BRIDGE_ADD, 1>(a, st);
            case SQ_BRIDGE_MUL: return launch_ct<SQ_BRIDGE_MUL, 1>(a, st);
            case SQ_BRIDGE_SUB: return launch_ct<SQ_BRIDGE_SUB, 1>(a, st);
            default: return launch_ct<SQ_BRIDGE_NONE, 1>(a, st);
        }
    }
    switch (bridge) {
        case SQ_BRIDGE_ADD: return launch_ct<SQ_BRIDGE_ADD, 2>(a, st);
        case SQ_BRIDGE_MUL: return launch_ct<SQ_BRIDGE_MUL, 2>(a, st);
        case SQ_BRIDGE_SUB: return launch_ct<SQ_BRIDGE_SUB, 2>(a, st);
        default: return launch_ct<SQ_BRIDGE_NONE, 2>(a, st);
    }
}
