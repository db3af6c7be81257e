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
#include "sq_common.h"

namespace {

constexpr int TH = 16, TW = 16;

template <int BN, int KS, int KC>
struct Cfg2 {
    static constexpr int HALO_W = TW + KS - 1;
    static constexpr int HALO_H = TH + KS - 1;
    static constexpr int HP = HALO_W * HALO_H;
    static constexpr int PS = KC + 2;                       // pixel stride (floats): conflict-free B reads
    static constexpr int BNS = (BN % 32 == 0) ? BN + 16 : BN;  // weight row stride: conflict-free A reads
    static constexpr int WROWS = KS * KS * KC;
    static constexpr int XS_FLOATS = HP * PS;
    static constexpr int LDS_BYTES = (XS_FLOATS + WROWS * BNS) * 4;
    static constexpr int QPP = KC / 4;                      // float4 per halo pixel
    static constexpr int XITEMS = HP * QPP;
    static constexpr int XSLOTS = (XITEMS + 255) / 256;
    static constexpr int WITEMS = WROWS * (BN / 4);
    static constexpr int WSLOTS = (WITEMS + 255) / 256;
    static constexpr int NSTEP = KS * KS * (KC / 4);
    static constexpr int OCC = BN >= 64 ? 2 : (BN >= 32 ? 3 : 4);  // blocks per CU (LDS-limited)
    static_assert((XS_FLOATS * 4) % 16 == 0, "weight slab must start 16-B aligned");
};

template <int BN, int KS, int KC>
__global__ __launch_bounds__(256, (Cfg2<BN, KS, KC>::OCC)) void conv_mfma_f32_v2_kernel(
    const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
    float *__restrict__ y, int N, int H, int W, int Cin, int Cout, float wscale, int act,
    int tiles_x, int tiles_y, int ntiles, int tiles_per_block) {
    using C = Cfg2<BN, KS, KC>;
    constexpr int NR = BN / 16;
    constexpr int PAD = KS / 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *xs = smem;
    float *ws = smem + C::XS_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, kk = lane >> 4;
    const int n0 = blockIdx.y * BN;
    const int vb = (int)sq_xcd_remap(blockIdx.x, gridDim.x);
    const int t_begin = vb * tiles_per_block;
    const int t_end = min(t_begin + tiles_per_block, ntiles);
    if (t_begin >= t_end) return;
    const int nchunk = Cin / KC;
    const int nitems = (t_end - t_begin) * nchunk;
    const bool restage_w = nchunk > 1;

    float4 xr[C::XSLOTS];
    float4 wr[C::WSLOTS];

    // Buffer resources: out-of-range offsets read as 0 / drop the store, so image borders,
    // ragged tiles and the "idx >= items" tail need no branches (hipcc otherwise wraps every
    // predicated load in its own s_cbranch_execz block and the loads stop overlapping).
    // All descriptor inputs are kernel arguments => provably wave-uniform (no waterfall loop).
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(x), 0, (int)((size_t)N * H * W * Cin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(w), 0, KS * KS * Cin * Cout * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(
        y, 0, (int)((size_t)N * H * W * Cout * 4), 0x00020000);
    constexpr unsigned OOB = 0x80000000u;   // tensors are < 2 GiB (checked on the host)

    // per-thread, tile-independent part of the halo addresses
    int xrel[C::XSLOTS], xpy[C::XSLOTS], xpx[C::XSLOTS];
#pragma unroll
    for (int sl = 0; sl < C::XSLOTS; ++sl) {
        const int idx = tid + sl * 256;
        const int pix = idx / C::QPP, q = idx % C::QPP;
        xpy[sl] = pix / C::HALO_W;
        xpx[sl] = pix % C::HALO_W;
        xrel[sl] = idx < C::XITEMS ? ((xpy[sl] * W + xpx[sl]) * Cin + q * 4) * 4 : (int)OOB;
    }
    int wrel[C::WSLOTS];
#pragma unroll
    for (int sl = 0; sl < C::WSLOTS; ++sl) {
        const int idx = tid + sl * 256;
        const int r = idx / (BN / 4), q4 = idx % (BN / 4);
        const int tap = r / KC, c = r % KC;
        const int co = n0 + q4 * 4;
        wrel[sl] = (idx < C::WITEMS && co < Cout) ? ((tap * Cin + c) * Cout + co) * 4 : (int)OOB;
    }

    // ---- issue the global loads of one work item (tile, chunk) into registers -------------
    auto issue = [&](int tile, int cc, bool want_w) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
        const int x0 = tx * TW - PAD, y0 = ty * TH - PAD;
        const int base = (((n * H + y0) * W + x0) * Cin + cc) * 4;     // may be "negative": wraps back
#pragma unroll
        for (int sl = 0; sl < C::XSLOTS; ++sl) {
            const bool inb = (unsigned)(y0 + xpy[sl]) < (unsigned)H && (unsigned)(x0 + xpx[sl]) < (unsigned)W &&
                             xrel[sl] != (int)OOB;
            const unsigned off = inb ? (unsigned)(base + xrel[sl]) : OOB;
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, off, 0, 0);
            xr[sl] = *reinterpret_cast<const float4 *>(&v);
        }
        if (want_w) {
            const int wbase = cc * Cout * 4;
#pragma unroll
            for (int sl = 0; sl < C::WSLOTS; ++sl) {
                const unsigned off = wrel[sl] != (int)OOB ? (unsigned)(wbase + wrel[sl]) : OOB;
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, off, 0, 0);
                wr[sl] = *reinterpret_cast<const float4 *>(&v);
            }
        }
    };
    // ---- registers -> LDS (same index map) ----------------------------------------------------
    auto commit = [&](bool want_w) {
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
        if (want_w) {
#pragma unroll
            for (int sl = 0; sl < C::WSLOTS; ++sl) {
                const int idx = tid + sl * 256;
                if (idx < C::WITEMS) {
                    const int r = idx / (BN / 4), q4 = idx % (BN / 4);
                    float4 v = wr[sl];
                    v.x *= wscale; v.y *= wscale; v.z *= wscale; v.w *= wscale;
                    *reinterpret_cast<float4 *>(ws + r * C::BNS + q4 * 4) = v;
                }
            }
        }
    };

    f32x4 acc[4][NR];
    const float *xb_lds = xs + ((4 * wv) * C::HALO_W + li) * C::PS + kk;
    const float *wa_lds = ws + kk * C::BNS + li;

    auto load_frag = [&](int st, float (&a)[NR], float (&b)[4]) {
        const int tap = st / (KC / 4), s = st % (KC / 4);
        const int ky = tap / KS, kx = tap % KS;
#pragma unroll
        for (int nb = 0; nb < NR; ++nb) a[nb] = wa_lds[(tap * KC + s * 4) * C::BNS + nb * 16];
#pragma unroll
        for (int r = 0; r < 4; ++r) b[r] = xb_lds[((r + ky) * C::HALO_W + kx) * C::PS + s * 4];
    };

    // activation as two selects (no per-element scalar branches in the store tail):
    //   v > 0 ? v : (relu ? +0 : v * slope),  slope = 1 (none) or 0.2 (leaky)
    const float slope = act == SQ_ACT_LEAKY ? 0.2f : 1.0f;
    const bool is_relu = act == SQ_ACT_RELU;
    auto actf = [&](float v) {
        const float neg = is_relu ? 0.0f : v * slope;
        return v > 0.0f ? v : neg;
    };

    auto epilogue = [&](int tile) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
        const int gx = tx * TW + li;
#pragma unroll
        for (int nb = 0; nb < NR; ++nb) {
            const int co = n0 + nb * 16 + 4 * kk;
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bias && co < Cout) bv = *reinterpret_cast<const float4 *>(bias + co);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gy = ty * TH + 4 * wv + r;
                f32x4 o;
                o[0] = actf(bias ? acc[r][nb][0] + bv.x : acc[r][nb][0]);
                o[1] = actf(bias ? acc[r][nb][1] + bv.y : acc[r][nb][1]);
                o[2] = actf(bias ? acc[r][nb][2] + bv.z : acc[r][nb][2]);
                o[3] = actf(bias ? acc[r][nb][3] + bv.w : acc[r][nb][3]);
                const bool ok = gy < H && gx < W && co < Cout;
                const unsigned off = ok ? (unsigned)((((n * H + gy) * W + gx) * Cout + co) * 4) : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(
                    *reinterpret_cast<__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned *>(&o),
                    yrsrc, off, 0, 0);
                acc[r][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};      // ready for the next tile
            }
        }
    };

    issue(t_begin, 0, true);
    commit(true);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int nb = 0; nb < NR; ++nb) acc[r][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    int tile = t_begin, chunk = 0;
    for (int it = 0; it < nitems; ++it) {
        // next work item
        int ntile = tile, nchk = chunk + 1;
        if (nchk == nchunk) { nchk = 0; ntile = tile + 1; }
        const bool has_next = it + 1 < nitems;
        if (has_next) issue(ntile, nchk * KC, restage_w);

        // ---- MFMA phase: NSTEP steps; the ds_reads of step s+1 are issued BEFORE the MFMAs
        // of step s (sched_barrier fences stop hipcc sinking them back to just-in-time) ----------
        {
            float a0[NR], b0[4], a1[NR], b1[4];
            load_frag(0, a0, b0);
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
        }
        // ---- stage the next item, THEN store this tile: the stores drain under the next MFMA
        // phase instead of being waited for together with the prefetch (vmcnt is in-order) -------
        if (has_next) {
            __syncthreads();            // every wave is done reading this item's LDS image
            commit(restage_w);
        }
        if (chunk == nchunk - 1) epilogue(tile);
        if (has_next) __syncthreads();
        tile = ntile;
        chunk = nchk;
    }
}

template <int BN, int KS, int KC>
int launch_v2(const float *x, const float *w, const float *bias, float *y, int N, int H, int W, int Cin,
              int Cout, float wscale, int act, hipStream_t st) {
    using C = Cfg2<BN, KS, KC>;
    static bool attr_set = false;
    auto kern = conv_mfma_f32_v2_kernel<BN, KS, KC>;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES) != hipSuccess) {
            sq_set_error("conv_mfma_f32_v2: cannot reserve %d bytes of LDS", C::LDS_BYTES);
            return SQ_ELAUNCH;
        }
        attr_set = true;
    }
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int ntiles = tiles_x * tiles_y * N;
    const int gy = (Cout + BN - 1) / BN;
    // persistent grid: about OCC resident blocks per CU over both grid dimensions
    int want = (256 * C::OCC + gy - 1) / gy;
    if (want < 1) want = 1;
    int tpb = (ntiles + want - 1) / want;
    if (tpb < 1) tpb = 1;
    const int gx = (ntiles + tpb - 1) / tpb;
    hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(256), C::LDS_BYTES, st, x, w, bias, y, N, H, W, Cin, Cout,
                       wscale, act, tiles_x, tiles_y, ntiles, tpb);
    return sq_check_launch("sq_conv2d_nhwc_fwd_f32(v2)");
}

template <int KS, int KC>
int dispatch_bn(const float *x, const float *w, const float *bias, float *y, int N, int H, int W, int Cin,
                int Cout, float wscale, int act, hipStream_t st) {
    if (Cout >= 64) return launch_v2<64, KS, KC>(x, w, bias, y, N, H, W, Cin, Cout, wscale, act, st);
    if (Cout > 16) return launch_v2<32, KS, KC>(x, w, bias, y, N, H, W, Cin, Cout, wscale, act, st);
    return launch_v2<16, KS, KC>(x, w, bias, y, N, H, W, Cin, Cout, wscale, act, st);
}

}  // namespace

// internal entry used by sq_conv2d_nhwc_fwd_f32's dispatcher (sq_conv_f32.hip)
int sq_conv_mfma_v2(const float *x, const float *w, const float *bias, float *y, int N, int H, int W,
                    int Cin, int Cout, int K, float wscale, int act, hipStream_t st) {
    if (Cin % 16 == 0)
        return K == 3 ? dispatch_bn<3, 16>(x, w, bias, y, N, H, W, Cin, Cout, wscale, act, st)
                      : dispatch_bn<1, 16>(x, w, bias, y, N, H, W, Cin, Cout, wscale, act, st);
    return K == 3 ? dispatch_bn<3, 8>(x, w, bias, y, N, H, W, Cin, Cout, wscale, act, st)
                  : dispatch_bn<1, 8>(x, w, bias, y, N, H, W, Cin, Cout, wscale, act, st);
}
