// 3x3 SAME convolution of a BATCH OF SMALL IMAGES (4x4 or 8x8 pixels, 256 .. 512 channels) on bf16 tensors, gfx950: the
// 4x4 / 8x8 levels of generator_network / discriminator_network (sequitr/networks/gan.py:149-316 at start_size, round 4).
//
// Why a kernel of its own.  conv_mfma_bf16_kernel walks these layers as a mosaic of 16x16-pixel tiles x 16-channel blocks:
// every 16-channel block re-reads the tile's input chunk from L2 (32 blocks per tile at 512 outputs) and every tile re-reads
// the whole filter -- 367 MB of L2 traffic for a 512 -> 512 layer on 64 8x8 images whose operands are 9 MB, and five LDS
// fragment reads per four MFMAs (2.5x the LDS rate the matrix pipe needs).  Measured (tools/r04_mosaic_sweep.py): 14 - 38 us per
// launch whatever the block width or the split, for 2 - 8 us of matrix work.
//
// Here a block owns 256 output pixels = G whole images (16 4x4 images or 4 8x8 images) x 64 output channels:
//   * the images sit in LDS with a one-pixel ZERO border each ((h+2) x (w+2) pixels, written once), so a tap is a constant
//     byte offset, there is no halo to fetch and no mosaic arithmetic: the global loads are the images' own contiguous pixels;
//   * a wave multiplies 4 pixel groups x 4 channel groups per tap: 8 fragment reads per 16 MFMAs (the LDS rate the pipe needs),
//     the input chunk is read by Cout / 64 blocks instead of Cout / 16, the filter by N / G blocks instead of one per tile;
//   * the reduction over the input-channel chunks is split over gridDim.z where the launch would otherwise leave most CUs idle
//     (f32 partial slices, added in order by conv_splitk_finish_bf16_kernel's layout-compatible finish below).
// Same operand rounding and f32 accumulation as the mosaic kernel (another order of additions: equal to f32 rounding).
#include "sq_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int KC = 32;                                          // input channels per chunk (the pack's chunk width at Cin % 32 == 0)
constexpr int KP = 288;                                         // packed k per chunk and output channel: 9 taps x 32 channels
constexpr int PSB = 96;                                         // LDS pixel stride in bytes (64 of data): conflict-free b128 reads
constexpr int WROWB = 608;                                      // LDS filter row stride in bytes (576 of data)
constexpr int BN = 64, NR = BN / 16;
constexpr unsigned OOB = 0x80000000u;

enum { SM_PLAIN = 0, SM_GATE = 1, SM_SPLITK = 2 };

template <int HW, int FORM>
__global__ __launch_bounds__(256, HW == 8 ? 2 : 1) void conv_small_bf16_kernel(
    const __bf16 *__restrict__ x, const __bf16 *__restrict__ wp, const float *__restrict__ bias,
    const __bf16 *__restrict__ gate, __bf16 *__restrict__ y, float *__restrict__ sk_ws, int Nimg, int Cin, int Cout, int act,
    float gscale, int chunks_per_block) {
    constexpr int PPI = HW * HW;                                // pixels per image
    constexpr int G = 256 / PPI;                                // images per block
    constexpr int PW = HW + 2;                                  // padded side
    constexpr int XPIX = G * PW * PW;
    constexpr int XS_BYTES = XPIX * PSB;
    constexpr int XSLOTS = 4;                                   // 256 pixels x 4 sixteen-byte quarters / 256 threads
    constexpr int WQ = KP / 8;                                  // 36 sixteen-byte items per filter row
    constexpr int WSLOTS = (BN * WQ + 255) / 256;               // 9
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *xs = smem;
    unsigned char *ws = smem + XS_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, kg = lane >> 4;
    const int img0 = blockIdx.x * G, n0 = blockIdx.y * BN;
    const int chunk0 = blockIdx.z * chunks_per_block;
    const int nchunk_all = Cin / KC;

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16 *>(x), 0, (int)((size_t)Nimg * PPI * Cin * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16 *>(wp), 0, (int)((size_t)nchunk_all * Cout * KP * 2), 0x00020000);

    // the images' zero borders (and everything else) once; the commits below only ever write interior pixels
    for (int i = tid * 16; i < XS_BYTES; i += 256 * 16) *reinterpret_cast<uint4 *>(xs + i) = make_uint4(0u, 0u, 0u, 0u);

    unsigned xg[XSLOTS];                                        // global byte offset of the slot's 16 bytes at chunk 0 (or OOB)
    int xl[XSLOTS];                                             // its LDS byte offset
#pragma unroll
    for (int sl = 0; sl < XSLOTS; ++sl) {
        const int idx = tid + sl * 256;
        const int pix = idx >> 2, q = idx & 3;
        const int g = pix / PPI, p = pix % PPI, py = p / HW, px = p % HW;
        const int img = img0 + g;
        xg[sl] = img < Nimg ? (unsigned)((((size_t)img * PPI + p) * Cin + q * 8) * 2) : OOB;
        xl[sl] = ((g * PW + py + 1) * PW + px + 1) * PSB + q * 16;
    }
    unsigned wg[WSLOTS];
    int wl[WSLOTS];
#pragma unroll
    for (int sl = 0; sl < WSLOTS; ++sl) {
        const int idx = tid + sl * 256;
        const int row = idx / WQ, q = idx % WQ;
        wg[sl] = (idx < BN * WQ && n0 + row < Cout) ? (unsigned)(((n0 + row) * KP + q * 8) * 2) : OOB;
        wl[sl] = row * WROWB + q * 16;
    }
    uint4 xr[XSLOTS], wr[WSLOTS];
    auto issue = [&](int chunk) {
#pragma unroll
        for (int sl = 0; sl < XSLOTS; ++sl) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, xg[sl] == OOB ? OOB : xg[sl] + (unsigned)(chunk * KC * 2), 0, 0);
            xr[sl] = *reinterpret_cast<const uint4 *>(&v);
        }
        const unsigned wbase = (unsigned)chunk * (unsigned)(Cout * KP * 2);
#pragma unroll
        for (int sl = 0; sl < WSLOTS; ++sl) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wg[sl] == OOB ? OOB : wbase + wg[sl], 0, 0);
            wr[sl] = *reinterpret_cast<const uint4 *>(&v);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int sl = 0; sl < XSLOTS; ++sl) *reinterpret_cast<uint4 *>(xs + xl[sl]) = xr[sl];
#pragma unroll
        for (int sl = 0; sl < WSLOTS; ++sl)
            if (tid + sl * 256 < BN * WQ) *reinterpret_cast<uint4 *>(ws + wl[sl]) = wr[sl];
    };

    // fragment addressing.  Row r of wave wv is the pixel group (4 wv + r): 16 consecutive pixels of the block's 256, lane li
    // one of them; its 3x3 window starts at the padded pixel (py, px) of its image (tap (ky, kx) = + (ky PW + kx) pixels)
    const unsigned char *xb[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int pix = (4 * wv + r) * 16 + li;
        const int g = pix / PPI, p = pix % PPI, py = p / HW, px = p % HW;
        xb[r] = xs + ((g * PW + py) * PW + px) * PSB + kg * 16;
    }
    const unsigned char *wa = ws + li * WROWB + kg * 16;

    f32x4 acc[4][NR];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int nb = 0; nb < NR; ++nb) acc[r][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    issue(chunk0);
    __builtin_amdgcn_s_waitcnt(0x0F70);                         // vmcnt(0)
    __syncthreads();                                            // the zero fill is complete
    commit();
    __syncthreads();
    for (int c = 0; c < chunks_per_block; ++c) {
        const bool has_next = c + 1 < chunks_per_block;
        if (has_next) issue(chunk0 + c + 1);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int toff = ((tap / 3) * PW + tap % 3) * PSB;
            bf16x8 a[NR], b[4];
#pragma unroll
            for (int nb = 0; nb < NR; ++nb) a[nb] = *reinterpret_cast<const bf16x8 *>(wa + nb * 16 * WROWB + tap * 64);
#pragma unroll
            for (int r = 0; r < 4; ++r) b[r] = *reinterpret_cast<const bf16x8 *>(xb[r] + toff);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int nb = 0; nb < NR; ++nb)
                    acc[r][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[nb], b[r], acc[r][nb], 0, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);                     // the prefetch has landed (stated outside the branch)
        if (has_next) {
            __syncthreads();
            commit();
            __syncthreads();
        }
    }

    // ---- epilogue: lane (li, kg) holds output channels n0 + 16 nb + 4 kg .. + 3 of pixel li of each of its four groups --------
    const float slope = act == SQ_ACT_LEAKY ? 0.2f : 1.0f;
    const bool is_relu = act == SQ_ACT_RELU;
    const size_t total = (size_t)Nimg * PPI * Cout;
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(y, 0, y ? (int)(total * 2) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16 *>(gate), 0, (FORM == SM_GATE && gate) ? (int)(total * 2) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t krsrc = __builtin_amdgcn_make_buffer_rsrc(
        FORM == SM_SPLITK ? sk_ws + (size_t)blockIdx.z * total : nullptr, 0, FORM == SM_SPLITK ? (int)(total * 4) : 0, 0x00020000);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int pix = (4 * wv + r) * 16 + li;
        const int img = img0 + pix / PPI, p = pix % PPI;
#pragma unroll
        for (int nb = 0; nb < NR; ++nb) {
            const int co = n0 + nb * 16 + 4 * kg;
            const bool ok = img < Nimg && co < Cout;
            const unsigned el = ok ? (unsigned)(((size_t)img * PPI + p) * Cout + co) : 0u;
            if constexpr (FORM == SM_SPLITK) {
                const float4 o = make_float4(acc[r][nb][0], acc[r][nb][1], acc[r][nb][2], acc[r][nb][3]);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(
                    __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, o), krsrc, ok ? el * 4u : OOB, 0, 0);
            } else if constexpr (FORM == SM_GATE) {
                const bf16x4 gv = __builtin_bit_cast(bf16x4, __builtin_amdgcn_raw_buffer_load_b64(grsrc, ok ? el * 2u : OOB, 0, 0));
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const __bf16 t = (__bf16)acc[r][nb][j];
                    o[j] = (float)gv[j] > 0.f ? t : (__bf16)((float)t * gscale);
                }
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(
                    __attribute__((__vector_size__(2 * sizeof(unsigned)))) unsigned, o), yrsrc, ok ? el * 2u : OOB, 0, 0);
            } else {
                float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (bias && co < Cout) bv = *reinterpret_cast<const float4 *>(bias + co);
                const float bq[4] = {bv.x, bv.y, bv.z, bv.w};
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = acc[r][nb][j] + bq[j];
                    o[j] = (__bf16)(v > 0.f ? v : (is_relu ? 0.f : v * slope));
                }
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(
                    __attribute__((__vector_size__(2 * sizeof(unsigned)))) unsigned, o), yrsrc, ok ? el * 2u : OOB, 0, 0);
            }
        }
    }
}

// y[p][c] = bf16(act(sum_s ws[s][p][c] + bias[c])), slices added in order; gate != NULL: t = bf16(sum), y = gate > 0 ? t : bf16(t * slope)
__global__ __launch_bounds__(256) void conv_small_finish_bf16_kernel(const float4 *__restrict__ ws, int S, int64_t n4, int C4,
                                                                      const float *__restrict__ bias, int act,
                                                                      const bf16x4 *__restrict__ gate, float slope,
                                                                      bf16x4 *__restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 v = ws[i];
        for (int s2 = 1; s2 < S; ++s2) {
            const float4 u = ws[(size_t)s2 * n4 + i];
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        bf16x4 o;
        if (gate) {
            const bf16x4 g = gate[i];
            const float t[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const __bf16 tb = (__bf16)t[j];
                o[j] = (float)g[j] > 0.f ? tb : (__bf16)((float)tb * slope);
            }
        } else {
            const int c = (int)(i % C4) * 4;
            const float4 b = bias ? *reinterpret_cast<const float4 *>(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            o[0] = (__bf16)sq_act(v.x + b.x, act);
            o[1] = (__bf16)sq_act(v.y + b.y, act);
            o[2] = (__bf16)sq_act(v.z + b.z, act);
            o[3] = (__bf16)sq_act(v.w + b.w, act);
        }
        y[i] = o;
    }
}

template <int HW, int FORM>
int launch_small(const __bf16 *x, const __bf16 *wp, const float *bias, const __bf16 *gate, __bf16 *y, float *sk_ws, int Nimg,
                 int Cin, int Cout, int act, float gscale, int S, hipStream_t st) {
    constexpr int G = 256 / (HW * HW), PW = HW + 2;
    constexpr int LDS = G * PW * PW * PSB + BN * WROWB;
    auto kern = conv_small_bf16_kernel<HW, FORM>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
            sq_set_error("conv_small_bf16: cannot reserve %d bytes of LDS", LDS);
            return SQ_ELAUNCH;
        }
        attr_set = true;
    }
    const int nchunk = Cin / KC;
    hipLaunchKernelGGL(kern, dim3((Nimg + G - 1) / G, (Cout + BN - 1) / BN, S), dim3(256), LDS, st, x, wp, bias, gate, y, sk_ws, Nimg,
                       Cin, Cout, act, gscale, nchunk / S);
    return sq_check_launch("sq_conv2d_small_bf16");
}

}  // namespace

// 1 when sq_conv2d_small_bf16 takes this layer: square images of 4 or 8 pixels, Cin % 32 == 0, Cout % 64 == 0
extern "C" int sq_conv2d_small_takes_bf16(int h, int w, int Cin, int Cout) {
    return (h == w && (h == 4 || h == 8) && Cin > 0 && Cin % 32 == 0 && Cout > 0 && Cout % 64 == 0) ? 1 : 0;
}

// weighted_conv2d (gan.py:61-99) of a batch of 4x4 or 8x8 images on bf16 tensors, K = 3, SAME: x (Nimg,h,h,Cin), wp the packed
// filter of sq_conv_pack_weights_bf16 (the equalised-LR scale folded in; the dgrad pack for an input gradient),
// y (Nimg,h,h,Cout) = act(conv + bias); gate != NULL: the act-gated dgrad instead (no bias: t = bf16(conv), y = gate > 0 ? t :
// bf16(t * slope(act)) -- sq_conv2d_nhwc_dgrad_actgate_bf16's two roundings).  workspace (may be NULL) / workspace_bytes: room for
// the split reduction's f32 slices (S <= 8 slices of Nimg*h*h*Cout floats); without it the reduction is not split.
extern "C" int sq_conv2d_small_bf16(const void *x, const void *wp, const float *bias, const void *gate, void *y, int Nimg, int h,
                                    int Cin, int Cout, int act, float *workspace, int64_t workspace_bytes, void *stream) {
    SQ_REQUIRE(x && wp && y && Nimg > 0, "sq_conv2d_small_bf16: null pointer / empty batch");
    SQ_REQUIRE(sq_conv2d_small_takes_bf16(h, h, Cin, Cout), "sq_conv2d_small_bf16: needs 4x4 or 8x8 images, Cin %% 32 == 0, Cout %% 64 == 0 "
               "(h=%d Cin=%d Cout=%d)", h, Cin, Cout);
    SQ_REQUIRE(act >= SQ_ACT_NONE && act <= SQ_ACT_LEAKY && (!gate || act != SQ_ACT_NONE), "sq_conv2d_small_bf16: bad activation %d", act);
    SQ_REQUIRE((size_t)Nimg * h * h * (size_t)(Cin > Cout ? Cin : Cout) * 4 < ((size_t)1 << 31), "sq_conv2d_small_bf16: tensors must be < 512 Mi elements");
    SQ_REQUIRE_ALIGNED(x); SQ_REQUIRE_ALIGNED(wp); SQ_REQUIRE_ALIGNED(y);
    if (bias) SQ_REQUIRE_ALIGNED(bias);
    if (gate) SQ_REQUIRE_ALIGNED(gate);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const __bf16 *xb = reinterpret_cast<const __bf16 *>(x), *wb = reinterpret_cast<const __bf16 *>(wp);
    const __bf16 *gb = reinterpret_cast<const __bf16 *>(gate);
    __bf16 *yb = reinterpret_cast<__bf16 *>(y);
    const int G = 256 / (h * h), nchunk = Cin / KC;
    const int64_t blocks = (int64_t)((Nimg + G - 1) / G) * ((Cout + BN - 1) / BN);
    const int64_t slice = (int64_t)Nimg * h * h * Cout * 4;
    int S = 1;
    if (workspace) {
        static const int force = [] { const char *e = getenv("SQ_SMALL_S"); return e ? atoi(e) : 0; }();   // experiment switch
        while (S < 8 && nchunk % (2 * S) == 0 && nchunk / (2 * S) >= 2 && blocks * S < 256 && slice * 2 * S <= workspace_bytes) S *= 2;
        if (force >= 1 && nchunk % force == 0 && slice * force <= workspace_bytes) S = force;
    }
    const float gscale = act == SQ_ACT_LEAKY ? 0.2f : 0.0f;
    int rc;
    if (S > 1) {
        SQ_REQUIRE_ALIGNED(workspace);
        rc = h == 4 ? launch_small<4, SM_SPLITK>(xb, wb, nullptr, nullptr, yb, workspace, Nimg, Cin, Cout, act, gscale, S, st)
                    : launch_small<8, SM_SPLITK>(xb, wb, nullptr, nullptr, yb, workspace, Nimg, Cin, Cout, act, gscale, S, st);
        if (rc) return rc;
        const int64_t n4 = slice / 16;
        int64_t nb = (n4 + 255) / 256;
        if (nb > 2048) nb = 2048;
        hipLaunchKernelGGL(conv_small_finish_bf16_kernel, dim3((unsigned)nb), dim3(256), 0, st, reinterpret_cast<const float4 *>(workspace), S,
                           n4, Cout / 4, gate ? nullptr : bias, gate ? (int)SQ_ACT_NONE : act, reinterpret_cast<const bf16x4 *>(gate), gscale,
                           reinterpret_cast<bf16x4 *>(yb));
        return sq_check_launch("sq_conv2d_small_bf16(split finish)");
    }
    if (gate)
        return h == 4 ? launch_small<4, SM_GATE>(xb, wb, nullptr, gb, yb, nullptr, Nimg, Cin, Cout, act, gscale, 1, st)
                      : launch_small<8, SM_GATE>(xb, wb, nullptr, gb, yb, nullptr, Nimg, Cin, Cout, act, gscale, 1, st);
    return h == 4 ? launch_small<4, SM_PLAIN>(xb, wb, bias, nullptr, yb, nullptr, Nimg, Cin, Cout, act, gscale, 1, st)
                  : launch_small<8, SM_PLAIN>(xb, wb, bias, nullptr, yb, nullptr, Nimg, Cin, Cout, act, gscale, 1, st);
}
