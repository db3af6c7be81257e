// bf16 NHWC convolution (3x3 / 1x1, SAME, stride 1) for gfx950 on v_mfma_f32_16x16x32_bf16:
// bf16 activations and weights, fp32 accumulation, + fp32 bias, activation, bf16 output.
// The training path of BASELINE configs 3-5 ("bf16 compute, fp32 master weights + accumulation").
//
// Same structure as sq_conv_f32_v2.hip (persistent blocks over 16x16 pixel tiles, register prefetch of
// the next work item under the MFMA phase, buffer loads/stores with out-of-range = 0 / dropped), same
// operand roles (A = weights [cout][k], B = pixels [k][pixel], so a lane ends with 4 consecutive output
// channels of one pixel -> one 8-byte bf16x4 store), but K = 32 per instruction:
//   KC = 32 (Cin % 32 == 0): one MFMA step = one tap x 32 channels;
//   KC = 16 (otherwise)    : one MFMA step = two taps x 16 channels (the 10th half-step has zero weights).
// Weights are pre-packed once per optimiser step into [chunk][cout][KP] bf16 (k contiguous per output
// channel), optionally rotated/transposed for dgrad: sq_conv_pack_weights_bf16.
//   KC = 8  (Cin % 16 == 8)  : one MFMA step = four taps x 8 channels (GAN 256x256 layers; taps 9..11 are zero).
// At levels 0-1 these kernels are HBM-bound (the matrix pipe is 16x the f32 rate); halving the bytes
// is the point of the bf16 path.
// TIO = float ("mixed"): the activations stay f32 in HBM, are rounded to bf16 (RNE) while they are staged into
// LDS, and the f32 accumulators are stored as f32 -- bf16 multiply / f32 accumulate behind an f32 graph (the GAN
// of BASELINE config 5, whose double-backward graph is built from f32 ops).
#include "sq_common.h"

#ifndef SQ_FIRST_UNROLL
#define SQ_FIRST_UNROLL 0      // 1: the single-channel first conv keeps its 144 weights in registers (155 VGPRs, occupancy 3)
#endif
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#ifndef SQ_CONV_EARLY_ISSUE
#define SQ_CONV_EARLY_ISSUE 2       // where the next item's loads are requested: 0 = top of the loop, 1 = after the commit
#endif                              // (before the epilogue), 2 = after the commit in the forms it was measured faster in

namespace {

constexpr int TH = 16, TW = 16;

__host__ __device__ constexpr int kp_for(int KS, int KC) { return ((KS * KS * KC + 31) / 32) * 32; }

// dropout fused into the epilogue (training conv_block: relu(conv2) -> dropout, unet.py:265-277): the
// element with flat NHWC index e is kept iff bit e % 4 of sq_dropout_keep4(key, e / 4, thr) is set (sq_common.h; thr in
// 16-bit units) -- the mask and the two roundings (bf16 activation, then * 1/(1-rate) -> bf16) are those of dropout_fwd_bf16_kernel, so the result
// equals the separate kernels bit for bit.  thr == 0: no dropout.
struct SqDropEpi {
    unsigned thr;
    float inv;
    unsigned seed;
    const int *step;
    float gscale = 1.f;     // factor applied where the gate passes (1: plain ReLU backward; 1/(1-rate): dropout + ReLU)
    // decoder-junction backward in the epilogue (j_g != NULL): the conv's output dM = d(merged) is not stored; with the
    // forward operands of merged = bridge(up, skip) the epilogue writes d_up in the space-to-depth layout the
    // transpose-conv gradients consume (j_g (N,H/2,W/2,4*Cout)) and d_skip (j_dskip (N,H,W,Cout)) -- what
    // sq_bridge_bwd_s2d_bf16 does in a pass of its own, same roundings
    const __bf16 *j_up = nullptr, *j_skip = nullptr;
    __bf16 *j_g = nullptr, *j_dskip = nullptr;
    int j_bridge = 0;
    // 2x2/s2 max pool of the block output in the same epilogue (pool != NULL; H, W even): the pooled tensor
    // (N,H/2,W/2,Cout) the next encoder level reads is written beside y, what sq_maxpool2x2_fwd_bf16 computes from y
    __bf16 *pool = nullptr;
    int pool_avg = 0;                                           // 1: the 2x2 AVERAGE pool instead -- bf16(((a + b) + (c + d)) * 0.25) of the stored
                                                                // bf16 values, sq_sumpool2x2_bf16's order and rounding (the GAN's discriminator blocks)
    // sign mask of a ReLU output, one bit per element in NHWC order (bit c & 7 of byte (pixel * C + c) >> 3; C % 16 == 0):
    // written by the forward conv beside y (FORM_MK), read by the dgrad that gates on that output (FORM_MG) in place
    // of the tensor itself -- 1/16 of its bytes
    unsigned char *mask = nullptr;
    const float *gate_f32 = nullptr;                            // FORM_GF (f32 tensors): (N,H,W,Cout) activation output, slope in gscale
    int gate_slope = 0;                                         // FORM_GB (bf16 tensors): `gate` is an activation output, slope in gscale
    // FORM_SK (bf16 tensors, mosaic): the reduction over the input-channel chunks is split over gridDim.z blocks; block z
    // multiplies chunks [z, z + 1) * sk_chunks and stores its raw f32 accumulators to sk_ws[z][pixel][cout] -- bias, activation,
    // gate and the bf16 rounding happen in conv_splitk_finish_bf16_kernel, which adds the slices in order
    float *sk_ws = nullptr;
    int sk_chunks = 0;
    // MOS (f32 tensors): the image the kernel tiles is a MOSAIC of mos_n small images (mos_h x mos_w, cells of pitch
    // h+1 / w+1 in a grid mos_cc wide, a zero row / column after every image standing in for the SAME padding --
    // sq_mosaic_pack_f32's layout) that is never materialised: loads, the gate and the stores address the compact
    // (mos_n, mos_h, mos_w, C) tensors directly, separator pixels read as zero and are not stored
    int mos_h = 0, mos_w = 0, mos_cc = 0, mos_n = 0;
    // FORM_FP: the block input y1 = relu(conv3x3(f_x, f_w) + f_b) of the single-channel f32 image f_x (N,H,W,1) is evaluated in
    // the block for the tile's 18 x 18 halo (conv_first_bf16_kernel's chain and rounding), written to LDS as the conv's input and,
    // for the tile's own pixels, to f_y1 (N,H,W,16) with its sign mask f_mask -- the first conv of down0 never runs as a launch
    const float *f_x = nullptr, *f_w = nullptr, *f_b = nullptr;
    __bf16 *f_y1 = nullptr;
    unsigned char *f_mask = nullptr;
    unsigned mos_mh = 0, mos_mw = 0;                            // ceil(2^16 / (h+1)), ceil(2^16 / (w+1)): q = (v * m) >> 16, exact for
                                                                // v < 2^13 at pitches <= 9 (set by the entry point)
    // FORM_PN: weighted_conv2d's pixel norm (gan.py:49-51, 96-97) in the epilogue.  One block holds ALL Cout <= 64 channels of
    // a pixel (gridDim.y == 1): pn_y (N,H,W,Cout) = bf16(v * r), r = 1 / sqrt(mean_c v^2 + pn_eps) of the STORED values
    // v = bf16(act(conv + bias)) -- what sq_pixelnorm_fwd_bf16 computes from y (its sum of squares is added in another order:
    // equal to f32 rounding of r).  y itself is written when pn_store_y (a backward pass will read it).
    __bf16 *pn_y = nullptr;
    float pn_eps = 0.f;
    int pn_store_y = 1;
};

template <int BN, int KS, int KC>
struct CfgB {
    static constexpr int HALO_W = TW + KS - 1;
    static constexpr int HP = HALO_W * (TH + KS - 1);
    static constexpr int PSB = KC == 8 ? 16 : (KC == 16 ? 32 : 96);   // pixel stride in BYTES (conflict-free b128 reads)
    static constexpr int KP = kp_for(KS, KC);                 // padded k per chunk
    static constexpr int WROWB = KP * 2 + ((KP * 2 / 16) % 16 == 6 ? 0 : ((6 - (KP * 2 / 16) % 16 + 16) % 16) * 16);
    static constexpr int XS_BYTES = HP * PSB;
    static constexpr int WS_BYTES = BN * WROWB;
    static constexpr int LDS_BYTES = XS_BYTES + WS_BYTES;
    static constexpr int XQ = KC / 8;                          // 16-byte items per halo pixel
    static constexpr int XITEMS = HP * XQ;
    static constexpr int XSLOTS = (XITEMS + 255) / 256;
    static constexpr int WQ = KP / 8;                          // 16-byte items per weight row
    static constexpr int WITEMS = BN * WQ;
    static constexpr int WSLOTS = (WITEMS + 255) / 256;
    static constexpr int NSTEP = KP / 32;
    static_assert(XS_BYTES % 16 == 0, "weight slab must start 16-B aligned");
    static_assert((WROWB / 16) % 16 == 6, "weight row stride must be 6 slots mod 16");
};

// packed weights: wp[chunk][co][k], k = tap*KC + c (zero padded to KP); source HWIO f32 (K,K,Cin,Cout).
// transform != 0: the dgrad filter, i.e. source element w[K-1-ky][K-1-kx][co][ci] (roles of Cin/Cout
// swapped: the packed tensor then has `Cin` = original Cout input channels, `Cout` = original Cin).
__global__ __launch_bounds__(256) void pack_weights_bf16_kernel(const float *__restrict__ w, __bf16 *__restrict__ wp,
                                                                 int K, int Cin, int Cout, int KC, int KP,
                                                                 float wscale, int transform) {
    const int64_t total = (int64_t)(Cin / KC) * Cout * KP;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int k = (int)(i % KP);
        const int co = (int)((i / KP) % Cout);
        const int chunk = (int)(i / ((int64_t)KP * Cout));
        float v = 0.f;
        if (k < K * K * KC) {
            const int tap = k / KC, c = chunk * KC + k % KC;
            if (!transform) {
                v = w[((size_t)tap * Cin + c) * Cout + co];
            } else {
                const int ky = tap / K, kx = tap % K;
                // original tensor is (K,K,Cout_packed_as_in?..): here Cin/Cout are those of the PACKED conv
                v = w[((size_t)((K - 1 - ky) * K + (K - 1 - kx)) * Cout + co) * Cin + c];
            }
            v *= wscale;
        }
        wp[i] = (__bf16)v;
    }
}

// All the packs (and plain bf16 casts) of one optimiser step in ONE launch.  table: n_entries x 8 int32
// {src offset (floats), dst offset (bf16), K, Cin, Cout of the PACKED conv, transform, first item, kind};
// kind 0 = pack as pack_weights_bf16_kernel, kind 1 = plain cast of Cin*Cout*K*K elements.  Items are
// numbered across entries; a thread finds its entry by scanning the (tiny) table in LDS.
__global__ __launch_bounds__(256) void pack_weights_multi_bf16_kernel(const float *__restrict__ base,
                                                                       __bf16 *__restrict__ out,
                                                                       const int *__restrict__ table, int n_entries,
                                                                       int total, const float *__restrict__ scales) {
    __shared__ int tb[128 * 8];
    for (int i = threadIdx.x; i < n_entries * 8; i += 256) tb[i] = table[i];
    __syncthreads();
    // a thread makes 8 consecutive packed elements (one 16-byte store; entries start and end on multiples of 8: KP is a
    // multiple of 32, channel counts of 8).  Which of (k-group, co) runs along the lanes follows the SOURCE layout:
    // forward packs read (tap, c, co) with co contiguous -> co along the lanes (each of the 8 loads is a coalesced row);
    // dgrad packs read (tap, co, c) with c contiguous -> the k-group along the lanes (32 contiguous source bytes each).
    // With one element per thread and k along the lanes, a forward pack read one float per 128-byte line.
    for (int u = blockIdx.x * 256 + threadIdx.x; u < total / 8; u += gridDim.x * 256) {
        const int i = u * 8;
        int lo = 0, hi = n_entries - 1;                        // last entry whose first item <= i
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (tb[mid * 8 + 6] <= i) lo = mid; else hi = mid - 1;
        }
        const int *e = tb + lo * 8;
        const float *w = base + e[0];
        __bf16 *wp = out + e[1];
        const int K = e[2], Cin = e[3], Cout = e[4], transform = e[5], j = i - e[6];
        typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
        bf16x8_t o;
        if (e[7] == 1) {
#pragma unroll
            for (int t = 0; t < 8; ++t) o[t] = (__bf16)w[j + t];
            *reinterpret_cast<bf16x8_t *>(wp + j) = o;
            continue;
        }
        const int KC = Cin % 32 == 0 ? 32 : (Cin % 16 == 0 ? 16 : 8), KP = kp_for(K, KC), KG = KP / 8;
        const int q = j / 8;
        int kg, co;
        if (transform) { kg = q % KG; co = (q / KG) % Cout; }
        else { co = q % Cout; kg = (q / Cout) % KG; }
        const int chunk = q / (KG * Cout), k0 = kg * 8;
        const float sc = scales ? scales[lo] : 1.0f;           // the equalised-LR factor, as pack_weights_bf16_kernel
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int k = k0 + t;
            float v = 0.f;
            if (k < K * K * KC) {
                const int tap = k / KC, c = chunk * KC + k % KC;
                if (!transform) {
                    v = w[((size_t)tap * Cin + c) * Cout + co];
                } else {
                    const int ky = tap / K, kx = tap % K;
                    v = w[((size_t)((K - 1 - ky) * K + (K - 1 - kx)) * Cout + co) * Cin + c];
                }
            }
            if (scales) v *= sc;
            o[t] = (__bf16)v;
        }
        *reinterpret_cast<bf16x8_t *>(wp + ((size_t)chunk * Cout + co) * KP + k0) = o;
    }
}

// FORM: epilogue variants, each its own instantiation so that the plain kernel keeps its register budget
//   1 JN: decoder-junction backward (SqDropEpi::j_*)        2 PL: max-pooled copy of the output (SqDropEpi::pool)
//   3 MK: sign mask of the output written beside it (mask)  4 MG: gate read from such a mask instead of a tensor
//   5 GF: f32 tensors (the GAN's mixed form): the result leaves through the backward of the activation whose output
//         `gate_f32` is -- dx = gate > 0 ? v : v * gscale -- the act_bwd pass that followed this dgrad
//   6 GB: the same on bf16 tensors (the GAN's bf16-storage form): t = bf16(v), dx = gate > 0 ? t : bf16(t * gscale) -- the two
//         roundings of the dgrad -> sq_act_bwd_bf16 pair it replaces
//   7 SK: split-K partial sums (SqDropEpi::sk_ws): the small-image levels of the GAN are 4 - 8 mosaic tiles x Cout / 16 blocks,
//         each walking all 16 channel chunks alone (one block per CU at best, 22 - 28 us of exposed round trips)
//   8 FP: FORM_PL whose input is made in the block from the single-channel image (SqDropEpi::f_*): conv_block of down0 with
//         its max pool as ONE launch (16 channels)
//   9 PN: pixel norm of the output in the same epilogue (SqDropEpi::pn_*): the generator's conv -> leaky -> pixel_norm
enum { FORM_PLAIN = 0, FORM_JN = 1, FORM_PL = 2, FORM_MK = 3, FORM_MG = 4, FORM_GF = 5, FORM_GB = 6, FORM_SK = 7, FORM_FP = 8, FORM_PN = 9 };
template <int BN, int KS, int KC, typename TIO, int FORM = FORM_PLAIN, bool MOS = false>
__global__ __launch_bounds__(256, 2) void conv_mfma_bf16_kernel(
    const TIO *__restrict__ x, const __bf16 *__restrict__ wp, const float *__restrict__ bias,
    TIO *__restrict__ y, int N, int H, int W, int Cin, int Cout, int act, int tiles_x, int tiles_y,
    int ntiles, int tiles_per_block, const __bf16 *__restrict__ gate, SqDropEpi drop) {
    using C = CfgB<BN, KS, KC>;
    constexpr int NR = BN / 16, PAD = KS / 2;
    constexpr bool FP = FORM == FORM_FP;
    static_assert(!FP || (BN == 16 && KS == 3 && KC == 16 && sizeof(TIO) == 2 && !MOS), "FORM_FP is the 16-channel level-0 block");
    constexpr bool JN = FORM == FORM_JN, PL = FORM == FORM_PL || FP, MK = FORM == FORM_MK, MG = FORM == FORM_MG;
    constexpr bool GF = FORM == FORM_GF, GB = FORM == FORM_GB, SK = FORM == FORM_SK, PN = FORM == FORM_PN;
    static_assert(!PN || (sizeof(TIO) == 2 && !MOS), "FORM_PN: bf16 tensors, plain addressing");
    constexpr bool F32IO = sizeof(TIO) == 4;                    // f32 activations in HBM, bf16 in LDS
    constexpr int ES = (int)sizeof(TIO), XV = F32IO ? 2 : 1;    // 16-byte loads per 8-channel LDS item
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *xs = smem;
    unsigned char *ws = smem + C::XS_BYTES;
    float *xin = reinterpret_cast<float *>(smem + C::LDS_BYTES);   // FP: the tile's 20 x 20 image patch
    constexpr int IN_W = TW + 4, IN_FLOATS = IN_W * IN_W;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, kg = lane >> 4;
    const int n0 = blockIdx.y * BN;
    const int vb = (int)sq_xcd_remap(blockIdx.x, gridDim.x);
    // tiles_per_block < 0: block b walks tiles b, b + G, b + 2G, ... (the G tiles in flight are a contiguous run of the image)
    const bool il = tiles_per_block < 0;
    const int tstride = il ? (int)gridDim.x : 1;
    const int t_begin = il ? vb : vb * tiles_per_block;
    const int t_count = il ? (ntiles - vb + tstride - 1) / tstride : min(tiles_per_block, ntiles - t_begin);
    if (t_begin >= ntiles || t_count <= 0) return;
    const int nchunk_all = Cin / KC;
    const int nchunk = SK ? drop.sk_chunks : nchunk_all;        // chunks this block reduces over ...
    const int chunk0 = SK ? (int)blockIdx.z * drop.sk_chunks : 0;   // ... starting here
    const int nitems = t_count * nchunk;
    const bool restage_w = nchunk > 1;

    const size_t io_pixels = MOS ? (size_t)drop.mos_n * drop.mos_h * drop.mos_w : (size_t)N * H * W;   // pixels behind x / y
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<TIO *>(x), 0, (int)(io_pixels * Cin * ES), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16 *>(wp), 0, (int)((size_t)nchunk_all * Cout * C::KP * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(
        y, 0, (int)(io_pixels * Cout * ES), 0x00020000);
    // dgrad fused with the upstream ReLU's backward: outputs pass only where gate (N,H,W,Cout) > 0
    const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16 *>(gate), 0, gate ? (int)(io_pixels * Cout * 2) : 0, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t mrsrc = __builtin_amdgcn_make_buffer_rsrc(
        drop.mask, 0, ((MK || MG) && drop.mask) ? (int)((size_t)N * H * W * Cout / 8) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc(
        drop.pool, 0, (PL && drop.pool) ? (int)((size_t)N * (H >> 1) * (W >> 1) * Cout * 2) : 0, 0x00020000);

    const __amdgpu_buffer_rsrc_t firsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(drop.f_x), 0, FP ? (int)((size_t)N * H * W * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t fyrsrc = __builtin_amdgcn_make_buffer_rsrc(
        drop.f_y1, 0, (FP && drop.f_y1) ? (int)((size_t)N * H * W * 16 * 2) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t fmrsrc = __builtin_amdgcn_make_buffer_rsrc(
        drop.f_mask, 0, (FP && drop.f_mask) ? (int)((size_t)N * H * W * 2) : 0, 0x00020000);
    float inr[2];                                               // FP: 400 patch pixels over 256 threads

    // MOS: compact pixel index of mosaic pixel (gy, gx), or -1 for a separator / padding cell / outside
    auto mos_pixel = [&](int gy, int gx) {
        if ((unsigned)gy >= (unsigned)H || (unsigned)gx >= (unsigned)W) return -1;
        const int cc = (int)(((unsigned)gx * drop.mos_mw) >> 16), xx = gx - cc * (drop.mos_w + 1);
        const int rr = (int)(((unsigned)gy * drop.mos_mh) >> 16), yy = gy - rr * (drop.mos_h + 1);
        const int im = rr * drop.mos_cc + cc;
        return (xx < drop.mos_w && yy < drop.mos_h && im < drop.mos_n) ? (im * drop.mos_h + yy) * drop.mos_w + xx : -1;
    };

    uint4 xr[C::XSLOTS][XV], wr[C::WSLOTS];
    int xrel[C::XSLOTS], xpy[C::XSLOTS], xpx[C::XSLOTS];
    int xpix[MOS ? C::XSLOTS : 1];                              // MOS: compact pixel of every slot, recomputed per TILE (chunk 0)
#pragma unroll
    for (int sl = 0; sl < C::XSLOTS; ++sl) {
        const int idx = tid + sl * 256;
        const int pix = idx / C::XQ, q = idx % C::XQ;
        xpy[sl] = pix / C::HALO_W;
        xpx[sl] = pix % C::HALO_W;
        xrel[sl] = idx < C::XITEMS ? (MOS ? q * 8 * ES : ((xpy[sl] * W + xpx[sl]) * Cin + q * 8) * ES) : (int)OOB;
    }
    int wrel[C::WSLOTS];
#pragma unroll
    for (int sl = 0; sl < C::WSLOTS; ++sl) {
        const int idx = tid + sl * 256;
        const int row = idx / C::WQ, q = idx % C::WQ;
        wrel[sl] = (idx < C::WITEMS && n0 + row < Cout) ? (((n0 + row) * C::KP) + q * 8) * 2 : (int)OOB;
    }

    auto issue = [&](int tile, int lchunk, bool want_w) {
        const int chunk = lchunk + chunk0;                      // lchunk counts within this block's share of the reduction
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
        const int x0 = tx * TW - PAD, y0 = ty * TH - PAD;
        const int base = (((n * H + y0) * W + x0) * Cin + chunk * KC) * ES;
        if constexpr (FP) {                                     // the 20 x 20 image patch around the tile (halo of the halo)
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                const int idx = tid + sl * 256;
                const int py = idx / IN_W, px = idx % IN_W;
                const bool inb = idx < IN_FLOATS && (unsigned)(y0 - 1 + py) < (unsigned)H && (unsigned)(x0 - 1 + px) < (unsigned)W;
                inr[sl] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    firsrc, inb ? (unsigned)(((n * H + y0 - 1 + py) * W + x0 - 1 + px) * 4) : OOB, 0, 0));
            }
        }
#pragma unroll
        for (int sl = 0; sl < (FP ? 0 : C::XSLOTS); ++sl) {
            bool inb;
            unsigned off;
            if constexpr (MOS) {
                if (lchunk == 0) xpix[sl] = xrel[sl] != (int)OOB ? mos_pixel(y0 + xpy[sl], x0 + xpx[sl]) : -1;
                const int px = xpix[sl];                        // items of a tile are issued chunk 0 first
                inb = px >= 0;
                off = (unsigned)((px * Cin + chunk * KC) * ES + xrel[sl]);
            } else {
                inb = (unsigned)(y0 + xpy[sl]) < (unsigned)H && (unsigned)(x0 + xpx[sl]) < (unsigned)W && xrel[sl] != (int)OOB;
                off = (unsigned)(base + xrel[sl]);
            }
#pragma unroll
            for (int h = 0; h < XV; ++h) {
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, inb ? off + 16 * h : OOB, 0, 0);
                xr[sl][h] = *reinterpret_cast<const uint4 *>(&v);
            }
        }
        if (want_w) {
            const int wbase = chunk * Cout * C::KP * 2;
#pragma unroll
            for (int sl = 0; sl < C::WSLOTS; ++sl) {
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(
                    wrsrc, wrel[sl] != (int)OOB ? (unsigned)(wbase + wrel[sl]) : OOB, 0, 0);
                wr[sl] = *reinterpret_cast<const uint4 *>(&v);
            }
        }
    };
    auto commit = [&](bool want_w) {
        if constexpr (FP) {
#pragma unroll
            for (int sl = 0; sl < 2; ++sl)
                if (tid + sl * 256 < IN_FLOATS) xin[tid + sl * 256] = inr[sl];
        }
#pragma unroll
        for (int sl = 0; sl < (FP ? 0 : C::XSLOTS); ++sl) {
            const int idx = tid + sl * 256;
            if (idx < C::XITEMS) {
                uint4 item;
                if constexpr (F32IO) {                          // 8 f32 channels -> 8 bf16 (round to nearest even)
                    const float4 lo = __builtin_bit_cast(float4, xr[sl][0]), hi = __builtin_bit_cast(float4, xr[sl][XV - 1]);
                    bf16x8 h;
                    h[0] = (__bf16)lo.x; h[1] = (__bf16)lo.y; h[2] = (__bf16)lo.z; h[3] = (__bf16)lo.w;
                    h[4] = (__bf16)hi.x; h[5] = (__bf16)hi.y; h[6] = (__bf16)hi.z; h[7] = (__bf16)hi.w;
                    item = __builtin_bit_cast(uint4, h);
                } else {
                    item = xr[sl][0];
                }
                *reinterpret_cast<uint4 *>(xs + (idx / C::XQ) * C::PSB + (idx % C::XQ) * 16) = item;
            }
        }
        if (want_w) {
#pragma unroll
            for (int sl = 0; sl < C::WSLOTS; ++sl) {
                const int idx = tid + sl * 256;
                if (idx < C::WITEMS)
                    *reinterpret_cast<uint4 *>(ws + (idx / C::WQ) * C::WROWB + (idx % C::WQ) * 16) = wr[sl];
            }
        }
    };

    // per-lane fragment addressing
    //   KC = 32: step = tap, the lane's 8 channels are 8*kg .. 8*kg+7
    //   KC = 16: step s covers taps 2s (kg 0,1) and 2s+1 (kg 2,3), channels 8*(kg&1) ..
    //   KC = 8 : step s covers taps 4s + kg, all 8 channels
    const unsigned char *xb = xs + ((4 * wv) * C::HALO_W + li) * C::PSB + (KC == 32 ? kg * 16 : (KC == 16 ? (kg & 1) * 16 : 0));
    const unsigned char *wa = ws + li * C::WROWB + kg * 16;
    int toff[C::NSTEP];
#pragma unroll
    for (int s = 0; s < C::NSTEP; ++s) {
        int tap = KC == 32 ? s : (KC == 16 ? 2 * s + (kg >> 1) : 4 * s + kg);
        if (tap > KS * KS - 1) tap = KS * KS - 1;                  // zero-weight padding step: any valid address
        toff[s] = ((tap / KS) * C::HALO_W + tap % KS) * C::PSB;
    }

    // FP: conv1 (3x3, 1 -> 16, bias, ReLU, bf16) of the 18 x 18 halo on the matrix cores, as the f32 FIRST form does it
    // (sq_conv_f32_v2.hip): the 9 taps are the reduction -- 3 steps of v_mfma_f32_16x16x4_f32, taps 9..11 with zero weights, so a
    // value is exactly conv_first_bf16_kernel's chain "acc = 0; fmaf over the taps in raster order", then + bias, ReLU, one
    // rounding.  21 column blocks of 16 halo pixels over the 4 waves.  Halo pixels outside the image are conv2's ZERO padding.
    float a1[3] = {0.f, 0.f, 0.f};
    int toff1[3] = {0, 0, 0};
    float4 b1v = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (FP) {
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) {
            const int tap = 4 * s3 + kg;
            a1[s3] = tap < 9 ? drop.f_w[tap * 16 + li] : 0.f;
            const int tc = tap < 9 ? tap : 8;
            toff1[s3] = (tc / 3) * IN_W + tc % 3;
        }
        if (drop.f_b) b1v = *reinterpret_cast<const float4 *>(drop.f_b + 4 * kg);
    }
    auto first_conv = [&](int tile) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
        constexpr int NBLK = (C::HP + 15) / 16;                 // 21 column blocks: 6 / 5 / 5 / 5 per wave, unrolled so that the
#pragma unroll                                                  // independent chains of a wave overlap (a rolled loop ran them serially)
        for (int u = 0; u < (NBLK + 3) / 4; ++u) {
            const int blk = wv + 4 * u;
            if (blk >= NBLK) break;
            const int pix = blk * 16 + li;
            const int pc = pix < C::HP ? pix : C::HP - 1;
            const int py = pc / C::HALO_W, px = pc % C::HALO_W;
            const float *src = xin + py * IN_W + px;
            f32x4 c1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3) c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s3], src[toff1[s3]], c1, 0, 0, 0);
            const int gy = ty * TH - 1 + py, gx = tx * TW - 1 + px;
            const bool inside = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            const float v[4] = {c1[0] + b1v.x, c1[1] + b1v.y, c1[2] + b1v.z, c1[3] + b1v.w};
            bf16x4 o;
            unsigned nib = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = (__bf16)((inside && v[j] > 0.f) ? v[j] : 0.f);
                nib |= ((float)o[j] > 0.f ? 1u : 0u) << j;
            }
            if (pix < C::HP) *reinterpret_cast<bf16x4 *>(xs + pix * C::PSB + 8 * kg) = o;
            // the tile's own pixels also go to HBM (conv2's weight gradient reads y1) with their sign mask (conv2's dgrad gate)
            const bool own = pix < C::HP && inside && py >= 1 && py <= TH && px >= 1 && px <= TW;
            const unsigned p = (unsigned)((n * H + gy) * W + gx);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(
                __attribute__((__vector_size__(2 * sizeof(unsigned)))) unsigned, o), fyrsrc, own ? p * 32u + 8u * kg : OOB, 0, 0);
            unsigned v16 = nib << (4 * kg);                     // the four kg lanes of a pixel: 16 channels, 16 bits
            v16 |= (unsigned)__shfl_xor((int)v16, 16);
            v16 |= (unsigned)__shfl_xor((int)v16, 32);
            __builtin_amdgcn_raw_buffer_store_b16((unsigned short)v16, fmrsrc, (own && kg == 0) ? p * 2u : OOB, 0, 0);
        }
    };

    f32x4 acc[4][NR];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int nb = 0; nb < NR; ++nb) acc[r][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const SqDropKey dkey = sq_dropout_key(drop.seed, drop.thr ? drop.step : nullptr);
    const float slope = act == SQ_ACT_LEAKY ? 0.2f : 1.0f;
    const bool is_relu = act == SQ_ACT_RELU;
    auto actf = [&](float v) {
        const float neg = is_relu ? 0.0f : v * slope;
        return v > 0.0f ? v : neg;
    };
    // the lane's bias quads, fetched once per block where the registers allow (BN <= 32): a load inside the epilogue
    // costs every tile an L2 round trip
    constexpr bool HOIST_BIAS = NR <= 2;
    float4 bvr[HOIST_BIAS ? NR : 1];
    if constexpr (HOIST_BIAS) {
#pragma unroll
        for (int nb = 0; nb < NR; ++nb) {
            const int co = n0 + nb * 16 + 4 * kg;
            bvr[nb] = (bias && co < Cout) ? *reinterpret_cast<const float4 *>(bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    // 1x1 form with a gate (the transpose conv's dgrad): the tile's gate values are requested BEFORE its last chunk is
    // multiplied and have landed by the vmcnt(0) that precedes the commit -- the epilogue used to request them itself
    // and wait, a third exposed HBM round trip per tile of a kernel whose tiles are two short chunks
    constexpr bool GATE_EARLY = KS == 1 && !F32IO && !JN;
    bf16x4 gpre[GATE_EARLY ? NR : 1][4];
    auto gate_fetch = [&](int tile) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
        const int gx = tx * TW + li;
#pragma unroll
        for (int nb = 0; nb < NR; ++nb) {
            const int co = n0 + nb * 16 + 4 * kg;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gy = ty * TH + 4 * wv + r;
                const bool ok = gy < H && gx < W && co < Cout;
                gpre[nb][r] = __builtin_bit_cast(bf16x4, __builtin_amdgcn_raw_buffer_load_b64(
                    grsrc, ok ? (unsigned)((((n * H + gy) * W + gx) * Cout + co) * 2) : OOB, 0, 0));
            }
        }
    };
    auto epilogue = [&](int tile) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
        const int gx = tx * TW + li;
        if constexpr (JN) {
            // junction form (dgrad: no bias, no activation): every forward operand of the tile is requested before the
            // first is used -- one exposed HBM round trip per tile, not one per store
            const int jb = (int)((size_t)N * H * W * Cout * 2);
            const bool mul = drop.j_bridge == SQ_BRIDGE_MUL;
            const __amdgpu_buffer_rsrc_t ursrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(drop.j_up), 0, mul ? jb : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t krsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(drop.j_skip), 0, mul ? jb : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t dsrsrc = __builtin_amdgcn_make_buffer_rsrc(drop.j_dskip, 0, jb, 0x00020000);
            const __amdgpu_buffer_rsrc_t g2rsrc = __builtin_amdgcn_make_buffer_rsrc(drop.j_g, 0, jb, 0x00020000);
            // NG channel blocks at a time: the operands of the whole 64-channel tile would not fit beside the
            // accumulators (BN = 64: 64 more registers -> scratch)
            constexpr int NG = NR > 2 ? 2 : NR;
#pragma unroll
            for (int nb0 = 0; nb0 < NR; nb0 += NG) {
                unsigned offs[NG][4];
                bf16x4 uv[NG][4], kv[NG][4];
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int co = n0 + (nb0 + g) * 16 + 4 * kg;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int gy = ty * TH + 4 * wv + r;
                        const bool ok = gy < H && gx < W && co < Cout;
                        offs[g][r] = ok ? (unsigned)((((n * H + gy) * W + gx) * Cout + co) * 2) : OOB;
                    }
                }
                if (mul) {
#pragma unroll
                    for (int g = 0; g < NG; ++g)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            uv[g][r] = __builtin_bit_cast(bf16x4, __builtin_amdgcn_raw_buffer_load_b64(ursrc, offs[g][r], 0, 0));
                            kv[g][r] = __builtin_bit_cast(bf16x4, __builtin_amdgcn_raw_buffer_load_b64(krsrc, offs[g][r], 0, 0));
                        }
                    __builtin_amdgcn_s_waitcnt(0x0F70);         // inside the branch, as for the gate below
                }
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int nb = nb0 + g;
                    const int co = n0 + nb * 16 + 4 * kg;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int gy = ty * TH + 4 * wv + r;
                        bf16x4 o, da, db;
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = (__bf16)acc[r][nb][j];
                        da = o, db = o;
                        if (mul) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                da[j] = (__bf16)((float)o[j] * (float)kv[g][r][j]);
                                db[j] = (__bf16)((float)o[j] * (float)uv[g][r][j]);
                            }
                        } else if (drop.j_bridge == SQ_BRIDGE_SUB) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) db[j] = (__bf16)(-(float)o[j]);
                        }
                        const unsigned off = offs[g][r];
                        const unsigned goff = off == OOB ? OOB :
                            (unsigned)((((((n * (H >> 1) + (gy >> 1)) * (W >> 1) + (gx >> 1)) * 4 + ((gy & 1) * 2 + (gx & 1))) * Cout) + co) * 2);
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(
                            __attribute__((__vector_size__(2 * sizeof(unsigned)))) unsigned, db), dsrsrc, off, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(
                            __attribute__((__vector_size__(2 * sizeof(unsigned)))) unsigned, da), g2rsrc, goff, 0, 0);
                        acc[r][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    }
                }
            }
            return;
        }
        if constexpr (PN) {
            // pass 1: the stored values v = bf16(act(acc + bias)) (kept in the accumulator registers) and their squares per
            // pixel row; the four kg lanes of a pixel hold its Cout channels between them
            float ss[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int nb = 0; nb < NR; ++nb) {
                const int co = n0 + nb * 16 + 4 * kg;
                float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (HOIST_BIAS) bv = bvr[nb];
                else if (bias && co < Cout) bv = *reinterpret_cast<const float4 *>(bias + co);
                const float bq[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v = co < Cout ? (float)(__bf16)actf(acc[r][nb][j] + bq[j]) : 0.f;
                        acc[r][nb][j] = v;
                        ss[r] = __builtin_fmaf(v, v, ss[r]);
                    }
            }
            const __amdgpu_buffer_rsrc_t nrsrc = __builtin_amdgcn_make_buffer_rsrc(drop.pn_y, 0, (int)(io_pixels * Cout * 2), 0x00020000);
            const float invC = 1.0f / (float)Cout;
            float rn[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ss[r] += __shfl_xor(ss[r], 16);
                ss[r] += __shfl_xor(ss[r], 32);
                rn[r] = 1.0f / __builtin_sqrtf(ss[r] * invC + drop.pn_eps);
            }
#pragma unroll
            for (int nb = 0; nb < NR; ++nb) {
                const int co = n0 + nb * 16 + 4 * kg;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gy = ty * TH + 4 * wv + r;
                    const bool ok = gy < H && gx < W && co < Cout;
                    const unsigned off = ok ? (unsigned)((((n * H + gy) * W + gx) * Cout + co) * 2) : OOB;
                    bf16x4 o, q;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        o[j] = (__bf16)acc[r][nb][j];
                        q[j] = (__bf16)(acc[r][nb][j] * rn[r]);
                    }
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(
                        __attribute__((__vector_size__(2 * sizeof(unsigned)))) unsigned, o), yrsrc, drop.pn_store_y ? off : OOB, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(
                        __attribute__((__vector_size__(2 * sizeof(unsigned)))) unsigned, q), nrsrc, off, 0, 0);
                    acc[r][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
            return;
        }
#pragma unroll
        for (int nb = 0; nb < NR; ++nb) {
            const int co = n0 + nb * 16 + 4 * kg;
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (HOIST_BIAS) bv = bvr[nb];
            else if (bias && co < Cout) bv = *reinterpret_cast<const float4 *>(bias + co);
            unsigned offs[4];
            bf16x4 gv[4];
            bf16x4 pprev = {};
            (void)pprev;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gy = ty * TH + 4 * wv + r;
                if constexpr (MOS) {
                    const int px = co < Cout ? mos_pixel(gy, gx) : -1;
                    offs[r] = px >= 0 ? (unsigned)((px * Cout + co) * ES) : OOB;
                } else {
                    const bool ok = gy < H && gx < W && co < Cout;
                    offs[r] = ok ? (unsigned)((((n * H + gy) * W + gx) * Cout + co) * ES) : OOB;
                }
            }
            if constexpr (SK) {                                 // raw f32 partial sums of this block's chunk range
                const __amdgpu_buffer_rsrc_t krsrc = __builtin_amdgcn_make_buffer_rsrc(
                    drop.sk_ws + (size_t)blockIdx.z * io_pixels * Cout, 0, (int)(io_pixels * Cout * 4), 0x00020000);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float4 o = make_float4(acc[r][nb][0], acc[r][nb][1], acc[r][nb][2], acc[r][nb][3]);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(
                        __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, o), krsrc,
                        offs[r] == OOB ? OOB : offs[r] * (4 / ES), 0, 0);
                    acc[r][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                continue;
            }
            if constexpr (F32IO) {                              // f32 store of the f32 accumulators; no dropout
                float4 gq[4];
                (void)gq;
                if constexpr (GF) {
                    const __amdgpu_buffer_rsrc_t gfrsrc = __builtin_amdgcn_make_buffer_rsrc(
                        const_cast<float *>(drop.gate_f32), 0, (int)(io_pixels * Cout * 4), 0x00020000);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const auto gl = __builtin_amdgcn_raw_buffer_load_b128(gfrsrc, offs[r], 0, 0);
                        gq[r] = *reinterpret_cast<const float4 *>(&gl);
                    }
                    __builtin_amdgcn_s_waitcnt(0x0F70);         // inside the branch, as for the bf16 gate below
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float4 o;
                    o.x = actf(acc[r][nb][0] + bv.x);
                    o.y = actf(acc[r][nb][1] + bv.y);
                    o.z = actf(acc[r][nb][2] + bv.z);
                    o.w = actf(acc[r][nb][3] + bv.w);
                    if constexpr (GF) {                         // one f32 multiply where the activation was not active: act_bwd's
                        o.x = gq[r].x > 0.f ? o.x : o.x * drop.gscale;
                        o.y = gq[r].y > 0.f ? o.y : o.y * drop.gscale;
                        o.z = gq[r].z > 0.f ? o.z : o.z * drop.gscale;
                        o.w = gq[r].w > 0.f ? o.w : o.w * drop.gscale;
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(
                        __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, o), yrsrc, offs[r], 0, 0);
                    acc[r][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                continue;
            }
            // (16-byte stores through v_permlane16_swap_b32 row pairs -- guide T21 -- were built and measured: +1.8 % on the training step,
            // +0.9 % on the GAN iteration; these epilogues are not bound by the number of store instructions.  Removed.)
            unsigned mbits[4];
            (void)mbits;
            if constexpr (MG) {
                // 16 channels of a pixel = one 16-bit word of the mask; the four kg lanes of a pixel read the same word
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    mbits[r] = (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(mrsrc, offs[r] == OOB ? OOB : ((offs[r] >> 4) & ~1u), 0, 0);
                __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
                for (int r = 0; r < 4; ++r) mbits[r] = (mbits[r] >> (4 * kg)) & 0xFu;
            } else if constexpr (GATE_EARLY) {
#pragma unroll
                for (int r = 0; r < 4; ++r) gv[r] = gpre[nb][r];
            } else if (gate) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    gv[r] = __builtin_bit_cast(bf16x4, __builtin_amdgcn_raw_buffer_load_b64(grsrc, offs[r], 0, 0));
                // waited for inside the same branch: otherwise "loaded but never consumed" is a path to the compiler
                // and the main loop's prefetch gets guarded by waits that drain the stores (HISTORY.md 4a)
                __builtin_amdgcn_s_waitcnt(0x0F70);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                bf16x4 o;
                o[0] = (__bf16)actf(acc[r][nb][0] + bv.x);
                o[1] = (__bf16)actf(acc[r][nb][1] + bv.y);
                o[2] = (__bf16)actf(acc[r][nb][2] + bv.z);
                o[3] = (__bf16)actf(acc[r][nb][3] + bv.w);
                if constexpr (MG) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (mbits[r] >> j) & 1u ? (__bf16)((float)o[j] * drop.gscale) : (__bf16)0.f;
                } else if (gate) {
                    if constexpr (GB) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = (float)gv[r][j] > 0.f ? o[j] : (__bf16)((float)o[j] * drop.gscale);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)     // bf16(o * 1.0f) == o: the plain ReLU gate costs no rounding
                            o[j] = (float)gv[r][j] > 0.f ? (__bf16)((float)o[j] * drop.gscale) : (__bf16)0.f;
                    }
                }
                if (drop.thr) {
                    const unsigned k4 = sq_dropout_keep4(dkey, offs[r] >> 3, drop.thr);    // quad index (offsets are in bytes)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        o[j] = (k4 >> j) & 1u ? (__bf16)((float)o[j] * drop.inv) : (__bf16)0.f;
                }
                const unsigned off = offs[r];
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(
                    __attribute__((__vector_size__(2 * sizeof(unsigned)))) unsigned, o), yrsrc, off, 0, 0);
                if constexpr (MK) {
                    unsigned nib = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) nib |= ((float)o[j] > 0.f ? 1u : 0u) << j;
                    unsigned v16 = nib << (4 * kg);             // fold the four kg lanes of the pixel: 16 channels, 16 bits
                    v16 |= (unsigned)__shfl_xor((int)v16, 16);
                    v16 |= (unsigned)__shfl_xor((int)v16, 32);
                    __builtin_amdgcn_raw_buffer_store_b16((unsigned short)v16, mrsrc, (off != OOB && kg == 0) ? (off >> 4) : OOB, 0, 0);
                }
                if constexpr (PL) {
                    // rows 4 wv + {0,1} and {2,3} are the two pool rows of this wave; the horizontal partner is the
                    // neighbouring lane (li ^ 1).  Every lane runs the exchange; even lanes of valid pixels store.
                    if ((r & 1) == 0) {
                        pprev = o;
                    } else {
                        bf16x4 m;
                        if (drop.pool_avg) {
                            // (a + b) + (c + d): the horizontal pairs first (the partner lane holds x + 1), rows r - 1 and r; odd lanes
                            // compute the mirrored sum and do not store
                            const uint2 tv = __builtin_bit_cast(uint2, pprev), bv = __builtin_bit_cast(uint2, o);
                            uint2 tp, bp;
                            tp.x = (unsigned)__shfl_xor((int)tv.x, 1); tp.y = (unsigned)__shfl_xor((int)tv.y, 1);
                            bp.x = (unsigned)__shfl_xor((int)bv.x, 1); bp.y = (unsigned)__shfl_xor((int)bv.y, 1);
                            const bf16x4 tq = __builtin_bit_cast(bf16x4, tp), bq = __builtin_bit_cast(bf16x4, bp);
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                m[j] = (__bf16)((((float)pprev[j] + (float)tq[j]) + ((float)o[j] + (float)bq[j])) * 0.25f);
                        } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) m[j] = (float)pprev[j] > (float)o[j] ? pprev[j] : o[j];
                        const uint2 mv = __builtin_bit_cast(uint2, m);
                        uint2 pv;
                        pv.x = (unsigned)__shfl_xor((int)mv.x, 1);
                        pv.y = (unsigned)__shfl_xor((int)mv.y, 1);
                        const bf16x4 pp = __builtin_bit_cast(bf16x4, pv);
#pragma unroll
                        for (int j = 0; j < 4; ++j) m[j] = (float)pp[j] > (float)m[j] ? pp[j] : m[j];
                        }
                        const int gy = ty * TH + 4 * wv + r;
                        const unsigned poff = (off != OOB && !(li & 1))
                            ? (unsigned)((((n * (H >> 1) + (gy >> 1)) * (W >> 1) + (gx >> 1)) * Cout + co) * 2) : OOB;
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(
                            __attribute__((__vector_size__(2 * sizeof(unsigned)))) unsigned, m), prsrc, poff, 0, 0);
                    }
                }
                acc[r][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    };

    issue(t_begin, 0, true);
    __builtin_amdgcn_s_waitcnt(0x0F70);                         // vmcnt(0), stated outside commit()'s branches (see the main loop)
    commit(true);
    __syncthreads();
    if constexpr (FP) {
        first_conv(t_begin);
        __syncthreads();
    }
    int tile = t_begin, chunk = 0;
    // EARLY: the loads of the next item are requested as soon as the staging registers are free again -- right after the
    // commit, BEFORE the epilogue of the tile that just finished -- so a block has a tile in flight during its epilogue
    // as well.  Measured per form at level 0 (r02t16 / r02t17): pool form 78 -> 69 us, junction form 124 -> 118 us, plain and
    // mask-out forms 62 -> 66 us (their short epilogues gain nothing and the request now queues behind the stores).
    constexpr bool EARLY = SQ_CONV_EARLY_ISSUE == 1 || (SQ_CONV_EARLY_ISSUE == 2 && (PL || JN));
    auto step = [&](int &t, int &c) { if (++c == nchunk) { c = 0; t += tstride; } };
    if (EARLY && nitems > 1) {
        int t1 = t_begin, c1 = 0;
        step(t1, c1);
        issue(t1, c1, restage_w);
    }
    for (int it = 0; it < nitems; ++it) {
        int ntile = tile, nchk = chunk;
        step(ntile, nchk);
        const bool has_next = it + 1 < nitems;
        if (!EARLY && has_next) issue(ntile, nchk, restage_w);
        if constexpr (GATE_EARLY) {
            if (gate && chunk == nchunk - 1) gate_fetch(tile);
        }
        __builtin_amdgcn_s_setprio(0);                          // see sq_conv_f32_v2.hip: low priority while only feeding MFMA
#pragma unroll
        for (int s = 0; s < C::NSTEP; ++s) {
            bf16x8 a[NR], b[4];
#pragma unroll
            for (int nb = 0; nb < NR; ++nb)
                a[nb] = *reinterpret_cast<const bf16x8 *>(wa + nb * 16 * C::WROWB + s * 64);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                b[r] = *reinterpret_cast<const bf16x8 *>(xb + r * C::HALO_W * C::PSB + toff[s]);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int nb = 0; nb < NR; ++nb)
                    acc[r][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[nb], b[r], acc[r][nb], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(3);
        // the prefetch has landed -- stated OUTSIDE the `has_next` branch: hipcc does not correlate the two tests, keeps the
        // prefetch registers marked as pending loads round the back edge and otherwise guards the next issue() with
        // s_waitcnt vmcnt(n), which at run time waits for the epilogue's stores (sq_conv_f32_v2.hip, HISTORY.md 4a)
        __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0); expcnt / lgkmcnt untouched
        if (has_next) {
            __syncthreads();
            commit(restage_w);
        }
        if (EARLY && it + 2 < nitems) {
            int t2 = ntile, c2 = nchk;
            step(t2, c2);
            issue(t2, c2, restage_w);
        }
        if constexpr (FP) {
            if (has_next) {
                __syncthreads();                                // the image patch is complete
                first_conv(ntile);
            }
        }
        if (chunk == nchunk - 1) epilogue(tile);
        if (has_next) __syncthreads();
        tile = ntile;
        chunk = nchk;
    }
}

// first layer (Cin = 1..7, f32 image in): direct 3x3 conv, 16 output channels per thread, bf16 out;
// one fmaf chain per output in (tap, channel) order, as the f32 direct kernel
template <int CIN>
__global__ __launch_bounds__(256) void conv_first_bf16_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                               const float *__restrict__ bias, __bf16 *__restrict__ y,
                                                               int N, int H, int W, int Cout, int act, int tiles_x,
                                                               int tiles_y, unsigned char *__restrict__ mask) {
    constexpr int HW = TW + 2;
    __shared__ float xs[HW * HW * CIN];
    __shared__ float wsh[9 * CIN * 16];
    const int tid = threadIdx.x, sp = blockIdx.x;
    const int tx = sp % tiles_x, ty = (sp / tiles_x) % tiles_y, n = sp / (tiles_x * tiles_y);
    const int x0 = tx * TW, y0 = ty * TH, n0 = blockIdx.y * 16;
    for (int idx = tid; idx < HW * HW * CIN; idx += 256) {
        const int pix = idx / CIN, c = idx % CIN;
        const int gy = y0 - 1 + pix / HW, gx = x0 - 1 + pix % HW;
        xs[idx] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? x[(((size_t)n * H + gy) * W + gx) * CIN + c] : 0.f;
    }
    for (int idx = tid; idx < 9 * CIN * 16; idx += 256)
        wsh[idx] = (n0 + idx % 16 < Cout) ? w[(size_t)(idx / 16) * Cout + n0 + idx % 16] : 0.f;
    __syncthreads();
    const int py = tid >> 4, px = tid & 15;
    float a[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) a[o] = 0.f;
    constexpr int TAP_UNROLL = (CIN == 1 && SQ_FIRST_UNROLL) ? 9 : 1;   // rolled taps keep the filter out of registers
#pragma unroll TAP_UNROLL
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int c = 0; c < CIN; ++c) {
            const float xv = xs[((py + t / 3) * HW + px + t % 3) * CIN + c];
#pragma unroll
            for (int o = 0; o < 16; ++o) a[o] = __builtin_fmaf(wsh[(t * CIN + c) * 16 + o], xv, a[o]);
        }
    }
    const int gy = y0 + py, gx = x0 + px;
    if (gy < H && gx < W) {
        __bf16 *yo = y + ((size_t)(n * H + gy) * W + gx) * Cout + n0;
        unsigned bits = 0;                                      // sign mask of the 16 channels (SqDropEpi::mask layout)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (n0 + q * 8 >= Cout) break;
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float v = a[q * 8 + j] + (bias ? bias[n0 + q * 8 + j] : 0.f);
                o[j] = (__bf16)sq_act(v, act);
                bits |= ((float)o[j] > 0.f ? 1u : 0u) << (q * 8 + j);
            }
            *reinterpret_cast<bf16x8 *>(yo + q * 8) = o;
        }
        if (mask)
            *reinterpret_cast<unsigned short *>(mask + ((((size_t)(n * H + gy) * W + gx) * Cout + n0) >> 3)) = (unsigned short)bits;
    }
}

template <int BN, int KS, int KC, typename TIO, int FORM = FORM_PLAIN, bool MOS = false>
int launch(const TIO *x, const __bf16 *wp, const float *bias, TIO *y, int N, int H, int W, int Cin, int Cout,
           int act, hipStream_t st, const __bf16 *gate, const SqDropEpi &drop) {
    if constexpr (FORM == FORM_PLAIN && !MOS && sizeof(TIO) == 4) {
        if (drop.mos_h && KS == 3)
            return drop.gate_f32 ? launch<BN, KS, KC, TIO, FORM_GF, true>(x, wp, bias, y, N, H, W, Cin, Cout, act, st, gate, drop)
                                 : launch<BN, KS, KC, TIO, FORM_PLAIN, true>(x, wp, bias, y, N, H, W, Cin, Cout, act, st, gate, drop);
        if (drop.gate_f32) return launch<BN, KS, KC, TIO, FORM_GF>(x, wp, bias, y, N, H, W, Cin, Cout, act, st, gate, drop);
    }
    if constexpr (FORM == FORM_PLAIN && !MOS && sizeof(TIO) == 2) {
        if (drop.mos_h && KS == 3 && drop.sk_ws)
            return launch<BN, KS, KC, TIO, FORM_SK, true>(x, wp, bias, y, N, H, W, Cin, Cout, act, st, gate, drop);
        if (drop.mos_h && KS == 3)
            return drop.gate_slope ? launch<BN, KS, KC, TIO, FORM_GB, true>(x, wp, bias, y, N, H, W, Cin, Cout, act, st, gate, drop)
                                   : launch<BN, KS, KC, TIO, FORM_PLAIN, true>(x, wp, bias, y, N, H, W, Cin, Cout, act, st, gate, drop);
        if (drop.gate_slope) return launch<BN, KS, KC, TIO, FORM_GB>(x, wp, bias, y, N, H, W, Cin, Cout, act, st, gate, drop);
    }
    if constexpr (FORM == FORM_PLAIN && KS == 3 && sizeof(TIO) == 2) {
        if constexpr (BN == 16 && KC == 16) {
            if (drop.f_x) return launch<BN, KS, KC, TIO, FORM_FP>(x, wp, bias, y, N, H, W, Cin, Cout, act, st, gate, drop);
        }
        if (drop.j_g) return launch<BN, KS, KC, TIO, FORM_JN>(x, wp, bias, y, N, H, W, Cin, Cout, act, st, gate, drop);
        if (drop.pool) return launch<BN, KS, KC, TIO, FORM_PL>(x, wp, bias, y, N, H, W, Cin, Cout, act, st, gate, drop);
        if (drop.pn_y) return launch<BN, KS, KC, TIO, FORM_PN>(x, wp, bias, y, N, H, W, Cin, Cout, act, st, gate, drop);
        if (drop.mask)
            return gate ? launch<BN, KS, KC, TIO, FORM_MG>(x, wp, bias, y, N, H, W, Cin, Cout, act, st, gate, drop)
                        : launch<BN, KS, KC, TIO, FORM_MK>(x, wp, bias, y, N, H, W, Cin, Cout, act, st, gate, drop);
    }
    using C = CfgB<BN, KS, KC>;
    static bool attr_set = false;
    static int occ = 2;                                         // resident blocks per CU (registers / LDS)
    auto kern = conv_mfma_bf16_kernel<BN, KS, KC, TIO, FORM, MOS>;
    constexpr int LDS = C::LDS_BYTES + (FORM == FORM_FP ? (TW + 4) * (TW + 4) * 4 : 0);
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                LDS) != hipSuccess) {
            sq_set_error("conv_mfma_bf16: cannot reserve %d bytes of LDS", LDS);
            return SQ_ELAUNCH;
        }
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(kern), 256, LDS) ==
                hipSuccess && nb >= 1)
            occ = nb > 8 ? 8 : nb;
        attr_set = true;
    }
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int ntiles = tiles_x * tiles_y * N;
    const int gy = (Cout + BN - 1) / BN;
    // persistent grid: every block slot the CU can hold (these kernels are latency / HBM bound at C <= 32:
    // the tiles in flight, not the MFMA rate, set their speed)
    int want = (256 * occ + gy - 1) / gy;
    if (want < 1) want = 1;
    int tpb = (ntiles + want - 1) / want;
    if (tpb < 1) tpb = 1;
    const int gx = (ntiles + tpb - 1) / tpb;
    const int gz = FORM == FORM_SK ? (Cin / KC) / drop.sk_chunks : 1;
    static const int il = [] { const char *e = getenv("SQ_CONV_BF16_INTERLEAVE"); return e ? atoi(e) : 1; }();   // 0: contiguous tile runs (A/B switch)
    hipLaunchKernelGGL(kern, dim3(gx, gy, gz), dim3(256), LDS, st, x, wp, bias, y, N, H, W, Cin, Cout, act,
                       tiles_x, tiles_y, ntiles, (il && !MOS) ? -tpb : tpb, gate, drop);
    return sq_check_launch("sq_conv2d_nhwc_fwd_bf16");
}

template <int KS, int KC, typename TIO>
int dispatch_bn(const TIO *x, const __bf16 *wp, const float *bias, TIO *y, int N, int H, int W, int Cin,
                int Cout, int act, hipStream_t st, const __bf16 *gate, const SqDropEpi &drop) {
    // narrow the channel block until the launch has ~2 blocks per CU (as sq_conv_f32_v2.hip): the GAN's 4x4 .. 32x32
    // levels are a handful of mosaic tiles x 512 .. 64 channels, and 64-channel blocks leave most of the chip idle
    const int64_t ntiles = (int64_t)((W + TW - 1) / TW) * ((H + TH - 1) / TH) * N;
    int bn = Cout >= 64 ? 64 : (Cout > 16 ? 32 : 16);
    static const int narrow = [] { const char *e = getenv("SQ_CONV_BF16_NARROW"); return e ? atoi(e) : 1; }();
    while (narrow && bn > 16 && !drop.pn_y && ntiles * ((Cout + bn - 1) / bn) < 2 * 256) bn >>= 1;   // (FORM_PN: one block per pixel's channels)
    static const int force_bn = [] { const char *e = getenv("SQ_MOS_BN"); return e ? atoi(e) : 0; }();   // experiment switch: block width of the mosaic launches
    if (drop.mos_h && !drop.pn_y && (force_bn == 16 || force_bn == 32 || force_bn == 64) && force_bn <= ((Cout + 15) / 16) * 16) bn = force_bn;
    if (bn == 64) return launch<64, KS, KC, TIO>(x, wp, bias, y, N, H, W, Cin, Cout, act, st, gate, drop);
    if (bn == 32) return launch<32, KS, KC, TIO>(x, wp, bias, y, N, H, W, Cin, Cout, act, st, gate, drop);
    return launch<16, KS, KC, TIO>(x, wp, bias, y, N, H, W, Cin, Cout, act, st, gate, drop);
}

inline int kc_for(int Cin) { return Cin % 32 == 0 ? 32 : (Cin % 16 == 0 ? 16 : 8); }

// split-K finish: y[p][c] = bf16(act(sum_s ws[s][p][c] + bias[c])), slices added in order; gate != NULL: the result is the gated
// dgrad instead (no bias / act): t = bf16(sum), y = gate > 0 ? t : bf16(t * slope) -- FORM_GB's two roundings
__global__ __launch_bounds__(256) void conv_splitk_finish_bf16_kernel(const float4 *__restrict__ ws, int S, int64_t n4, int C4,
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

template <typename TIO>
int dispatch_kc(const TIO *x, const __bf16 *wp, const float *bias, TIO *y, int N, int H, int W, int Cin, int Cout, int K,
                int act, hipStream_t st, const __bf16 *gate, const SqDropEpi &drop) {
    switch (kc_for(Cin)) {
    case 32:
        return K == 3 ? dispatch_bn<3, 32, TIO>(x, wp, bias, y, N, H, W, Cin, Cout, act, st, gate, drop)
                      : dispatch_bn<1, 32, TIO>(x, wp, bias, y, N, H, W, Cin, Cout, act, st, gate, drop);
    case 16:
        return K == 3 ? dispatch_bn<3, 16, TIO>(x, wp, bias, y, N, H, W, Cin, Cout, act, st, gate, drop)
                      : dispatch_bn<1, 16, TIO>(x, wp, bias, y, N, H, W, Cin, Cout, act, st, gate, drop);
    default:
        return K == 3 ? dispatch_bn<3, 8, TIO>(x, wp, bias, y, N, H, W, Cin, Cout, act, st, gate, drop)
                      : dispatch_bn<1, 8, TIO>(x, wp, bias, y, N, H, W, Cin, Cout, act, st, gate, drop);
    }
}

}  // namespace

extern "C" int64_t sq_conv_packed_weights_elems_bf16(int K, int Cin, int Cout) {
    if ((K != 1 && K != 3) || Cin <= 0 || Cin % 8 || Cout <= 0) return -1;
    const int KC = kc_for(Cin);
    return (int64_t)(Cin / KC) * Cout * kp_for(K, KC);
}

// f32 HWIO (K,K,Cin,Cout) [* wscale] -> packed bf16 filter of the conv Cin -> Cout.
// transform != 0: `w` is the filter of the FORWARD conv Cout -> Cin, i.e. stored (K,K,Cout,Cin); the packed
// result is its dgrad filter (taps rotated by 180 degrees, channel roles swapped).
extern "C" int sq_conv_pack_weights_bf16(const float *w, void *wp, int K, int Cin, int Cout, float wscale,
                                         int transform, void *stream) {
    SQ_REQUIRE(w && wp, "sq_conv_pack_weights_bf16: null pointer");
    const int64_t n = sq_conv_packed_weights_elems_bf16(K, Cin, Cout);
    SQ_REQUIRE(n > 0, "sq_conv_pack_weights_bf16: unsupported K=%d Cin=%d Cout=%d (Cin %% 8 == 0)", K, Cin, Cout);
    const int KC = kc_for(Cin);
    int64_t nb = (n + 255) / 256;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(pack_weights_bf16_kernel, dim3((unsigned)nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       w, reinterpret_cast<__bf16 *>(wp), K, Cin, Cout, KC, kp_for(K, KC), wscale, transform);
    return sq_check_launch("sq_conv_pack_weights_bf16");
}

// every pack / cast of a training step in one launch (table layout: pack_weights_multi_bf16_kernel)
extern "C" int sq_conv_pack_weights_multi_bf16(const float *base, void *out, const int32_t *table, int n_entries,
                                               int total_items, void *stream) {
    SQ_REQUIRE(base && out && table && n_entries > 0 && n_entries <= 128 && total_items > 0,
               "sq_conv_pack_weights_multi_bf16: bad arguments (at most 128 entries)");
    SQ_REQUIRE(total_items % 8 == 0 && (((uintptr_t)out) & 15u) == 0,
               "sq_conv_pack_weights_multi_bf16: every entry must hold a multiple of 8 elements, out 16-byte aligned");
    int nb = (total_items / 8 + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(pack_weights_multi_bf16_kernel, dim3((unsigned)nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       base, reinterpret_cast<__bf16 *>(out), table, n_entries, total_items, (const float *)nullptr);
    return sq_check_launch("sq_conv_pack_weights_multi_bf16");
}

// the same with one factor per entry (scales: n_entries floats on the device): every filter pack of a GAN solver step
// -- forward and dgrad form of each weighted_conv2d kernel, equalised-LR factor folded in (gan.py:75-79) -- in ONE launch
extern "C" int sq_conv_pack_weights_multi_scaled_bf16(const float *base, void *out, const int32_t *table,
                                                      const float *scales, int n_entries, int total_items, void *stream) {
    SQ_REQUIRE(base && out && table && scales && n_entries > 0 && n_entries <= 128 && total_items > 0,
               "sq_conv_pack_weights_multi_scaled_bf16: bad arguments (at most 128 entries)");
    SQ_REQUIRE(total_items % 8 == 0 && (((uintptr_t)out) & 15u) == 0,
               "sq_conv_pack_weights_multi_scaled_bf16: every entry must hold a multiple of 8 elements, out 16-byte aligned");
    int nb = (total_items / 8 + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(pack_weights_multi_bf16_kernel, dim3((unsigned)nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       base, reinterpret_cast<__bf16 *>(out), table, n_entries, total_items, scales);
    return sq_check_launch("sq_conv_pack_weights_multi_scaled_bf16");
}

// conv_layer / weighted_conv2d on bf16 tensors: x (N,H,W,Cin) bf16, wp from sq_conv_pack_weights_bf16,
// bias f32 or NULL, y (N,H,W,Cout) bf16.  Cin % 16 == 0, Cout % 4 == 0.
static int conv_fwd_bf16_impl(const void *x, const void *wp, const float *bias, void *y, int N, int H, int W, int Cin,
                              int Cout, int K, int act, void *stream, const void *gate,
                              const SqDropEpi &drop = SqDropEpi{0u, 1.f, 0u, nullptr}) {
    SQ_REQUIRE(x && wp && y, "sq_conv2d_nhwc_fwd_bf16: null tensor pointer");
    SQ_REQUIRE(N > 0 && H > 0 && W > 0 && (K == 1 || K == 3), "sq_conv2d_nhwc_fwd_bf16: bad shape / K");
    SQ_REQUIRE(Cin % 8 == 0 && Cin > 0 && Cout % 4 == 0 && Cout > 0,
               "sq_conv2d_nhwc_fwd_bf16: Cin=%d (multiple of 8), Cout=%d (multiple of 4)", Cin, Cout);
    SQ_REQUIRE((size_t)N * H * W * (size_t)(Cin > Cout ? Cin : Cout) * 2 < ((size_t)1 << 31),
               "sq_conv2d_nhwc_fwd_bf16: tensors must be < 2 GiB");
    SQ_REQUIRE(act >= SQ_ACT_NONE && act <= SQ_ACT_LEAKY, "sq_conv2d_nhwc_fwd_bf16: bad activation %d", act);
    SQ_REQUIRE_ALIGNED(x); SQ_REQUIRE_ALIGNED(wp); SQ_REQUIRE_ALIGNED(y);
    if (bias) SQ_REQUIRE_ALIGNED(bias);
    if (gate) SQ_REQUIRE_ALIGNED(gate);
    const __bf16 *gb = reinterpret_cast<const __bf16 *>(gate);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const __bf16 *xb = reinterpret_cast<const __bf16 *>(x), *wb = reinterpret_cast<const __bf16 *>(wp);
    __bf16 *yb = reinterpret_cast<__bf16 *>(y);
    return dispatch_kc<__bf16>(xb, wb, bias, yb, N, H, W, Cin, Cout, K, act, st, gb, drop);
}

// "mixed" convolution: f32 activations in and out, operands rounded to bf16 on the way into LDS, f32 accumulation
// (v_mfma_f32_16x16x32_bf16).  wp from sq_conv_pack_weights_bf16 (which also folds the equalised-LR scale in).
// The drop-in for sq_conv2d_nhwc_fwd_f32 behind an f32 graph that wants the bf16 matrix rate (GAN, config 5).
extern "C" int sq_conv2d_nhwc_fwd_mixed_f32(const float *x, const void *wp, const float *bias, float *y, int N, int H,
                                            int W, int Cin, int Cout, int K, int act, void *stream) {
    SQ_REQUIRE(x && wp && y, "sq_conv2d_nhwc_fwd_mixed_f32: null tensor pointer");
    SQ_REQUIRE(N > 0 && H > 0 && W > 0 && (K == 1 || K == 3), "sq_conv2d_nhwc_fwd_mixed_f32: bad shape / K");
    SQ_REQUIRE(Cin % 8 == 0 && Cin > 0 && Cout % 4 == 0 && Cout > 0,
               "sq_conv2d_nhwc_fwd_mixed_f32: Cin=%d (multiple of 8), Cout=%d (multiple of 4)", Cin, Cout);
    SQ_REQUIRE((size_t)N * H * W * (size_t)(Cin > Cout ? Cin : Cout) * 4 < ((size_t)1 << 31),
               "sq_conv2d_nhwc_fwd_mixed_f32: tensors must be < 2 GiB");
    SQ_REQUIRE(act >= SQ_ACT_NONE && act <= SQ_ACT_LEAKY, "sq_conv2d_nhwc_fwd_mixed_f32: bad activation %d", act);
    SQ_REQUIRE_ALIGNED(x); SQ_REQUIRE_ALIGNED(wp); SQ_REQUIRE_ALIGNED(y);
    if (bias) SQ_REQUIRE_ALIGNED(bias);
    return dispatch_kc<float>(x, reinterpret_cast<const __bf16 *>(wp), bias, y, N, H, W, Cin, Cout, K, act,
                              reinterpret_cast<hipStream_t>(stream), nullptr, SqDropEpi{0u, 1.f, 0u, nullptr});
}

// dgrad of a conv (f32 tensors, bf16 multiply) whose input was the output `gate` of a leaky-ReLU / ReLU: the result leaves
// through that activation's backward, dx = gate > 0 ? v : v * slope (slope 0.2 / 0) -- sq_conv2d_nhwc_fwd_mixed_f32 on dY
// with the dgrad pack followed by sq_act_bwd_f32, in one kernel, same bits
extern "C" int sq_conv2d_nhwc_dgrad_actgate_mixed_f32(const float *dy, const void *wp_t, const float *gate, int act, float *dx,
                                                      int N, int H, int W, int Cin, int Cout, int K, void *stream) {
    SQ_REQUIRE(dy && wp_t && gate && dx, "sq_conv2d_nhwc_dgrad_actgate_mixed_f32: null tensor pointer");
    SQ_REQUIRE(N > 0 && H > 0 && W > 0 && (K == 1 || K == 3), "sq_conv2d_nhwc_dgrad_actgate_mixed_f32: bad shape / K");
    SQ_REQUIRE(Cin % 8 == 0 && Cin > 0 && Cout % 4 == 0 && Cout > 0,
               "sq_conv2d_nhwc_dgrad_actgate_mixed_f32: Cin=%d (multiple of 8), Cout=%d (multiple of 4)", Cin, Cout);
    SQ_REQUIRE((size_t)N * H * W * (size_t)(Cin > Cout ? Cin : Cout) * 4 < ((size_t)1 << 31),
               "sq_conv2d_nhwc_dgrad_actgate_mixed_f32: tensors must be < 2 GiB");
    SQ_REQUIRE(act == SQ_ACT_RELU || act == SQ_ACT_LEAKY, "sq_conv2d_nhwc_dgrad_actgate_mixed_f32: activation %d has no gate", act);
    SQ_REQUIRE_ALIGNED(dy); SQ_REQUIRE_ALIGNED(wp_t); SQ_REQUIRE_ALIGNED(gate); SQ_REQUIRE_ALIGNED(dx);
    SqDropEpi d{0u, 1.f, 0u, nullptr};
    d.gate_f32 = gate;
    d.gscale = act == SQ_ACT_LEAKY ? 0.2f : 0.0f;
    return dispatch_kc<float>(dy, reinterpret_cast<const __bf16 *>(wp_t), nullptr, dx, N, H, W, Cin, Cout, K, SQ_ACT_NONE,
                              reinterpret_cast<hipStream_t>(stream), nullptr, d);
}

// the two mixed-precision forms above on a BATCH OF SMALL IMAGES convolved as one mosaic image (R x Cc cells of pitch
// h+1 / w+1, sq_mosaic_pack_f32's layout) without building the mosaic: x, gate, y / dx are the compact (Nimg, h, w, C)
// tensors.  Same fmaf chain per output as the packed route (separator pixels read as zero, are not stored).  gate == NULL:
// plain forward (bias / act apply); gate != NULL: the act-gated dgrad (bias and act ignored, `act` names the gate's activation).
extern "C" int sq_conv2d_nhwc_mixed_mosaic_f32(const float *x, const void *wp, const float *bias, const float *gate, float *y,
                                               int Nimg, int h, int w, int Cin, int Cout, int act, int R, int Cc,
                                               void *stream) {
    SQ_REQUIRE(x && wp && y, "sq_conv2d_nhwc_mixed_mosaic_f32: null tensor pointer");
    SQ_REQUIRE(Nimg > 0 && h > 0 && w > 0 && R > 0 && Cc > 0 && (int64_t)R * Cc >= Nimg,
               "sq_conv2d_nhwc_mixed_mosaic_f32: need R * Cc >= Nimg (Nimg=%d R=%d Cc=%d)", Nimg, R, Cc);
    SQ_REQUIRE(Cin % 8 == 0 && Cin > 0 && Cout % 4 == 0 && Cout > 0,
               "sq_conv2d_nhwc_mixed_mosaic_f32: Cin=%d (multiple of 8), Cout=%d (multiple of 4)", Cin, Cout);
    SQ_REQUIRE(act >= SQ_ACT_NONE && act <= SQ_ACT_LEAKY && (!gate || act != SQ_ACT_NONE),
               "sq_conv2d_nhwc_mixed_mosaic_f32: bad activation %d", act);
    const int H = R * (h + 1), W = Cc * (w + 1);
    SQ_REQUIRE((size_t)H * W * (size_t)(Cin > Cout ? Cin : Cout) * 4 < ((size_t)1 << 31), "sq_conv2d_nhwc_mixed_mosaic_f32: too large");
    SQ_REQUIRE_ALIGNED(x); SQ_REQUIRE_ALIGNED(wp); SQ_REQUIRE_ALIGNED(y);
    if (bias) SQ_REQUIRE_ALIGNED(bias);
    if (gate) SQ_REQUIRE_ALIGNED(gate);
    SqDropEpi d{0u, 1.f, 0u, nullptr};
    SQ_REQUIRE(h <= 8 && w <= 8 && H < (1 << 13) && W < (1 << 13), "sq_conv2d_nhwc_mixed_mosaic_f32: images up to 8 x 8, mosaic < 8192");
    d.mos_h = h; d.mos_w = w; d.mos_cc = Cc; d.mos_n = Nimg;
    d.mos_mh = (65536u + (unsigned)h) / (unsigned)(h + 1);      // ceil(2^16 / (h+1))
    d.mos_mw = (65536u + (unsigned)w) / (unsigned)(w + 1);
    if (gate) {
        d.gate_f32 = gate;
        d.gscale = act == SQ_ACT_LEAKY ? 0.2f : 0.0f;
    }
    return dispatch_kc<float>(x, reinterpret_cast<const __bf16 *>(wp), gate ? nullptr : bias, y, 1, H, W, Cin, Cout, 3,
                              gate ? (int)SQ_ACT_NONE : act, reinterpret_cast<hipStream_t>(stream), nullptr, d);
}

extern "C" int sq_conv2d_nhwc_fwd_bf16(const void *x, const void *wp, const float *bias, void *y, int N, int H,
                                       int W, int Cin, int Cout, int K, int act, void *stream) {
    return conv_fwd_bf16_impl(x, wp, bias, y, N, H, W, Cin, Cout, K, act, stream, nullptr);
}

// dgrad of a conv on bf16 tensors whose input was the output `gate` of a leaky-ReLU / ReLU (the GAN's conv -> leaky -> conv
// chains under bf16 storage): sq_conv2d_nhwc_fwd_bf16 on dY with the dgrad pack followed by sq_act_bwd_bf16, in one kernel,
// same two roundings, same bits
extern "C" int sq_conv2d_nhwc_dgrad_actgate_bf16(const void *dy, const void *wp_t, const void *gate, int act, void *dx, int N,
                                                 int H, int W, int Cin, int Cout, int K, void *stream) {
    SQ_REQUIRE(gate, "sq_conv2d_nhwc_dgrad_actgate_bf16: null gate");
    SQ_REQUIRE(act == SQ_ACT_RELU || act == SQ_ACT_LEAKY, "sq_conv2d_nhwc_dgrad_actgate_bf16: activation %d has no gate", act);
    SqDropEpi d{0u, 1.f, 0u, nullptr};
    d.gate_slope = 1;
    d.gscale = act == SQ_ACT_LEAKY ? 0.2f : 0.0f;
    return conv_fwd_bf16_impl(dy, wp_t, nullptr, dx, N, H, W, Cin, Cout, K, SQ_ACT_NONE, stream, gate, d);
}

// sq_conv2d_nhwc_mixed_mosaic_f32 on bf16 tensors: a batch of small images (Nimg, h, w, C) convolved as one mosaic image of
// R x Cc cells that is never built.  gate == NULL: plain forward (bias / act apply); gate != NULL: the act-gated dgrad above.
// workspace (may be NULL) / workspace_bytes: room for split-K partial sums.  Where the launch would leave most of the chip idle
// (few mosaic tiles x Cout / 16 blocks, each walking every input-channel chunk alone) and the workspace holds S >= 2 slices of
// Nimg*h*w*Cout floats, the reduction is split S ways over the grid and a finish kernel adds the slices in order: same value
// to f32 rounding (another summation order than the unsplit kernel), run-to-run identical.
extern "C" int sq_conv2d_nhwc_mosaic_bf16(const void *x, const void *wp, const float *bias, const void *gate, void *y, int Nimg,
                                          int h, int w, int Cin, int Cout, int act, int R, int Cc, float *workspace,
                                          int64_t workspace_bytes, void *stream) {
    SQ_REQUIRE(Nimg > 0 && h > 0 && w > 0 && R > 0 && Cc > 0 && (int64_t)R * Cc >= Nimg,
               "sq_conv2d_nhwc_mosaic_bf16: need R * Cc >= Nimg (Nimg=%d R=%d Cc=%d)", Nimg, R, Cc);
    SQ_REQUIRE(act >= SQ_ACT_NONE && act <= SQ_ACT_LEAKY && (!gate || act != SQ_ACT_NONE), "sq_conv2d_nhwc_mosaic_bf16: bad activation %d", act);
    const int H = R * (h + 1), W = Cc * (w + 1);
    SQ_REQUIRE(h <= 8 && w <= 8 && H < (1 << 13) && W < (1 << 13), "sq_conv2d_nhwc_mosaic_bf16: images up to 8 x 8, mosaic < 8192");
    SqDropEpi d{0u, 1.f, 0u, nullptr};
    d.mos_h = h; d.mos_w = w; d.mos_cc = Cc; d.mos_n = Nimg;
    d.mos_mh = (65536u + (unsigned)h) / (unsigned)(h + 1);      // ceil(2^16 / (h+1))
    d.mos_mw = (65536u + (unsigned)w) / (unsigned)(w + 1);
    if (gate) {
        d.gate_slope = 1;
        d.gscale = act == SQ_ACT_LEAKY ? 0.2f : 0.0f;
    }
    // split-K: blocks the unsplit launch would have (16-channel blocks once narrowed) vs the chip
    const int nchunk = Cin / kc_for(Cin);
    const int64_t blocks = (int64_t)((H + TH - 1) / TH) * ((W + TW - 1) / TW) * ((Cout + 15) / 16);
    const int64_t slice = (int64_t)Nimg * h * w * Cout * 4;
    int S = 1;
    static const int sk_on = [] { const char *e = getenv("SQ_CONV_SPLITK"); return e ? atoi(e) : 1; }();
    if (sk_on && workspace && Cout % 4 == 0 && blocks < 256 && nchunk >= 4)
        while (S < 8 && nchunk % (2 * S) == 0 && nchunk / (2 * S) >= 2 && blocks * S < 512 && slice * 2 * S <= workspace_bytes) S *= 2;
    static const int force_s = [] { const char *e = getenv("SQ_MOS_S"); return e ? atoi(e) : 0; }();   // experiment switch (tools/r04_mosaic_sweep.py)
    if (force_s >= 1 && workspace && nchunk % force_s == 0 && slice * force_s <= workspace_bytes) S = force_s;
    if (S == 1)
        return conv_fwd_bf16_impl(x, wp, gate ? nullptr : bias, y, 1, H, W, Cin, Cout, 3, gate ? (int)SQ_ACT_NONE : act, stream, gate, d);
    SQ_REQUIRE_ALIGNED(workspace);
    SQ_REQUIRE(slice * S < ((int64_t)1 << 31), "sq_conv2d_nhwc_mosaic_bf16: split-K workspace too large");
    d.sk_ws = workspace;
    d.sk_chunks = nchunk / S;
    int rc = conv_fwd_bf16_impl(x, wp, nullptr, y, 1, H, W, Cin, Cout, 3, SQ_ACT_NONE, stream, nullptr, d);
    if (rc) return rc;
    const int64_t n4 = slice / 16;
    int64_t nb = (n4 + 255) / 256;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(conv_splitk_finish_bf16_kernel, dim3((unsigned)nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const float4 *>(workspace), S, n4, Cout / 4, gate ? nullptr : bias, gate ? (int)SQ_ACT_NONE : act,
                       reinterpret_cast<const bf16x4 *>(gate), d.gscale, reinterpret_cast<bf16x4 *>(y));
    return sq_check_launch("sq_conv2d_nhwc_mosaic_bf16(split-K finish)");
}

// conv + bias + act + dropout in one kernel (training): y = dropout(act(conv(x))) with the counter-hash mask of
// sq_dropout_fwd_bf16 (seed, optional device step counter); no mask tensor is written -- for a ReLU the
// backward can gate on y itself (y > 0 <=> kept and active), see sq_relu_scale_bwd_bf16.
extern "C" int sq_conv2d_nhwc_fwd_dropout_bf16(const void *x, const void *wp, const float *bias, void *y, int N, int H,
                                               int W, int Cin, int Cout, int K, int act, float rate, uint32_t seed,
                                               const int32_t *step_dev, void *stream) {
    SQ_REQUIRE(rate > 0.f && rate < 1.f, "sq_conv2d_nhwc_fwd_dropout_bf16: rate must be in (0, 1)");
    SqDropEpi d;
    d.thr = (unsigned)(rate * 65536.0f);                       // = sq_dropout_thr16
    d.inv = 1.0f / (1.0f - rate);
    d.seed = seed;
    d.step = step_dev;
    SQ_REQUIRE(d.thr != 0u, "sq_conv2d_nhwc_fwd_dropout_bf16: rate too small for the 16-bit threshold");
    return conv_fwd_bf16_impl(x, wp, bias, y, N, H, W, Cin, Cout, K, act, stream, nullptr, d);
}

// the same, with the 2x2/s2 max pool of the result written beside it (ypool (N,H/2,W/2,Cout); H, W even; K = 3): the
// encoder's conv_block -> max_pool pair (unet.py:241-243) in one kernel.  rate == 0: no dropout (plain conv + act).
extern "C" int sq_conv2d_nhwc_fwd_dropout_pool_bf16(const void *x, const void *wp, const float *bias, void *y, void *ypool,
                                                    int N, int H, int W, int Cin, int Cout, int K, int act, float rate,
                                                    uint32_t seed, const int32_t *step_dev, void *stream) {
    SQ_REQUIRE(rate >= 0.f && rate < 1.f, "sq_conv2d_nhwc_fwd_dropout_pool_bf16: rate must be in [0, 1)");
    SQ_REQUIRE(ypool && K == 3 && H % 2 == 0 && W % 2 == 0, "sq_conv2d_nhwc_fwd_dropout_pool_bf16: K = 3, even H and W, ypool");
    SQ_REQUIRE_ALIGNED(ypool);
    SqDropEpi d{0u, 1.f, 0u, nullptr};
    if (rate > 0.f) {
        d.thr = (unsigned)(rate * 65536.0f);
        d.inv = 1.0f / (1.0f - rate);
        d.seed = seed;
        d.step = step_dev;
        SQ_REQUIRE(d.thr != 0u, "sq_conv2d_nhwc_fwd_dropout_pool_bf16: rate too small for the 16-bit threshold");
    }
    d.pool = reinterpret_cast<__bf16 *>(ypool);
    return conv_fwd_bf16_impl(x, wp, bias, y, N, H, W, Cin, Cout, K, act, stream, nullptr, d);
}

// conv_block of down0 + max_pool_layer (sequitr/networks/unet.py:238-243, 265-277) for a single-channel f32 image and 16
// filters, the training form, in ONE launch: y1 = relu(conv3x3(x, w1) + b1) (sq_conv3x3_first_fwd_mask_bf16's tensor and sign
// mask, bit for bit: the weight gradient of conv2 and the gate of its dgrad need them) is evaluated per tile for the 18 x 18
// halo and never read back from HBM; y = dropout(relu(conv3x3(y1, wp2) + b2)) and ypool as sq_conv2d_nhwc_fwd_dropout_pool_bf16.
extern "C" int sq_conv3x3_first_block_dropout_pool_bf16(const float *x, const float *w1, const float *b1, void *y1, void *mask1,
                                                        const void *wp2, const float *b2, void *y, void *ypool, int N, int H,
                                                        int W, float rate, uint32_t seed, const int32_t *step_dev,
                                                        void *stream) {
    SQ_REQUIRE(x && w1 && y1 && mask1 && wp2 && y && ypool, "sq_conv3x3_first_block_dropout_pool_bf16: null tensor pointer");
    SQ_REQUIRE(rate >= 0.f && rate < 1.f, "sq_conv3x3_first_block_dropout_pool_bf16: rate must be in [0, 1)");
    SQ_REQUIRE(N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "sq_conv3x3_first_block_dropout_pool_bf16: even H and W");
    SQ_REQUIRE_ALIGNED(w1); SQ_REQUIRE_ALIGNED(y1); SQ_REQUIRE_ALIGNED(ypool);
    if (b1) SQ_REQUIRE_ALIGNED(b1);
    SqDropEpi d{0u, 1.f, 0u, nullptr};
    if (rate > 0.f) {
        d.thr = (unsigned)(rate * 65536.0f);
        d.inv = 1.0f / (1.0f - rate);
        d.seed = seed;
        d.step = step_dev;
        SQ_REQUIRE(d.thr != 0u, "sq_conv3x3_first_block_dropout_pool_bf16: rate too small for the 16-bit threshold");
    }
    d.pool = reinterpret_cast<__bf16 *>(ypool);
    d.f_x = x; d.f_w = w1; d.f_b = b1;
    d.f_y1 = reinterpret_cast<__bf16 *>(y1);
    d.f_mask = reinterpret_cast<unsigned char *>(mask1);
    // the kernel's `x` is y1: never read (the halo is made in the block), it only passes the pointer checks
    return conv_fwd_bf16_impl(y1, wp2, b2, y, N, H, W, 16, 16, 3, SQ_ACT_RELU, stream, nullptr, d);
}

// weighted_conv2d followed by the discriminator's 2x2 average pool (gan.py:171-192) on bf16 tensors: y (N,H,W,Cout) = act(conv + bias)
// AND ypool (N,H/2,W/2,Cout) = the average pool of the stored y (sq_sumpool2x2_bf16(y, 0.25), bit for bit) from one kernel.
// K = 3, H and W even.
extern "C" int sq_conv2d_nhwc_fwd_avgpool_bf16(const void *x, const void *wp, const float *bias, void *y, void *ypool, int N, int H,
                                               int W, int Cin, int Cout, int act, void *stream) {
    SQ_REQUIRE(ypool && H % 2 == 0 && W % 2 == 0, "sq_conv2d_nhwc_fwd_avgpool_bf16: even H and W, ypool");
    SQ_REQUIRE_ALIGNED(ypool);
    SqDropEpi d{0u, 1.f, 0u, nullptr};
    d.pool = reinterpret_cast<__bf16 *>(ypool);
    d.pool_avg = 1;
    return conv_fwd_bf16_impl(x, wp, bias, y, N, H, W, Cin, Cout, 3, act, stream, nullptr, d);
}

// weighted_conv2d with norm=True (gan.py:86-97) on bf16 tensors: y (N,H,W,Cout) = act(conv3x3 + bias) AND ynorm = pixel_norm(y, eps)
// of the stored y from one kernel (the generator's blocks at Cout <= 64, the levels whose tensors are largest).  y may be NULL:
// then only ynorm is written (inference, or a forward pass nobody differentiates).  Cout % 8 == 0, Cout <= 64.
extern "C" int sq_conv2d_nhwc_fwd_pixelnorm_bf16(const void *x, const void *wp, const float *bias, void *y, void *ynorm, int N,
                                                 int H, int W, int Cin, int Cout, int act, float eps, void *stream) {
    SQ_REQUIRE(ynorm && Cout % 8 == 0 && Cout <= 64, "sq_conv2d_nhwc_fwd_pixelnorm_bf16: ynorm, Cout %% 8 == 0, Cout <= 64 (Cout=%d)", Cout);
    SQ_REQUIRE(eps >= 0.f, "sq_conv2d_nhwc_fwd_pixelnorm_bf16: eps must not be negative");
    SQ_REQUIRE_ALIGNED(ynorm);
    SqDropEpi d{0u, 1.f, 0u, nullptr};
    d.pn_y = reinterpret_cast<__bf16 *>(ynorm);
    d.pn_eps = eps;
    d.pn_store_y = y != nullptr;
    return conv_fwd_bf16_impl(x, wp, bias, y ? y : ynorm, N, H, W, Cin, Cout, 3, act, stream, nullptr, d);
}

// conv + bias + ReLU with the sign mask of the output beside it (mask: N*H*W*Cout/8 bytes, bit c & 7 of byte
// (pixel * Cout + c) >> 3; K = 3, Cout % 16 == 0): what the dgrad of the NEXT conv gates on, at 1/16 of the tensor's bytes
extern "C" int sq_conv2d_nhwc_fwd_mask_bf16(const void *x, const void *wp, const float *bias, void *y, void *mask, int N,
                                            int H, int W, int Cin, int Cout, int K, int act, void *stream) {
    SQ_REQUIRE(mask && K == 3 && Cout % 16 == 0, "sq_conv2d_nhwc_fwd_mask_bf16: mask, K = 3, Cout %% 16 == 0");
    SqDropEpi d{0u, 1.f, 0u, nullptr};
    d.mask = reinterpret_cast<unsigned char *>(mask);
    return conv_fwd_bf16_impl(x, wp, bias, y, N, H, W, Cin, Cout, K, act, stream, nullptr, d);
}

// sq_conv2d_nhwc_dgrad_gate_bf16 with the gate read from such a mask: dx passes (times scale) where the bit is set
extern "C" int sq_conv2d_nhwc_dgrad_maskgate_bf16(const void *dy, const void *wp_t, const void *mask, float scale, void *dx,
                                                  int N, int H, int W, int Cin, int Cout, int K, void *stream) {
    SQ_REQUIRE(mask && K == 3 && Cout % 16 == 0, "sq_conv2d_nhwc_dgrad_maskgate_bf16: mask, K = 3, Cout %% 16 == 0");
    SQ_REQUIRE(scale > 0.f, "sq_conv2d_nhwc_dgrad_maskgate_bf16: scale must be positive");
    SqDropEpi d{0u, 1.f, 0u, nullptr};
    d.gscale = scale;
    d.mask = const_cast<unsigned char *>(reinterpret_cast<const unsigned char *>(mask));
    // `gate` only selects the form here (non-NULL = read the mask); the tensor behind it is not touched
    return conv_fwd_bf16_impl(dy, wp_t, nullptr, dx, N, H, W, Cin, Cout, K, SQ_ACT_NONE, stream, mask, d);
}

// dX of a convolution whose INPUT was the ReLU output `gate` (N,H,W,Cout): the dgrad convolution of dy with
// the transposed packed filter, passed only where gate > 0 -- the upstream activation's backward fused in,
// one pass over dX saved.
extern "C" int sq_conv2d_nhwc_dgrad_relu_bf16(const void *dy, const void *wp_t, const void *gate, void *dx, int N, int H,
                                              int W, int Cin, int Cout, int K, void *stream) {
    SQ_REQUIRE(gate, "sq_conv2d_nhwc_dgrad_relu_bf16: null gate");
    return conv_fwd_bf16_impl(dy, wp_t, nullptr, dx, N, H, W, Cin, Cout, K, SQ_ACT_NONE, stream, gate);
}

// the same with a factor on what passes: dx = gate > 0 ? bf16(bf16(dgrad) * gate_scale) : 0 -- the backward of
// dropout(ReLU(.)) whose output is `gate` (gate > 0 <=> kept and active, gate_scale = 1 / (1 - rate)), with the two
// roundings of the stand-alone sq_relu_scale_bwd_bf16 pass it replaces (bit-identical to dgrad followed by that pass).
extern "C" int sq_conv2d_nhwc_dgrad_gate_bf16(const void *dy, const void *wp_t, const void *gate, float gate_scale,
                                              void *dx, int N, int H, int W, int Cin, int Cout, int K, void *stream) {
    SQ_REQUIRE(gate, "sq_conv2d_nhwc_dgrad_gate_bf16: null gate");
    SQ_REQUIRE(gate_scale > 0.f, "sq_conv2d_nhwc_dgrad_gate_bf16: gate_scale must be positive");
    SqDropEpi d{0u, 1.f, 0u, nullptr};
    d.gscale = gate_scale;
    return conv_fwd_bf16_impl(dy, wp_t, nullptr, dx, N, H, W, Cin, Cout, K, SQ_ACT_NONE, stream, gate, d);
}

// dgrad of the first conv of a decoder block, fused with the backward of the junction that produced its input
// (merged = bridge(up, skip), unet.py:312-319): g (N,H/2,W/2,4*Cout) = d_up in space-to-depth layout, dskip (N,H,W,Cout);
// dM itself is never written.  Bit-identical to sq_conv2d_nhwc_fwd_bf16 (dgrad pack) + sq_bridge_bwd_s2d_bf16.
extern "C" int sq_conv2d_nhwc_dgrad_junction_bf16(const void *dy, const void *wp_t, const void *up, const void *skip, void *g,
                                                  void *dskip, int N, int H, int W, int Cin, int Cout, int K, int bridge,
                                                  void *stream) {
    SQ_REQUIRE(g && dskip, "sq_conv2d_nhwc_dgrad_junction_bf16: null output");
    SQ_REQUIRE(K == 3, "sq_conv2d_nhwc_dgrad_junction_bf16: K=%d (the junction epilogue exists for the 3x3 form)", K);
    SQ_REQUIRE(bridge >= SQ_BRIDGE_ADD && bridge <= SQ_BRIDGE_SUB, "sq_conv2d_nhwc_dgrad_junction_bf16: bad bridge %d", bridge);
    SQ_REQUIRE(bridge != SQ_BRIDGE_MUL || (up && skip), "sq_conv2d_nhwc_dgrad_junction_bf16: eltwise_mul needs both forward operands");
    SQ_REQUIRE(H % 2 == 0 && W % 2 == 0 && Cout % 4 == 0, "sq_conv2d_nhwc_dgrad_junction_bf16: even H, W; Cout %% 4 == 0");
    SQ_REQUIRE_ALIGNED(g); SQ_REQUIRE_ALIGNED(dskip);
    SqDropEpi d{0u, 1.f, 0u, nullptr};
    d.j_up = reinterpret_cast<const __bf16 *>(up);
    d.j_skip = reinterpret_cast<const __bf16 *>(skip);
    d.j_g = reinterpret_cast<__bf16 *>(g);
    d.j_dskip = reinterpret_cast<__bf16 *>(dskip);
    d.j_bridge = bridge;
    return conv_fwd_bf16_impl(dy, wp_t, nullptr, dskip /* placeholder: never stored through */, N, H, W, Cin, Cout, K,
                              SQ_ACT_NONE, stream, nullptr, d);
}

// first conv of down0 in the bf16 graph: f32 image (1..7 channels) in, bf16 activation out.
extern "C" int sq_conv3x3_first_fwd_bf16(const float *x, const float *w, const float *bias, void *y, int N, int H,
                                         int W, int Cin, int Cout, int act, void *stream) {
    return sq_conv3x3_first_fwd_mask_bf16(x, w, bias, y, nullptr, N, H, W, Cin, Cout, act, stream);
}

// ... and the sign mask of the output (1 bit per element, NHWC order, Cout % 16 == 0) beside it; mask may be NULL
extern "C" int sq_conv3x3_first_fwd_mask_bf16(const float *x, const float *w, const float *bias, void *y, void *mask, int N,
                                              int H, int W, int Cin, int Cout, int act, void *stream) {
    SQ_REQUIRE(x && w && y && N > 0 && H > 0 && W > 0 && Cin >= 1 && Cin <= 7 && Cout > 0 && Cout % 8 == 0,
               "sq_conv3x3_first_fwd_bf16: bad arguments (Cin 1..7, Cout %% 8 == 0)");
    SQ_REQUIRE(!mask || Cout % 16 == 0, "sq_conv3x3_first_fwd_mask_bf16: the mask needs Cout %% 16 == 0");
    unsigned char *mk = reinterpret_cast<unsigned char *>(mask);
    SQ_REQUIRE_ALIGNED(y);
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const dim3 grid((unsigned)(tiles_x * tiles_y) * N, (Cout + 15) / 16);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    __bf16 *yo = reinterpret_cast<__bf16 *>(y);
#define SQ_FIRST(C)                                                                                                \
    case C:                                                                                                        \
        hipLaunchKernelGGL(conv_first_bf16_kernel<C>, grid, dim3(256), 0, st, x, w, bias, yo, N, H, W, Cout, act,  \
                           tiles_x, tiles_y, mk);                                                                  \
        break;
    switch (Cin) { SQ_FIRST(1) SQ_FIRST(2) SQ_FIRST(3) SQ_FIRST(4) SQ_FIRST(5) SQ_FIRST(6) SQ_FIRST(7) }
#undef SQ_FIRST
    return sq_check_launch("sq_conv3x3_first_fwd_bf16");
}
