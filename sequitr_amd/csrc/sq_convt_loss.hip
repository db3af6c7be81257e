// conv_transpose_layer hook (sequitr/networks/unet.py:336-338) fused with the up_layer
// bridge (unet.py:312-319), and the weighted softmax cross-entropy.
#include <stdlib.h>
#include "sq_common.h"

namespace {

// ---- 2x2 stride-2 transpose convolution as a GEMM on v_mfma_f32_16x16x4_f32 --------
// Non-overlapping: every input pixel p=(n,i,j) produces the 2x2 output patch
//   y[n,2i+a,2j+b,o] = chain_c fmaf(w[a,b,o,c], x[p,c]) + bias[o]   (c ascending)
// i.e. D[rho][p] with rho = (2a+b)*Cout + o and Wt[rho][c] = the TF kernel (2,2,Cout,Cin)
// read flat.  Block = 64 rows x 64 input pixels, wave w owns rows 16w..16w+15 and four
// 16-pixel column blocks; a lane ends up with 4 consecutive o of one output pixel, so the
// bridge operand is one 16-B load and the result one 16-B store.
// KCH input channels are staged per barrier pair: 16 (any Cin % 16 == 0) or 32 when Cin % 32 == 0 (same chain, c
// ascending).  Measured on the decoder's three launches (Cin 64 / 128 / 256): KCH 16: 124 us average, KCH 64: 149 us
// (33.8 KB of LDS per block: half the resident blocks, and these launches live on occupancy).
// Round 3: the block's memory round trips no longer queue up behind each other.  Before, a block paid one HBM latency per
// staged chunk (load -> LDS -> barrier -> MFMA -> barrier, nothing in flight during the MFMAs) and one more in the
// epilogue for the bridge operand: Cin / KCH + 1 serial round trips per block, 3.2 TB/s over the decoder's three
// launches.  Now (i) the four bridge quads of a lane are requested before the first chunk and are in registers when the
// epilogue needs them, (ii) chunk c + 1 is requested into registers before the MFMAs of chunk c and committed to LDS
// after them (the T14 split staging of the conv kernels).  Same fmaf chains, same bits.
template <int KCH>
__global__ __launch_bounds__(256) void convT2x2_mfma_f32_kernel(
    const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
    const float *__restrict__ skip, float *__restrict__ y, int64_t P, int H, int W, int Cin,
    int Cout, int bridge) {
    constexpr int CT_PS = KCH + 2;  // chunk + 2 pad floats: conflict-free ds_read_b32 (stride = 2 mod 32)
    __shared__ __attribute__((aligned(16))) float as[64 * CT_PS];
    __shared__ __attribute__((aligned(16))) float xs[64 * CT_PS];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, kk = lane >> 4;
    const int64_t p0 = (int64_t)blockIdx.x * 64;
    const int r0 = blockIdx.y * 64;

    // ---- epilogue geometry first: the bridge operand is requested NOW -----------------------------------
    const int rho = r0 + 16 * wv + 4 * kk;
    const bool row_ok = rho < 4 * Cout;
    const int ab = row_ok ? rho / Cout : 0, o = row_ok ? rho % Cout : 0;
    const int pa = ab >> 1, pb = ab & 1;
    size_t off[4];
    bool ok[4];
    float4 sk[4];
    // pixel -> (n, i, j): ONE division pair per lane (32-bit whenever the pixel count allows: a 64-bit division is
    // ~100 VALU instructions and the kernel issued 6 of them per lane for every 16 MFMAs), then the three other
    // column blocks by carrying 16 pixels forward
    int64_t nn;
    int ii, jj;
    {
        const int64_t pl = p0 + li < P ? p0 + li : 0;
        if (P < ((int64_t)1 << 31)) {
            const unsigned pu = (unsigned)pl, t = pu / (unsigned)W;
            jj = (int)(pu - t * (unsigned)W);
            const unsigned n32 = t / (unsigned)H;
            ii = (int)(t - n32 * (unsigned)H);
            nn = n32;
        } else {
            jj = (int)(pl % W);
            const int64_t t = pl / W;
            ii = (int)(t % H);
            nn = t / H;
        }
    }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        const int64_t p = p0 + cb * 16 + li;
        ok[cb] = row_ok && p < P;
        off[cb] = ((size_t)(nn * 2 * H + 2 * ii + pa) * (2 * W) + 2 * jj + pb) * Cout + o;
        sk[cb] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bridge != SQ_BRIDGE_NONE && ok[cb]) sk[cb] = *reinterpret_cast<const float4 *>(skip + off[cb]);
        jj += 16;                                               // next column block: 16 pixels on
        while (jj >= W) {
            jj -= W;
            if (++ii == H) { ii = 0; ++nn; }
        }
    }

    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias && row_ok) bv = *reinterpret_cast<const float4 *>(bias + o);

    f32x4 acc[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) acc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    constexpr int QPR = KCH / 4;              // float4 per staged row
    constexpr int NIT = QPR / 4;              // staging: 64 rows x QPR float4 over 256 threads
    float4 vr[NIT], ur[NIT];
    auto issue = [&](int cc) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * 256, srow = idx / QPR, sq = idx % QPR;
            vr[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            ur[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r0 + srow < 4 * Cout)                           // ragged last row block (Cout % 16 != 0)
                vr[it] = *reinterpret_cast<const float4 *>(w + (size_t)(r0 + srow) * Cin + cc + sq * 4);
            if (p0 + srow < P)
                ur[it] = *reinterpret_cast<const float4 *>(x + (size_t)(p0 + srow) * Cin + cc + sq * 4);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * 256, srow = idx / QPR, sq = idx % QPR;
            float *d = as + srow * CT_PS + sq * 4;
            *reinterpret_cast<float2 *>(d) = make_float2(vr[it].x, vr[it].y);
            *reinterpret_cast<float2 *>(d + 2) = make_float2(vr[it].z, vr[it].w);
            float *e = xs + srow * CT_PS + sq * 4;
            *reinterpret_cast<float2 *>(e) = make_float2(ur[it].x, ur[it].y);
            *reinterpret_cast<float2 *>(e + 2) = make_float2(ur[it].z, ur[it].w);
        }
    };
    issue(0);
    for (int cc = 0; cc < Cin; cc += KCH) {
        commit();
        __syncthreads();
        if (cc + KCH < Cin) issue(cc + KCH);                    // in flight during this chunk's MFMAs
#pragma unroll 4
        for (int s = 0; s < QPR; ++s) {
            const float a = as[(16 * wv + li) * CT_PS + s * 4 + kk];
            float b[4];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) b[cb] = xs[(cb * 16 + li) * CT_PS + s * 4 + kk];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
                acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[cb], acc[cb], 0, 0, 0);
        }
        __syncthreads();
    }

    if (!row_ok) return;                                        // after the last barrier: safe to leave
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        if (!ok[cb]) continue;
        float4 v = make_float4(acc[cb][0] + bv.x, acc[cb][1] + bv.y, acc[cb][2] + bv.z, acc[cb][3] + bv.w);
        if (!bias) v = make_float4(acc[cb][0], acc[cb][1], acc[cb][2], acc[cb][3]);
        if (bridge != SQ_BRIDGE_NONE) {
            const float4 k = sk[cb];
            if (bridge == SQ_BRIDGE_ADD) v = make_float4(v.x + k.x, v.y + k.y, v.z + k.z, v.w + k.w);
            else if (bridge == SQ_BRIDGE_MUL) v = make_float4(v.x * k.x, v.y * k.y, v.z * k.z, v.w * k.w);
            else v = make_float4(v.x - k.x, v.y - k.y, v.z - k.z, v.w - k.w);
        }
        *reinterpret_cast<float4 *>(y + off[cb]) = v;
    }
}

// ---- weighted softmax cross-entropy, forward + backward in one pass over the logits -----
// Pure HBM: per pixel reads C f32 logits + C u8 labels + 1 f32 weight, writes C f32 dlogits.
// Loss partials are summed in fp64 in a fixed order (thread grid-stride order, then a
// fixed LDS tree, then one thread over the block partials) => run-to-run reproducible.
constexpr int CE_MAXC = 8;

__global__ __launch_bounds__(256) void wsoftmax_ce_f32_kernel(
    const float *__restrict__ z, const uint8_t *__restrict__ yoh, const float *__restrict__ wgt,
    int64_t npix, int C, float gscale, double *__restrict__ partials, float *__restrict__ dz) {
    __shared__ double red[256];
    double tsum = 0.0;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (int64_t)gridDim.x * 256) {
        float zc[CE_MAXC], yc[CE_MAXC], dzc[CE_MAXC];
#pragma unroll
        for (int c = 0; c < CE_MAXC; ++c) {
            zc[c] = yc[c] = 0.f;
            if (c < C) {
                zc[c] = z[p * C + c];
                yc[c] = (float)yoh[p * C + c];
            }
        }
        const float wp = wgt[p];
        tsum += (double)sq_wce_pixel<CE_MAXC>(zc, yc, C, wp, wp * gscale, dz ? dzc : nullptr);
        if (dz) {
#pragma unroll
            for (int c = 0; c < CE_MAXC; ++c)
                if (c < C) dz[p * C + c] = dzc[c];
        }
    }
    red[threadIdx.x] = tsum;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void ce_finish_kernel(const double *__restrict__ partials, int n, double inv_npix,
                                                        double *__restrict__ loss) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += partials[i];      // fixed assignment, fixed tree
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = red[0] * inv_npix;
}

inline int64_t ce_blocks(int64_t npix) {
    int64_t b = (npix + 255) / 256;
    return b > 2048 ? 2048 : (b < 1 ? 1 : b);
}

}  // namespace

int sq_convT_v2_launch(const float *x, const float *w, const float *bias, const float *skip, float *y, int N, int H,
                       int W, int Cin, int Cout, int bridge, hipStream_t st);

extern "C" int sq_convT2x2s2_nhwc_fwd_f32(const float *x, const float *w, const float *bias,
                                          const float *skip, float *y, int N, int H, int W, int Cin,
                                          int Cout, int bridge, void *stream) {
    SQ_REQUIRE(x && w && y, "sq_convT2x2s2_nhwc_fwd_f32: null tensor pointer");
    SQ_REQUIRE(N > 0 && H > 0 && W > 0, "sq_convT2x2s2_nhwc_fwd_f32: bad shape");
    SQ_REQUIRE(Cin % 16 == 0 && Cout % 4 == 0 && Cin > 0 && Cout > 0,
               "sq_convT2x2s2_nhwc_fwd_f32: Cin=%d must be a multiple of 16, Cout=%d of 4", Cin, Cout);
    SQ_REQUIRE(bridge >= SQ_BRIDGE_NONE && bridge <= SQ_BRIDGE_SUB, "sq_convT2x2s2_nhwc_fwd_f32: bad bridge %d", bridge);
    SQ_REQUIRE(bridge == SQ_BRIDGE_NONE || skip, "sq_convT2x2s2_nhwc_fwd_f32: bridge needs a skip tensor");
    SQ_REQUIRE_ALIGNED(x); SQ_REQUIRE_ALIGNED(w); SQ_REQUIRE_ALIGNED(y);
    if (bias) SQ_REQUIRE_ALIGNED(bias);
    if (skip) SQ_REQUIRE_ALIGNED(skip);
    {
        const int r = sq_convT_v2_launch(x, w, bias, skip, y, N, H, W, Cin, Cout, bridge, reinterpret_cast<hipStream_t>(stream));
        if (r != SQ_NOT_MINE) return r;
    }
    const int64_t P = (int64_t)N * H * W;
    dim3 grid((unsigned)((P + 63) / 64), (unsigned)((4 * Cout + 63) / 64));
    static const int kch = [] { const char *e = getenv("SQ_CONVT_KCH"); return e ? atoi(e) : 32; }();   // A/B switch
    if (Cin % 32 == 0 && kch == 32)
        hipLaunchKernelGGL(convT2x2_mfma_f32_kernel<32>, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                           x, w, bias, skip, y, P, H, W, Cin, Cout, bridge);
    else
        hipLaunchKernelGGL(convT2x2_mfma_f32_kernel<16>, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                           x, w, bias, skip, y, P, H, W, Cin, Cout, bridge);
    return sq_check_launch("sq_convT2x2s2_nhwc_fwd_f32");
}

extern "C" int64_t sq_wsoftmax_ce_partials(int64_t npix) { return npix > 0 ? ce_blocks(npix) : 0; }

extern "C" int sq_wsoftmax_ce_fwd_bwd_f32(const float *logits, const uint8_t *onehot,
                                          const float *weights, int64_t npix, int C, float grad_scale,
                                          double *partials, double *loss, float *dlogits, void *stream) {
    SQ_REQUIRE(logits && onehot && weights && partials && loss, "sq_wsoftmax_ce_fwd_bwd_f32: null pointer");
    SQ_REQUIRE(npix > 0 && C >= 1 && C <= CE_MAXC, "sq_wsoftmax_ce_fwd_bwd_f32: C=%d unsupported (1..%d)", C, CE_MAXC);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nb = (int)ce_blocks(npix);
    hipLaunchKernelGGL(wsoftmax_ce_f32_kernel, dim3(nb), dim3(256), 0, st, logits, onehot, weights, npix, C,
                       grad_scale / (float)npix, partials, dlogits);
    int rc = sq_check_launch("sq_wsoftmax_ce_fwd_bwd_f32");
    if (rc) return rc;
    hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(256), 0, st, partials, nb, 1.0 / (double)npix, loss);
    return sq_check_launch("sq_wsoftmax_ce_fwd_bwd_f32(finish)");
}
