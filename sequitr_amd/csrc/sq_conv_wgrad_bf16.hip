// Weight gradient of the KxK SAME convolution on bf16 activations, gfx950:
//   dW[tap][ci][co] = sum_p X[p + tap][ci] * dY[p][co],  db[co] = sum_p dY[p][co]   (fp32 results)
// on v_mfma_f32_16x16x32_bf16 with the reduction over PIXELS (32 per instruction).  Both operands are
// needed "channel-major over pixels" while NHWC tiles in LDS are pixel-major: ds_read_b64_tr_b16 (the
// gfx950 transposing LDS read) delivers, per 16-lane group, a 4-pixel x 16-channel block with lane i
// receiving channel i of the 4 pixels -- exactly the MFMA fragment, no shuffles, no transposed copy.
// Semantics were confirmed on hardware with tools/probe_tr16.hip.
//
// Block = (16-channel ci chunk, 16-channel co chunk) pair x a run of 16x16 pixel tiles (persistent,
// accumulators stay in registers across the run); wave w owns tile rows 4w..4w+3 = two 32-pixel k-steps.
// Pixel <-> k mapping of one k-step (rows r, r+1 of the tile), lane group g, element e:
//     row = r + (g >> 1),  x = 4*(g & 1) + 8*(e >> 2) + (e & 3)
// chosen so the two groups of a 32-lane half read blocks 128 B apart (conflict-free).
// Partials + fixed-order finish kernel as in sq_conv_wgrad_f32.hip (no float atomics).
#include "sq_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int TH = 16, TW = 16, PSB = 32;          // 16 bf16 channels per pixel row in LDS

template <int KS>
struct WB {
    static constexpr int HALO_W = TW + KS - 1;
    static constexpr int HP = HALO_W * (TH + KS - 1);
    static constexpr int NTAP = KS * KS;
    static constexpr int XS_BYTES = HP * PSB;
    static constexpr int YS_BYTES = TH * TW * PSB;
    static constexpr int ROWS = NTAP * 16 + 1;
    static constexpr int RED_FLOATS = ROWS * 16;
    static constexpr int LDS_BYTES = (XS_BYTES + YS_BYTES) > RED_FLOATS * 4 ? (XS_BYTES + YS_BYTES) : RED_FLOATS * 4;
    static constexpr int XITEMS = HP * 2, XSLOTS = (XITEMS + 255) / 256;
    static constexpr int YITEMS = TH * TW * 2, YSLOTS = YITEMS / 256;
};

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char *p0, const unsigned char *p1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p1);
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

template <int KS>
__global__ __launch_bounds__(256, 2) void conv_wgrad_bf16_kernel(
    const __bf16 *__restrict__ x, const __bf16 *__restrict__ dy, float *__restrict__ partials, int N, int H,
    int W, int Cin, int Cout, int tiles_x, int tiles_y, int ntiles, int tiles_per_block) {
    using C = WB<KS>;
    constexpr int PAD = KS / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *xs = smem, *ys = smem + C::XS_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, kg = lane >> 4, q = li >> 2, p = li & 3;
    const int nco = Cout / 16;
    const int ci0 = (blockIdx.y / nco) * 16, co0 = (blockIdx.y % nco) * 16;
    const int t_begin = blockIdx.x * tiles_per_block, t_end = min(t_begin + tiles_per_block, ntiles);

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16 *>(x), 0, (int)((size_t)N * H * W * Cin * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16 *>(dy), 0, (int)((size_t)N * H * W * Cout * 2), 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    int xrel[C::XSLOTS], xpy[C::XSLOTS], xpx[C::XSLOTS];
#pragma unroll
    for (int sl = 0; sl < C::XSLOTS; ++sl) {
        const int idx = tid + sl * 256, pix = idx >> 1, h = idx & 1;
        xpy[sl] = pix / C::HALO_W;
        xpx[sl] = pix % C::HALO_W;
        xrel[sl] = idx < C::XITEMS ? ((xpy[sl] * W + xpx[sl]) * Cin + ci0 + h * 8) * 2 : (int)OOB;
    }
    int yrel[C::YSLOTS], ypy[C::YSLOTS], ypx[C::YSLOTS];
#pragma unroll
    for (int sl = 0; sl < C::YSLOTS; ++sl) {
        const int idx = tid + sl * 256, pix = idx >> 1, h = idx & 1;
        ypy[sl] = pix / TW;
        ypx[sl] = pix % TW;
        yrel[sl] = ((ypy[sl] * W + ypx[sl]) * Cout + co0 + h * 8) * 2;
    }
    uint4 xr[C::XSLOTS], yr[C::YSLOTS];
    auto issue = [&](int tile) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
        const int x0 = tx * TW, y0 = ty * TH;
        const int xbase = (((n * H + y0 - PAD) * W + x0 - PAD) * Cin) * 2;
        const int ybase = (((n * H + y0) * W + x0) * Cout) * 2;
#pragma unroll
        for (int sl = 0; sl < C::XSLOTS; ++sl) {
            const bool inb = (unsigned)(y0 - PAD + xpy[sl]) < (unsigned)H &&
                             (unsigned)(x0 - PAD + xpx[sl]) < (unsigned)W && xrel[sl] != (int)OOB;
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, inb ? (unsigned)(xbase + xrel[sl]) : OOB, 0, 0);
            xr[sl] = *reinterpret_cast<const uint4 *>(&v);
        }
#pragma unroll
        for (int sl = 0; sl < C::YSLOTS; ++sl) {
            const bool inb = (y0 + ypy[sl]) < H && (x0 + ypx[sl]) < W;
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(yrsrc, inb ? (unsigned)(ybase + yrel[sl]) : OOB, 0, 0);
            yr[sl] = *reinterpret_cast<const uint4 *>(&v);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int sl = 0; sl < C::XSLOTS; ++sl) {
            const int idx = tid + sl * 256;
            if (idx < C::XITEMS) *reinterpret_cast<uint4 *>(xs + idx * 16) = xr[sl];
        }
#pragma unroll
        for (int sl = 0; sl < C::YSLOTS; ++sl) *reinterpret_cast<uint4 *>(ys + (tid + sl * 256) * 16) = yr[sl];
    };

    f32x4 acc[C::NTAP];
#pragma unroll
    for (int t = 0; t < C::NTAP; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;

    // this lane's tr-read address inside a k-step: pixel (row kg>>1, x 4*(kg&1) + q), 8-byte piece p;
    // the second read of a fragment is 8 pixels further along the row
    const int lane_px = (kg >> 1) * TW + 4 * (kg & 1) + q;                 // in the dY tile
    const int lane_hx = (kg >> 1) * C::HALO_W + 4 * (kg & 1) + q;          // in the X halo
    const unsigned char *yb = ys + ((4 * wv) * TW + lane_px) * PSB + p * 8;
    const unsigned char *xa = xs + ((4 * wv) * C::HALO_W + lane_hx) * PSB + p * 8;

    if (t_begin < t_end) {
        issue(t_begin);
        commit();
    }
    __syncthreads();
    for (int tile = t_begin; tile < t_end; ++tile) {
        const bool has_next = tile + 1 < t_end;
        if (has_next) issue(tile + 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const unsigned char *yk = yb + (2 * ks) * TW * PSB;
            const bf16x8 b = tr_frag(yk, yk + 8 * PSB);
#pragma unroll
            for (int e = 0; e < 8; ++e) bsum += (float)b[e];
#pragma unroll
            for (int t = 0; t < C::NTAP; ++t) {
                const unsigned char *xk = xa + ((2 * ks + t / KS) * C::HALO_W + t % KS) * PSB;
                const bf16x8 a = tr_frag(xk, xk + 8 * PSB);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[t], 0, 0, 0);
            }
        }
        __syncthreads();
        if (has_next) {
            commit();
            __syncthreads();
        }
    }

    // cross-wave reduction in wave order, then one partial per block (layout of the f32 kernel)
    float *red = reinterpret_cast<float *>(smem);
    bsum += __shfl_xor(bsum, 16);
    bsum += __shfl_xor(bsum, 32);
    for (int w = 0; w < 4; ++w) {
        if (wv == w) {
#pragma unroll
            for (int t = 0; t < C::NTAP; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float *d = red + (t * 16 + 4 * kg + j) * 16 + li;
                    *d = (w == 0) ? acc[t][j] : *d + acc[t][j];
                }
            if (kg == 0) {
                float *d = red + (C::NTAP * 16) * 16 + li;
                *d = (w == 0) ? bsum : *d + bsum;
            }
        }
        __syncthreads();
    }
    float *out = partials + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * C::RED_FLOATS;
    for (int i = tid; i < C::RED_FLOATS; i += 256) out[i] = red[i];
}

template <int KS>
__global__ __launch_bounds__(256) void conv_wgrad_bf16_finish_kernel(const float *__restrict__ partials,
                                                                      float *__restrict__ dw, float *__restrict__ db,
                                                                      int nblk, int Cin, int Cout, int G) {
    using C = WB<KS>;
    const int nco = Cout / 16, npairs = (Cin / 16) * nco;
    const int total = C::NTAP * Cin * Cout;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int i = t / G, g = t % G;
    const size_t stride = (size_t)npairs * C::RED_FLOATS;
    if (i < total) {
        const int co = i % Cout, ci = (i / Cout) % Cin, tap = i / (Cout * Cin);
        const size_t off = (size_t)((ci / 16) * nco + co / 16) * C::RED_FLOATS + (tap * 16 + ci % 16) * 16 + co % 16;
        const float s = sq_group_reduce(partials + off, stride, nblk, g, G);
        if (g == 0) dw[i] = s;
    } else if (i < total + Cout) {
        const int co = i - total;
        const size_t off = (size_t)(co / 16) * C::RED_FLOATS + (C::NTAP * 16) * 16 + co % 16;
        const float s = sq_group_reduce(partials + off, stride, nblk, g, G);
        if (g == 0 && db) db[co] = s;
    }
}

template <int KS>
void plan(int N, int H, int W, int Cin, int Cout, int *gx, int *tpb, int64_t *ws_floats) {
    const int ntiles = ((W + TW - 1) / TW) * ((H + TH - 1) / TH) * N;
    const int npairs = (Cin / 16) * (Cout / 16);
    int want = (512 + npairs - 1) / npairs;
    if (want < 1) want = 1;
    int t = (ntiles + want - 1) / want;
    if (t < 1) t = 1;
    *tpb = t;
    *gx = (ntiles + t - 1) / t;
    *ws_floats = (int64_t)(*gx) * npairs * WB<KS>::RED_FLOATS;
}

template <int KS>
int launch(const __bf16 *x, const __bf16 *dy, float *dw, float *db, float *ws, int N, int H, int W, int Cin,
           int Cout, hipStream_t st) {
    using C = WB<KS>;
    int gx, tpb;
    int64_t wsf;
    plan<KS>(N, H, W, Cin, Cout, &gx, &tpb, &wsf);
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int npairs = (Cin / 16) * (Cout / 16);
    hipLaunchKernelGGL(conv_wgrad_bf16_kernel<KS>, dim3(gx, npairs), dim3(256), C::LDS_BYTES, st, x, dy, ws, N, H, W,
                       Cin, Cout, tiles_x, tiles_y, tiles_x * tiles_y * N, tpb);
    int rc = sq_check_launch("sq_conv2d_nhwc_wgrad_bf16");
    if (rc) return rc;
    const int G = sq_group_size(gx);
    const int64_t total = ((int64_t)KS * KS * Cin * Cout + Cout) * G;
    hipLaunchKernelGGL(conv_wgrad_bf16_finish_kernel<KS>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, ws, dw,
                       db, gx, Cin, Cout, G);
    return sq_check_launch("sq_conv2d_nhwc_wgrad_bf16(finish)");
}

bool ok_shape(int N, int H, int W, int Cin, int Cout, int K) {
    return N > 0 && H > 0 && W > 0 && (K == 1 || K == 3) && Cin > 0 && Cin % 16 == 0 && Cout > 0 && Cout % 16 == 0 &&
           (size_t)N * H * W * (size_t)(Cin > Cout ? Cin : Cout) * 2 < ((size_t)1 << 31);
}

}  // namespace

extern "C" int64_t sq_conv2d_nhwc_wgrad_workspace_bf16(int N, int H, int W, int Cin, int Cout, int K) {
    if (!ok_shape(N, H, W, Cin, Cout, K)) return -1;
    int gx, tpb;
    int64_t wsf;
    if (K == 3) plan<3>(N, H, W, Cin, Cout, &gx, &tpb, &wsf);
    else plan<1>(N, H, W, Cin, Cout, &gx, &tpb, &wsf);
    return wsf * 4;
}

// dW (K,K,Cin,Cout) f32 and db (Cout) f32 from bf16 X (N,H,W,Cin) and bf16 dY (N,H,W,Cout).
extern "C" int sq_conv2d_nhwc_wgrad_bf16(const void *x, const void *dy, float *dw, float *db, float *workspace,
                                         int N, int H, int W, int Cin, int Cout, int K, void *stream) {
    SQ_REQUIRE(x && dy && dw && workspace, "sq_conv2d_nhwc_wgrad_bf16: null pointer");
    SQ_REQUIRE(ok_shape(N, H, W, Cin, Cout, K),
               "sq_conv2d_nhwc_wgrad_bf16: unsupported shape Cin=%d Cout=%d K=%d (both %% 16, K 1|3, < 2 GiB)", Cin, Cout, K);
    SQ_REQUIRE_ALIGNED(x); SQ_REQUIRE_ALIGNED(dy); SQ_REQUIRE_ALIGNED(workspace);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const __bf16 *xb = reinterpret_cast<const __bf16 *>(x), *yb = reinterpret_cast<const __bf16 *>(dy);
    return K == 3 ? launch<3>(xb, yb, dw, db, workspace, N, H, W, Cin, Cout, st)
                  : launch<1>(xb, yb, dw, db, workspace, N, H, W, Cin, Cout, st);
}
