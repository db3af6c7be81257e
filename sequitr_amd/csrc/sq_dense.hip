// Dense layer with a long reduction and very few rows (tf.layers.dense of discriminator_network,
// sequitr/networks/gan.py:226-237: 16*(f+1) = 8208 -> 512 on a batch of 32..96 samples).  As a 1x1
// convolution it is 32 output-channel blocks each walking all 8208 inputs in order (450 us for 0.27 GFLOP);
// the weights (16.8 MB) are the only real traffic, so the reduction is split over thread blocks instead:
//   part[s][m][n] = fmaf chain over k in slice s (k ascending), y[m][n] = act(sum_s part[s][m][n] + bias[n])
// with the slices added in order: deterministic, but NOT the single chain of the convolution kernels -- this
// path is used only for dense layers (ops.conv2d: K == 1, <= 128 rows, Cin >= 1024), whose parity is checked
// against fp64 with a tolerance, never bit for bit.
#include "sq_common.h"

namespace {

constexpr int DK = 64;       // reduction slice per block
constexpr int DM = 32;       // rows per block

__global__ __launch_bounds__(256) void dense_partial_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                            float *__restrict__ part, int M, int K, int N, float wscale) {
    __shared__ float4 xs[DK / 4][DM];                            // [k/4][m] -> the 4 consecutive k of row m
    const int s = blockIdx.x, n = blockIdx.y * 256 + threadIdx.x, m0 = blockIdx.z * DM;
    const int k0 = s * DK;
    for (int i = threadIdx.x; i < DM * (DK / 4); i += 256) {
        const int m = i % DM, kq = i / DM;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const int k = k0 + 4 * kq;
        if (m0 + m < M) {
            const float *src = x + (size_t)(m0 + m) * K + k;
            if (k + 3 < K) v = *reinterpret_cast<const float4 *>(src);
            else {
                if (k < K) v.x = src[0];
                if (k + 1 < K) v.y = src[1];
                if (k + 2 < K) v.z = src[2];
            }
        }
        xs[kq][m] = v;
    }
    __syncthreads();
    float acc[DM];
#pragma unroll
    for (int m = 0; m < DM; ++m) acc[m] = 0.f;
    if (n < N) {
        // 16 weights (four k-quads) of this column are in flight while the previous 16 are multiplied: loading four at a
        // time inside the loop exposed an HBM round trip per quad (40 us for 7 us of arithmetic)
        constexpr int WB = 16;
        float wn[WB];
#pragma unroll
        for (int j = 0; j < WB; ++j) wn[j] = (k0 + j < K) ? w[(size_t)(k0 + j) * N + n] : 0.f;
#pragma unroll 1
        for (int kb = 0; kb < DK; kb += WB) {
            float wc[WB];
#pragma unroll
            for (int j = 0; j < WB; ++j) wc[j] = wn[j] * wscale;
            if (kb + WB < DK) {
#pragma unroll
                for (int j = 0; j < WB; ++j) wn[j] = (k0 + kb + WB + j < K) ? w[(size_t)(k0 + kb + WB + j) * N + n] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < WB / 4; ++q) {
#pragma unroll
                for (int m = 0; m < DM; ++m) {
                    const float4 xv = xs[kb / 4 + q][m];
                    acc[m] = fmaf(wc[4 * q + 0], xv.x, acc[m]);
                    acc[m] = fmaf(wc[4 * q + 1], xv.y, acc[m]);
                    acc[m] = fmaf(wc[4 * q + 2], xv.z, acc[m]);
                    acc[m] = fmaf(wc[4 * q + 3], xv.w, acc[m]);
                }
            }
        }
#pragma unroll
        for (int m = 0; m < DM; ++m)
            if (m0 + m < M) part[((size_t)s * M + m0 + m) * N + n] = acc[m];
    }
}

// G thread groups per block: group g adds slices g, g + G, ... in order (four loads in flight), a fixed LDS tree folds the
// groups; threads of a group walk consecutive outputs (coalesced rows of 256 / G floats).  One thread walking all S slices
// was a chain of S dependent round trips (31 us for S = 129).
__global__ __launch_bounds__(256) void dense_finish_kernel(const float *__restrict__ part, const float *__restrict__ bias,
                                                           float *__restrict__ y, int S, int M, int N, int act, int G) {
    __shared__ float red[256];
    const int64_t total = (int64_t)M * N;
    const int OUT = 256 / G, ol = threadIdx.x % OUT, g = threadIdx.x / OUT;
    const int64_t i = (int64_t)blockIdx.x * OUT + ol;
    float s = 0.f;
    if (i < total) {
        const float *p = part + i;
        int b = g;
        for (; b + 3 * G < S; b += 4 * G) {
            const float v0 = p[(size_t)b * total], v1 = p[(size_t)(b + G) * total], v2 = p[(size_t)(b + 2 * G) * total],
                        v3 = p[(size_t)(b + 3 * G) * total];
            s = (((s + v0) + v1) + v2) + v3;
        }
        for (; b < S; b += G) s += p[(size_t)b * total];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int m = G >> 1; m > 0; m >>= 1) {
        if (g < m) red[threadIdx.x] += red[threadIdx.x + m * OUT];
        __syncthreads();
    }
    if (g != 0 || i >= total) return;
    float v = red[threadIdx.x];
    if (bias) v += bias[i % N];
    y[i] = sq_act(v, act);
}

// Weight gradient of such a layer: dW (K,N) = scale * x^T dY, db (N) = column sums of dY, x (M,K), dY (M,N), M <= 128 rows.
// As the 1x1 weight gradient of a "64-pixel image" it was 8208 blocks x one quarter-filled tile + a finish pass (61 us for
// 0.5 GFLOP); here a block owns WK rows of dW: its WK columns of x sit in LDS (read as broadcasts), a thread owns one
// output column and walks the M rows of dY once (fmaf chain in row order).  acc: bit 0 dW += , bit 1 db += .
constexpr int WK = 32;
__global__ __launch_bounds__(256) void dense_wgrad_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                          float *__restrict__ dw, float *__restrict__ db, int M, int K, int N,
                                                          float scale, int acc) {
    __shared__ float xs[128][WK];
    const int k0 = blockIdx.x * WK, n = blockIdx.y * 256 + threadIdx.x;
    for (int i = threadIdx.x; i < M * WK; i += 256) {
        const int m = i / WK, kk = i % WK;
        xs[m][kk] = k0 + kk < K ? x[(size_t)m * K + k0 + kk] : 0.f;
    }
    __syncthreads();
    if (n >= N) return;
    float a[WK], bs = 0.f;
#pragma unroll
    for (int kk = 0; kk < WK; ++kk) a[kk] = 0.f;
#pragma unroll 4
    for (int m = 0; m < M; ++m) {
        const float d = dy[(size_t)m * N + n];
        bs += d;
#pragma unroll
        for (int kk = 0; kk < WK; ++kk) a[kk] = fmaf(xs[m][kk], d, a[kk]);
    }
#pragma unroll
    for (int kk = 0; kk < WK; ++kk)
        if (k0 + kk < K) {
            const size_t o = (size_t)(k0 + kk) * N + n;
            const float v = scale == 1.0f ? a[kk] : a[kk] * scale;
            dw[o] = (acc & 1) ? dw[o] + v : v;
        }
    if (db && blockIdx.x == 0) db[n] = (acc & 2) ? db[n] + bs : bs;
}

}  // namespace

// dW (K,N) = scale * x^T dY and db (N, may be NULL) = column sums of dY for a dense layer's M <= 128 rows (f32, exact fmaf
// chains in row order); accumulate: bit 0 -- add to the contents of dw, bit 1 -- of db.
extern "C" int sq_dense_wgrad_f32(const float *x, const float *dy, float *dw, float *db, int M, int K, int N, float scale,
                                  int accumulate, void *stream) {
    SQ_REQUIRE(x && dy && dw && M > 0 && M <= 128 && K > 0 && N > 0, "sq_dense_wgrad_f32: bad arguments (1 <= M <= 128)");
    hipLaunchKernelGGL(dense_wgrad_kernel, dim3((K + WK - 1) / WK, (N + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, dy, dw,
                       db, M, K, N, scale, accumulate);
    return sq_check_launch("sq_dense_wgrad_f32");
}

extern "C" int64_t sq_dense_workspace_f32(int M, int K, int N) {
    if (M <= 0 || K <= 0 || N <= 0) return -1;
    return (int64_t)((K + DK - 1) / DK) * M * N * 4;
}

// y (M,N) = act(x (M,K) @ fl(w (K,N) * wscale) + bias): weighted dense layer with a split reduction
extern "C" int sq_dense_fwd_f32(const float *x, const float *w, const float *bias, float *y, float *workspace, int M, int K,
                                int N, float wscale, int act, void *stream) {
    SQ_REQUIRE(x && w && y && workspace && M > 0 && K > 0 && N > 0, "sq_dense_fwd_f32: bad arguments");
    SQ_REQUIRE(K % 4 == 0, "sq_dense_fwd_f32: K=%d must be a multiple of 4", K);
    SQ_REQUIRE(act >= SQ_ACT_NONE && act <= SQ_ACT_LEAKY, "sq_dense_fwd_f32: bad activation %d", act);
    SQ_REQUIRE_ALIGNED(x);
    hipStream_t st = (hipStream_t)stream;
    const int S = (K + DK - 1) / DK;
    dim3 grid(S, (N + 255) / 256, (M + DM - 1) / DM);
    hipLaunchKernelGGL(dense_partial_kernel, grid, dim3(256), 0, st, x, w, workspace, M, K, N, wscale);
    int rc = sq_check_launch("sq_dense_fwd_f32(partial)");
    if (rc) return rc;
    int G = sq_group_size(S);
    if (G > 8) G = 8;                                           // >= 32 consecutive floats per load row
    const int64_t nb = ((int64_t)M * N + 256 / G - 1) / (256 / G);
    hipLaunchKernelGGL(dense_finish_kernel, dim3((unsigned)nb), dim3(256), 0, st, workspace, bias, y, S, M, N, act, G);
    return sq_check_launch("sq_dense_fwd_f32");
}
