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

}  // namespace

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
