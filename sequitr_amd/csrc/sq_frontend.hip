// Tile front end on the GPU (the caller side of the inference hot path).
// Raw camera frames (uint8 / uint16 / float32, single channel; OctopusData .dat, dataio/octopus.py:231-245)
// -> ImageNorm (sequitr/pipeline.py:350-356: (x - mean) / (1e-99 + std) per frame, float32)
// -> fixed-size network tiles (N,T,T,1) f32 in HBM;  and back: tile masks -> full-frame masks.
//
// ImageNorm parity is BIT-EXACT with numpy, which means reproducing numpy's float32 summation order:
// np.mean / np.std reduce a contiguous float32 array in chunks of 8192 elements (the ufunc buffer size),
//     res = 0;  for every chunk: res = res + pairwise(chunk)
// and pairwise() splits recursively at n2 = n/2 - (n/2) % 8 down to blocks of <= 128 elements, each summed
// with 8 interleaved accumulators r[j] += a[8 i + j] combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)).
// A full chunk is a perfect binary tree over 64 blocks: one wave per chunk (lane = block, 16-byte loads feed
// the 8 accumulators directly), adjacent-pair shuffles for the tree.  The ragged last chunk follows the
// recursion literally on one thread.  HBM-bound: the frame is read 3 times (mean, variance, normalise) at
// 1-4 B/pixel and the tiles written once at 4 B/pixel.
#include "sq_common.h"

// every multiply and add below is a separate, correctly rounded operation as in numpy: no fused contraction
#pragma clang fp contract(off)

namespace {

constexpr int CHUNK = 8192, LEAF = 128;

template <typename T> __device__ __forceinline__ float ld(const T *p, int64_t i) { return (float)p[i]; }

// element transform: plain value (mean pass) or squared deviation (variance pass), all in float32
template <bool SQ> __device__ __forceinline__ float xf(float v, float mean) {
    if (SQ) {
        const float d = v - mean;
        return d * d;
    }
    return v;
}

// numpy's pairwise block for n <= 128 (n >= 8 takes the 8-accumulator path)
template <typename T, bool SQ>
__device__ float pw_block(const T *a, int n, float mean) {
    if (n < 8) {
        float res = 0.f;
        for (int i = 0; i < n; ++i) res += xf<SQ>(ld(a, i), mean);
        return res;
    }
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = xf<SQ>(ld(a, j), mean);
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] += xf<SQ>(ld(a, i + j), mean);
    }
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += xf<SQ>(ld(a, i), mean);
    return res;
}

template <typename T, bool SQ>
__device__ float pw_rec(const T *a, int n, float mean) {
    if (n <= LEAF) return pw_block<T, SQ>(a, n, mean);
    int n2 = n / 2;
    n2 -= n2 % 8;
    const float l = pw_rec<T, SQ>(a, n2, mean);
    const float r = pw_rec<T, SQ>(a + n2, n - n2, mean);
    return l + r;
}

// chunk_sums[f][c] for every 8192-element chunk c of frame f.  blockDim = 256 = 4 waves = 4 chunks.
template <typename T, bool SQ>
__global__ __launch_bounds__(256) void frame_chunk_sums_kernel(const T *__restrict__ frames,
                                                               const float *__restrict__ mean,
                                                               float *__restrict__ chunk_sums, int64_t npix,
                                                               int nchunks) {
    const int f = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= nchunks) return;
    const T *a = frames + (size_t)f * npix + (size_t)c * CHUNK;
    const int64_t left = npix - (int64_t)c * CHUNK;
    const float m = SQ ? mean[f] : 0.f;
    float s;
    if (left >= CHUNK) {
        s = pw_block<T, SQ>(a + lane * LEAF, LEAF, m);         // lane = block of the perfect tree
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const float o = __shfl_down(s, d);
            s = s + o;                                          // only lanes that are multiples of 2d matter
        }
    } else {
        s = lane == 0 ? pw_rec<T, SQ>(a, (int)left, m) : 0.f;   // ragged tail: the recursion, literally
    }
    if (lane == 0) chunk_sums[(size_t)f * nchunks + c] = s;
}

// res = 0; res += chunk (in order); then mean = res / n   or   std = sqrt(res / n)
__global__ void frame_stats_finish_kernel(const float *__restrict__ chunk_sums, float *__restrict__ out, int F,
                                          int nchunks, float n, int take_sqrt) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    float res = 0.f;
    for (int c = 0; c < nchunks; ++c) res = res + chunk_sums[(size_t)f * nchunks + c];
    res = res / n;
    out[f] = take_sqrt ? sqrtf(res) : res;
}

// tile t = (f, ty, tx): out[t][y][x] = (frame[f][oy[ty]+y][ox[tx]+x] - mean[f]) / std[f]
template <typename T>
__global__ __launch_bounds__(256) void tiles_norm_kernel(const T *__restrict__ frames, const float *__restrict__ mean,
                                                         const float *__restrict__ stdv, const int *__restrict__ oy,
                                                         const int *__restrict__ ox, float *__restrict__ tiles, int F,
                                                         int H, int W, int TR, int TC, int TS, int normalise) {
    const int64_t total = (int64_t)F * TR * TC * TS * TS;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % TS);
        int64_t t = i / TS;
        const int y = (int)(t % TS);
        t /= TS;
        const int tx = (int)(t % TC);
        t /= TC;
        const int ty = (int)(t % TR);
        const int f = (int)(t / TR);
        const float v = (float)frames[((size_t)f * H + oy[ty] + y) * W + ox[tx] + x];
        tiles[i] = normalise ? (v - mean[f]) / stdv[f] : v;
    }
}

// full-frame mask: pixel (y, x) takes the tile that owns it (ymap / xmap: tile index << 16 | local coordinate)
__global__ __launch_bounds__(256) void stitch_masks_kernel(const uint8_t *__restrict__ tile_masks,
                                                           const int *__restrict__ ymap, const int *__restrict__ xmap,
                                                           uint8_t *__restrict__ out, int F, int H, int W, int TR, int TC,
                                                           int TS) {
    const int64_t total = (int64_t)F * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)((i / W) % H), f = (int)(i / ((int64_t)H * W));
        const int ym = ymap[y], xm = xmap[x];
        const size_t t = ((size_t)f * TR + (ym >> 16)) * TC + (xm >> 16);
        out[i] = tile_masks[(t * TS + (ym & 0xffff)) * TS + (xm & 0xffff)];
    }
}

inline unsigned fe_grid(int64_t items) {
    int64_t b = (items + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

template <typename T>
int stats_launch(const T *frames, float *mean, float *stdv, float *ws, int F, int64_t npix, hipStream_t st) {
    const int nchunks = (int)((npix + CHUNK - 1) / CHUNK);
    dim3 grid((nchunks + 3) / 4, F);
    hipLaunchKernelGGL((frame_chunk_sums_kernel<T, false>), grid, dim3(256), 0, st, frames, (const float *)nullptr, ws,
                       npix, nchunks);
    hipLaunchKernelGGL(frame_stats_finish_kernel, dim3((F + 63) / 64), dim3(64), 0, st, ws, mean, F, nchunks, (float)npix, 0);
    hipLaunchKernelGGL((frame_chunk_sums_kernel<T, true>), grid, dim3(256), 0, st, frames, mean, ws, npix, nchunks);
    hipLaunchKernelGGL(frame_stats_finish_kernel, dim3((F + 63) / 64), dim3(64), 0, st, ws, stdv, F, nchunks, (float)npix, 1);
    return sq_check_launch("sq_frame_stats");
}

template <typename T>
int tiles_launch(const T *frames, const float *mean, const float *stdv, const int *oy, const int *ox, float *tiles, int F,
                 int H, int W, int TR, int TC, int TS, int normalise, hipStream_t st) {
    hipLaunchKernelGGL(tiles_norm_kernel<T>, dim3(fe_grid((int64_t)F * TR * TC * TS * TS)), dim3(256), 0, st, frames, mean,
                       stdv, oy, ox, tiles, F, H, W, TR, TC, TS, normalise);
    return sq_check_launch("sq_frames_to_tiles");
}

}  // namespace

extern "C" int64_t sq_frame_stats_workspace(int F, int H, int W) {
    if (F <= 0 || H <= 0 || W <= 0) return -1;
    const int64_t npix = (int64_t)H * W;
    if (npix > (1 << 24)) return -1;                            // n must be exact in float32 (numpy divides by it)
    return (int64_t)F * ((npix + CHUNK - 1) / CHUNK) * 4;
}

extern "C" int sq_frame_stats(const void *frames, int dtype, float *mean, float *stdv, void *workspace, int F, int H,
                              int W, void *stream) {
    SQ_REQUIRE(frames && mean && stdv && workspace, "sq_frame_stats: null pointer");
    SQ_REQUIRE(sq_frame_stats_workspace(F, H, W) > 0, "sq_frame_stats: need F, H, W > 0 and H*W <= 2^24");
    SQ_REQUIRE_ALIGNED(frames);
    hipStream_t st = (hipStream_t)stream;
    const int64_t npix = (int64_t)H * W;
    float *ws = reinterpret_cast<float *>(workspace);
    switch (dtype) {
    case SQ_PIX_U8: return stats_launch(reinterpret_cast<const uint8_t *>(frames), mean, stdv, ws, F, npix, st);
    case SQ_PIX_U16: return stats_launch(reinterpret_cast<const uint16_t *>(frames), mean, stdv, ws, F, npix, st);
    case SQ_PIX_F32: return stats_launch(reinterpret_cast<const float *>(frames), mean, stdv, ws, F, npix, st);
    }
    sq_set_error("sq_frame_stats: unknown pixel type %d", dtype);
    return SQ_EINVAL;
}

extern "C" int sq_frames_to_tiles(const void *frames, int dtype, const float *mean, const float *stdv, const int32_t *oy,
                                  const int32_t *ox, float *tiles, int F, int H, int W, int TR, int TC, int TS,
                                  void *stream) {
    SQ_REQUIRE(frames && oy && ox && tiles, "sq_frames_to_tiles: null pointer");
    SQ_REQUIRE((mean == nullptr) == (stdv == nullptr), "sq_frames_to_tiles: give both mean and std, or neither");
    SQ_REQUIRE(F > 0 && TR > 0 && TC > 0 && TS > 0 && TS <= H && TS <= W, "sq_frames_to_tiles: tile %d does not fit %dx%d",
               TS, H, W);
    hipStream_t st = (hipStream_t)stream;
    const int normalise = mean != nullptr;
    switch (dtype) {
    case SQ_PIX_U8:
        return tiles_launch(reinterpret_cast<const uint8_t *>(frames), mean, stdv, oy, ox, tiles, F, H, W, TR, TC, TS,
                            normalise, st);
    case SQ_PIX_U16:
        return tiles_launch(reinterpret_cast<const uint16_t *>(frames), mean, stdv, oy, ox, tiles, F, H, W, TR, TC, TS,
                            normalise, st);
    case SQ_PIX_F32:
        return tiles_launch(reinterpret_cast<const float *>(frames), mean, stdv, oy, ox, tiles, F, H, W, TR, TC, TS,
                            normalise, st);
    }
    sq_set_error("sq_frames_to_tiles: unknown pixel type %d", dtype);
    return SQ_EINVAL;
}

extern "C" int sq_stitch_masks_u8(const uint8_t *tile_masks, const int32_t *ymap, const int32_t *xmap, uint8_t *out, int F,
                                  int H, int W, int TR, int TC, int TS, void *stream) {
    SQ_REQUIRE(tile_masks && ymap && xmap && out, "sq_stitch_masks_u8: null pointer");
    SQ_REQUIRE(F > 0 && H > 0 && W > 0 && TR > 0 && TC > 0 && TS > 0 && TS < 65536, "sq_stitch_masks_u8: bad geometry");
    hipLaunchKernelGGL(stitch_masks_kernel, dim3(fe_grid((int64_t)F * H * W)), dim3(256), 0, (hipStream_t)stream, tile_masks,
                       ymap, xmap, out, F, H, W, TR, TC, TS);
    return sq_check_launch("sq_stitch_masks_u8");
}
