// Mask -> connected components -> centroids on the GPU: the step after the segmentation hot path
// (CentroidWriter.write, sequitr/utils.py:531-578: per frame and per class c > 0,
// scipy.ndimage.label(out == c) with the default 4-connectivity, then center_of_mass of every label).
//
// All classes are labelled in one pass: two pixels are connected iff they are 4-neighbours AND carry the
// same class value > 0.  Union-find with the smaller linear index as the root, so a component's root is
// its first pixel in raster order -- scipy numbers its labels in exactly that order.
//   1. row scan   : one wave per image row; parent = first pixel of the horizontal run (ballot + clz, no
//                   atomics); accumulators zeroed at run starts (a root is always one)
//   2. merge      : one union per place where a run starts to overlap a run of the row above
//   3. compress   : parent = root
//   4. accumulate : row scan again; one atomic triple (count, sum row, sum col) per run segment, integer
//                   arithmetic => exact and order-independent
//   5. emit       : every root writes [frame, x = row centre, y = col centre, 0, class] and its sort key
// HBM-bound integer/byte work: ~1 B/pixel of mask read three times + 4 B/pixel of parent written and read
// a few times; the accumulators are touched only at roots.
#include "sq_common.h"

namespace {

typedef unsigned long long u64;

__device__ __forceinline__ int cc_find(const int *parent, int a) {
    int p = parent[a];
    while (p != a) {
        a = p;
        p = parent[a];
    }
    return a;
}

__device__ __forceinline__ void cc_unite(int *parent, int a, int b) {
    for (;;) {
        a = cc_find(parent, a);
        b = cc_find(parent, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }          // a = larger root, hangs under b
        const int old = atomicMin(&parent[a], b);
        if (old == a) return;
        a = old;                                               // someone re-rooted a meanwhile: retry from there
    }
}

// Row scan shared by kernels 1 and 4.  For the 64-pixel segment starting at column c0 of one row:
// v = class of this lane's pixel (0 outside the row), `same` = continues the run of the pixel to its left,
// j = lane index where this lane's run starts inside the segment, or -1 when it started in an earlier
// segment (then `carry` = that run's start column).
struct SegScan {
    int v;
    bool same;
    int j;
};

__device__ __forceinline__ SegScan seg_scan(const uint8_t *__restrict__ row, int W, int c0, int lane, int prev_last) {
    SegScan s;
    const int col = c0 + lane;
    s.v = col < W ? (int)row[col] : 0;
    int left = __shfl_up(s.v, 1);
    if (lane == 0) left = prev_last;
    s.same = s.v != 0 && s.v == left;
    const u64 B = __ballot(s.same);
    const u64 upto = lane == 63 ? ~0ULL : ((2ULL << lane) - 1ULL);
    const u64 m = ~B & upto;                                   // lanes <= mine that START something
    s.j = m ? 63 - __clzll((long long)m) : -1;
    return s;
}

__global__ __launch_bounds__(256) void cc_rowscan_kernel(const uint8_t *__restrict__ mask, int *__restrict__ parent,
                                                         unsigned *__restrict__ cnt, u64 *__restrict__ sums,
                                                         int rows, int W) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const uint8_t *row = mask + (size_t)r * W;
    const int base = r * W;
    int prev_last = 0, carry = 0;
    for (int c0 = 0; c0 < W; c0 += 64) {
        const SegScan s = seg_scan(row, W, c0, lane, prev_last);
        const int start = s.j >= 0 ? c0 + s.j : carry;
        const int col = c0 + lane;
        if (col < W) {
            const int g = base + col;
            parent[g] = s.v ? base + start : -1;
            if (s.v && start == col) {                          // a root is always the first pixel of a run
                cnt[g] = 0u;
                sums[3 * (size_t)g] = 0ULL;
                sums[3 * (size_t)g + 1] = 0ULL;
                sums[3 * (size_t)g + 2] = 0ULL;
            }
        }
        prev_last = __shfl(s.v, 63);
        carry = __shfl(start, 63);
    }
}

// planes: 1 for images; for volumes every frame is `planes` consecutive (H, W) planes and voxels are also
// linked to the same-class voxel of the previous plane (6-connectivity, scipy's default 3-D structure)
__global__ __launch_bounds__(256) void cc_merge_kernel(const uint8_t *__restrict__ mask, int *__restrict__ parent,
                                                       int64_t total, int planes, int H, int W) {
    const int64_t HW = (int64_t)H * W;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
        const int col = (int)(g % W), rowi = (int)((g / W) % H), plane = (int)((g / HW) % planes);
        const int v = mask[g];
        if (v == 0) continue;
        const bool left_same = col > 0 && mask[g - 1] == v;
        // the pixel to the left makes the same link when it is in my run and also touches the same neighbour run
        if (rowi > 0 && mask[g - W] == v && !(left_same && mask[g - W - 1] == v)) cc_unite(parent, (int)g, (int)(g - W));
        if (plane > 0 && mask[g - HW] == v && !(left_same && mask[g - HW - 1] == v)) cc_unite(parent, (int)g, (int)(g - HW));
    }
}

__global__ __launch_bounds__(256) void cc_compress_kernel(int *__restrict__ parent, int64_t total) {
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
        const int p = parent[g];
        if (p >= 0) parent[g] = cc_find(parent, p);
    }
}

__global__ __launch_bounds__(256) void cc_accumulate_kernel(const uint8_t *__restrict__ mask,
                                                            const int *__restrict__ parent, unsigned *__restrict__ cnt,
                                                            u64 *__restrict__ sums, int rows, int planes, int H,
                                                            int W) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const uint8_t *row = mask + (size_t)r * W;
    const int base = r * W, rowi = r % H, plane = (r / H) % planes;
    int prev_last = 0;
    for (int c0 = 0; c0 < W; c0 += 64) {
        const SegScan s = seg_scan(row, W, c0, lane, prev_last);
        const bool next_same = __shfl_down((int)s.same, 1) != 0 && lane != 63;
        if (s.v != 0 && !next_same) {                           // last lane of a run segment
            const int j = s.j >= 0 ? s.j : 0;                    // the part of the run inside this segment
            const unsigned len = (unsigned)(lane - j + 1);
            const int root = parent[base + c0 + lane];
            atomicAdd(&cnt[root], len);
            atomicAdd(&sums[3 * (size_t)root], (u64)len * (u64)rowi);
            atomicAdd(&sums[3 * (size_t)root + 1], (u64)len * (u64)(2 * c0 + lane + j) / 2ULL);
            if (planes > 1) atomicAdd(&sums[3 * (size_t)root + 2], (u64)len * (u64)plane);
        }
        prev_last = __shfl(s.v, 63);
    }
}

__global__ __launch_bounds__(256) void cc_emit_kernel(const uint8_t *__restrict__ mask, const int *__restrict__ parent,
                                                      const unsigned *__restrict__ cnt, const u64 *__restrict__ sums,
                                                      int64_t total, int planes, int H, int W,
                                                      int *__restrict__ count, float *__restrict__ out,
                                                      int *__restrict__ keys, int max_out) {
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
        if (parent[g] != (int)g) continue;
        const int idx = atomicAdd(count, 1);
        if (idx >= max_out) continue;
        const double c = (double)mask[g];
        // scipy.ndimage.center_of_mass(out, labels, index): sum(out * grid) / sum(out) in float64, out == c
        const double norm = c * (double)cnt[g];
        const float crow = (float)(c * (double)sums[3 * (size_t)g] / norm);
        const float ccol = (float)(c * (double)sums[3 * (size_t)g + 1] / norm);
        out[5 * (size_t)idx + 0] = (float)(g / ((int64_t)planes * H * W));
        if (planes > 1) {                                       // (x, y, z) = the three axes of the volume in order
            out[5 * (size_t)idx + 1] = (float)(c * (double)sums[3 * (size_t)g + 2] / norm);
            out[5 * (size_t)idx + 2] = crow;
            out[5 * (size_t)idx + 3] = ccol;
        } else {
            out[5 * (size_t)idx + 1] = crow;
            out[5 * (size_t)idx + 2] = ccol;
            out[5 * (size_t)idx + 3] = 0.0f;
        }
        out[5 * (size_t)idx + 4] = (float)c;
        keys[idx] = (int)g;
    }
}

inline unsigned cc_grid(int64_t items) {
    int64_t b = (items + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

static int centroids_launch(const uint8_t *mask, int N, int planes, int H, int W, void *workspace, int32_t *count, float *out,
                            int32_t *keys, int max_out, void *stream, const char *who) {
    SQ_REQUIRE(mask && workspace && count && out && keys && max_out > 0, "%s: null pointer", who);
    SQ_REQUIRE(N > 0 && planes > 0 && H > 0 && W > 0 && (int64_t)N * planes * H * W < ((int64_t)1 << 31),
               "%s: the mask must have fewer than 2^31 elements", who);
    SQ_REQUIRE((((uintptr_t)workspace) & 15u) == 0, "%s: workspace must be 16-byte aligned", who);
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = (int64_t)N * planes * H * W;
    unsigned long long *sums = reinterpret_cast<unsigned long long *>(workspace);     // 8-byte items first
    int *parent = reinterpret_cast<int *>(sums + 3 * total);
    unsigned *cnt = reinterpret_cast<unsigned *>(parent + total);
    const int rows = N * planes * H;
    if (hipMemsetAsync(count, 0, sizeof(int32_t), st) != hipSuccess) {
        sq_set_error("%s: cannot clear the counter", who);
        return SQ_ELAUNCH;
    }
    hipLaunchKernelGGL(cc_rowscan_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, mask, parent, cnt, sums, rows, W);
    hipLaunchKernelGGL(cc_merge_kernel, dim3(cc_grid(total)), dim3(256), 0, st, mask, parent, total, planes, H, W);
    hipLaunchKernelGGL(cc_compress_kernel, dim3(cc_grid(total)), dim3(256), 0, st, parent, total);
    hipLaunchKernelGGL(cc_accumulate_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, mask, parent, cnt, sums, rows, planes, H,
                       W);
    hipLaunchKernelGGL(cc_emit_kernel, dim3(cc_grid(total)), dim3(256), 0, st, mask, parent, cnt, sums, total, planes, H, W,
                       count, out, keys, max_out);
    return sq_check_launch(who);
}

extern "C" int64_t sq_mask_centroids_workspace(int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0 || (int64_t)N * H * W >= ((int64_t)1 << 31)) return -1;
    return (int64_t)N * H * W * (4 + 4 + 24);                  // parent, count, three coordinate sums
}

extern "C" int sq_mask_centroids_u8(const uint8_t *mask, int N, int H, int W, void *workspace, int32_t *count,
                                    float *out, int32_t *keys, int max_out, void *stream) {
    return centroids_launch(mask, N, 1, H, W, workspace, count, out, keys, max_out, stream, "sq_mask_centroids_u8");
}

// volumes: mask (N, D0, D1, D2) uint8, 6-connectivity; rows [frame, x, y, z, class] with (x, y, z) the centre along
// (D0, D1, D2); workspace = sq_mask_centroids_workspace(N * D0, D1, D2)
extern "C" int sq_volume_centroids_u8(const uint8_t *mask, int N, int D0, int D1, int D2, void *workspace, int32_t *count,
                                      float *out, int32_t *keys, int max_out, void *stream) {
    return centroids_launch(mask, N, D0, D1, D2, workspace, count, out, keys, max_out, stream, "sq_volume_centroids_u8");
}
