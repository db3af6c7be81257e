// Shared host/device helpers for libsequitr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/sequitr_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SQ_NOT_MINE 1   /* internal: a specialised launcher declines a shape, the caller takes the generic kernel */

void sq_set_error(const char *fmt, ...);

#define SQ_REQUIRE(cond, ...)            \
    do {                                 \
        if (!(cond)) {                   \
            sq_set_error(__VA_ARGS__);   \
            return SQ_EINVAL;            \
        }                                \
    } while (0)

#define SQ_ALIGNED16(p) ((((uintptr_t)(p)) & 15u) == 0)

#define SQ_REQUIRE_ALIGNED(p)                                     \
    do {                                                          \
        if (!SQ_ALIGNED16(p)) {                                   \
            sq_set_error("%s: pointer %s not 16-byte aligned", __func__, #p); \
            return SQ_EALIGN;                                     \
        }                                                         \
    } while (0)

static inline int sq_check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        sq_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return SQ_ELAUNCH;
    }
    return SQ_OK;
}

__device__ __forceinline__ float sq_act(float v, int act) {
    if (act == SQ_ACT_RELU) return v > 0.0f ? v : 0.0f;
    if (act == SQ_ACT_LEAKY) return v > 0.0f ? v : 0.2f * v;
    return v;
}

// XCD-aware, bijective block remap (8 XCDs, blocks dealt round-robin): gives each
// XCD a contiguous run of tiles so neighbouring tiles' halos hit the same L2.
// Speed only; any placement is correct.
__device__ __forceinline__ unsigned sq_xcd_remap(unsigned bid, unsigned nblk) {
    const unsigned q = nblk >> 3, r = nblk & 7u, xcd = bid & 7u, k = bid >> 3;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

// Counter-based dropout mask (tf.nn.dropout's semantics, sequitr/networks/unet.py:274-276; the random stream is the
// build's own): one 32-bit hash per QUAD of consecutive elements 4q .. 4q+3 yields two words = four 16-bit uniforms,
// element kept iff its uniform >= thr16 = rate * 2^16.  v_mul_lo_u32 is quarter rate on CDNA, and the fused
// conv + dropout epilogue is VALU-bound on it at the shallow levels: three multiplies per four elements here instead of
// four per element.  The key carries (seed, step) into the mix at two places, so that two layers' (or steps') masks are
// not shifted copies of each other.  Every dropout kernel of the library uses this one definition.
struct SqDropKey { unsigned s1, s2; };
__device__ __forceinline__ SqDropKey sq_dropout_key(unsigned seed, const int *__restrict__ step) {
    if (step) seed += (unsigned)step[0] * 0x9E3779B9u;          // a fresh mask on every replayed step
    SqDropKey k;
    k.s1 = seed * 0x9E3779B1u + 0x7F4A7C15u;
    k.s2 = (seed ^ 0x68E31DA4u) * 0x85EBCA6Bu;
    k.s2 ^= k.s2 >> 13;
    return k;
}
__device__ __forceinline__ unsigned sq_dropout_thr16(float rate) { return (unsigned)(rate * 65536.0f); }
// bit j of the result: element 4q + j is kept
__device__ __forceinline__ unsigned sq_dropout_keep4(SqDropKey k, unsigned q, unsigned thr16) {
    unsigned h = q + k.s1;
    h ^= h >> 16; h *= 0x7FEB352Du; h = h ^ (h >> 15) ^ k.s2; h *= 0x846CA68Bu; h ^= h >> 16;
    unsigned g = (h ^ 0x5BD1E995u) * 0x2C1B3C6Du;
    g ^= g >> 15;
    return (unsigned)((h & 0xFFFFu) >= thr16) | ((unsigned)((h >> 16) >= thr16) << 1) |
           ((unsigned)((g & 0xFFFFu) >= thr16) << 2) | ((unsigned)((g >> 16) >= thr16) << 3);
}

// Weighted softmax cross-entropy of ONE pixel: returns w * (lse(z) * sum(y) - <y, z>) and, when dz is
// given, dz[c] = g * (softmax(z)[c] * sum(y) - y[c]).  One definition for sq_wsoftmax_ce_fwd_bwd_f32 and for the head +
// loss kernels (sq_conv1x1_head_wce_*_bf16), contraction off, so that both evaluate the same roundings.
template <int MAXC>
__device__ __forceinline__ float sq_wce_pixel(const float (&zc)[MAXC], const float (&yc)[MAXC], int C, float wp, float g,
                                              float *__restrict__ dz) {
#pragma clang fp contract(off)
    float m = -INFINITY, yt = 0.f, dot = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
        if (c < C) {
            m = zc[c] > m ? zc[c] : m;
            yt += yc[c];
            dot = __builtin_fmaf(yc[c], zc[c], dot);
        }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
        if (c < C) s += expf(zc[c] - m);
    const float lse = m + logf(s);
    if (dz) {
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C) dz[c] = g * (expf(zc[c] - lse) * yt - yc[c]);
    }
    return wp * (lse * yt - dot);
}

// Fixed-order reduction of `nblk` block partials per output by a group of G lanes (G a power of two
// <= 64, the same for a given problem size, so results are run-to-run reproducible): lane g sums
// partials g, g+G, ... in order, then an xor butterfly folds the group.
__device__ __forceinline__ float sq_group_reduce(const float *__restrict__ p, size_t stride, int nblk, int g, int G) {
    float s = 0.f;
    int b = g;
    // eight loads in flight, added in the same order as one at a time (a lane's 32 partials were 32 serial round trips)
    for (; b + 7 * G < nblk; b += 8 * G) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(b + u * G) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; b < nblk; b += G) s += p[(size_t)b * stride];
    for (int m = G >> 1; m > 0; m >>= 1) s += __shfl_xor(s, m);
    return s;
}
static inline int sq_group_size(int nblk) {
    int G = 1;
    while (G < 64 && G * 2 <= nblk) G *= 2;
    return G;
}
