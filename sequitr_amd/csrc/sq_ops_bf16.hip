// bf16-activation variants of the HBM-bound ops and of the transpose convolution / head, gfx950.
// 16 B per lane = 8 bf16; arithmetic in f32 registers, one rounding (RNE) when a value is stored.
#include "sq_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

namespace {

inline unsigned grid_for(int64_t items) {
    int64_t b = (items + 255) / 256;
    if (b > 2048) b = 2048;
    return (unsigned)(b < 1 ? 1 : b);
}
#define SQ_GRID_STRIDE(i, n) \
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (n); i += (int64_t)gridDim.x * 256)

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float4 *__restrict__ x, bf16x4 *__restrict__ y, int64_t n4) {
    SQ_GRID_STRIDE(i, n4) {
        const float4 v = x[i];
        y[i] = (bf16x4){(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    }
}
__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const bf16x4 *__restrict__ x, float4 *__restrict__ y, int64_t n4) {
    SQ_GRID_STRIDE(i, n4) {
        const bf16x4 v = x[i];
        y[i] = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
    }
}

__global__ __launch_bounds__(256) void maxpool_bf16_kernel(const bf16x8 *__restrict__ x, bf16x8 *__restrict__ y, int N,
                                                            int H, int W, int C8) {
    const int Ho = H >> 1, Wo = W >> 1;
    const int64_t total = (int64_t)N * Ho * Wo * C8;
    SQ_GRID_STRIDE(i, total) {
        const int c = (int)(i % C8);
        int64_t t = i / C8;
        const int xo = (int)(t % Wo);
        t /= Wo;
        const int yo = (int)(t % Ho);
        const int n = (int)(t / Ho);
        const int64_t b = (((int64_t)n * H + 2 * yo) * W + 2 * xo) * C8 + c;
        const bf16x8 a = x[b], bb = x[b + C8], d = x[b + (int64_t)W * C8], e = x[b + (int64_t)W * C8 + C8];
        bf16x8 r;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float m = (float)a[j];
            const float f1 = (float)bb[j], f2 = (float)d[j], f3 = (float)e[j];
            m = f1 > m ? f1 : m; m = f2 > m ? f2 : m; m = f3 > m ? f3 : m;
            r[j] = (__bf16)m;
        }
        y[i] = r;
    }
}

__global__ __launch_bounds__(256) void maxpool_bwd_bf16_kernel(const bf16x8 *__restrict__ x, const bf16x8 *__restrict__ dy,
                                                                bf16x8 *__restrict__ dx, int N, int H, int W, int C8) {
    const int Ho = H >> 1, Wo = W >> 1;
    const int64_t total = (int64_t)N * Ho * Wo * C8;
    SQ_GRID_STRIDE(i, total) {
        const int c = (int)(i % C8);
        int64_t t = i / C8;
        const int xo = (int)(t % Wo);
        t /= Wo;
        const int yo = (int)(t % Ho);
        const int n = (int)(t / Ho);
        const int64_t b00 = (((int64_t)n * H + 2 * yo) * W + 2 * xo) * C8 + c;
        const int64_t b01 = b00 + C8, b10 = b00 + (int64_t)W * C8, b11 = b10 + C8;
        const bf16x8 a = x[b00], b = x[b01], d = x[b10], e = x[b11], g = dy[i];
        bf16x8 ra, rb, rd, re;
        const __bf16 z = (__bf16)0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float m = (float)a[j];
            int k = 0;
            if ((float)b[j] > m) { m = (float)b[j]; k = 1; }
            if ((float)d[j] > m) { m = (float)d[j]; k = 2; }
            if ((float)e[j] > m) { m = (float)e[j]; k = 3; }
            ra[j] = k == 0 ? g[j] : z; rb[j] = k == 1 ? g[j] : z; rd[j] = k == 2 ? g[j] : z; re[j] = k == 3 ? g[j] : z;
        }
        dx[b00] = ra; dx[b01] = rb; dx[b10] = rd; dx[b11] = re;
    }
}

__global__ __launch_bounds__(256) void act_bwd_bf16_kernel(const bf16x8 *__restrict__ dy, const bf16x8 *__restrict__ y,
                                                            bf16x8 *__restrict__ dx, int64_t n8, int act) {
    const float slope = act == SQ_ACT_LEAKY ? 0.2f : (act == SQ_ACT_RELU ? 0.0f : 1.0f);
    SQ_GRID_STRIDE(i, n8) {
        const bf16x8 g = dy[i], v = y[i];
        bf16x8 r;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = (float)v[j] > 0.f ? g[j] : (__bf16)((float)g[j] * slope);
        dx[i] = r;
    }
}

__global__ __launch_bounds__(256) void bridge_bf16_kernel(const bf16x8 *__restrict__ a, const bf16x8 *__restrict__ b,
                                                           bf16x8 *__restrict__ y, int64_t n8, int op) {
    SQ_GRID_STRIDE(i, n8) {
        const bf16x8 u = a[i], v = b[i];
        bf16x8 r;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float p = (float)u[j], q = (float)v[j];
            r[j] = (__bf16)(op == SQ_BRIDGE_ADD ? p + q : (op == SQ_BRIDGE_MUL ? p * q : (op == SQ_BRIDGE_SUB ? p - q : p)));
        }
        y[i] = r;
    }
}

__global__ __launch_bounds__(256) void bridge_bwd_bf16_kernel(const bf16x8 *__restrict__ dy, const bf16x8 *__restrict__ a,
                                                               const bf16x8 *__restrict__ b, bf16x8 *__restrict__ da,
                                                               bf16x8 *__restrict__ db, int64_t n8, int op) {
    SQ_GRID_STRIDE(i, n8) {
        const bf16x8 g = dy[i];
        if (op == SQ_BRIDGE_MUL) {
            const bf16x8 u = a[i], v = b[i];
            bf16x8 ra, rb;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                ra[j] = (__bf16)((float)g[j] * (float)v[j]);
                rb[j] = (__bf16)((float)g[j] * (float)u[j]);
            }
            da[i] = ra; db[i] = rb;
        } else {
            da[i] = g;
            if (op == SQ_BRIDGE_SUB) {
                bf16x8 r;
#pragma unroll
                for (int j = 0; j < 8; ++j) r[j] = (__bf16)(-(float)g[j]);
                db[i] = r;
            } else {
                db[i] = g;
            }
        }
    }
}

// Decoder junction backward (merged = bridge(up, skip), up = transpose conv output): one pass writes
//   g    (N,H,W,4C): d_up already in the space-to-depth layout the transpose-conv backward consumes
//                    (g[n,i,j,(2a+b)C + c] = d_up[n,2i+a,2j+b,c]), and
//   dskip (N,2H,2W,C): the gradient of the skip operand.
// Same arithmetic and rounding as bridge_bwd_bf16_kernel followed by space_to_depth2.
__global__ __launch_bounds__(256) void bridge_bwd_s2d_bf16_kernel(const bf16x8 *__restrict__ dy, const bf16x8 *__restrict__ up,
                                                                   const bf16x8 *__restrict__ skip, bf16x8 *__restrict__ g,
                                                                   bf16x8 *__restrict__ dskip, int N, int H, int W, int C8,
                                                                   int op) {
    const int64_t total = (int64_t)N * 2 * H * 2 * W * C8;      // H, W = the LOW-resolution side
    SQ_GRID_STRIDE(i, total) {
        const int c = (int)(i % C8);
        int64_t t = i / C8;
        const int x = (int)(t % (2 * W));
        t /= 2 * W;
        const int y = (int)(t % (2 * H));
        const int n = (int)(t / (2 * H));
        const bf16x8 gy = dy[i];
        bf16x8 da, db;
        if (op == SQ_BRIDGE_MUL) {
            const bf16x8 u = up[i], v = skip[i];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                da[j] = (__bf16)((float)gy[j] * (float)v[j]);
                db[j] = (__bf16)((float)gy[j] * (float)u[j]);
            }
        } else {
            da = gy;
            db = gy;
            if (op == SQ_BRIDGE_SUB) {
#pragma unroll
                for (int j = 0; j < 8; ++j) db[j] = (__bf16)(-(float)gy[j]);
            }
        }
        dskip[i] = db;
        const int ab = (y & 1) * 2 + (x & 1);
        g[((((int64_t)n * H + (y >> 1)) * W + (x >> 1)) * 4 + ab) * C8 + c] = da;
    }
}

// max-pool backward that also adds a second gradient of the pooled tensor's INPUT (the skip path of the
// U-Net): dx = scatter(dy) + add, the bf16 sum torch's gradient accumulation would form in its own kernel
// gscale > 0: x is the dropout(ReLU(.)) output of the conv block in front of the pool, and the gradient is handed on
// already through that gate, dx = (x > 0) ? bf16(bf16(scatter + add) * gscale) : 0 -- the two roundings of the
// stand-alone sq_relu_scale_bwd_bf16 pass it replaces (same bits), with x already in registers for the arg-max.
__global__ __launch_bounds__(256) void maxpool_bwd_add_bf16_kernel(const bf16x8 *__restrict__ x, const bf16x8 *__restrict__ dy,
                                                                    const bf16x8 *__restrict__ add, bf16x8 *__restrict__ dx,
                                                                    int N, int H, int W, int C8, float gscale) {
    const int Ho = H >> 1, Wo = W >> 1;
    const int64_t total = (int64_t)N * Ho * Wo * C8;
    SQ_GRID_STRIDE(i, total) {
        const int c = (int)(i % C8);
        int64_t t = i / C8;
        const int xo = (int)(t % Wo);
        t /= Wo;
        const int yo = (int)(t % Ho);
        const int n = (int)(t / Ho);
        const int64_t b00 = (((int64_t)n * H + 2 * yo) * W + 2 * xo) * C8 + c;
        const int64_t b01 = b00 + C8, b10 = b00 + (int64_t)W * C8, b11 = b10 + C8;
        const bf16x8 a = x[b00], b = x[b01], d = x[b10], e = x[b11], gy = dy[i];
        const bf16x8 sa = add[b00], sb = add[b01], sd = add[b10], se = add[b11];
        bf16x8 ra, rb, rd, re;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float m = (float)a[j];
            int k = 0;
            if ((float)b[j] > m) { m = (float)b[j]; k = 1; }
            if ((float)d[j] > m) { m = (float)d[j]; k = 2; }
            if ((float)e[j] > m) { m = (float)e[j]; k = 3; }
            const float gg = (float)gy[j];
            ra[j] = (__bf16)((k == 0 ? gg : 0.f) + (float)sa[j]);
            rb[j] = (__bf16)((k == 1 ? gg : 0.f) + (float)sb[j]);
            rd[j] = (__bf16)((k == 2 ? gg : 0.f) + (float)sd[j]);
            re[j] = (__bf16)((k == 3 ? gg : 0.f) + (float)se[j]);
            if (gscale > 0.f) {
                ra[j] = (float)a[j] > 0.f ? (__bf16)((float)ra[j] * gscale) : (__bf16)0.f;
                rb[j] = (float)b[j] > 0.f ? (__bf16)((float)rb[j] * gscale) : (__bf16)0.f;
                rd[j] = (float)d[j] > 0.f ? (__bf16)((float)rd[j] * gscale) : (__bf16)0.f;
                re[j] = (float)e[j] > 0.f ? (__bf16)((float)re[j] * gscale) : (__bf16)0.f;
            }
        }
        dx[b00] = ra; dx[b01] = rb; dx[b10] = rd; dx[b11] = re;
    }
}

// 8 elements per thread (16 B of data, 8 B of mask); the mask of sq_dropout_keep4 (sq_common.h), as the f32 kernel
__global__ __launch_bounds__(256) void dropout_fwd_bf16_kernel(const bf16x8 *__restrict__ x, bf16x8 *__restrict__ y,
                                                                uint2 *__restrict__ mask, int64_t n8, float rate,
                                                                unsigned seed, int mask_given, const int *__restrict__ step) {
    const SqDropKey key = sq_dropout_key(seed, step);
    const unsigned thr = sq_dropout_thr16(rate);
    const float inv = 1.0f / (1.0f - rate);
    SQ_GRID_STRIDE(i, n8) {
        uint2 m;
        if (mask_given) m = mask[i];
        else {
            const unsigned a = sq_dropout_keep4(key, (unsigned)(i * 2), thr), b = sq_dropout_keep4(key, (unsigned)(i * 2 + 1), thr);
            m.x = (a & 1u) | ((a & 2u) << 7) | ((a & 4u) << 14) | ((a & 8u) << 21);
            m.y = (b & 1u) | ((b & 2u) << 7) | ((b & 4u) << 14) | ((b & 8u) << 21);
            mask[i] = m;
        }
        const bf16x8 v = x[i];
        bf16x8 r;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned keep = ((j < 4 ? m.x : m.y) >> (8 * (j & 3))) & 0xFFu;
            r[j] = keep ? (__bf16)((float)v[j] * inv) : (__bf16)0.f;
        }
        y[i] = r;
    }
}
__global__ __launch_bounds__(256) void dropout_bwd_bf16_kernel(const bf16x8 *__restrict__ dy, const uint2 *__restrict__ mask,
                                                                bf16x8 *__restrict__ dx, int64_t n8, float rate) {
    const float inv = 1.0f / (1.0f - rate);
    SQ_GRID_STRIDE(i, n8) {
        const uint2 m = mask[i];
        const bf16x8 v = dy[i];
        bf16x8 r;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned keep = ((j < 4 ? m.x : m.y) >> (8 * (j & 3))) & 0xFFu;
            r[j] = keep ? (__bf16)((float)v[j] * inv) : (__bf16)0.f;
        }
        dx[i] = r;
    }
}

// backward of y = dropout(relu(z)) given only y: y > 0 <=> (kept and z > 0), so dx = y > 0 ? dy * scale : 0
__global__ __launch_bounds__(256) void relu_scale_bwd_bf16_kernel(const bf16x8 *__restrict__ dy, const bf16x8 *__restrict__ y,
                                                                   bf16x8 *__restrict__ dx, int64_t n8, float scale) {
    SQ_GRID_STRIDE(i, n8) {
        const bf16x8 g = dy[i], v = y[i];
        bf16x8 r;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = (float)v[j] > 0.f ? (__bf16)((float)g[j] * scale) : (__bf16)0.f;
        dx[i] = r;
    }
}

// dropout backward and the activation backward of the layer in front of it in ONE pass:
// dx = act'(y) * (keep ? dy / (1 - rate) : 0), y = the activation's output (the dropout's input)
__global__ __launch_bounds__(256) void act_dropout_bwd_bf16_kernel(const bf16x8 *__restrict__ dy,
                                                                    const uint2 *__restrict__ mask,
                                                                    const bf16x8 *__restrict__ y, bf16x8 *__restrict__ dx,
                                                                    int64_t n8, float rate, int act) {
    const float inv = 1.0f / (1.0f - rate);
    const float slope = act == SQ_ACT_LEAKY ? 0.2f : (act == SQ_ACT_RELU ? 0.0f : 1.0f);
    SQ_GRID_STRIDE(i, n8) {
        const uint2 m = mask[i];
        const bf16x8 v = dy[i], yy = y[i];
        bf16x8 r;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned keep = ((j < 4 ? m.x : m.y) >> (8 * (j & 3))) & 0xFFu;
            const __bf16 d = keep ? (__bf16)((float)v[j] * inv) : (__bf16)0.f;      // rounded as dropout_bwd does
            r[j] = (float)yy[j] > 0.f ? d : (__bf16)((float)d * slope);
        }
        dx[i] = r;
    }
}

// ---- 2x2/s2 transpose conv + bias + bridge on v_mfma_f32_16x16x32_bf16 ------------------------------------
// D[rho][p] = sum_c Wt[rho][c] X[p][c], rho = (2a+b)*Cout + o; Wt = the (2,2,Cout,Cin) kernel read flat (k = c is
// contiguous for both operands: 16-byte fragment reads).  Block = 64 rows x 64 input pixels, 32-channel chunks.
constexpr int CT_ROWB = 96;                                   // 64 B of data + pad: conflict-free b128 reads
constexpr int CT_STGB = 144;                                  // output staging: 64 rows x (128 B + pad)
__global__ __launch_bounds__(256) void convT_bf16_kernel(const __bf16 *__restrict__ x, const __bf16 *__restrict__ w,
                                                          const float *__restrict__ bias, const __bf16 *__restrict__ skip,
                                                          __bf16 *__restrict__ y, int64_t P, int H, int W, int Cin, int Cout,
                                                          int bridge, __bf16 *__restrict__ up_out) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 64 * CT_ROWB];
    unsigned char *as = lds, *xs = lds + 64 * CT_ROWB;
    static_assert(64 * CT_STGB <= 2 * 64 * CT_ROWB, "the output staging tile reuses the operand tiles");
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, kg = lane >> 4;
    const int64_t p0 = (int64_t)blockIdx.x * 64;
    const int r0 = blockIdx.y * 64;
    f32x4 acc[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) acc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int srow = tid >> 2, sq = tid & 3;                  // 64 rows x 4 sixteen-byte pieces
    for (int cc = 0; cc < Cin; cc += 32) {
        *reinterpret_cast<uint4 *>(as + srow * CT_ROWB + sq * 16) =
            *reinterpret_cast<const uint4 *>(w + (size_t)(r0 + srow) * Cin + cc + sq * 8);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (p0 + srow < P) v = *reinterpret_cast<const uint4 *>(x + (size_t)(p0 + srow) * Cin + cc + sq * 8);
        *reinterpret_cast<uint4 *>(xs + srow * CT_ROWB + sq * 16) = v;
        __syncthreads();
        const bf16x8 a = *reinterpret_cast<const bf16x8 *>(as + (16 * wv + li) * CT_ROWB + kg * 16);
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const bf16x8 b = *reinterpret_cast<const bf16x8 *>(xs + (cb * 16 + li) * CT_ROWB + kg * 16);
            acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[cb], 0, 0, 0);
        }
        __syncthreads();
    }
    // epilogue through LDS: the accumulator layout gives a lane 4 channels (8 B) of one sub-pixel, 16 lanes 16 output
    // pixels a stride apart -- scattered 8-byte accesses for the skip read and both stores.  Staged as [pixel][rho]
    // (up-scaled value, already rounded to bf16) and read back in OUTPUT order, the block's results are runs of whole
    // output-row segments: 16 bytes per lane, consecutive lanes consecutive addresses.
    const int rho = 16 * wv + 4 * kg;                          // block-local row
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) bv = *reinterpret_cast<const float4 *>(bias + (r0 + rho) % Cout);
    unsigned char *stg = lds;                                  // the operand tiles are dead (loop ended on a barrier)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
        *reinterpret_cast<bf16x4 *>(stg + (cb * 16 + li) * CT_STGB + rho * 2) =
            (bf16x4){(__bf16)(acc[cb][0] + bv.x), (__bf16)(acc[cb][1] + bv.y), (__bf16)(acc[cb][2] + bv.z),
                     (__bf16)(acc[cb][3] + bv.w)};
    __syncthreads();
    // rows (2a+b)*Cout + o: for one a, (b, o) is contiguous in the output over min(64, 2*Cout) rows of this block
    const int RB = 2 * Cout < 64 ? 2 * Cout : 64, R8 = RB >> 3;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int idx = tid + it * 256;                         // 512 sixteen-byte pieces
        const int ai = idx / (64 * R8), rem = idx % (64 * R8);
        const int pl = rem / R8, rl = ai * RB + (rem % R8) * 8;
        const int64_t pp = p0 + pl;
        if (pp >= P) continue;
        const int rg = r0 + rl, ab = rg / Cout, o = rg % Cout;
        const int jx = (int)(pp % W);
        const int64_t t = pp / W;
        const int iy = (int)(t % H);
        const int64_t n = t / H;
        const size_t off = ((size_t)(n * 2 * H + 2 * iy + (ab >> 1)) * (2 * W) + 2 * jx + (ab & 1)) * Cout + o;
        bf16x8 u = *reinterpret_cast<const bf16x8 *>(stg + pl * CT_STGB + rl * 2);
        if (up_out) *reinterpret_cast<bf16x8 *>(up_out + off) = u;    // training keeps the up-scaled tensor for the bridge backward
        if (bridge != SQ_BRIDGE_NONE) {
            const bf16x8 k = *reinterpret_cast<const bf16x8 *>(skip + off);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                // the up-scaled value was rounded to bf16 first: bit-identical to the unfused convT -> bridge pair
                const float uu = (float)u[e], sk = (float)k[e];
                u[e] = (__bf16)(bridge == SQ_BRIDGE_ADD ? uu + sk : (bridge == SQ_BRIDGE_MUL ? uu * sk : uu - sk));
            }
        }
        *reinterpret_cast<bf16x8 *>(y + off) = u;
    }
}

// to_image head on a bf16 activation: f32 logits (+ uint8 argmax mask), f32 (Cin, COUT) weights
template <int COUT>
__global__ __launch_bounds__(256) void head_fwd_bf16_kernel(const __bf16 *__restrict__ x, const float *__restrict__ w,
                                                             const float *__restrict__ bias, float *__restrict__ logits,
                                                             uint8_t *__restrict__ mask, int64_t npix, int Cin) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= npix) return;
    float acc[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) acc[o] = 0.f;
    for (int c8 = 0; c8 < Cin / 8; ++c8) {
        const bf16x8 v = *reinterpret_cast<const bf16x8 *>(x + p * Cin + c8 * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int o = 0; o < COUT; ++o) acc[o] = __builtin_fmaf(w[(c8 * 8 + j) * COUT + o], (float)v[j], acc[o]);
    }
    int best = 0;
    float bestv = 0.f;
#pragma unroll
    for (int o = 0; o < COUT; ++o) {
        const float v = acc[o] + (bias ? bias[o] : 0.f);
        logits[p * COUT + o] = v;
        if (o == 0 || v > bestv) { bestv = v; best = o; }
    }
    if (mask) mask[p] = (uint8_t)best;
}

__device__ __forceinline__ float wave_sum(float v) {
    v += __shfl_xor(v, 32); v += __shfl_xor(v, 16); v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4);  v += __shfl_xor(v, 2);  v += __shfl_xor(v, 1);
    return v;
}

// head + weighted softmax-CE, forward: the logits (head_fwd_bf16_kernel's fmaf chain) live in registers only; the loss
// partials are those of wsoftmax_ce_f32_kernel on the same logits (same grid, same per-thread order, fp64)
struct SqHeadCE {
    const float *bias;
    const uint8_t *yoh;
    const float *wgt;
    const float *dloss;         // device scalar: the gradient arriving at the loss (backward only)
    float inv_npix;
    double *lparts;             // backward only, may be NULL: the block's loss partial, as head_wce_fwd_bf16_kernel writes it
};

template <int COUT>
__global__ __launch_bounds__(256) void head_wce_fwd_bf16_kernel(const __bf16 *__restrict__ x, const float *__restrict__ w,
                                                                 SqHeadCE ce, double *__restrict__ partials, int64_t npix,
                                                                 int Cin) {
    __shared__ double red[256];
    double tsum = 0.0;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (int64_t)gridDim.x * 256) {
        float acc[COUT], yc[COUT];
#pragma unroll
        for (int o = 0; o < COUT; ++o) acc[o] = 0.f;
        for (int c8 = 0; c8 < Cin / 8; ++c8) {
            const bf16x8 v = *reinterpret_cast<const bf16x8 *>(x + p * Cin + c8 * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int o = 0; o < COUT; ++o) acc[o] = __builtin_fmaf(w[(c8 * 8 + j) * COUT + o], (float)v[j], acc[o]);
        }
#pragma unroll
        for (int o = 0; o < COUT; ++o) {
            acc[o] = acc[o] + (ce.bias ? ce.bias[o] : 0.f);
            yc[o] = (float)ce.yoh[p * COUT + o];
        }
        tsum += (double)sq_wce_pixel<COUT>(acc, yc, COUT, ce.wgt[p], 0.f, nullptr);
    }
    red[threadIdx.x] = tsum;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void head_wce_finish_kernel(const double *__restrict__ partials, int n, double inv_npix,
                                                              float *__restrict__ loss) {      // = ce_finish_kernel, f32 out
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += partials[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = (float)(red[0] * inv_npix);
}

// head backward: x bf16 (CIN ch), dz f32 (COUT ch) -> dx bf16, block partials of dW (CIN,COUT) and db.
// CE: dz is not read but formed here, from the logits recomputed out of x: dz = (w_p / npix * (softmax * sum(y) - y))
// * dloss, the two roundings of wsoftmax_ce_f32_kernel followed by the tape's multiplication with the incoming gradient
template <int CIN, int COUT, bool CE = false>
__global__ __launch_bounds__(256) void head_bwd_bf16_kernel(const __bf16 *__restrict__ x, const float *__restrict__ w,
                                                             const float *__restrict__ dz, __bf16 *__restrict__ dx,
                                                             float *__restrict__ partials, int64_t npix, float gscale,
                                                             SqHeadCE ce = SqHeadCE{}) {
    constexpr int NVAL = CIN * COUT + COUT;
    __shared__ float red[4][NVAL];
    double tsum = 0.0;                                          // CE with lparts: this thread's pixels in head_wce_fwd's order
    float gw[CIN][COUT], gb[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) gb[o] = 0.f;
#pragma unroll
    for (int c = 0; c < CIN; ++c)
#pragma unroll
        for (int o = 0; o < COUT; ++o) gw[c][o] = 0.f;
    const int64_t stride = (int64_t)gridDim.x * 256;
    // the operands of a thread's NEXT pixel are requested before this pixel is worked on: the loop was load -> ~150 VALU ->
    // store with nothing in flight meanwhile (3.6 TB/s of its bytes at level 0); same pixels, same order of the sums
    bf16x8 xn[CIN / 8] = {};
    float yn[COUT] = {}, wn = 0.f, gn[COUT] = {};
    auto fetch = [&](int64_t q) {
        if (q < npix) {
#pragma unroll
            for (int c8 = 0; c8 < CIN / 8; ++c8) xn[c8] = *reinterpret_cast<const bf16x8 *>(x + q * CIN + c8 * 8);
            if constexpr (CE) {
#pragma unroll
                for (int o = 0; o < COUT; ++o) yn[o] = (float)ce.yoh[q * COUT + o];
                wn = ce.wgt[q];
            } else {
#pragma unroll
                for (int o = 0; o < COUT; ++o) gn[o] = dz[q * COUT + o];
            }
        }
    };
    fetch((int64_t)blockIdx.x * 256 + threadIdx.x);
    for (int64_t base = (int64_t)blockIdx.x * 256; base < npix; base += stride) {
        const int64_t p = base + threadIdx.x;
        float g[COUT], ycur[COUT];
        bf16x8 xv[CIN / 8];
#pragma unroll
        for (int c8 = 0; c8 < CIN / 8; ++c8) xv[c8] = xn[c8];
#pragma unroll
        for (int o = 0; o < COUT; ++o) { ycur[o] = yn[o]; g[o] = gn[o]; }
        const float wcur = wn;
        fetch(p + stride);
        if (p < npix) {
            if constexpr (CE) {
                float acc[COUT], yc[COUT];
#pragma unroll
                for (int o = 0; o < COUT; ++o) acc[o] = 0.f;
#pragma unroll
                for (int c8 = 0; c8 < CIN / 8; ++c8)
#pragma unroll
                    for (int j = 0; j < 8; ++j)
#pragma unroll
                        for (int o = 0; o < COUT; ++o)
                            acc[o] = __builtin_fmaf(w[(c8 * 8 + j) * COUT + o], (float)xv[c8][j], acc[o]);
#pragma unroll
                for (int o = 0; o < COUT; ++o) {
                    acc[o] = acc[o] + (ce.bias ? ce.bias[o] : 0.f);
                    yc[o] = ycur[o];
                }
                const float wp = wcur;
                tsum += (double)sq_wce_pixel<COUT>(acc, yc, COUT, wp, wp * ce.inv_npix, g);
                const float up = ce.dloss[0];
#pragma unroll
                for (int o = 0; o < COUT; ++o) g[o] = g[o] * up;
            }
#pragma unroll
            for (int o = 0; o < COUT; ++o) gb[o] += g[o];
#pragma unroll
            for (int c8 = 0; c8 < CIN / 8; ++c8) {
                const bf16x8 v = xv[c8];
                bf16x8 r;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float s = 0.f;
#pragma unroll
                    for (int o = 0; o < COUT; ++o) {
                        s = __builtin_fmaf(g[o], w[(c8 * 8 + j) * COUT + o], s);
                        gw[c8 * 8 + j][o] = __builtin_fmaf((float)v[j], g[o], gw[c8 * 8 + j][o]);
                    }
                    r[j] = (__bf16)s;
                    // gscale > 0: x is a dropout(ReLU(.)) block output; dx leaves already through that gate (the
                    // two roundings of the sq_relu_scale_bwd_bf16 pass this replaces)
                    if (gscale > 0.f) r[j] = (float)v[j] > 0.f ? (__bf16)((float)r[j] * gscale) : (__bf16)0.f;
                }
                if (dx) *reinterpret_cast<bf16x8 *>(dx + p * CIN + c8 * 8) = r;
            }
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < CIN; ++c)
#pragma unroll
        for (int o = 0; o < COUT; ++o) {
            const float v = wave_sum(gw[c][o]);
            if (lane == 0) red[wv][c * COUT + o] = v;
        }
#pragma unroll
    for (int o = 0; o < COUT; ++o) {
        const float v = wave_sum(gb[o]);
        if (lane == 0) red[wv][CIN * COUT + o] = v;
    }
    __syncthreads();
    if ((int)threadIdx.x < NVAL)
        partials[(size_t)blockIdx.x * NVAL + threadIdx.x] =
            ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    if constexpr (CE) {
        if (ce.lparts) {                                        // the loss the forward pass did not compute: same tree, same bits
            __shared__ double lred[256];
            lred[threadIdx.x] = tsum;
            __syncthreads();
            for (int s = 128; s > 0; s >>= 1) {
                if ((int)threadIdx.x < s) lred[threadIdx.x] += lred[threadIdx.x + s];
                __syncthreads();
            }
            if (threadIdx.x == 0) ce.lparts[blockIdx.x] = lred[0];
        }
    }
}
__global__ __launch_bounds__(256) void head_finish2_kernel(const float *__restrict__ partials, float *__restrict__ dw,
                                                           float *__restrict__ db, int nblk, int nw, int nb, int G) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int i = t / G, g = t % G;
    if (i >= nw + nb) return;
    const float s = sq_group_reduce(partials + i, (size_t)(nw + nb), nblk, g, G);
    if (g != 0) return;
    if (i < nw) dw[i] = s;
    else if (db) db[i - nw] = s;
}

inline int head_blocks(int64_t npix) {
    int64_t b = (npix + 255) / 256;
    return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

}  // namespace

#define SQ_ST(s) reinterpret_cast<hipStream_t>(s)
#define BF(p) reinterpret_cast<const __bf16 *>(p)
#define BFM(p) reinterpret_cast<__bf16 *>(p)

extern "C" int sq_cast_f32_to_bf16(const float *x, void *y, int64_t n, void *stream) {
    SQ_REQUIRE(x && y && n > 0 && n % 4 == 0, "sq_cast_f32_to_bf16: bad arguments (n %% 4 == 0)");
    SQ_REQUIRE_ALIGNED(x);
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(n / 4)), dim3(256), 0, SQ_ST(stream),
                       reinterpret_cast<const float4 *>(x), reinterpret_cast<bf16x4 *>(y), n / 4);
    return sq_check_launch("sq_cast_f32_to_bf16");
}
extern "C" int sq_cast_bf16_to_f32(const void *x, float *y, int64_t n, void *stream) {
    SQ_REQUIRE(x && y && n > 0 && n % 4 == 0, "sq_cast_bf16_to_f32: bad arguments (n %% 4 == 0)");
    SQ_REQUIRE_ALIGNED(y);
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid_for(n / 4)), dim3(256), 0, SQ_ST(stream),
                       reinterpret_cast<const bf16x4 *>(x), reinterpret_cast<float4 *>(y), n / 4);
    return sq_check_launch("sq_cast_bf16_to_f32");
}

extern "C" int sq_maxpool2x2_fwd_bf16(const void *x, void *y, int N, int H, int W, int C, void *stream) {
    SQ_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && H % 2 == 0 && W % 2 == 0 && C % 8 == 0,
               "sq_maxpool2x2_fwd_bf16: need even H,W and C %% 8 == 0");
    SQ_REQUIRE_ALIGNED(x); SQ_REQUIRE_ALIGNED(y);
    hipLaunchKernelGGL(maxpool_bf16_kernel, dim3(grid_for((int64_t)N * (H / 2) * (W / 2) * (C / 8))), dim3(256), 0,
                       SQ_ST(stream), reinterpret_cast<const bf16x8 *>(x), reinterpret_cast<bf16x8 *>(y), N, H, W, C / 8);
    return sq_check_launch("sq_maxpool2x2_fwd_bf16");
}
extern "C" int sq_maxpool2x2_bwd_bf16(const void *x, const void *dy, void *dx, int N, int H, int W, int C, void *stream) {
    SQ_REQUIRE(x && dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && H % 2 == 0 && W % 2 == 0 && C % 8 == 0,
               "sq_maxpool2x2_bwd_bf16: need even H,W and C %% 8 == 0");
    SQ_REQUIRE_ALIGNED(x); SQ_REQUIRE_ALIGNED(dy); SQ_REQUIRE_ALIGNED(dx);
    hipLaunchKernelGGL(maxpool_bwd_bf16_kernel, dim3(grid_for((int64_t)N * (H / 2) * (W / 2) * (C / 8))), dim3(256), 0,
                       SQ_ST(stream), reinterpret_cast<const bf16x8 *>(x), reinterpret_cast<const bf16x8 *>(dy),
                       reinterpret_cast<bf16x8 *>(dx), N, H, W, C / 8);
    return sq_check_launch("sq_maxpool2x2_bwd_bf16");
}
extern "C" int sq_maxpool2x2_bwd_add_bf16(const void *x, const void *dy, const void *add, void *dx, int N, int H, int W,
                                          int C, void *stream) {
    return sq_maxpool2x2_bwd_add_gate_bf16(x, dy, add, dx, N, H, W, C, 0.f, stream);
}
extern "C" int sq_maxpool2x2_bwd_add_gate_bf16(const void *x, const void *dy, const void *add, void *dx, int N, int H,
                                               int W, int C, float gate_scale, void *stream) {
    SQ_REQUIRE(x && dy && add && dx, "sq_maxpool2x2_bwd_add_bf16: null tensor pointer");
    SQ_REQUIRE(gate_scale >= 0.f, "sq_maxpool2x2_bwd_add_gate_bf16: gate_scale must be >= 0 (0 = no gate)");
    SQ_REQUIRE(N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && C % 8 == 0,
               "sq_maxpool2x2_bwd_add_bf16: H, W even, C %% 8 == 0");
    SQ_REQUIRE_ALIGNED(x); SQ_REQUIRE_ALIGNED(dy); SQ_REQUIRE_ALIGNED(add); SQ_REQUIRE_ALIGNED(dx);
    hipLaunchKernelGGL(maxpool_bwd_add_bf16_kernel, dim3(grid_for((int64_t)N * (H / 2) * (W / 2) * (C / 8))), dim3(256), 0,
                       SQ_ST(stream), reinterpret_cast<const bf16x8 *>(x), reinterpret_cast<const bf16x8 *>(dy),
                       reinterpret_cast<const bf16x8 *>(add), reinterpret_cast<bf16x8 *>(dx), N, H, W, C / 8, gate_scale);
    return sq_check_launch("sq_maxpool2x2_bwd_add_bf16");
}
extern "C" int sq_bridge_bwd_s2d_bf16(const void *dy, const void *up, const void *skip, void *g, void *dskip, int N, int H,
                                      int W, int C, int bridge, void *stream) {
    SQ_REQUIRE(dy && g && dskip, "sq_bridge_bwd_s2d_bf16: null tensor pointer");
    SQ_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "sq_bridge_bwd_s2d_bf16: C %% 8 == 0");
    SQ_REQUIRE(bridge >= SQ_BRIDGE_ADD && bridge <= SQ_BRIDGE_SUB, "sq_bridge_bwd_s2d_bf16: bad bridge %d", bridge);
    SQ_REQUIRE(bridge != SQ_BRIDGE_MUL || (up && skip), "sq_bridge_bwd_s2d_bf16: eltwise_mul needs both forward operands");
    SQ_REQUIRE_ALIGNED(dy); SQ_REQUIRE_ALIGNED(g); SQ_REQUIRE_ALIGNED(dskip);
    hipLaunchKernelGGL(bridge_bwd_s2d_bf16_kernel, dim3(grid_for((int64_t)N * 4 * H * W * (C / 8))), dim3(256), 0,
                       SQ_ST(stream), reinterpret_cast<const bf16x8 *>(dy), reinterpret_cast<const bf16x8 *>(up),
                       reinterpret_cast<const bf16x8 *>(skip), reinterpret_cast<bf16x8 *>(g),
                       reinterpret_cast<bf16x8 *>(dskip), N, H, W, C / 8, bridge);
    return sq_check_launch("sq_bridge_bwd_s2d_bf16");
}
extern "C" int sq_act_bwd_bf16(const void *dy, const void *y, void *dx, int64_t n, int act, void *stream) {
    SQ_REQUIRE(dy && y && dx && n > 0 && n % 8 == 0, "sq_act_bwd_bf16: bad arguments (n %% 8 == 0)");
    SQ_REQUIRE_ALIGNED(dy); SQ_REQUIRE_ALIGNED(y); SQ_REQUIRE_ALIGNED(dx);
    hipLaunchKernelGGL(act_bwd_bf16_kernel, dim3(grid_for(n / 8)), dim3(256), 0, SQ_ST(stream),
                       reinterpret_cast<const bf16x8 *>(dy), reinterpret_cast<const bf16x8 *>(y),
                       reinterpret_cast<bf16x8 *>(dx), n / 8, act);
    return sq_check_launch("sq_act_bwd_bf16");
}
extern "C" int sq_bridge_fwd_bf16(const void *a, const void *b, void *y, int64_t n, int bridge, void *stream) {
    SQ_REQUIRE(a && b && y && n > 0 && n % 8 == 0, "sq_bridge_fwd_bf16: bad arguments (n %% 8 == 0)");
    SQ_REQUIRE_ALIGNED(a); SQ_REQUIRE_ALIGNED(b); SQ_REQUIRE_ALIGNED(y);
    hipLaunchKernelGGL(bridge_bf16_kernel, dim3(grid_for(n / 8)), dim3(256), 0, SQ_ST(stream),
                       reinterpret_cast<const bf16x8 *>(a), reinterpret_cast<const bf16x8 *>(b),
                       reinterpret_cast<bf16x8 *>(y), n / 8, bridge);
    return sq_check_launch("sq_bridge_fwd_bf16");
}
extern "C" int sq_bridge_bwd_bf16(const void *dy, const void *a, const void *b, void *da, void *db, int64_t n, int bridge,
                                  void *stream) {
    SQ_REQUIRE(dy && da && db && n > 0 && n % 8 == 0, "sq_bridge_bwd_bf16: bad arguments (n %% 8 == 0)");
    SQ_REQUIRE(bridge >= SQ_BRIDGE_ADD && bridge <= SQ_BRIDGE_SUB, "sq_bridge_bwd_bf16: bad bridge %d", bridge);
    SQ_REQUIRE(bridge != SQ_BRIDGE_MUL || (a && b), "sq_bridge_bwd_bf16: eltwise_mul needs both forward operands");
    hipLaunchKernelGGL(bridge_bwd_bf16_kernel, dim3(grid_for(n / 8)), dim3(256), 0, SQ_ST(stream),
                       reinterpret_cast<const bf16x8 *>(dy), reinterpret_cast<const bf16x8 *>(a),
                       reinterpret_cast<const bf16x8 *>(b), reinterpret_cast<bf16x8 *>(da), reinterpret_cast<bf16x8 *>(db),
                       n / 8, bridge);
    return sq_check_launch("sq_bridge_bwd_bf16");
}
extern "C" int sq_dropout_fwd_bf16(const void *x, void *y, uint8_t *mask, int64_t n, float rate, uint32_t seed,
                                   int mask_given, const int32_t *step_dev, void *stream) {
    SQ_REQUIRE(x && y && mask && n > 0 && n % 8 == 0 && rate >= 0.f && rate < 1.f, "sq_dropout_fwd_bf16: bad arguments (n %% 8 == 0)");
    SQ_REQUIRE_ALIGNED(x); SQ_REQUIRE_ALIGNED(y);
    hipLaunchKernelGGL(dropout_fwd_bf16_kernel, dim3(grid_for(n / 8)), dim3(256), 0, SQ_ST(stream),
                       reinterpret_cast<const bf16x8 *>(x), reinterpret_cast<bf16x8 *>(y), reinterpret_cast<uint2 *>(mask),
                       n / 8, rate, seed, mask_given, step_dev);
    return sq_check_launch("sq_dropout_fwd_bf16");
}
extern "C" int sq_dropout_bwd_bf16(const void *dy, const uint8_t *mask, void *dx, int64_t n, float rate, void *stream) {
    SQ_REQUIRE(dy && mask && dx && n > 0 && n % 8 == 0 && rate >= 0.f && rate < 1.f, "sq_dropout_bwd_bf16: bad arguments (n %% 8 == 0)");
    SQ_REQUIRE_ALIGNED(dy); SQ_REQUIRE_ALIGNED(dx);
    hipLaunchKernelGGL(dropout_bwd_bf16_kernel, dim3(grid_for(n / 8)), dim3(256), 0, SQ_ST(stream),
                       reinterpret_cast<const bf16x8 *>(dy), reinterpret_cast<const uint2 *>(mask),
                       reinterpret_cast<bf16x8 *>(dx), n / 8, rate);
    return sq_check_launch("sq_dropout_bwd_bf16");
}

extern "C" int sq_relu_scale_bwd_bf16(const void *dy, const void *y, void *dx, int64_t n, float scale, void *stream) {
    SQ_REQUIRE(dy && y && dx && n > 0 && n % 8 == 0, "sq_relu_scale_bwd_bf16: bad arguments (n %% 8 == 0)");
    SQ_REQUIRE_ALIGNED(dy); SQ_REQUIRE_ALIGNED(y); SQ_REQUIRE_ALIGNED(dx);
    hipLaunchKernelGGL(relu_scale_bwd_bf16_kernel, dim3(grid_for(n / 8)), dim3(256), 0, SQ_ST(stream),
                       reinterpret_cast<const bf16x8 *>(dy), reinterpret_cast<const bf16x8 *>(y),
                       reinterpret_cast<bf16x8 *>(dx), n / 8, scale);
    return sq_check_launch("sq_relu_scale_bwd_bf16");
}

extern "C" int sq_act_dropout_bwd_bf16(const void *dy, const uint8_t *mask, const void *y, void *dx, int64_t n, float rate,
                                       int act, void *stream) {
    SQ_REQUIRE(dy && mask && y && dx && n > 0 && n % 8 == 0, "sq_act_dropout_bwd_bf16: bad arguments (n %% 8 == 0)");
    SQ_REQUIRE(rate >= 0.f && rate < 1.f, "sq_act_dropout_bwd_bf16: rate must be in [0, 1)");
    SQ_REQUIRE_ALIGNED(dy); SQ_REQUIRE_ALIGNED(y); SQ_REQUIRE_ALIGNED(dx);
    SQ_REQUIRE((((uintptr_t)mask) & 7u) == 0, "sq_act_dropout_bwd_bf16: mask must be 8-byte aligned");
    hipLaunchKernelGGL(act_dropout_bwd_bf16_kernel, dim3(grid_for(n / 8)), dim3(256), 0, SQ_ST(stream),
                       reinterpret_cast<const bf16x8 *>(dy), reinterpret_cast<const uint2 *>(mask),
                       reinterpret_cast<const bf16x8 *>(y), reinterpret_cast<bf16x8 *>(dx), n / 8, rate, act);
    return sq_check_launch("sq_act_dropout_bwd_bf16");
}

extern "C" int sq_convT2x2s2_nhwc_fwd_bf16(const void *x, const void *w, const float *bias, const void *skip, void *y,
                                           int N, int H, int W, int Cin, int Cout, int bridge, void *stream) {
    SQ_REQUIRE(x && w && y, "sq_convT2x2s2_nhwc_fwd_bf16: null tensor pointer");
    SQ_REQUIRE(N > 0 && H > 0 && W > 0 && Cin % 32 == 0 && Cin > 0 && Cout % 16 == 0 && Cout > 0,
               "sq_convT2x2s2_nhwc_fwd_bf16: Cin=%d (multiple of 32), Cout=%d (multiple of 16)", Cin, Cout);
    SQ_REQUIRE(bridge >= SQ_BRIDGE_NONE && bridge <= SQ_BRIDGE_SUB, "sq_convT2x2s2_nhwc_fwd_bf16: bad bridge %d", bridge);
    SQ_REQUIRE(bridge == SQ_BRIDGE_NONE || skip, "sq_convT2x2s2_nhwc_fwd_bf16: bridge needs a skip tensor");
    SQ_REQUIRE_ALIGNED(x); SQ_REQUIRE_ALIGNED(w); SQ_REQUIRE_ALIGNED(y);
    const int64_t P = (int64_t)N * H * W;
    dim3 grid((unsigned)((P + 63) / 64), (unsigned)(4 * Cout / 64));
    hipLaunchKernelGGL(convT_bf16_kernel, grid, dim3(256), 0, SQ_ST(stream), BF(x), BF(w), bias, BF(skip), BFM(y), P, H, W,
                       Cin, Cout, bridge, (__bf16 *)nullptr);
    return sq_check_launch("sq_convT2x2s2_nhwc_fwd_bf16");
}

// the training form: writes BOTH the up-scaled tensor (needed by the bridge backward) and bridge(up, skip) from one
// pass -- the separate bridge kernel's re-read of `up` and its launch disappear; same bits as the two kernels
extern "C" int sq_convT2x2s2_bridge_both_fwd_bf16(const void *x, const void *w, const float *bias, const void *skip,
                                                  void *up, void *merged, int N, int H, int W, int Cin, int Cout,
                                                  int bridge, void *stream) {
    SQ_REQUIRE(x && w && skip && up && merged, "sq_convT2x2s2_bridge_both_fwd_bf16: null tensor pointer");
    SQ_REQUIRE(N > 0 && H > 0 && W > 0 && Cin % 32 == 0 && Cin > 0 && Cout % 16 == 0 && Cout > 0,
               "sq_convT2x2s2_bridge_both_fwd_bf16: Cin=%d (multiple of 32), Cout=%d (multiple of 16)", Cin, Cout);
    SQ_REQUIRE(bridge > SQ_BRIDGE_NONE && bridge <= SQ_BRIDGE_SUB, "sq_convT2x2s2_bridge_both_fwd_bf16: bad bridge %d", bridge);
    SQ_REQUIRE_ALIGNED(x); SQ_REQUIRE_ALIGNED(w); SQ_REQUIRE_ALIGNED(up); SQ_REQUIRE_ALIGNED(merged); SQ_REQUIRE_ALIGNED(skip);
    const int64_t P = (int64_t)N * H * W;
    dim3 grid((unsigned)((P + 63) / 64), (unsigned)(4 * Cout / 64));
    hipLaunchKernelGGL(convT_bf16_kernel, grid, dim3(256), 0, SQ_ST(stream), BF(x), BF(w), bias, BF(skip), BFM(merged), P, H,
                       W, Cin, Cout, bridge, BFM(up));
    return sq_check_launch("sq_convT2x2s2_bridge_both_fwd_bf16");
}

extern "C" int sq_conv1x1_head_fwd_bf16(const void *x, const float *w, const float *bias, float *logits, uint8_t *mask,
                                        int64_t npix, int Cin, int Cout, void *stream) {
    SQ_REQUIRE(x && w && logits && npix > 0 && Cin > 0 && Cin % 8 == 0, "sq_conv1x1_head_fwd_bf16: bad arguments (Cin %% 8)");
    SQ_REQUIRE_ALIGNED(x);
    const unsigned nb = (unsigned)((npix + 255) / 256);
    hipStream_t st = SQ_ST(stream);
    switch (Cout) {
    case 1: hipLaunchKernelGGL(head_fwd_bf16_kernel<1>, dim3(nb), dim3(256), 0, st, BF(x), w, bias, logits, mask, npix, Cin); break;
    case 2: hipLaunchKernelGGL(head_fwd_bf16_kernel<2>, dim3(nb), dim3(256), 0, st, BF(x), w, bias, logits, mask, npix, Cin); break;
    case 3: hipLaunchKernelGGL(head_fwd_bf16_kernel<3>, dim3(nb), dim3(256), 0, st, BF(x), w, bias, logits, mask, npix, Cin); break;
    case 4: hipLaunchKernelGGL(head_fwd_bf16_kernel<4>, dim3(nb), dim3(256), 0, st, BF(x), w, bias, logits, mask, npix, Cin); break;
    case 5: hipLaunchKernelGGL(head_fwd_bf16_kernel<5>, dim3(nb), dim3(256), 0, st, BF(x), w, bias, logits, mask, npix, Cin); break;
    default: sq_set_error("sq_conv1x1_head_fwd_bf16: Cout=%d unsupported (1..5)", Cout); return SQ_EINVAL;
    }
    return sq_check_launch("sq_conv1x1_head_fwd_bf16");
}

extern "C" int64_t sq_conv1x1_head_bwd_workspace_bf16(int64_t npix, int Cin, int Cout) {
    if (npix <= 0 || Cin <= 0 || Cout <= 0) return -1;
    return (int64_t)head_blocks(npix) * (Cin * Cout + Cout) * 4;
}

extern "C" int sq_conv1x1_head_bwd_bf16(const void *x, const float *w, const float *dz, void *dx, float *dw, float *db,
                                        float *workspace, int64_t npix, int Cin, int Cout, void *stream) {
    return sq_conv1x1_head_bwd_gate_bf16(x, w, dz, dx, dw, db, workspace, npix, Cin, Cout, 0.f, stream);
}
template <bool CE>
static int head_bwd_impl(const void *x, const float *w, const float *dz, void *dx, float *dw, float *db, float *workspace,
                         int64_t npix, int Cin, int Cout, float gate_scale, void *stream, const SqHeadCE &ce) {
    SQ_REQUIRE(x && w && (CE || dz) && dw && workspace && npix > 0, "sq_conv1x1_head_bwd_bf16: null pointer");
    SQ_REQUIRE(gate_scale >= 0.f, "sq_conv1x1_head_bwd_gate_bf16: gate_scale must be >= 0 (0 = no gate)");
    SQ_REQUIRE((Cin == 16 || Cin == 32) && Cout >= 1 && Cout <= 5, "sq_conv1x1_head_bwd_bf16: Cin=%d (16|32), Cout=%d (1..5)", Cin, Cout);
    const int nb = head_blocks(npix);
    hipStream_t st = SQ_ST(stream);
#define SQ_HEAD_BWD(CI, CO) hipLaunchKernelGGL((head_bwd_bf16_kernel<CI, CO, CE>), dim3(nb), dim3(256), 0, st, BF(x), w, dz, BFM(dx), workspace, npix, gate_scale, ce)
    if (Cin == 16) {
        switch (Cout) {
        case 1: SQ_HEAD_BWD(16, 1); break;
        case 2: SQ_HEAD_BWD(16, 2); break;
        case 3: SQ_HEAD_BWD(16, 3); break;
        case 4: SQ_HEAD_BWD(16, 4); break;
        default: SQ_HEAD_BWD(16, 5); break;
        }
    } else {
        switch (Cout) {
        case 1: SQ_HEAD_BWD(32, 1); break;
        case 2: SQ_HEAD_BWD(32, 2); break;
        case 3: SQ_HEAD_BWD(32, 3); break;
        case 4: SQ_HEAD_BWD(32, 4); break;
        default: SQ_HEAD_BWD(32, 5); break;
        }
    }
#undef SQ_HEAD_BWD
    int rc = sq_check_launch("sq_conv1x1_head_bwd_bf16");
    if (rc) return rc;
    const int nw = Cin * Cout;
    { const int G = sq_group_size(nb); hipLaunchKernelGGL(head_finish2_kernel, dim3(((nw + Cout) * G + 255) / 256), dim3(256), 0, st, workspace, dw, db, nb, nw, Cout, G); }
    return sq_check_launch("sq_conv1x1_head_bwd_bf16(finish)");
}

extern "C" int sq_conv1x1_head_bwd_gate_bf16(const void *x, const float *w, const float *dz, void *dx, float *dw, float *db,
                                             float *workspace, int64_t npix, int Cin, int Cout, float gate_scale,
                                             void *stream) {
    return head_bwd_impl<false>(x, w, dz, dx, dw, db, workspace, npix, Cin, Cout, gate_scale, stream, SqHeadCE{});
}

// ---- to_image head + weighted softmax-CE without the logits round trip (training) -----------------------------------
// forward: loss (f32 device scalar) = mean over pixels of w * CE(softmax(head(x)), y); partials: sq_wsoftmax_ce_partials(npix)
// doubles.  Same logits as sq_conv1x1_head_fwd_bf16 and same loss as sq_wsoftmax_ce_fwd_bwd_f32 on them, bit for bit.
extern "C" int sq_conv1x1_head_wce_fwd_bf16(const void *x, const float *w, const float *bias, const uint8_t *onehot,
                                            const float *weights, double *partials, float *loss, int64_t npix, int Cin,
                                            int Cout, void *stream) {
    SQ_REQUIRE(x && w && onehot && weights && partials && loss && npix > 0, "sq_conv1x1_head_wce_fwd_bf16: null pointer");
    SQ_REQUIRE(Cin > 0 && Cin % 8 == 0 && Cout >= 1 && Cout <= 5, "sq_conv1x1_head_wce_fwd_bf16: Cin=%d (%% 8), Cout=%d (1..5)", Cin, Cout);
    SQ_REQUIRE_ALIGNED(x);
    const int64_t nb64 = sq_wsoftmax_ce_partials(npix);
    const unsigned nb = (unsigned)nb64;
    hipStream_t st = SQ_ST(stream);
    const SqHeadCE ce{bias, onehot, weights, nullptr, 0.f, nullptr};
    switch (Cout) {
    case 1: hipLaunchKernelGGL(head_wce_fwd_bf16_kernel<1>, dim3(nb), dim3(256), 0, st, BF(x), w, ce, partials, npix, Cin); break;
    case 2: hipLaunchKernelGGL(head_wce_fwd_bf16_kernel<2>, dim3(nb), dim3(256), 0, st, BF(x), w, ce, partials, npix, Cin); break;
    case 3: hipLaunchKernelGGL(head_wce_fwd_bf16_kernel<3>, dim3(nb), dim3(256), 0, st, BF(x), w, ce, partials, npix, Cin); break;
    case 4: hipLaunchKernelGGL(head_wce_fwd_bf16_kernel<4>, dim3(nb), dim3(256), 0, st, BF(x), w, ce, partials, npix, Cin); break;
    default: hipLaunchKernelGGL(head_wce_fwd_bf16_kernel<5>, dim3(nb), dim3(256), 0, st, BF(x), w, ce, partials, npix, Cin); break;
    }
    int rc = sq_check_launch("sq_conv1x1_head_wce_fwd_bf16");
    if (rc) return rc;
    hipLaunchKernelGGL(head_wce_finish_kernel, dim3(1), dim3(256), 0, st, partials, (int)nb, 1.0 / (double)npix, loss);
    return sq_check_launch("sq_conv1x1_head_wce_fwd_bf16(finish)");
}

// backward of the pair above: dloss = the gradient arriving at the loss (f32 device scalar); dx / dW / db as
// sq_conv1x1_head_bwd_gate_bf16 fed with sq_wsoftmax_ce_fwd_bwd_f32's dlogits * dloss, bit for bit
extern "C" int sq_conv1x1_head_wce_bwd_bf16(const void *x, const float *w, const float *bias, const uint8_t *onehot,
                                            const float *weights, const float *dloss, void *dx, float *dw, float *db,
                                            float *workspace, int64_t npix, int Cin, int Cout, float gate_scale,
                                            void *stream) {
    SQ_REQUIRE(onehot && weights && dloss, "sq_conv1x1_head_wce_bwd_bf16: null pointer");
    const SqHeadCE ce{bias, onehot, weights, dloss, 1.0f / (float)npix, nullptr};
    return head_bwd_impl<true>(x, w, nullptr, dx, dw, db, workspace, npix, Cin, Cout, gate_scale, stream, ce);
}

// the same backward pass that ALSO leaves the loss (sq_conv1x1_head_wce_fwd_bf16's value, bit for bit: same grid, same
// per-thread pixel order, same fp64 tree): a training step that always runs the backward right after the forward skips the
// forward kernel -- one read of the level-0 activation less.  partials: sq_wsoftmax_ce_partials(npix) doubles.
extern "C" int sq_conv1x1_head_wce_bwd_loss_bf16(const void *x, const float *w, const float *bias, const uint8_t *onehot,
                                                 const float *weights, const float *dloss, void *dx, float *dw, float *db,
                                                 float *workspace, double *partials, float *loss, int64_t npix, int Cin,
                                                 int Cout, float gate_scale, void *stream) {
    SQ_REQUIRE(onehot && weights && dloss && partials && loss, "sq_conv1x1_head_wce_bwd_loss_bf16: null pointer");
    SQ_REQUIRE(sq_wsoftmax_ce_partials(npix) == head_blocks(npix), "sq_conv1x1_head_wce_bwd_loss_bf16: grids differ");
    const SqHeadCE ce{bias, onehot, weights, dloss, 1.0f / (float)npix, partials};
    int rc = head_bwd_impl<true>(x, w, nullptr, dx, dw, db, workspace, npix, Cin, Cout, gate_scale, stream, ce);
    if (rc) return rc;
    hipLaunchKernelGGL(head_wce_finish_kernel, dim3(1), dim3(256), 0, SQ_ST(stream), partials, head_blocks(npix),
                       1.0 / (double)npix, loss);
    return sq_check_launch("sq_conv1x1_head_wce_bwd_loss_bf16(finish)");
}
