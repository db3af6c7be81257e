/*
 * sq_oracle.c -- CPU restatement of the sequitr per-tile network hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under sequitr_amd/ may import, link or
 * call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, as the checker.
 *
 * PARITY STATUS: "parity unpinned" for the U-Net leaf ops and the loss -- the
 * reference leaves them abstract (sequitr/networks/unet.py:326-343 raise
 * NotImplementedError, no model_fn / loss in the tree) and ships no tests or
 * golden vectors.  The wiring follows
 * sequitr/networks/unet.py:224-322 and the GAN leaf ops follow
 * sequitr/networks/gan.py:44-136 literally.  This restatement is pinned
 * against an independent fp64 torch-CPU implementation in
 * tests/test_oracle.py instead.
 *
 * NUMERICS CONTRACT (shared with the HIP kernels):
 * every convolution output is ONE single-precision fmaf chain that starts at
 * +0.0f and visits the reduction index in this order:
 *
 *     for chunk in range(0, Cin, KC):          KC = min(Cin, 16)
 *       for tap = ky*KW + kx (raster order):
 *         for c in chunk .. chunk+KC-1:
 *           acc = fmaf(w[ky,kx,c,o] * 1, x[n, y+ky-p, x+kx-p, c] or 0, acc)
 *
 * then  y = acc + bias[o]  (one rounding), then the activation.  Zero padding
 * takes part in the chain as an explicit 0 operand.  This is exactly what a
 * k-ordered v_mfma_f32_16x16x4_f32 accumulation produces on gfx950 (the
 * instruction is bit-for-bit an fmaf chain), so the GPU logits are compared
 * BIT-EXACT against this file, not within a tolerance.
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>

#define SQ_ACT_NONE 0
#define SQ_ACT_RELU 1
#define SQ_ACT_LEAKY 2 /* alpha = 0.2, sequitr/networks/gan.py:44-46 */

#define SQ_BRIDGE_NONE 0
#define SQ_BRIDGE_ADD 1
#define SQ_BRIDGE_MUL 2
#define SQ_BRIDGE_SUB 3

static inline float act_f32(float v, int act) {
    if (act == SQ_ACT_RELU) return v > 0.0f ? v : 0.0f;
    if (act == SQ_ACT_LEAKY) return v > 0.0f ? v : 0.2f * v;
    return v;
}

/*
 * conv_layer / conv_layer_1x1 / weighted_conv2d core.
 * KxK SAME stride-1 NHWC convolution, HWIO weights, optional runtime weight
 * scale (w' = fl(w * wscale), sequitr/networks/gan.py:75-79), bias, activation.
 *   x: (N,H,W,Cin)  w: (K,K,Cin,Cout)  bias: (Cout) or NULL  y: (N,H,W,Cout)
 * Hooks restated: sequitr/networks/unet.py:326-333 (build default),
 *                 sequitr/networks/gan.py:61-99.
 */
int oracle_conv2d_nhwc_f32(const float *x, const float *w, const float *bias,
                           float *y, int N, int H, int W, int Cin, int Cout,
                           int K, float wscale, int act) {
    if (K != 1 && K != 3) return -1;
    const int pad = K / 2;
    const int KC = Cin < 16 ? Cin : 16;
    if (Cin % KC) return -2;
    if (Cout > 1024) return -3;
    /* w' = fl(w * wscale) once, as the reference scales the kernel tensor
     * before the convolution (sequitr/networks/gan.py:79). */
    const size_t nw = (size_t)K * K * Cin * Cout;
    float *ws = (float *)malloc(nw * sizeof(float));
    if (!ws) return -4;
    for (size_t i = 0; i < nw; ++i) ws[i] = w[i] * wscale;
    const long rows = (long)N * H;
#pragma omp parallel for schedule(static)
    for (long r = 0; r < rows; ++r) {
        const int n = (int)(r / H), yy = (int)(r % H);
        float acc[1024];
        for (int xx = 0; xx < W; ++xx) {
            float *yo = y + (((size_t)n * H + yy) * W + xx) * Cout;
            for (int o = 0; o < Cout; ++o) acc[o] = 0.0f;
            /* every acc[o] is its own chain; the o loop is only SIMD width */
            for (int cc = 0; cc < Cin; cc += KC) {
                for (int ky = 0; ky < K; ++ky) {
                    for (int kx = 0; kx < K; ++kx) {
                        const int sy = yy + ky - pad, sx = xx + kx - pad;
                        const int inb = (sy >= 0 && sy < H && sx >= 0 && sx < W);
                        const float *xi = inb ? x + (((size_t)n * H + sy) * W + sx) * Cin : NULL;
                        const float *wk = ws + ((size_t)(ky * K + kx) * Cin) * Cout;
                        for (int c = cc; c < cc + KC; ++c) {
                            const float xv = inb ? xi[c] : 0.0f;
                            const float *wr = wk + (size_t)c * Cout;
                            for (int o = 0; o < Cout; ++o) acc[o] = fmaf(wr[o], xv, acc[o]);
                        }
                    }
                }
            }
            for (int o = 0; o < Cout; ++o) {
                float v = acc[o];
                if (bias) v = v + bias[o];
                yo[o] = act_f32(v, act);
            }
        }
    }
    free(ws);
    return 0;
}

/* pixel_norm, sequitr/networks/gan.py:49-51:
 *   y = x * rsqrt(mean_c(x^2) + eps); the mean is a sequential f32 sum of
 *   fl(x*x) over c, divided by C; rsqrt = 1/sqrtf (correctly rounded both). */
int oracle_pixelnorm_f32(const float *x, float *y, long npix, int C, float eps) {
#pragma omp parallel for schedule(static)
    for (long p = 0; p < npix; ++p) {
        const float *xi = x + (size_t)p * C;
        float s = 0.0f;
        for (int c = 0; c < C; ++c) s = fmaf(xi[c], xi[c], s);
        const float r = 1.0f / sqrtf(s / (float)C + eps);
        for (int c = 0; c < C; ++c) y[(size_t)p * C + c] = xi[c] * r;
    }
    return 0;
}

/* max_pool_layer / pool_layer hook (sequitr/networks/unet.py:242,340-342):
 * 2x2 stride-2 VALID max pooling, NHWC. */
int oracle_maxpool2x2_f32(const float *x, float *y, int N, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2;
#pragma omp parallel for schedule(static)
    for (long r = 0; r < (long)N * Ho; ++r) {
        const int n = (int)(r / Ho), yo = (int)(r % Ho);
        for (int xo = 0; xo < Wo; ++xo)
            for (int c = 0; c < C; ++c) {
                const float *p = x + (((size_t)n * H + 2 * yo) * W + 2 * xo) * C + c;
                float m = p[0];
                const float b = p[C], d = p[(size_t)W * C], e = p[(size_t)W * C + C];
                m = b > m ? b : m;
                m = d > m ? d : m;
                m = e > m ? e : m;
                y[(((size_t)n * Ho + yo) * Wo + xo) * C + c] = m;
            }
    }
    return 0;
}

/* 2x2 stride-2 average pooling (tf.layers.average_pooling2d,
 * sequitr/networks/gan.py:189-192): ((a+b)+(c+d))*0.25f in that order. */
int oracle_avgpool2x2_f32(const float *x, float *y, int N, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2;
#pragma omp parallel for schedule(static)
    for (long r = 0; r < (long)N * Ho; ++r) {
        const int n = (int)(r / Ho), yo = (int)(r % Ho);
        for (int xo = 0; xo < Wo; ++xo)
            for (int c = 0; c < C; ++c) {
                const float *p = x + (((size_t)n * H + 2 * yo) * W + 2 * xo) * C + c;
                const float s = (p[0] + p[C]) + (p[(size_t)W * C] + p[(size_t)W * C + C]);
                y[(((size_t)n * Ho + yo) * Wo + xo) * C + c] = s * 0.25f;
            }
    }
    return 0;
}

/* conv_transpose_layer hook (sequitr/networks/unet.py:336-338; build default
 * SURVEY A.1): 2x2 stride-2 transpose conv, TF kernel layout (2,2,Cout,Cin),
 *   y[n,2i+a,2j+b,o] = chain_c fmaf(w[a,b,o,c], x[n,i,j,c]) + bias[o]
 * followed by the bridge (sequitr/networks/unet.py:190-200) against
 * `skip` (N,2H,2W,Cout):  add / mul / sub with the up-scaled value FIRST. */
int oracle_convT2x2s2_nhwc_f32(const float *x, const float *w, const float *bias,
                               const float *skip, float *y, int N, int H, int W,
                               int Cin, int Cout, int bridge) {
    const int Ho = 2 * H, Wo = 2 * W;
#pragma omp parallel for schedule(static)
    for (long r = 0; r < (long)N * Ho; ++r) {
        const int n = (int)(r / Ho), yo = (int)(r % Ho);
        const int i = yo >> 1, a = yo & 1;
        for (int xo = 0; xo < Wo; ++xo) {
            const int j = xo >> 1, b = xo & 1;
            const float *xi = x + (((size_t)n * H + i) * W + j) * Cin;
            const size_t ob = (((size_t)n * Ho + yo) * Wo + xo) * Cout;
            for (int o = 0; o < Cout; ++o) {
                const float *wk = w + ((size_t)(a * 2 + b) * Cout + o) * Cin;
                float acc = 0.0f;
                for (int c = 0; c < Cin; ++c) acc = fmaf(wk[c], xi[c], acc);
                float v = acc;
                if (bias) v = v + bias[o];
                if (bridge == SQ_BRIDGE_ADD) v = v + skip[ob + o];
                else if (bridge == SQ_BRIDGE_MUL) v = v * skip[ob + o];
                else if (bridge == SQ_BRIDGE_SUB) v = v - skip[ob + o];
                y[ob + o] = v;
            }
        }
    }
    return 0;
}

/* prediction: argmax over the channel axis, ties -> lowest index. */
int oracle_argmax_u8(const float *logits, uint8_t *mask, long npix, int C) {
#pragma omp parallel for schedule(static)
    for (long p = 0; p < npix; ++p) {
        const float *z = logits + (size_t)p * C;
        int best = 0;
        for (int c = 1; c < C; ++c)
            if (z[c] > z[best]) best = c;
        mask[p] = (uint8_t)best;
    }
    return 0;
}

/* nearest-neighbour 2x up-sampling (double_size, sequitr/networks/gan.py:133-136:
 * resize_nearest_neighbor(align_corners=True) to 2H: src = floor(d/2)). */
int oracle_upsample_nn2x_f32(const float *x, float *y, int N, int H, int W, int C) {
    const int Ho = 2 * H, Wo = 2 * W;
#pragma omp parallel for schedule(static)
    for (long r = 0; r < (long)N * Ho; ++r) {
        const int n = (int)(r / Ho), yo = (int)(r % Ho);
        for (int xo = 0; xo < Wo; ++xo)
            memcpy(y + (((size_t)n * Ho + yo) * Wo + xo) * C,
                   x + (((size_t)n * H + (yo >> 1)) * W + (xo >> 1)) * C,
                   sizeof(float) * (size_t)C);
    }
    return 0;
}

/*
 * Weighted softmax cross-entropy (tensor contract from
 * sequitr/networks/unet.py:395-401: one-hot uint8 labels (N,H,W,C),
 * float32 weights (N,H,W,1)).  fp64 accumulation in the oracle.
 *   loss = 1/P * sum_p w_p (logsumexp(z_p) - sum_c y_pc z_pc)
 *   dz_pc = w_p (softmax(z_p)_c - y_pc) / P
 * dlogits may be NULL.
 */
int oracle_wsoftmax_ce_f32(const float *logits, const uint8_t *onehot,
                           const float *weights, long npix, int C,
                           double *loss_out, float *dlogits) {
    double total = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : total)
    for (long p = 0; p < npix; ++p) {
        const float *z = logits + (size_t)p * C;
        double m = z[0];
        for (int c = 1; c < C; ++c) m = z[c] > m ? z[c] : m;
        double s = 0.0, dot = 0.0;
        for (int c = 0; c < C; ++c) {
            s += exp((double)z[c] - m);
            dot += (double)onehot[(size_t)p * C + c] * (double)z[c];
        }
        const double lse = m + log(s);
        double yt = 0.0;
        for (int c = 0; c < C; ++c) yt += (double)onehot[(size_t)p * C + c];
        const double wp = weights[p];
        total += wp * (lse * yt - dot);
        if (dlogits)
            for (int c = 0; c < C; ++c) {
                const double sm = exp((double)z[c] - lse);
                dlogits[(size_t)p * C + c] =
                    (float)(wp * (sm * yt - (double)onehot[(size_t)p * C + c]) / (double)npix);
            }
    }
    *loss_out = total / (double)npix;
    return 0;
}

/* Batch normalisation between conv and activation (SURVEY.md A.1 optional `batch_norm`;
 * tf.layers.batch_normalization: y = gamma*(x-mean)/sqrt(var+eps)+beta).  The reference leaves the layer
 * abstract (sequitr/networks/unet.py:326-328) -> parity unpinned; the restatement fixes the arithmetic:
 *   stats: mean = sum(x)/M, var = sum(x^2)/M - mean^2 in double, rounded to float (population variance)
 *   fold : scale = gamma / sqrtf(var + eps); shift = fmaf(-mean, scale, beta)
 *   apply: y = act(fmaf(x, scale[c], shift[c])) */
int oracle_bn_stats_f32(const float *x, float *mean, float *var, long npix, int C) {
    for (int c = 0; c < C; ++c) {
        double s = 0.0, q = 0.0;
        for (long p = 0; p < npix; ++p) {
            const double v = (double)x[p * C + c];
            s += v;
            q += v * v;
        }
        const double m = s / (double)npix;
        const double v = q / (double)npix - m * m;
        mean[c] = (float)m;
        var[c] = (float)(v > 0.0 ? v : 0.0);
    }
    return 0;
}

int oracle_bn_fold_f32(const float *gamma, const float *beta, const float *mean, const float *var, float eps,
                       float *scale, float *shift, int C) {
    for (int c = 0; c < C; ++c) {
        const float s = gamma[c] / sqrtf(var[c] + eps);
        scale[c] = s;
        shift[c] = fmaf(-mean[c], s, beta[c]);
    }
    return 0;
}

int oracle_bn_apply_f32(const float *x, const float *scale, const float *shift, float *y, long npix, int C, int act) {
#pragma omp parallel for schedule(static)
    for (long p = 0; p < npix; ++p)
        for (int c = 0; c < C; ++c)
            y[p * C + c] = act_f32(fmaf(x[p * C + c], scale[c], shift[c]), act);
    return 0;
}

int oracle_version(void) { return 2; }
