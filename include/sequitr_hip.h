/*
 * sequitr_hip.h -- C-ABI of libsequitr_hip.so, the MI355X (gfx950) back end of
 * the sequitr per-tile network hot path.
 *
 * The reference has NO FFI: its leaf operators are Python hooks
 * on the UNet class (sequitr/networks/unet.py:326-343) and module-level
 * functions in sequitr/networks/gan.py:44-136 that call TensorFlow.  Each entry
 * point below names the hook / function whose arithmetic it replaces; the
 * Python side that binds them with ctypes is sequitr_amd/_lib.py +
 * sequitr_amd/ops.py, and INTEGRATION.md shows the stub a maintainer of the
 * reference would add.
 *
 * Conventions
 *   - every function returns int: 0 = ok, negative = SQ_E*; sq_last_error()
 *     returns a thread-local message for the last failure on this thread;
 *   - all tensor pointers are DEVICE pointers (e.g. torch.Tensor.data_ptr());
 *     the caller owns every byte, kernels never allocate;
 *   - layout is NHWC, weights HWIO (kh,kw,Cin,Cout), transpose-conv weights in
 *     TensorFlow's (kh,kw,Cout,Cin);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *     launches are asynchronous and graph-capturable (no sync, no malloc);
 *   - no global mutable state: one host thread per GPU process may call in.
 *
 * Numerics contract for the f32 entry points: every
 * convolution output is one f32 fmaf chain from +0, reduction order
 * (16-channel chunk, tap, channel), then "+ bias" (one rounding), then the
 * activation.  oracle/sq_oracle.c restates it; parity is bit-exact.
 */
#ifndef SEQUITR_HIP_H
#define SEQUITR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SQ_OK 0
#define SQ_EINVAL (-1)   /* bad shape / null pointer / unsupported combination */
#define SQ_ELAUNCH (-2)  /* HIP reported a launch error */
#define SQ_EALIGN (-3)   /* pointer not 16-byte aligned */

/* activations */
#define SQ_ACT_NONE 0
#define SQ_ACT_RELU 1    /* UNet._activation, sequitr/networks/unet.py:142 */
#define SQ_ACT_LEAKY 2   /* k_leaky_relu_alpha, alpha 0.2, sequitr/networks/gan.py:44-46 */

/* bridges, sequitr/networks/unet.py:42,190-200 (up-scaled operand first) */
#define SQ_BRIDGE_NONE 0
#define SQ_BRIDGE_ADD 1
#define SQ_BRIDGE_MUL 2
#define SQ_BRIDGE_SUB 3

int sq_version(void);
const char *sq_last_error(void);

/*
 * conv_layer / conv_layer_1x1 hooks (sequitr/networks/unet.py:326-333) and
 * weighted_conv2d / to_image / from_image (sequitr/networks/gan.py:61-125).
 * KxK (K = 1 or 3) SAME stride-1 convolution + bias + activation.
 *   x (N,H,W,Cin)  w (K,K,Cin,Cout)  bias (Cout) or NULL  y (N,H,W,Cout)
 * wscale: runtime equalised-LR scale, w' = fl(w*wscale) (gan.py:75-79); 1.0f for the U-Net.
 * Supported: Cin in 1..8 or Cin % 16 == 0; Cout % 4 == 0, or Cout <= 7 with K == 1 and Cin % 4 == 0.
 */
int sq_conv2d_nhwc_fwd_f32(const float *x, const float *w, const float *bias, float *y,
                           int N, int H, int W, int Cin, int Cout, int K,
                           float wscale, int act, void *stream);

/*
 * max_pool_layer / pool_layer hook (sequitr/networks/unet.py:242,340-342):
 * 2x2 stride-2 VALID max pooling.  x (N,H,W,C) -> y (N,H/2,W/2,C); H,W even, C % 4 == 0.
 */
int sq_maxpool2x2_fwd_f32(const float *x, float *y, int N, int H, int W, int C, void *stream);

/* tf.layers.average_pooling2d(2,2) in the discriminator (sequitr/networks/gan.py:189-192). */
int sq_avgpool2x2_fwd_f32(const float *x, float *y, int N, int H, int W, int C, void *stream);

/*
 * conv_transpose_layer hook (sequitr/networks/unet.py:336-338) fused with the
 * bridge of up_layer (unet.py:312-319): 2x2 stride-2 transpose convolution +
 * bias, then bridge(upscale, skip).
 *   x (N,H,W,Cin)  w (2,2,Cout,Cin)  bias (Cout) or NULL
 *   skip (N,2H,2W,Cout) or NULL when bridge == SQ_BRIDGE_NONE   y (N,2H,2W,Cout)
 * Cin % 16 == 0, Cout % 4 == 0.
 */
int sq_convT2x2s2_nhwc_fwd_f32(const float *x, const float *w, const float *bias,
                               const float *skip, float *y, int N, int H, int W,
                               int Cin, int Cout, int bridge, void *stream);

/* UNet.bridge as a stand-alone op (sequitr/networks/unet.py:190-200), y = a (op) b, n elements, n % 4 == 0. */
int sq_bridge_fwd_f32(const float *a, const float *b, float *y, int64_t n, int bridge, void *stream);

/*
 * to_image head of UNet.build (sequitr/networks/unet.py:252-253) fused with the
 * prediction argmax: 1x1 conv Cin -> Cout (Cout <= 7) writes f32 logits and the
 * uint8 class mask (ties -> lowest index) in one pass.  mask may be NULL.
 */
int sq_conv1x1_argmax_fwd_f32(const float *x, const float *w, const float *bias,
                              float *logits, uint8_t *mask, int N, int H, int W,
                              int Cin, int Cout, void *stream);

/* argmax over the channel axis of (npix, C) f32 logits -> uint8, ties -> lowest index. */
int sq_argmax_u8(const float *logits, uint8_t *mask, int64_t npix, int C, void *stream);

/* pixel_norm (sequitr/networks/gan.py:49-51): y = x * rsqrt(mean_c(x^2) + eps). C % 4 == 0. */
int sq_pixelnorm_fwd_f32(const float *x, float *y, int64_t npix, int C, float eps, void *stream);

/* double_size (sequitr/networks/gan.py:133-136): nearest-neighbour 2x up-sampling. C % 4 == 0. */
int sq_upsample_nn2x_f32(const float *x, float *y, int N, int H, int W, int C, void *stream);

/*
 * Weighted softmax cross-entropy, forward + backward in one pass (SURVEY.md A.3;
 * tensor contract sequitr/networks/unet.py:395-401).
 *   logits (npix,C) f32, onehot (npix,C) u8, weights (npix) f32
 *   partials: workspace of sq_wsoftmax_ce_partials(npix) doubles (block sums, fixed order)
 *   loss: 1 double on the device = sum(partials)/npix, written by a second tiny kernel
 *   dlogits (npix,C) f32 or NULL:  w_p (softmax_c * sum(y) - y_c) * grad_scale / npix
 */
int64_t sq_wsoftmax_ce_partials(int64_t npix);
int sq_wsoftmax_ce_fwd_bwd_f32(const float *logits, const uint8_t *onehot, const float *weights,
                               int64_t npix, int C, float grad_scale, double *partials,
                               double *loss, float *dlogits, void *stream);

/* ------------------------------------------------------------------------------------------
 * Training side.  The reference trains through TensorFlow's automatic differentiation of the
 * same graph (sequitr/networks/gan.py:721,740-751; the U-Net model_fn is absent);
 * these entry points are the hand-written gradients of the forward operators above.
 * Gradient reductions use fixed-order two-stage sums (no float atomics): reproducible.
 * ---------------------------------------------------------------------------------------- */

/* dgrad filter: wt[ky][kx][co][ci] = w[K-1-ky][K-1-kx][ci][co]; then
 * dX = sq_conv2d_nhwc_fwd_f32(dY, wt, NULL, ..., Cin := Cout, Cout := Cin, act NONE). */
int sq_conv_weight_transform_f32(const float *w, float *wt, int K, int Cin, int Cout, void *stream);

/* dW (K,K,Cin,Cout) and db (Cout, may be NULL) of the KxK SAME convolution from X (N,H,W,Cin)
 * and dY (N,H,W,Cout).  Cin in 1..7 (K=3 only), 8, or Cin % 16 == 0; Cout % 4 == 0.
 * workspace: sq_conv2d_nhwc_wgrad_workspace_f32(...) bytes (returns -1 for unsupported shapes). */
int64_t sq_conv2d_nhwc_wgrad_workspace_f32(int N, int H, int W, int Cin, int Cout, int K);
int sq_conv2d_nhwc_wgrad_f32(const float *x, const float *dy, float *dw, float *db, float *workspace,
                             int N, int H, int W, int Cin, int Cout, int K, void *stream);

/* d(pre-activation) = dY * act'(.), decided from the activation OUTPUT y (y > 0 <=> pre > 0). */
int sq_act_bwd_f32(const float *dy, const float *y, float *dx, int64_t n, int act, void *stream);

/* max-pool backward: gradient to the first maximum in raster order; x (N,H,W,C) is the pool input. */
int sq_maxpool2x2_bwd_f32(const float *x, const float *dy, float *dx, int N, int H, int W, int C, void *stream);

/* dst (N,H,W,C) = scale * src (N,H/2,W/2,C) broadcast over 2x2: avg-pool backward (scale 0.25)
 * and double_size forward (scale 1).  sq_sumpool2x2: y = scale * sum of each 2x2 patch (double_size
 * backward).  Any C (2-channel images take a scalar path). */
int sq_broadcast2x2_f32(const float *src, float *dst, int N, int H, int W, int C, float scale, void *stream);
int sq_sumpool2x2_f32(const float *x, float *y, int N, int H, int W, int C, float scale, void *stream);

/* bridge backward: (da, db) from dY and the forward operands a (up-scaled) and b (skip). */
int sq_bridge_bwd_f32(const float *dy, const float *a, const float *b, float *da, float *db, int64_t n,
                      int bridge, void *stream);

/* g[n,i,j,(2a+b)*C+c] = dy[n,2i+a,2j+b,c]: makes the 2x2/s2 transpose-conv backward two 1x1 convs. */
int sq_space_to_depth2_f32(const float *dy, float *g, int N, int H, int W, int C, void *stream);

/* to_image head backward (1x1 conv, Cin in {8,16,32}, Cout <= 4): dx (may be NULL), dw (Cin,Cout), db. */
int64_t sq_conv1x1_small_bwd_workspace_f32(int64_t npix, int Cin, int Cout);
int sq_conv1x1_small_bwd_f32(const float *x, const float *w, const float *dz, float *dx, float *dw, float *db,
                             float *workspace, int64_t npix, int Cin, int Cout, void *stream);

/* tf.layers.dropout (sequitr/networks/unet.py:274-276): y = x * keep / (1 - rate).  mask (u8, n) is
 * written from a counter-based hash of (seed, index), or read when mask_given != 0.  step_dev (may be
 * NULL): device int32 whose value is folded into the seed, so a captured hipGraph draws a new mask on
 * every replay. */
int sq_dropout_fwd_f32(const float *x, float *y, uint8_t *mask, int64_t n, float rate, uint32_t seed,
                       int mask_given, const int32_t *step_dev, void *stream);
int sq_dropout_bwd_f32(const float *dy, const uint8_t *mask, float *dx, int64_t n, float rate, void *stream);

/* tf.train.AdamOptimizer update over a flat parameter buffer (sequitr/networks/gan.py:736-751):
 * g is first multiplied by grad_scale (1/world for data-parallel averaging); step counts from 1. */
int sq_adam_step_f32(float *p, const float *g, float *m, float *v, int64_t n, float lr, float beta1,
                     float beta2, float eps, int step, float grad_scale, void *stream);
/* y += alpha * x (flat fp32): accumulates the gradient bucket over the micro-batches of one optimiser step when a
 * rank's share of the global batch is larger than one launch batch (BASELINE config 4 on fewer than 8 GPUs). */
int sq_axpy_f32(float *y, const float *x, float alpha, int64_t n, void *stream);
/* hipGraph-safe form: state = 2 x int32 in device memory {step counter, lr_t bits}; the call increments
 * the counter on the device, so a captured step replays with the right bias correction. */
int sq_adam_step_dev_f32(float *p, const float *g, float *m, float *v, int64_t n, float lr, float beta1,
                         float beta2, float eps, int32_t *state, float grad_scale, void *stream);
/* its two halves, for optimisers with per-variable slots (the GAN's two AdamOptimizers, gan.py:736-751): one advance
 * of {step, lr_t} per minimize(), then one apply per tensor reading it. */
int sq_adam_advance_dev(int32_t *state, float lr, float beta1, float beta2, void *stream);
/* the same advance with a linear learning-rate warm-up evaluated ON THE DEVICE from the step counter:
 * lr_t = lr * min(1, t / warmup_steps) * sqrt(1 - beta2^t) / (1 - beta1^t); warmup_steps = 0 is sq_adam_advance_dev.
 * A captured training step walks the schedule by itself when replayed (the U-Net trainer's default: the reference's
 * learning_rate 0.01, sequitr/utils.py:289, diverges on step 2 of the 5-level net without it -- HISTORY.md section 8). */
int sq_adam_advance_warmup_dev(int32_t *state, float lr, float beta1, float beta2, int warmup_steps, void *stream);
int sq_adam_apply_dev_f32(float *p, const float *g, float *m, float *v, int64_t n, float beta1, float beta2,
                          float eps, const int32_t *state, float grad_scale, void *stream);
/* the same apply over a list of tensors in ONE launch: table (device memory, 8-byte aligned) = n_entries x
 * {p, g, m, v, element count, first chunk} as 64-bit words, chunks of sq_adam_multi_chunk() elements numbered across
 * the entries, total_chunks of them; element for element the update of sq_adam_apply_dev_f32 */
int sq_adam_multi_chunk(void);
int sq_adam_apply_multi_dev_f32(const void *table, int n_entries, int64_t total_chunks, float beta1, float beta2, float eps,
                                const int32_t *state, float grad_scale, void *stream);

/* conv_transpose_layer with a 3x3 kernel (SURVEY.md A.1 `up_kernel` = (3,3); hook at
 * sequitr/networks/unet.py:336-338): TF's conv2d_transpose(k=3, s=2, SAME) equals a SAME 3x3
 * convolution (sq_conv2d_nhwc_fwd_f32, filter = sq_conv_weight_transform_f32 of the TF (3,3,Cout,Cin)
 * kernel) of the zero-inserted input u[n,2i+1,2j+1,:] = x[n,i,j,:].  H, W are the SMALL side; C % 4 == 0.
 * sq_gather_odd2x_f32 is the adjoint (dx[n,i,j,:] = du[n,2i+1,2j+1,:]). */
int sq_zero_insert2x_f32(const float *x, float *u, int N, int H, int W, int C, void *stream);
int sq_gather_odd2x_f32(const float *du, float *dx, int N, int H, int W, int C, void *stream);

/* ------------------------------------------------------------------------------------------
 * Batch normalisation between a convolution and its activation: the optional `batch_norm` of the
 * conv_layer hook (sequitr/networks/unet.py:326-328 leaves the layer abstract; SURVEY.md A.1 pins
 * tf.layers.batch_normalization defaults: epsilon 1e-3, momentum 0.99).  x (npix, C) NHWC-flat,
 * C % 4 == 0, C <= 1024.  workspace: sq_bn_workspace_f32 bytes, 8-byte aligned.
 * ---------------------------------------------------------------------------------------- */
int64_t sq_bn_workspace_f32(int64_t npix, int C);
/* batch mean and POPULATION variance per channel (tf.nn.moments); fp64 fixed-order accumulation */
int sq_bn_stats_f32(const float *x, float *mean, float *var, void *workspace, int64_t npix, int C, void *stream);
/* scale = gamma / sqrtf(var + eps), shift = fmaf(-mean, scale, beta)  (batch or moving statistics) */
int sq_bn_fold_f32(const float *gamma, const float *beta, const float *mean, const float *var, float eps,
                   float *scale, float *shift, int C, void *stream);
/* moving -= (moving - batch) * (1 - momentum); the variance enters with Bessel's correction
 * npix/(npix-1) as TF's fused kernel does */
int sq_bn_update_moving_f32(float *moving_mean, float *moving_var, const float *mean, const float *var,
                            float momentum, int64_t npix, int C, void *stream);
/* y = act(fmaf(x, scale[c], shift[c])) */
int sq_bn_apply_f32(const float *x, const float *scale, const float *shift, float *y, int64_t npix, int C, int act,
                    void *stream);
/* gradients of y = act(BN(x)) given dy: the activation is differentiated through y_act (= y; NULL when
 * act == SQ_ACT_NONE); dx (npix,C), dgamma (C), dbeta (C). */
int sq_bn_bwd_f32(const float *x, const float *dy, const float *y_act, int act, const float *mean, const float *var,
                  const float *gamma, float eps, float *dx, float *dgamma, float *dbeta, void *workspace,
                  int64_t npix, int C, void *stream);
/* The same three passes on bf16 tensors (`batch_norm` in the bf16 training graph, BASELINE configs 3-4; hook
 * sequitr/networks/unet.py:326-328 + SURVEY.md A.1): elements are widened to f32, statistics accumulate in f64, parameters
 * and gradients of gamma / beta stay f32, y / dx are rounded to bf16 once.  fold / update_moving are the f32 entry points;
 * workspace as sq_bn_workspace_f32. */
int sq_bn_stats_bf16(const void *x, float *mean, float *var, void *workspace, int64_t npix, int C, void *stream);
int sq_bn_apply_bf16(const void *x, const float *scale, const float *shift, void *y, int64_t npix, int C, int act,
                     void *stream);
int sq_bn_bwd_bf16(const void *x, const void *dy, const void *y_act, int act, const float *mean, const float *var,
                   const float *gamma, float eps, void *dx, float *dgamma, float *dbeta, void *workspace, int64_t npix,
                   int C, void *stream);

/* ------------------------------------------------------------------------------------------
 * GAN side (sequitr/networks/gan.py).  weighted_conv2d / to_image / from_image are
 * sq_conv2d_nhwc_fwd_f32 with wscale + act; the entries below are the remaining leaf ops, their
 * gradients and the second-order pieces the WGAN-GP penalty needs (gan.py:719-729).
 * ---------------------------------------------------------------------------------------- */

/* pixel_norm gradients (gan.py:49-51): bwd: dx from (x, dy); bwd2: with v = dL/d(dx) returns
 * dg = dL/d(dy) and dx2 = dL/dx through the backward expression. */
int sq_pixelnorm_bwd_f32(const float *x, const float *dy, float *dx, int64_t npix, int C, float eps, void *stream);
int sq_pixelnorm_bwd2_f32(const float *x, const float *g, const float *v, float *dg, float *dx2, int64_t npix,
                          int C, float eps, void *stream);
/* pixel_norm backward + the backward of the activation that produced x (conv -> leaky -> pixel_norm, gan.py:90-98) in
 * one pass: dx = act'(x) * pixelnorm_bwd(x, dy) */
int sq_pixelnorm_bwd_act_f32(const float *x, const float *dy, float *dx, int64_t npix, int C, float eps, int act,
                             void *stream);
/* avg-pool backward (scale * 2x nearest up-sampling of src (N,H/2,W/2,C)) + the backward of the activation whose
 * output is `gate` (N,H,W,C): the discriminator block's conv2 -> leaky -> avg-pool tail (gan.py:171-192) */
int sq_broadcast2x2_act_bwd_f32(const float *src, const float *gate, float *dst, int N, int H, int W, int C, float scale,
                                int act, void *stream);

/* half_size / any tf.image.resize_nearest_neighbor(align_corners=True) (gan.py:128-136). */
int sq_resize_nearest_f32(const float *x, float *y, int N, int Hi, int Wi, int Ho, int Wo, int C, void *stream);

/* y = alpha*a + (1-alpha)*b: fade-in (gan.py:687-694, scalar alpha) and the real/fake interpolation
 * (gan.py:709-714, alpha_per_sample (N) != NULL, per_sample = elements per sample). */
int sq_lerp_f32(const float *a, const float *b, float *y, int64_t n, int64_t per_sample, float alpha,
                const float *alpha_per_sample, void *stream);
/* y = k*x, k = s or s_per_sample[n] (or 1-k when one_minus): the gradients of sq_lerp_f32. */
int sq_scale_f32(const float *x, float *y, int64_t n, int64_t per_sample, float s, const float *s_per_sample,
                 int one_minus, void *stream);

/* stand-alone activation (k_leaky_relu_alpha, gan.py:44-46) */
int sq_act_fwd_f32(const float *x, float *y, int64_t n, int act, void *stream);

/* out[n] = sum_i a[n,i]*b[n,i] (b == a: squared gradient norm of gan.py:722); fixed-order two-stage sum. */
int64_t sq_dot_per_sample_workspace_f32(int N);
int sq_dot_per_sample_f32(const float *a, const float *b, float *out, float *workspace, int N, int64_t per_sample,
                          void *stream);

/* minibatch stdev scalar (gan.py:204-211): sqrt(mean over positions of the batch variance).
 * workspace: 256 floats. */
int sq_mbstd_fwd_f32(const float *x, float *out, float *workspace, int N, int64_t per_sample, void *stream);
/* The discriminator's minibatch-stdev FEATURE MAP (gan.py:204-212) up to second order (two launches per call):
 * x (groups*n, per_sample) -> y (groups*n, cells) filled with its group's statistic (cells = 16: the hard-coded
 * (N,4,4,1) map); bwd: dx from (x, dy); bwd2: with v = dL/d(dx) returns ddy = dL/d(dy) and dx2 = dL/dx.  groups = 2
 * when D(Gz) and D(X) run as one stacked pass (each minibatch its own statistic). */
int64_t sq_mbstd_map_workspace(int groups);      /* bytes of the `workspace` the three calls take */
int sq_mbstd_map_fwd_f32(const float *x, float *y, float *workspace, int groups, int n, int64_t per_sample, int cells,
                         void *stream);
int sq_mbstd_map_bwd_f32(const float *x, const float *dy, float *dx, float *workspace, int groups, int n,
                         int64_t per_sample, int cells, void *stream);
int sq_mbstd_map_bwd2_f32(const float *x, const float *dy, const float *v, float *ddy, float *dx2, float *workspace,
                          int groups, int n, int64_t per_sample, int cells, void *stream);
/* the same three with the feature tensors (x, v, dx, dx2) in bf16 storage -- f32 arithmetic, results rounded once; y, dy, ddy f32 */
int sq_mbstd_map_fwd_bf16(const void *x, float *y, float *workspace, int groups, int n, int64_t per_sample, int cells, void *stream);
int sq_mbstd_map_bwd_bf16(const void *x, const float *dy, void *dx, float *workspace, int groups, int n, int64_t per_sample,
                          int cells, void *stream);
int sq_mbstd_map_bwd2_bf16(const void *x, const float *dy, const void *v, float *ddy, void *dx2, float *workspace, int groups,
                           int n, int64_t per_sample, int cells, void *stream);
/* WGAN-GP loss algebra of gan.py:715-729 in one launch: out2 = {d_loss, g_loss} from Dz, Dx (N) and gn2 (N) = squared
 * norm of dD(mix)/dmix per sample (one-sided penalty, lambda 10, eps drift 0.001 Dx^2); Dx = gn2 = NULL: g_loss only.
 * bwd: g_dloss / g_gloss = upstream gradients (device scalars, NULL = 0) -> dDz, dDx, dgn2. */
int sq_wgan_losses_fwd_f32(const float *Dz, const float *Dx, const float *gn2, float *out2, int N, void *stream);
int sq_wgan_losses_bwd_f32(const float *Dz, const float *Dx, const float *gn2, const float *g_dloss, const float *g_gloss,
                           float *dDz, float *dDx, float *dgn2, int N, void *stream);

/* Small-image batches (the 4x4 / 8x8 levels of generator_network / discriminator_network,
 * gan.py:246-316,149-240) as ONE image of R x Cc cells of pitch (H+1, W+1): m (1, R*(H+1), Cc*(W+1), C),
 * image n at cell (n / Cc, n % Cc), a zero row and column after every image = the SAME padding.  A KxK
 * (K <= 3) SAME convolution of m equals the per-image convolutions on the image cells, bit for bit. */
int sq_mosaic_pack_f32(const float *x, float *m, int N, int H, int W, int C, int R, int Cc, void *stream);
int sq_mosaic_unpack_f32(const float *m, float *y, int N, int H, int W, int C, int R, int Cc, void *stream);

/* tf.layers.dense with a long reduction and few rows (discriminator_network, gan.py:226-237: 8208 -> 512 on
 * 32..96 samples): y (M,N) = act(x (M,K) @ fl(w (K,N) * wscale) + bias), the reduction split into 64-wide
 * slices over thread blocks and the slices added in order (deterministic; not the convolution's single
 * chain).  K % 4 == 0; workspace sq_dense_workspace_f32 bytes. */
int64_t sq_dense_workspace_f32(int M, int K, int N);
int sq_dense_fwd_f32(const float *x, const float *w, const float *bias, float *y, float *workspace, int M, int K, int N,
                     float wscale, int act, void *stream);
/* its weight gradient: dW (K,N) = scale * x^T dY, db (N) or NULL = column sums of dY; x (M,K), dY (M,N), M <= 128 rows.
 * accumulate: bit 0 -- dW is added to the contents of dw, bit 1 -- db to those of db. */
int sq_dense_wgrad_f32(const float *x, const float *dy, float *dw, float *db, int M, int K, int N, float scale, int accumulate,
                       void *stream);

/* M (Ca,Cb) = sum_p a[p,:]^T b[p,:] with Ca <= 7, Cb % 4 == 0: weight gradient of to_image / from_image. */
int64_t sq_wgrad1x1_small_workspace_f32(int64_t npix, int Ca, int Cb);
int sq_wgrad1x1_small_f32(const float *a, const float *b, float *m, float *workspace, int64_t npix, int Ca, int Cb,
                          void *stream);

/* ------------------------------------------------------------------------------------------
 * Fused inference variants of the 3x3 convolution (bit-identical to the unfused sequence).
 * ---------------------------------------------------------------------------------------- */

/* conv_layer on the concat bridge (sequitr/networks/unet.py:196-197, 321): y = act(conv(concat([xa, xb], -1)) + bias);
 * xa, xb (N,H,W,Ca) each (the up-scaled tensor first, the skip tensor second), w (K,K,2*Ca,Cout); the concatenated
 * tensor is never written.  Bit-identical to sq_conv2d_nhwc_fwd_f32 on the materialised concat. */
int sq_conv2d_concat_nhwc_fwd_f32(const float *xa, const float *xb, const float *w, const float *bias, float *y, int N,
                                  int H, int W, int Ca, int Cout, int K, int act, void *stream);

/* conv_block tail + max_pool_layer (sequitr/networks/unet.py:241-243,265-277): 3x3 conv + bias + act
 * that writes y (N,H,W,Cout) AND its 2x2 max-pool (N,H/2,W/2,Cout) from the same accumulators. */
int sq_conv3x3_pool_fwd_f32(const float *x, const float *w, const float *bias, float *y, float *pooled,
                            int N, int H, int W, int Cin, int Cout, int act, void *stream);

/* last conv_layer of up0 + conv_layer_1x1 + prediction (unet.py:252-253,321): 3x3 conv Cin -> 16
 * + bias + act, then the 1x1 head (16 -> head_c <= 4, HWIO head_w (1,1,16,head_c)) and the argmax
 * mask in the epilogue; the 16-channel activation never reaches HBM.  mask may be NULL. */
int sq_conv3x3_head_fwd_f32(const float *x, const float *w, const float *bias, const float *head_w,
                            const float *head_b, float *logits, uint8_t *mask, int N, int H, int W,
                            int Cin, int head_c, int act, void *stream);

/* conv_block of down0 for a 1-channel input (unet.py:238): conv1 (3x3, 1 -> 16, bias, ReLU) is
 * evaluated in LDS, conv2 (3x3, 16 -> 16, bias, ReLU) on the matrix cores; writes y (N,H,W,16) and,
 * when pooled != NULL, its 2x2 max-pool. */
int sq_conv3x3_first_block_fwd_f32(const float *x, const float *w1, const float *b1, const float *w2,
                                   const float *b2, float *y, float *pooled, int N, int H, int W,
                                   void *stream);

/* conv_transpose_layer + bridge + first conv_layer of up0 (unet.py:299-322) for the level-0 shape: the
 * transpose conv (32 -> 16 channels) and the bridge are evaluated for each tile's halo inside the 3x3
 * convolution's staging; the up-scaled / merged tensor never reaches HBM.
 *   x_low (N,H/2,W/2,32); wt (2,2,16,32) TF layout, bt (16) or NULL; skip (N,H,W,16); bridge SQ_BRIDGE_*;
 *   w (3,3,16,16), bias (16) or NULL; y (N,H,W,16) = act(conv3x3(bridge(convT(x_low) + bt, skip)) + bias).
 * Same bits as sq_convT2x2s2_nhwc_fwd_f32 followed by sq_conv2d_nhwc_fwd_f32. */
int sq_convT_conv3x3_fwd_f32(const float *x_low, const float *wt, const float *bt, const float *skip, int bridge,
                             const float *w, const float *bias, float *y, int N, int H, int W, int act, void *stream);

/* ------------------------------------------------------------------------------------------
 * bf16 path (BASELINE configs 3-5: bf16 compute, fp32 master weights and accumulation).
 * bf16 tensors are passed as void* (2-byte elements, NHWC); biases / logits / weight grads stay f32.
 * ---------------------------------------------------------------------------------------- */

/* Packed bf16 filter of the KxK conv Cin -> Cout: [Cin/KC][Cout][KP] with k = tap*KC + c, KC = 32 when
 * Cin % 32 == 0, 16 when Cin % 16 == 0, else 8 (Cin % 8 == 0); KP = K*K*KC rounded up to 32.
 * sq_conv_packed_weights_elems_bf16 gives the element count (-1: unsupported).  transform != 0 packs the dgrad filter of the forward conv
 * Cout -> Cin whose f32 HWIO weights (K,K,Cout,Cin) are passed in `w`. */
int64_t sq_conv_packed_weights_elems_bf16(int K, int Cin, int Cout);
int sq_conv_pack_weights_bf16(const float *w, void *wp, int K, int Cin, int Cout, float wscale, int transform,
                              void *stream);
/* Every pack (and plain f32 -> bf16 cast) of one optimiser step in one launch.  base: the flat fp32 parameter
 * buffer; out: one bf16 buffer; table (device, n_entries x 8 int32, n_entries <= 128):
 * {src offset in floats, dst offset in bf16 elements, K, Cin, Cout of the packed conv, transform, index of the
 * entry's first item, kind (0 = pack as sq_conv_pack_weights_bf16, 1 = plain cast of K*K*Cin*Cout values)};
 * total_items = sum of the entries' item counts (packed elements, or values for kind 1). */
int sq_conv_pack_weights_multi_bf16(const float *base, void *out, const int32_t *table, int n_entries,
                                    int total_items, void *stream);
/* the same with a factor per entry (`scales`, n_entries floats on the device) applied as sq_conv_pack_weights_bf16's
 * wscale: all filter packs of a GAN solver step in one launch */
int sq_conv_pack_weights_multi_scaled_bf16(const float *base, void *out, const int32_t *table, const float *scales,
                                           int n_entries, int total_items, void *stream);

/* conv_layer / weighted_conv2d on bf16 activations: y = act(conv(x, wp) + bias), fp32 accumulate. */
int sq_conv2d_nhwc_fwd_bf16(const void *x, const void *wp, const float *bias, void *y, int N, int H, int W,
                            int Cin, int Cout, int K, int act, void *stream);

/* "Mixed" convolution behind an f32 graph (the GAN of BASELINE config 5; weighted_conv2d, gan.py:61-99): f32
 * activations in and out, both operands rounded to bf16 (RNE) on the way into LDS, f32 accumulation on
 * v_mfma_f32_16x16x32_bf16.  wp = sq_conv_pack_weights_bf16(w, K, Cin, Cout, wscale, transform) (Cin % 8 == 0);
 * Cout % 4 == 0.  The drop-in for sq_conv2d_nhwc_fwd_f32 where the bf16 matrix rate is wanted. */
int sq_conv2d_nhwc_fwd_mixed_f32(const float *x, const void *wp, const float *bias, float *y, int N, int H, int W,
                                 int Cin, int Cout, int K, int act, void *stream);
/* dgrad of such a conv whose INPUT was the output `gate` (N,H,W,Cout) of a ReLU / leaky-ReLU (act): the result leaves through
 * that activation's backward in the epilogue, dx = gate > 0 ? v : v * slope -- the sq_act_bwd_f32 pass that followed the
 * dgrad in the GAN's first-order backward passes (gan.py:90-98 chains), same bits */
int sq_conv2d_nhwc_dgrad_actgate_mixed_f32(const float *dy, const void *wp_t, const float *gate, int act, float *dx, int N,
                                           int H, int W, int Cin, int Cout, int K, void *stream);
/* both forms on a batch of small images (Nimg, h, w, C) convolved as ONE mosaic image of R x Cc cells (sq_mosaic_pack_f32's
 * layout, 3x3 only) without building it: loads, gate and stores address the compact tensors.  gate == NULL: forward with
 * bias / act; gate != NULL: the act-gated dgrad (`act` = the gate's activation).  Same bits as pack -> conv -> unpack. */
int sq_conv2d_nhwc_mixed_mosaic_f32(const float *x, const void *wp, const float *bias, const float *gate, float *y, int Nimg,
                                    int h, int w, int Cin, int Cout, int act, int R, int Cc, void *stream);
/* its weight gradient: dW (K,K,Cin,Cout) f32, db (Cout) f32 or NULL from f32 X and f32 dY (rounded to bf16 in
 * LDS); Cin % 16 == 0, Cout % 16 == 0. */
int64_t sq_conv2d_nhwc_wgrad_workspace_mixed_f32(int N, int H, int W, int Cin, int Cout, int K);
int sq_conv2d_nhwc_wgrad_mixed_f32(const float *x, const float *dy, float *dw, float *db, float *workspace, int N,
                                   int H, int W, int Cin, int Cout, int K, void *stream);
/* the same with dW (not db) multiplied by dw_scale in the finish kernel: the gradient of an equalised-learning-rate
 * kernel is wscale * raw dW (gan.py:75-79); saves the scalar-multiply pass over every weight gradient */
int sq_conv2d_nhwc_wgrad_scaled_mixed_f32(const float *x, const float *dy, float *dw, float *db, float *workspace, int N,
                                          int H, int W, int Cin, int Cout, int K, float dw_scale, void *stream);
/* ... on a batch of small images (Nimg, h, w, C) taken as one mosaic image of R x Cc cells without building the mosaics
 * (see sq_conv2d_nhwc_mixed_mosaic_f32); workspace as for (1, R*(h+1), Cc*(w+1), Cin, Cout, K = 3) */
int sq_conv2d_nhwc_wgrad_mixed_mosaic_f32(const float *x, const float *dy, float *dw, float *db, float *workspace, int Nimg,
                                          int h, int w, int Cin, int Cout, int R, int Cc, float dw_scale, void *stream);
int sq_conv2d_nhwc_wgrad_scaled_f32(const float *x, const float *dy, float *dw, float *db, float *workspace, int N, int H,
                                    int W, int Cin, int Cout, int K, float dw_scale, void *stream);

/* first conv of down0 in the bf16 graph: f32 (N,H,W,Cin) image, Cin 1..7 -> bf16 (N,H,W,Cout), 3x3, f32 HWIO
 * weights (`num_inputs`, unet.py:131). */
int sq_conv3x3_first_fwd_bf16(const float *x, const float *w, const float *bias, void *y, int N, int H, int W,
                              int Cin, int Cout, int act, void *stream);
/* Sign masks of ReLU outputs: one bit per element in NHWC order (bit c & 7 of byte (pixel * Cout + c) >> 3, Cout % 16 == 0,
 * N*H*W*Cout/8 bytes).  The forward convs of a conv_block's conv1 (unet.py:265-270) write the mask beside their output;
 * the dgrad of conv2, which lets gradient through where conv1's ReLU was active, reads the mask instead of the tensor
 * (1/16 of the bytes).  sq_conv2d_nhwc_dgrad_maskgate_bf16 == sq_conv2d_nhwc_dgrad_gate_bf16 on the tensor, bit for bit. */
int sq_conv3x3_first_fwd_mask_bf16(const float *x, const float *w, const float *bias, void *y, void *mask, int N, int H,
                                   int W, int Cin, int Cout, int act, void *stream);
int sq_conv2d_nhwc_fwd_mask_bf16(const void *x, const void *wp, const float *bias, void *y, void *mask, int N, int H, int W,
                                 int Cin, int Cout, int K, int act, void *stream);
int sq_conv2d_nhwc_dgrad_maskgate_bf16(const void *dy, const void *wp_t, const void *mask, float scale, void *dx, int N,
                                       int H, int W, int Cin, int Cout, int K, void *stream);

/* dW (K,K,Cin,Cout) f32 and db (Cout, may be NULL) f32 from bf16 X (N,H,W,Cin) and bf16 dY (N,H,W,Cout);
 * Cin % 16 == 0, Cout % 16 == 0.  Transposing LDS reads (ds_read_b64_tr_b16) feed the MFMA. */
int64_t sq_conv2d_nhwc_wgrad_workspace_bf16(int N, int H, int W, int Cin, int Cout, int K);
int sq_conv2d_nhwc_wgrad_bf16(const void *x, const void *dy, float *dw, float *db, float *workspace, int N,
                              int H, int W, int Cin, int Cout, int K, void *stream);
/* parameter gradients of the 2x2/s2 transpose conv from x (N,H,W,Cin) bf16 and the output gradient in space-to-depth
 * form g (N,H,W,4*Cout) bf16: dW (2,2,Cout,Cin) f32, db (Cout) f32 or NULL; workspace as for the 1x1 wgrad Cin -> 4*Cout */
int sq_convT2x2s2_wgrad_bf16(const void *x, const void *g, float *dw, float *db, float *workspace, int N, int H, int W,
                             int Cin, int Cout, void *stream);

/* The concatenation in front of the discriminator's dense head (gan.py:213-226: tf.concat([conv, minibatch_stdev], -1) then
 * reshape) with bf16 features: flat f32 (npix, C + 1) = [float(conv (npix, C)), mb (npix)] in one pass, and its adjoint
 * (dconv rounded to bf16, dmb f32) -- each is the other's derivative, so the WGAN-GP second-order pass stays closed. */
int sq_head_concat_fwd_bf16(const void *conv, const float *mb, float *flat, int64_t npix, int C, void *stream);
int sq_head_concat_bwd_bf16(const float *dflat, void *dconv, float *dmb, int64_t npix, int C, void *stream);

/* bf16 <-> f32 casts (RNE), n % 4 == 0 */
int sq_cast_f32_to_bf16(const float *x, void *y, int64_t n, void *stream);
int sq_cast_bf16_to_f32(const void *x, float *y, int64_t n, void *stream);

/* bf16 variants of the streaming ops (C % 8 == 0 / n % 8 == 0): same semantics as the f32 entries */
int sq_maxpool2x2_fwd_bf16(const void *x, void *y, int N, int H, int W, int C, void *stream);
int sq_maxpool2x2_bwd_bf16(const void *x, const void *dy, void *dx, int N, int H, int W, int C, void *stream);
/* decoder junction backward in one pass (merged = bridge(up, skip), unet.py:312-319): g (N,H,W,4C) = d_up in the
 * space-to-depth layout sq_conv2d_nhwc_wgrad_bf16 / the 1x1 dgrad of the transpose conv consume, dskip
 * (N,2H,2W,C) = gradient of the skip operand.  H, W = the LOW-resolution side; up / skip needed for eltwise_mul. */
int sq_bridge_bwd_s2d_bf16(const void *dy, const void *up, const void *skip, void *g, void *dskip, int N, int H, int W,
                           int C, int bridge, void *stream);
/* max-pool backward + the other gradient of the pooled tensor (the U-Net skip path): dx = scatter(dy) + add */
int sq_maxpool2x2_bwd_add_bf16(const void *x, const void *dy, const void *add, void *dx, int N, int H, int W, int C,
                               void *stream);
/* ... and through the gate of the conv block that produced x = dropout(ReLU(.)) (unet.py:265-277): dx = x > 0 ?
 * (scatter(dy) + add) * gate_scale : 0, gate_scale = 1 / (1 - rate); 0 = no gate.  Replaces a stand-alone
 * sq_relu_scale_bwd_bf16 pass bit for bit (x is already read for the arg-max). */
int sq_maxpool2x2_bwd_add_gate_bf16(const void *x, const void *dy, const void *add, void *dx, int N, int H, int W, int C,
                                    float gate_scale, void *stream);
int sq_act_bwd_bf16(const void *dy, const void *y, void *dx, int64_t n, int act, void *stream);
/* dropout backward + the backward of the activation in front of it, one pass:
 * dx = act'(y) * (mask ? dy / (1 - rate) : 0), y = the activation output that entered the dropout */
int sq_act_dropout_bwd_bf16(const void *dy, const uint8_t *mask, const void *y, void *dx, int64_t n, float rate,
                            int act, void *stream);
/* conv + bias + act + dropout in one kernel: the mask is the counter hash of sq_dropout_fwd_bf16 over the flat
 * NHWC element index (same seed / step_dev semantics, same two roundings => same bits as the two kernels);
 * no mask tensor is written.  sq_relu_scale_bwd_bf16 is the matching backward for act == ReLU:
 * dx = y > 0 ? dy * scale : 0 with scale = 1 / (1 - rate)  (y > 0 <=> kept and active). */
int sq_conv2d_nhwc_fwd_dropout_bf16(const void *x, const void *wp, const float *bias, void *y, int N, int H, int W,
                                    int Cin, int Cout, int K, int act, float rate, uint32_t seed,
                                    const int32_t *step_dev, void *stream);
/* ... and the 2x2/s2 max pool of the result beside it (ypool (N,H/2,W/2,Cout); K = 3, H and W even; rate 0 = no dropout):
 * conv_block -> max_pool of an encoder level (unet.py:241-243, 265-277) in one kernel; ypool = sq_maxpool2x2_fwd_bf16(y) */
int sq_conv2d_nhwc_fwd_dropout_pool_bf16(const void *x, const void *wp, const float *bias, void *y, void *ypool, int N, int H,
                                         int W, int Cin, int Cout, int K, int act, float rate, uint32_t seed,
                                         const int32_t *step_dev, void *stream);
/* conv_block of down0 + max_pool_layer (unet.py:238-243, 265-277), training form, ONE launch for a single-channel f32 image
 * and 16 filters: y1 = relu(conv3x3(x, w1) + b1) and its sign mask mask1 (sq_conv3x3_first_fwd_mask_bf16's outputs, bit for
 * bit) are made per tile and never read back; y, ypool as sq_conv2d_nhwc_fwd_dropout_pool_bf16(y1, wp2, b2, relu). */
int sq_conv3x3_first_block_dropout_pool_bf16(const float *x, const float *w1, const float *b1, void *y1, void *mask1,
                                             const void *wp2, const float *b2, void *y, void *ypool, int N, int H, int W,
                                             float rate, uint32_t seed, const int32_t *step_dev, void *stream);
int sq_relu_scale_bwd_bf16(const void *dy, const void *y, void *dx, int64_t n, float scale, void *stream);
/* dX of a convolution whose input was the ReLU output `gate` (same shape as dx): sq_conv2d_nhwc_fwd_bf16 of dy
 * with the transposed packed filter, passed only where gate > 0 (the upstream ReLU backward fused in) */
int sq_conv2d_nhwc_dgrad_relu_bf16(const void *dy, const void *wp_t, const void *gate, void *dx, int N, int H, int W,
                                   int Cin, int Cout, int K, void *stream);
/* the same with a factor on what passes (backward of dropout(ReLU(.)) = `gate`): gate > 0 ? dgrad * gate_scale : 0 */
int sq_conv2d_nhwc_dgrad_gate_bf16(const void *dy, const void *wp_t, const void *gate, float gate_scale, void *dx, int N,
                                   int H, int W, int Cin, int Cout, int K, void *stream);
/* dgrad of a decoder block's first conv with the junction backward (merged = bridge(up, skip), unet.py:312-319) in
 * its epilogue: writes g (N,H/2,W/2,4*Cout) = d_up in the space-to-depth layout and dskip (N,H,W,Cout); d(merged) is
 * never stored.  Same bits as the dgrad followed by sq_bridge_bwd_s2d_bf16. */
int sq_conv2d_nhwc_dgrad_junction_bf16(const void *dy, const void *wp_t, const void *up, const void *skip, void *g,
                                       void *dskip, int N, int H, int W, int Cin, int Cout, int K, int bridge, void *stream);
int sq_bridge_fwd_bf16(const void *a, const void *b, void *y, int64_t n, int bridge, void *stream);
int sq_bridge_bwd_bf16(const void *dy, const void *a, const void *b, void *da, void *db, int64_t n, int bridge,
                       void *stream);
int sq_dropout_fwd_bf16(const void *x, void *y, uint8_t *mask, int64_t n, float rate, uint32_t seed,
                        int mask_given, const int32_t *step_dev, void *stream);
int sq_dropout_bwd_bf16(const void *dy, const uint8_t *mask, void *dx, int64_t n, float rate, void *stream);

/* conv_transpose_layer + bridge on bf16 tensors; w = bf16 copy of the (2,2,Cout,Cin) kernel, bias f32.
 * Cin % 32 == 0, Cout % 16 == 0.  The up-scaled value is rounded to bf16 before the bridge. */
int sq_convT2x2s2_nhwc_fwd_bf16(const void *x, const void *w, const float *bias, const void *skip, void *y,
                                int N, int H, int W, int Cin, int Cout, int bridge, void *stream);
/* training form of the decoder junction (unet.py:312-319): one pass writes the up-scaled tensor `up` (kept for the
 * bridge backward) AND merged = bridge(up, skip); same bits as sq_convT2x2s2_nhwc_fwd_bf16 + sq_bridge_fwd_bf16. */
int sq_convT2x2s2_bridge_both_fwd_bf16(const void *x, const void *w, const float *bias, const void *skip, void *up,
                                       void *merged, int N, int H, int W, int Cin, int Cout, int bridge, void *stream);

/* to_image head on a bf16 activation: f32 (Cin,Cout<=4) weights, f32 logits + uint8 mask (may be NULL);
 * backward: dz f32 -> dx bf16 (may be NULL), dw (Cin,Cout) f32, db f32; Cin in {16,32}, Cout <= 2. */
int sq_conv1x1_head_fwd_bf16(const void *x, const float *w, const float *bias, float *logits, uint8_t *mask,
                             int64_t npix, int Cin, int Cout, void *stream);
int64_t sq_conv1x1_head_bwd_workspace_bf16(int64_t npix, int Cin, int Cout);
int sq_conv1x1_head_bwd_bf16(const void *x, const float *w, const float *dz, void *dx, float *dw, float *db,
                             float *workspace, int64_t npix, int Cin, int Cout, void *stream);
/* gate_scale > 0: x = dropout(ReLU(.)) of the last conv block; dx leaves through that gate (x > 0 ? dx * gate_scale : 0) */
int sq_conv1x1_head_bwd_gate_bf16(const void *x, const float *w, const float *dz, void *dx, float *dw, float *db,
                                  float *workspace, int64_t npix, int Cin, int Cout, float gate_scale, void *stream);
/* to_image head + weighted softmax-CE of the training step as one forward and one backward kernel: the logits are never
 * written (sequitr/networks/unet.py:252-253 followed by the loss of SURVEY.md A.3).  forward: loss = f32 device scalar,
 * partials = sq_wsoftmax_ce_partials(npix) doubles; backward: dloss = f32 device scalar (gradient arriving at the loss),
 * workspace = sq_conv1x1_head_bwd_workspace_bf16 bytes, gate_scale as above.  Bit-identical to
 * sq_conv1x1_head_fwd_bf16 -> sq_wsoftmax_ce_fwd_bwd_f32 -> (* dloss) -> sq_conv1x1_head_bwd_gate_bf16. */
int sq_conv1x1_head_wce_fwd_bf16(const void *x, const float *w, const float *bias, const uint8_t *onehot,
                                 const float *weights, double *partials, float *loss, int64_t npix, int Cin, int Cout,
                                 void *stream);
int sq_conv1x1_head_wce_bwd_bf16(const void *x, const float *w, const float *bias, const uint8_t *onehot,
                                 const float *weights, const float *dloss, void *dx, float *dw, float *db, float *workspace,
                                 int64_t npix, int Cin, int Cout, float gate_scale, void *stream);
/* the backward pass that also leaves the LOSS (sq_conv1x1_head_wce_fwd_bf16's value, bit for bit) in `loss`: a training
 * step that runs the backward right behind the forward skips the forward kernel and its read of the level-0 activation
 * (the loss of unet.py:395-401's tensor contract is only a reported number there).  partials as in the forward. */
int sq_conv1x1_head_wce_bwd_loss_bf16(const void *x, const float *w, const float *bias, const uint8_t *onehot,
                                      const float *weights, const float *dloss, void *dx, float *dw, float *db,
                                      float *workspace, double *partials, float *loss, int64_t npix, int Cin, int Cout,
                                      float gate_scale, void *stream);

/* weight gradient of the first (Cin -> Cout, Cin 1..7) 3x3 convolution from the f32 image and a bf16 dY:
 * dW (3,3,Cin,Cout) f32, db (Cout) f32 or NULL */
int64_t sq_conv3x3_first_wgrad_workspace_bf16(int N, int H, int W, int Cin, int Cout);
int sq_conv3x3_first_wgrad_bf16(const float *x, const void *dy, float *dw, float *db, float *workspace, int N,
                                int H, int W, int Cin, int Cout, void *stream);

/* ------------------------------------------------------------------------------------------
 * Mask -> connected components -> centroids (the step after the hot path).
 * CentroidWriter.write, sequitr/utils.py:531-578: per frame, per class c > 0,
 * scipy.ndimage.label(mask == c) (4-connectivity) and center_of_mass of every label.
 *   mask (N,H,W) uint8 class labels on the device; workspace sq_mask_centroids_workspace bytes, 16-B aligned.
 *   count (device int32): number of components found (may exceed max_out: then re-run with more room).
 *   out (max_out,5) f32 rows [frame, x = row centre, y = column centre, 0, class] in NO particular order;
 *   keys (max_out) int32 = linear index of each component's first pixel in raster order: sorting rows by
 *   (frame, class, key) gives the reference's order (scipy numbers labels by first pixel).
 * ---------------------------------------------------------------------------------------- */
int64_t sq_mask_centroids_workspace(int N, int H, int W);
int sq_mask_centroids_u8(const uint8_t *mask, int N, int H, int W, void *workspace, int32_t *count, float *out,
                         int32_t *keys, int max_out, void *stream);
/* volumetric form (CentroidWriter.write on (N,Z,X,Y) input, utils.py:511-521, after its swapaxes(1,-1)):
 * mask (N,D0,D1,D2), 6-connectivity; rows [frame, x, y, z, class] = centre along (D0, D1, D2);
 * workspace: sq_mask_centroids_workspace(N*D0, D1, D2). */
int sq_volume_centroids_u8(const uint8_t *mask, int N, int D0, int D1, int D2, void *workspace, int32_t *count,
                           float *out, int32_t *keys, int max_out, void *stream);

/* ------------------------------------------------------------------------------------------
 * EDT weight maps (the step in front of the training hot path).
 * ImageWeightMap.pipe, sequitr/pipeline.py:475-479:
 *   d = distance_transform_edt(1 - image);  out = w0*(1-image)*exp(-(d*d)/(2 sigma^2 + 1e-99)) + image + 1
 * img (N,H,W) f32 binary label images (values 0 / 1; a pixel is a feature iff 1 - image == 0), one map per
 * image.  workspace: sq_weightmap_workspace bytes, 16-B aligned.  H, W < 30000.
 *   sq_edt_sq_f32        : d2 (N,H,W) int32 = EXACT squared Euclidean distance to the nearest feature
 *                          (an image with no feature reproduces scipy's artefact: distance to index (-1, 0))
 *   sq_weightmap_edt_f32 : out64 (N,H,W) f64 as the reference computes it and / or out32 (N,H,W) f32, the
 *                          `weights` tensor the loss kernel takes (sq_wsoftmax_ce_fwd_bwd_f32)
 * ---------------------------------------------------------------------------------------- */
int64_t sq_weightmap_workspace(int N, int H, int W);
int sq_edt_sq_f32(const float *img, int32_t *d2, void *workspace, int N, int H, int W, void *stream);
int sq_weightmap_edt_f32(const float *img, double *out64, float *out32, void *workspace, int N, int H, int W,
                         double w0, double sigma, void *stream);

/* ImageWeightMap2 (sequitr/pipeline.py:482-571), the per-pixel part on the device: `simplices` (nsimp,7) int32 rows
 * {tile, x0, y0, x1, y1, x2, y2} (x = row, y = column, as np.where orders them) and `longest` (nsimp) float64 = the
 * longest edge of each simplex, from scipy.spatial.Delaunay of the boundary points (host); img (N,H,W) binary f32.
 * Rasterises the simplices (point location; a pixel covered by several takes the largest value), builds the
 * pre-filter map (1024 where uncovered, 0 on foreground), applies scipy's gaussian_filter(sigma = 1) and the
 * reference's float64 expression.  out64 (N,H,W) and / or out32; workspace of sq_weightmap2_workspace bytes. */
int64_t sq_weightmap2_workspace(int N, int H, int W);
/* ImageWeightMap2's boundary points on the device (sequitr/pipeline.py:516-528: erosion outline of the label XOR the
 * outline of the label dilated three times, von Neumann element, border value 0): points (N,H,W) uint8, 1 = a vertex
 * of the triangulation.  img (N,H,W) binary f32. */
int sq_wm2_boundary_points_u8(const float *img, uint8_t *points, int N, int H, int W, void *stream);
/* HOST function (no GPU work): exact Delaunay triangulation of each tile's boundary points in place of
 * scipy.spatial.Delaunay (pipeline.py:531-537) -- integer predicates in 128-bit arithmetic, incremental insertion in scan
 * order, tiles on a small pool of host threads (SQ_HOST_THREADS, default 16).  xy = the (row, column) int32 pairs of
 * `nsets` tiles back to back, tile s = points offsets[s] .. offsets[s+1] (0 <= coordinate < 32768, distinct); writes
 * the (tile, x0, y0, x1, y1, x2, y2) rows and longest edges sq_weightmap2_delaunay_f32 takes.  A triangulation of n points
 * has < 2 n triangles: tile s owns rows 2 offsets[s] .. 2 offsets[s+1] - 1, written by the worker that triangulated it (no
 * second pass); the few rows a tile does not need are padding with tile = -1, which sq_weightmap2_delaunay_f32 skips.
 * Returns the number of rows = 2 * offsets[nsets] (cap must hold them), or a negative SQ_E* code. */
int64_t sq_delaunay2d_batch_i32(const int32_t *xy, const int64_t *offsets, int nsets, int32_t *simplices, double *longest,
                                int64_t cap);
int sq_weightmap2_delaunay_f32(const float *img, const int32_t *simplices, const double *longest, int nsimp, double *out64,
                               float *out32, void *workspace, int N, int H, int W, double w0, double sigma, void *stream);

/* ------------------------------------------------------------------------------------------
 * GAN operators on bf16 FEATURE tensors (BASELINE config 5 with bf16 storage; sequitr/networks/gan.py:44-136, 149-316):
 * activations and activation gradients are bf16 in HBM, every kernel computes in f32 and rounds once per stored value;
 * parameters, images (<= 4 channels), the discriminator's outputs and the losses stay f32.  C % 8 == 0 everywhere.
 *   pixel_norm (gan.py:49-51): fwd; bwd (act != NONE: x is that activation's output and dx also leaves through its
 *   backward, second rounding kept: == sq_pixelnorm_bwd_bf16(NONE) then sq_act_bwd_bf16, bit for bit); bwd2 as
 *   sq_pixelnorm_bwd2_f32.
 *   half_size by averaging / double_size (gan.py:133-136, 189-192): sumpool (scale 0.25 = average pool) and its adjoint
 *   broadcast (scale 1 = nearest-neighbour up-sampling), H and W are the LARGER tensor's size; *_act_bwd: the
 *   up-sampled gradient additionally passes the backward of the activation whose output is `gate`.
 *   to_image / from_image (gan.py:102-125) and their gradients: 1x1 convolutions between an f32 image side with
 *   <= 4 channels and a bf16 feature side -- smallin (image -> features), smallout (features -> image),
 *   wgrad1x1_small (m (Ca, C) = scale * sum_p a[p]^T b[p]; a == NULL: per-channel sums of b, from_image's bias gradient;
 *   asum != NULL: (Ca) per-channel sums of a from the same pass, to_image's bias gradient).
 *   Weights are f32 row-major (in, out), multiplied by wscale on the fly (equalised learning rate, gan.py:75-79).
 * ---------------------------------------------------------------------------------------- */
int sq_pixelnorm_fwd_bf16(const void *x, void *y, int64_t npix, int C, float eps, void *stream);
int sq_pixelnorm_bwd_bf16(const void *x, const void *dy, void *dx, int64_t npix, int C, float eps, int act, void *stream);
int sq_pixelnorm_bwd2_bf16(const void *x, const void *g, const void *v, void *dg, void *dx2, int64_t npix, int C, float eps,
                           void *stream);
int sq_sumpool2x2_bf16(const void *x, void *y, int N, int H, int W, int C, float scale, void *stream);
int sq_broadcast2x2_bf16(const void *src, void *dst, int N, int H, int W, int C, float scale, void *stream);
int sq_broadcast2x2_act_bwd_bf16(const void *src, const void *gate, void *dst, int N, int H, int W, int C, float scale,
                                 int act, void *stream);
int sq_act_fwd_bf16(const void *x, void *y, int64_t n, int act, void *stream);
int sq_conv1x1_smallin_fwd_bf16(const float *x, const float *w, const float *bias, void *y, int64_t npix, int Ca, int C,
                                float wscale, int act, void *stream);
int sq_conv1x1_smallout_fwd_bf16(const void *x, const float *w, const float *bias, float *y, int64_t npix, int C, int Co,
                                 float wscale, int act, void *stream);
int64_t sq_wgrad1x1_small_workspace_bf16(int64_t npix, int Ca, int C);
int sq_wgrad1x1_small_bf16(const float *a, const void *b, float *m, float *asum, float *workspace, int64_t npix, int Ca,
                           int C, float scale, void *stream);
/* weighted_conv2d (gan.py:61-99) on bf16 tensors: the forward is sq_conv2d_nhwc_fwd_bf16; these are the forms the
 * mixed (f32 tensor) GAN path has beside it -- the dgrad that leaves through the previous activation's backward
 * (== dgrad then sq_act_bwd_bf16, same two roundings), the small-image batch addressed as one mosaic (forward, or with
 * `gate` the gated dgrad; with a workspace the reduction over input channels is split over the grid where the launch would
 * otherwise be a handful of blocks, slices added in order by a finish kernel), and the weight gradient with the equalised-LR
 * factor in the finish kernel, plain or mosaic.
 * sq_conv2d_nhwc_wgrad_bf16 and the scaled form take channel counts that are multiples of 8 (8 mod 16: the ragged form). */
/* a discriminator block's second conv with the 2x2 average pool that follows it (gan.py:171-192) written from the same kernel:
 * y (N,H,W,Cout) and ypool (N,H/2,W/2,Cout) == sq_sumpool2x2_bf16(y, 0.25).  K = 3, even H and W. */
int sq_conv2d_nhwc_fwd_avgpool_bf16(const void *x, const void *wp, const float *bias, void *y, void *ypool, int N, int H, int W,
                                    int Cin, int Cout, int act, void *stream);
/* weighted_conv2d with norm=True (gan.py:86-97: conv -> bias -> activation -> pixel_norm, one op upstream) from one kernel:
 * y (N,H,W,Cout) = act(conv3x3(x, wp) + bias) and ynorm = pixel_norm(y, eps) of the STORED y (== sq_pixelnorm_fwd_bf16(y) up
 * to the f32 rounding of the per-pixel factor: the squares are added in another order).  One block holds all channels of a
 * pixel: Cout % 8 == 0, Cout <= 64 (the generator's 32x32 .. 256x256 levels).  y may be NULL (only ynorm is wanted). */
int sq_conv2d_nhwc_fwd_pixelnorm_bf16(const void *x, const void *wp, const float *bias, void *y, void *ynorm, int N, int H, int W,
                                      int Cin, int Cout, int act, float eps, void *stream);
int sq_conv2d_nhwc_dgrad_actgate_bf16(const void *dy, const void *wp_t, const void *gate, int act, void *dx, int N, int H,
                                      int W, int Cin, int Cout, int K, void *stream);
int sq_conv2d_nhwc_mosaic_bf16(const void *x, const void *wp, const float *bias, const void *gate, void *y, int Nimg, int h,
                               int w, int Cin, int Cout, int act, int R, int Cc, float *workspace, int64_t workspace_bytes,
                               void *stream);
int sq_conv2d_nhwc_wgrad_scaled_bf16(const void *x, const void *dy, float *dw, float *db, float *workspace, int N, int H,
                                     int W, int Cin, int Cout, int K, float dw_scale, void *stream);
int sq_conv2d_nhwc_wgrad_mosaic_bf16(const void *x, const void *dy, float *dw, float *db, float *workspace, int Nimg, int h,
                                     int w, int Cin, int Cout, int R, int Cc, float dw_scale, void *stream);

/* Weight gradients of SEVERAL layers in one launch (+ one finish launch) per kernel block shape: the deep layers of a
 * training step are each a ~37 us launch for ~10 us of matrix work (ramp-up and the cross-wave reduction run with the chip
 * idle, and every one of a layer's ~512 blocks ends with a cross-wave reduction); sharing one grid the layers are cut into a
 * quarter as many, longer blocks.  Every item is what sq_conv2d_nhwc_wgrad_scaled_bf16 (convT_cout == 0) or
 * sq_convT2x2s2_wgrad_bf16 (convT_cout > 0: K = 1, Cout = 4 * convT_cout, dW (2,2,convT_cout,Cin)) computes, to f32 rounding
 * (the same products; more of them summed per block, fewer block partials in the fixed-order finish); run-to-run identical.
 * bf16 X (N,H,W,Cin) and dY (N,H,W,Cout), channel counts multiples of 16 (plain items: of 8, the ragged form), db may be NULL.  accumulate: bit 0 -- dW is ADDED to
 * the contents of dw, bit 1 -- db to the contents of db (a parameter used by several passes of one step: the first item
 * writes, the later ones accumulate; items naming the same destination run in item order, in separate launches).
 * The caller keeps X and dY alive until the launch has run. */
typedef struct sq_wgrad_item {
    const void *x, *dy;
    float *dw, *db;
    int32_t N, H, W, Cin, Cout, K;
    int32_t convT_cout;
    float dw_scale;
    int32_t accumulate;
    int32_t mosaic_R, mosaic_Cc;   /* > 0: X / dY are N small images (H, W <= 8) taken as one mosaic of R x Cc cells (K = 3) */
    int32_t reserved;
} sq_wgrad_item;
int64_t sq_conv2d_nhwc_wgrad_group_workspace_bf16(const sq_wgrad_item *items, int n);
int sq_conv2d_nhwc_wgrad_group_bf16(const sq_wgrad_item *items, int n, float *workspace, void *stream);

/* ------------------------------------------------------------------------------------------
 * Tile front end: raw single-channel camera frames in HBM (OctopusData .dat memmap,
 * sequitr/dataio/octopus.py:231-245) -> ImageNorm (sequitr/pipeline.py:350-356) -> network tiles, and the
 * tile masks back to full-frame masks.
 *   sq_frame_stats     : per-frame float32 mean and std EXACTLY as numpy's np.mean / np.std of the float32
 *                        frame (8192-element chunks, pairwise blocks of 128 with 8 accumulators); H*W <= 2^24.
 *                        workspace: sq_frame_stats_workspace bytes.
 *   sq_frames_to_tiles : tiles (F*TR*TC, TS, TS) f32, tile (f,ty,tx) = frame f at origin (oy[ty], ox[tx]),
 *                        value (x - mean[f]) / std[f], or the plain cast when mean == std == NULL.
 *   sq_stitch_masks_u8 : out (F,H,W): pixel (y,x) = tile_masks[(f, ymap[y]>>16, xmap[x]>>16)][ymap[y]&0xffff][xmap[x]&0xffff]
 * ---------------------------------------------------------------------------------------- */
#define SQ_PIX_U8 0
#define SQ_PIX_U16 1
#define SQ_PIX_F32 2
int64_t sq_frame_stats_workspace(int F, int H, int W);
int sq_frame_stats(const void *frames, int dtype, float *mean, float *stdv, void *workspace, int F, int H, int W,
                   void *stream);
int sq_frames_to_tiles(const void *frames, int dtype, const float *mean, const float *stdv, const int32_t *oy,
                       const int32_t *ox, float *tiles, int F, int H, int W, int TR, int TC, int TS, void *stream);
int sq_stitch_masks_u8(const uint8_t *tile_masks, const int32_t *ymap, const int32_t *xmap, uint8_t *out, int F, int H,
                       int W, int TR, int TC, int TS, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SEQUITR_HIP_H */
