// Epilogue / fusion operands shared by the f32 implicit-GEMM convolution kernels (sq_conv_f32_v2.hip, sq_conv_f32_l0.hip).
#pragma once
#include "sq_common.h"

struct SqConvEpi {
    float *pooled;          // (N,H/2,W/2,Cout) or NULL
    const float *head_w;    // (Cout, head_c) 1x1 head or NULL (needs Cout == 16)
    const float *head_b;    // (head_c) or NULL
    float *logits;          // (N,H,W,head_c)
    uint8_t *mask;          // (N,H,W) or NULL
    int head_c;
    int store_y;
    const float *first_w;   // FIRST only: (3,3,1,16) and (16)
    const float *first_b;
    const float *up_x;      // UP only: low-resolution input (N,H/2,W/2,32)
    const float *up_w;      //          transpose-conv kernel (2,2,16,32) and bias (16)
    const float *up_b;
    int up_bridge;          //          SQ_BRIDGE_*: merged = bridge(convT(up_x), x)
    const float *x2;        // concat bridge (unet.py:196-197): channels [Cin/2, Cin) of the input come from this second
                            // tensor (N,H,W,Cin/2), channels [0, Cin/2) from x: tf.concat([upscale, skip], -1) never exists
};

#define SQ_L0_NOT_MINE 1
// the level-0 (16 -> 16 channel) kernel family of sq_conv_f32_l0.hip: returns SQ_L0_NOT_MINE when the shape is not its own
// (the caller then launches the generic kernel), otherwise the launch status
// mode: 0 plain, 1 FIRST (x = the single-channel image), 2 UP (x = the skip tensor); cout: 16, or 32 for the plain form
int sq_conv_l0_launch(int mode, const float *x, const float *w, const float *bias, float *y, int N, int H, int W,
                      int cout, int act, const SqConvEpi &epi, hipStream_t st);
