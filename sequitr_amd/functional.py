"""Differentiable wrappers: torch.autograd is the TAPE only (plumbing) -- every forward and every
gradient is one of the hand-written HIP kernels in ops.py.  These let the reference's hook-style
graph code (UNet.build, sequitr/networks/unet.py:224-322) be trained exactly as written: a hook
may be overridden with any differentiable composition and gradients still flow.
"""
import torch

from . import ops


class _Conv2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, act, wscale):
        y = ops.conv2d(x, w, bias, act=act, wscale=wscale)
        ctx.act, ctx.wscale, ctx.has_bias = act, wscale, bias is not None
        ctx.save_for_backward(x, w, y if ops.ACT[act] else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        dy = dy.contiguous()
        dpre = ops.act_bwd(dy, y, ctx.act) if y is not None else dy
        K = w.shape[0]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.conv2d_dgrad(dpre, w, wscale=ctx.wscale)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw, db = ops.conv2d_wgrad(x, dpre, K, want_bias=ctx.has_bias)
            if ctx.wscale != 1.0:
                dw = dw * ctx.wscale                     # w' = w * wscale (gan.py:79)
        return dx, dw, db, None, None


def conv2d(x, w, bias=None, act=None, wscale=1.0):
    return _Conv2d.apply(x, w, bias, act, float(wscale))


class _Head(torch.autograd.Function):
    """1x1 conv to <= 4 channels (to_image), no activation."""

    @staticmethod
    def forward(ctx, x, w, bias):
        y = ops.conv2d(x, w, bias, act=None)
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dz):
        x, w = ctx.saved_tensors
        dx, dw, db = ops.conv1x1_small_bwd(x, w, dz.contiguous(), want_dx=ctx.needs_input_grad[0])
        return dx, dw, (db if ctx.has_bias else None)


def conv1x1_head(x, w, bias=None):
    return _Head.apply(x, w, bias)


class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.maxpool2x2(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.maxpool2x2_bwd(x, dy.contiguous())


def maxpool2x2(x):
    return _MaxPool.apply(x)


class _ConvT(torch.autograd.Function):
    """2x2/s2 transpose conv + bias.  Backward = space-to-depth, then a 1x1 dgrad and a 1x1 wgrad:
    convT(x) == depth_to_space(conv1x1(x, W')) with W'[c][(2a+b)*Cout + o] = W[a,b,o,c]."""

    @staticmethod
    def forward(ctx, x, w, bias):
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return ops.convT2x2s2(x, w, bias)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        Cout, Cin = w.shape[2], w.shape[3]
        g = ops.space_to_depth2(dy.contiguous())                         # (N,H,W,4*Cout)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            # dX[p,c] = sum_rho g[p,rho] * W[rho,c]: a 1x1 conv with HWIO filter (1,1,4Cout,Cin) = W flat
            dx = ops.conv2d(g, w.reshape(1, 1, 4 * Cout, Cin), None, act=None)
        if ctx.needs_input_grad[1] or ctx.has_bias:
            dwp, dbp = ops.conv2d_wgrad(x, g, 1, want_bias=ctx.has_bias)  # (1,1,Cin,4Cout), (4Cout)
            dw = dwp.reshape(Cin, 2, 2, Cout).permute(1, 2, 3, 0).contiguous()
            if ctx.has_bias:
                db = dbp.reshape(4, Cout).sum(0)
        return dx, dw, db


def convT2x2s2(x, w, bias=None):
    return _ConvT.apply(x, w, bias)


class _Bridge(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, kind):
        ctx.kind = kind
        ctx.save_for_backward(*((a, b) if kind == 'eltwise_mul' else ()))
        return ops.bridge(a, b, kind)

    @staticmethod
    def backward(ctx, dy):
        a, b = ctx.saved_tensors if ctx.kind == 'eltwise_mul' else (None, None)
        da, db = ops.bridge_bwd(dy.contiguous(), a, b, ctx.kind)
        return da, db, None


def bridge(a, b, kind):
    return _Bridge.apply(a, b, kind)


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rate, seed, mask):
        y, m = ops.dropout_fwd(x, rate, seed=seed, mask=mask)
        ctx.rate = rate
        ctx.save_for_backward(m)
        return y

    @staticmethod
    def backward(ctx, dy):
        (m,) = ctx.saved_tensors
        return ops.dropout_bwd(dy.contiguous(), m, ctx.rate), None, None, None


def dropout(x, rate, seed=0, mask=None):
    if rate <= 0.0:
        return x
    return _Dropout.apply(x, float(rate), int(seed), mask)


class _WeightedSoftmaxCE(torch.autograd.Function):
    """loss (0-d float32 tensor) with the fused forward+backward kernel; d loss/d logits is
    computed in the same pass and scaled by the incoming gradient in backward."""

    @staticmethod
    def forward(ctx, logits, onehot, weights):
        loss, dz = ops.wsoftmax_ce(logits, onehot, weights, want_grad=True)
        ctx.save_for_backward(dz)
        return loss.to(torch.float32)

    @staticmethod
    def backward(ctx, dloss):
        (dz,) = ctx.saved_tensors
        return dz * dloss, None, None


def weighted_softmax_cross_entropy(logits, onehot, weights):
    return _WeightedSoftmaxCE.apply(logits, onehot, weights)
