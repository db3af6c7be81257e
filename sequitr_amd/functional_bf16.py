"""Differentiable wrappers for the bf16 path (BASELINE configs 3-5): bf16 activations, fp32 master
weights / biases / gradients-of-weights.  First order only (the U-Net training graph); torch.autograd
is the tape, every forward and gradient is a HIP kernel of ops_bf16.py."""
import numpy as np
import contextlib

import torch
from torch.autograd.function import once_differentiable

from . import ops
from . import ops_bf16 as ob
from .functional import grad_sink


class _Conv2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, act):
        K, _, Cin, Cout = w.shape
        y = ob.conv2d(x, ob.pack_weights(w), bias, K, Cout, act=act)
        ctx.act, ctx.has_bias = act, bias is not None
        ctx.sinks = (grad_sink(w), grad_sink(bias))
        ctx.save_for_backward(x, w, y if ops.ACT[act] else None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        K, _, Cin, Cout = w.shape
        dpre = ob.act_bwd(dy.contiguous(), y, ctx.act) if y is not None else dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ob.conv2d(dpre, ob.pack_weights(w, transform=True), None, K, Cin)
        sw, sb = ctx.sinks
        if ctx.needs_input_grad[1] or ctx.has_bias:
            dw, db = ob.conv2d_wgrad(x, dpre, K, want_bias=ctx.has_bias, dw_out=sw, db_out=sb)
        return dx, (None if sw is not None else dw), (None if sb is not None else db), None


def conv2d(x, w, bias=None, act=None):
    return _Conv2d.apply(x, w, bias, act)


class _ConvFirst(torch.autograd.Function):
    """first conv of down0: f32 single-channel image -> bf16 activation."""

    @staticmethod
    def forward(ctx, x, w, bias, act):
        y = ob.conv3x3_first(x, w, bias, act=act)
        ctx.act = act
        ctx.sinks = (grad_sink(w), grad_sink(bias))
        ctx.save_for_backward(x, y)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        dpre = ob.act_bwd(dy.contiguous(), y, ctx.act) if ops.ACT[ctx.act] else dy.contiguous()   # act None: a BN layer follows
        sw, sb = ctx.sinks
        dw, db = ob.conv3x3_first_wgrad(x, dpre, dw_out=sw, db_out=sb)
        return None, (None if sw is not None else dw), (None if sb is not None else db), None


def conv3x3_first(x, w, bias, act='relu'):
    return _ConvFirst.apply(x, w, bias, act)


class BlockGate(object):
    """The backward gate of a conv block's output, out = dropout(ReLU(conv2)): d(pre-activation) = out > 0 ?
    d(out) * scale : 0 (scale = 1 / (1 - rate); 1 without dropout).  It is a pointwise function of `out` alone, so
    the kernel that PRODUCES d(out) can apply it in its epilogue -- every such kernel reads `out` anyway or for 2
    bytes per element -- instead of a stand-alone pass (read d(out), read out, write: sq_relu_scale_bwd_bf16, 7 %
    of the step).  conv_block() hangs one of these on its output; the tape entries that consume that tensor as
    their ONLY differentiable use (the pool + skip junction pair of an encoder level, the transpose conv of the
    next decoder level, the 1x1 head) pick it up, apply the gate and set `applied`; _ConvBlock.backward then skips
    its own pass.  Same two bf16 roundings as the stand-alone pass: the gradients are bit-identical."""

    def __init__(self, scale):
        self.scale, self.applied = float(scale), False

    def take(self):
        """the consumer's side: returns the factor to apply and marks the gate as applied"""
        self.applied = True
        return self.scale


def _gate_of(t):
    g = getattr(t, '_sq_gate', None)
    return g if isinstance(g, BlockGate) else None


FUSE_GATE = __import__("os").environ.get("SQ_FUSE_GATE", "1") != "0"    # A/B switch
FUSE_JUNCTION = __import__("os").environ.get("SQ_FUSE_JUNCTION", "1") != "0"


class JunctionHandoff(object):
    """The decoder junction's backward as the epilogue of the kernel that produces its incoming gradient.
    merged = bridge(up, skip) feeds exactly one op, conv1 of the block (unet.py:312-321); conv1's dgrad kernel can
    therefore form d_up (in the space-to-depth layout the transpose-conv gradients take) and d_skip in its epilogue and
    never write d(merged) (sq_conv2d_nhwc_dgrad_junction_bf16): one write and one read of a full-resolution tensor
    less per decoder level than the stand-alone sq_bridge_bwd_s2d_bf16 pass, same roundings.  up_junction() hangs
    one of these on `merged`; conv_block() passes it to its tape entry, whose backward leaves (g, dskip) in
    `result` and hands autograd a zero-stride placeholder as the gradient of `merged`; _UpJunction.backward picks
    the result up instead of reading that gradient."""

    def __init__(self):
        self.up = self.skip = self.kind = self.result = None


_ZEROS = {}


def _placeholder_grad(like):
    key = (like.device, like.dtype)
    z = _ZEROS.get(key)
    if z is None:
        z = _ZEROS[key] = torch.zeros((), dtype=like.dtype, device=like.device)
    return z.expand(like.shape)


class _ConvBlock(torch.autograd.Function):
    """conv_block of the U-Net (sequitr/networks/unet.py:265-277) as ONE tape entry:
    relu(conv1) -> relu(conv2) -> dropout.  The two activations between the three ops have exactly one
    consumer each, which lets the backward fuse what separate tape entries cannot:
      * dropout backward + conv2's ReLU backward: one pass (sq_act_dropout_bwd_bf16),
      * conv1's ReLU backward: the epilogue of conv2's dgrad kernel (sq_conv2d_nhwc_dgrad_relu_bf16).
    x is the bf16 block input, or the f32 single-channel image for down0."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, rate, seed, mask, step_dev, gate, junction, poolbox):
        first = x.dtype == torch.float32
        ctx.gate, ctx.junction = gate, junction
        f = w1.shape[3]
        m1 = None
        pooling = poolbox is not None and mask is None and x.shape[1] % 2 == 0 and x.shape[2] % 2 == 0
        if FUSE_FIRST and FUSE_MASK and first and pooling and ob.conv_first_block_takes(x, w1, w2):
            # down0 with its pool in ONE launch: y1 is made per tile inside conv2's kernel (and stored for the weight gradient)
            y1, m1, out, pooled = ob.conv_first_block_dropout_pool(x, w1, b1, ob.pack_weights(w2), b2, rate, seed=seed,
                                                                   step_dev=step_dev)
            poolbox.append(pooled)
            ctx.rate, ctx.first = rate, first
            ctx.sinks = (grad_sink(w1), grad_sink(b1), grad_sink(w2), grad_sink(b2))
            ctx.save_for_backward(x, w1, w2, y1, out, None, m1)
            return out
        if FUSE_MASK and f % 16 == 0:
            # conv1's ReLU sign mask (1 bit per element) leaves its epilogue: conv2's dgrad gates on it in the backward
            # instead of reading y1 again (y1 itself stays: it is conv2's wgrad operand)
            y1, m1 = ob.conv3x3_first_mask(x, w1, b1) if first else ob.conv2d_mask(x, ob.pack_weights(w1), b1, 3, f)
        else:
            y1 = ob.conv3x3_first(x, w1, b1, act='relu') if first else ob.conv2d(x, ob.pack_weights(w1), b1, 3, f, act='relu')
        m = y2 = None
        if poolbox is not None and mask is None and y1.shape[1] % 2 == 0 and y1.shape[2] % 2 == 0:
            # an encoder level: the max-pooled copy the next level reads leaves conv2's epilogue with the block output
            # (poolbox: a one-slot list the pool layer looks into -- a side channel, not a second autograd output)
            out, pooled = ob.conv2d_dropout_pool(y1, ob.pack_weights(w2), b2, 3, f, 'relu', rate, seed=seed, step_dev=step_dev)
            poolbox.append(pooled)
            if rate <= 0.0:
                y2 = out
        elif rate > 0.0 and mask is None:
            # conv2 + ReLU + dropout in ONE kernel; neither y2 nor a mask is stored: out > 0 <=> kept and active
            out = ob.conv2d_dropout(y1, ob.pack_weights(w2), b2, 3, f, 'relu', rate, seed=seed, step_dev=step_dev)
        else:
            y2 = ob.conv2d(y1, ob.pack_weights(w2), b2, 3, f, act='relu')
            out = y2
            if rate > 0.0:                                      # pinned masks (parity tests): separate kernels
                out, m = ob.dropout_fwd(y2, rate, seed=seed, mask=mask, step_dev=step_dev)
        ctx.rate, ctx.first = rate, first
        ctx.sinks = (grad_sink(w1), grad_sink(b1), grad_sink(w2), grad_sink(b2))
        ctx.save_for_backward(x, w1, w2, y1, y2 if m is not None else out, m, m1)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        x, w1, w2, y1, y2, m, m1 = ctx.saved_tensors
        s1w, s1b, s2w, s2b = ctx.sinks
        f = w2.shape[3]
        dout = dout.contiguous()
        pre = ctx.gate is not None and ctx.gate.applied
        if ctx.gate is not None:
            ctx.gate.applied = False
        if pre:
            d2 = dout                                           # the producer of dout applied the gate (BlockGate)
        elif m is not None:
            d2 = ob.act_dropout_bwd(dout, m, y2, ctx.rate, 'relu')
        elif ctx.rate > 0.0:
            inv = float(np.float32(1.0) / (np.float32(1.0) - np.float32(ctx.rate)))    # as the kernels compute it
            d2 = ob.relu_scale_bwd(dout, y2, inv)                           # y2 holds the block output here
        else:
            d2 = ob.act_bwd(dout, y2, 'relu')
        dw2, db2 = ob.conv2d_wgrad(y1, d2, 3, want_bias=True, dw_out=s2w, db_out=s2b)
        if m1 is not None:
            d1 = ob.conv2d_dgrad_mask(d2, ob.pack_weights(w2, transform=True), m1, 3, f)
        else:
            d1 = ob.conv2d_dgrad_relu(d2, ob.pack_weights(w2, transform=True), y1, 3)
        if ctx.first:
            dw1, db1 = ob.conv3x3_first_wgrad(x, d1, dw_out=s1w, db_out=s1b)
            dx = None
        else:
            dw1, db1 = ob.conv2d_wgrad(x, d1, 3, want_bias=True, dw_out=s1w, db_out=s1b)
            dx = None
            if ctx.needs_input_grad[0] and ctx.junction is not None:
                j = ctx.junction                                # x = merged: the junction's backward rides in this dgrad
                j.result = ob.conv2d_dgrad_junction(d1, ob.pack_weights(w1, transform=True), j.up, j.skip, j.kind, 3,
                                                    x.shape[3])
                dx = _placeholder_grad(x)
            elif ctx.needs_input_grad[0]:
                dx = ob.conv2d(d1, ob.pack_weights(w1, transform=True), None, 3, x.shape[3])
        return (dx, None if s1w is not None else dw1, None if s1b is not None else db1,
                None if s2w is not None else dw2, None if s2b is not None else db2, None, None, None, None, None, None, None)


FUSE_POOL = __import__("os").environ.get("SQ_FUSE_POOL", "1") != "0"
FUSE_MASK = __import__("os").environ.get("SQ_FUSE_MASK", "1") != "0"
# down0 as ONE launch (sq_conv3x3_first_block_dropout_pool_bf16): bit-identical, measured 127 us against 48 + 65 us for the two
# kernels (the per-tile conv1 phase sits on the block's critical path, which is one HBM round trip per tile) -- off by default
FUSE_FIRST = __import__("os").environ.get("SQ_FUSE_FIRST", "0") != "0"


def conv_block(x, w1, b1, w2, b2, rate=0.0, seed=0, mask=None, step_dev=None, pool_follows=False):
    gate = None
    if FUSE_GATE and mask is None:                               # pinned masks keep the separate mask kernels
        rate = float(rate)
        gate = BlockGate(float(np.float32(1.0) / (np.float32(1.0) - np.float32(rate))) if rate > 0.0 else 1.0)
    junction = getattr(x, '_sq_junction', None)
    if not isinstance(junction, JunctionHandoff):
        junction = None
    poolbox = [] if (pool_follows and FUSE_POOL) else None
    out = _ConvBlock.apply(x, w1, b1, w2, b2, float(rate), int(seed), mask, step_dev, gate, junction, poolbox)
    if gate is not None:
        out._sq_gate = gate
    if poolbox:
        out._sq_pooled = poolbox                                # maxpool2x2(out) returns it instead of running the pool kernel
    return out


class _MaxPool(torch.autograd.Function):
    """2x2/s2 max pool.  `box` (a dict shared with the decoder junction that also consumes x, or None): when
    the junction has left the skip gradient in box['dskip'] -- its backward always runs first, it is downstream
    of everything below this pool -- the pool's backward adds it in the same pass and x receives ONE gradient
    instead of two that autograd would have to sum in an extra kernel."""

    @staticmethod
    def forward(ctx, x, box, gate, pooled):
        ctx.save_for_backward(x)
        ctx.box, ctx.gate = box, gate
        if pooled:                                              # written by the conv block's epilogue (conv_block(pool_follows=True))
            return pooled.pop()
        return ob.maxpool2x2(x)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        other = ctx.box.pop('dskip', None) if ctx.box is not None else None
        if other is not None:
            # x has exactly two differentiable uses, this pool and the junction whose gradient is `other`: the sum is
            # the whole gradient of x, so the block gate of x (if it is a conv block's output) is applied here too
            gs = ctx.gate.take() if ctx.gate is not None else 0.0
            return ob.maxpool2x2_bwd_add(x, dy.contiguous(), other, gate_scale=gs), None, None, None
        return ob.maxpool2x2_bwd(x, dy.contiguous()), None, None, None


def maxpool2x2(x, box=None):
    return _MaxPool.apply(x, box, _gate_of(x) if box is not None else None, getattr(x, '_sq_pooled', None))


_UPJ_BOTH = __import__("os").environ.get("SQ_UPJ_BOTH", "1") != "0"   # A/B: dual-output convT+bridge kernel


class _UpJunction(torch.autograd.Function):
    """conv_transpose_layer + bridge of a decoder level (unet.py:312-319) as one tape entry: the forward is
    the two kernels as before; the backward produces d_up directly in the space-to-depth layout the
    transpose-conv gradients consume (one pass instead of bridge backward + space-to-depth) and, when the
    skip tensor's pool shares a `box`, hands d_skip to that pool's backward instead of to autograd."""

    @staticmethod
    def forward(ctx, x, w, bias, skip, kind, box, gate, handoff):
        ctx.gate, ctx.handoff = gate, handoff
        if kind == 'eltwise_mul' and _UPJ_BOTH:                 # `up` is needed by the backward: both from one pass
            up, merged = ob.convT2x2s2_bridge_both(x, ob.to_bf16(w), bias, skip, kind)
        elif kind == 'eltwise_mul':
            up = ob.convT2x2s2(x, ob.to_bf16(w), bias)
            merged = ob.bridge(up, skip, kind)
        else:                                                   # add / sub: the up-scaled tensor is not kept at all
            up, merged = None, ob.convT2x2s2(x, ob.to_bf16(w), bias, skip, kind)
        keep = kind == 'eltwise_mul'
        if handoff is not None:
            handoff.up, handoff.skip, handoff.kind = (up if keep else None), (skip if keep else None), kind
        ctx.save_for_backward(x, w, up if keep else None, skip if keep else None)
        ctx.kind, ctx.box, ctx.has_bias = kind, box, bias is not None
        ctx.sinks = (grad_sink(w), grad_sink(bias))
        return merged

    @staticmethod
    @once_differentiable
    def backward(ctx, dm):
        x, w, up, skip = ctx.saved_tensors
        Cout, Cin = w.shape[2], w.shape[3]
        if ctx.handoff is not None and ctx.handoff.result is not None:
            g, dskip = ctx.handoff.result                       # formed in conv1's dgrad epilogue (JunctionHandoff)
            ctx.handoff.result = None
        else:
            g, dskip = ob.bridge_bwd_s2d(dm.contiguous(), up, skip, ctx.kind)
        dx = None
        if ctx.needs_input_grad[0]:
            packs = getattr(w, '_sq_packs', None) or {}
            wp = packs.get('convT_dgrad')
            if wp is None:
                wp = ob.pack_weights(w.reshape(1, 1, 4 * Cout, Cin))
            if ctx.gate is not None:        # x = the previous block's output, used here only: its gate in the epilogue
                dx = ob.conv2d_dgrad_relu(g, wp, x, 1, scale=ctx.gate.take())
            else:
                dx = ob.conv2d(g, wp, None, 1, Cin)
        sw, sb = ctx.sinks
        dw, db = ob.convT_wgrad(x, g, Cout, want_bias=ctx.has_bias, dw_out=sw, db_out=sb if ctx.has_bias else None)
        dw, db = (None if sw is not None else dw), (None if (sb is not None or not ctx.has_bias) else db)
        if ctx.box is not None and ctx.needs_input_grad[3]:
            ctx.box['dskip'] = dskip                            # picked up by the skip tensor's pool backward
            dskip = None
        return dx, dw, db, dskip, None, None, None, None


def up_junction(x, w, bias, skip, kind, box=None, x_single_use=False, merged_single_use=False):
    """x_single_use: the caller guarantees this junction is the only differentiable consumer of x (U-Net wiring:
    net[-1] feeds the next up_layer and nothing else), which lets x's block gate ride in the dgrad epilogue.
    merged_single_use: the result feeds one conv_block and nothing else (unet.py:319-321): its backward may then be
    formed in that block's dgrad epilogue (JunctionHandoff)."""
    handoff = JunctionHandoff() if (merged_single_use and FUSE_JUNCTION) else None
    merged = _UpJunction.apply(x, w, bias, skip, kind, box, _gate_of(x) if x_single_use else None, handoff)
    if handoff is not None:
        merged._sq_junction = handoff
    return merged


class _ConvT(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias):
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        ctx.sinks = (grad_sink(w), grad_sink(bias))
        return ob.convT2x2s2(x, ob.to_bf16(w), bias)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        Cout, Cin = w.shape[2], w.shape[3]
        g = ob.space_to_depth2(dy.contiguous())                              # (N,H,W,4*Cout) bf16
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            packs = getattr(w, '_sq_packs', None) or {}
            wp = packs.get('convT_dgrad')
            if wp is None:
                wp = ob.pack_weights(w.reshape(1, 1, 4 * Cout, Cin))          # 1x1 conv 4Cout -> Cin
            dx = ob.conv2d(g, wp, None, 1, Cin)
        sw, sb = ctx.sinks
        dw, db = ob.convT_wgrad(x, g, Cout, want_bias=ctx.has_bias, dw_out=sw, db_out=sb if ctx.has_bias else None)
        dw, db = (None if sw is not None else dw), (None if (sb is not None or not ctx.has_bias) else db)
        return dx, dw, db


def convT2x2s2(x, w, bias=None):
    return _ConvT.apply(x, w, bias)


class _Bridge(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, kind):
        ctx.kind = kind
        ctx.save_for_backward(*((a, b) if kind == 'eltwise_mul' else ()))
        return ob.bridge(a, b, kind)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        a, b = ctx.saved_tensors if ctx.kind == 'eltwise_mul' else (None, None)
        da, db = ob.bridge_bwd(dy.contiguous(), a, b, ctx.kind)
        return da, db, None


def bridge(a, b, kind):
    return _Bridge.apply(a, b, kind)


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rate, seed, mask, step_dev):
        y, m = ob.dropout_fwd(x, rate, seed=seed, mask=mask, step_dev=step_dev)
        ctx.rate = rate
        ctx.save_for_backward(m)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (m,) = ctx.saved_tensors
        return ob.dropout_bwd(dy.contiguous(), m, ctx.rate), None, None, None, None


def dropout(x, rate, seed=0, mask=None, step_dev=None):
    if rate <= 0.0:
        return x
    return _Dropout.apply(x, float(rate), int(seed), mask, step_dev)


class _Head(torch.autograd.Function):
    """to_image on a bf16 activation: f32 logits out, f32 dlogits in."""

    @staticmethod
    def forward(ctx, x, w, bias, gate):
        logits, _ = ob.head_fwd(x, w, bias, want_mask=False)
        ctx.gate = gate
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        ctx.sinks = (grad_sink(w), grad_sink(bias))
        return logits

    @staticmethod
    @once_differentiable
    def backward(ctx, dz):
        x, w = ctx.saved_tensors
        sw, sb = ctx.sinks
        gs = ctx.gate.take() if (ctx.gate is not None and ctx.needs_input_grad[0]) else 0.0
        dx, dw, db = ob.head_bwd(x, w, dz.contiguous(), want_dx=ctx.needs_input_grad[0], dw_out=sw,
                                 db_out=sb if ctx.has_bias else None, gate_scale=gs)
        return dx, (None if sw is not None else dw), (db if ctx.has_bias and sb is None else None), None


def conv1x1_head(x, w, bias=None, x_single_use=False):
    return _Head.apply(x, w, bias, _gate_of(x) if x_single_use else None)


_DEFER_LOSS = [False]


@contextlib.contextmanager
def deferred_loss(on=True):
    """Inside: conv1x1_head_loss() returns an UNINITIALISED loss tensor that its backward fills in (the forward kernel's
    value, bit for bit) -- for callers that run .backward() before they read the loss (UNetTrainer.forward_backward): the
    forward kernel, one read of the level-0 activation, is not launched.  Without gradients (validation) nothing changes."""
    prev = _DEFER_LOSS[0]
    _DEFER_LOSS[0] = bool(on)
    try:
        yield
    finally:
        _DEFER_LOSS[0] = prev


class _HeadLoss(torch.autograd.Function):
    """to_image + weighted softmax cross-entropy as one tape entry: the logits are recomputed from x in backward
    instead of being written, read by the loss, and their gradient written and read again (three kernels and two f32
    tensors less per step); same loss and same gradients, bit for bit, as _Head followed by F.weighted_softmax_cross_entropy."""

    @staticmethod
    def forward(ctx, x, w, bias, onehot, weights, gate, defer):
        ctx.gate = gate
        ctx.save_for_backward(x, w, bias, onehot, weights)
        ctx.sinks = (grad_sink(w), grad_sink(bias))
        ctx.loss_out = None
        if defer:
            # deferred_loss(): the caller runs the backward before it looks at the loss -- the backward kernel recomputes
            # the logits anyway and leaves the forward kernel's value in this tensor (until then it is uninitialised)
            loss = torch.empty((), dtype=torch.float32, device=x.device)
            ctx.loss_out = loss.detach()                        # an alias without the autograd node: no reference cycle
            return loss
        return ob.head_wce_fwd(x, w, bias, onehot, weights)

    @staticmethod
    @once_differentiable
    def backward(ctx, dloss):
        x, w, bias, onehot, weights = ctx.saved_tensors
        sw, sb = ctx.sinks
        gs = ctx.gate.take() if (ctx.gate is not None and ctx.needs_input_grad[0]) else 0.0
        dx, dw, db = ob.head_wce_bwd(x, w, bias, onehot, weights, dloss.contiguous(), want_dx=ctx.needs_input_grad[0],
                                     dw_out=sw, db_out=sb if bias is not None else None, gate_scale=gs,
                                     loss_out=ctx.loss_out)
        return (dx, (None if sw is not None else dw), (db if bias is not None and sb is None else None), None, None,
                None, None)


def conv1x1_head_loss(x, w, bias, onehot, weights, x_single_use=False):
    # (grad mode is read HERE: inside Function.forward it is always off)
    defer = _DEFER_LOSS[0] and torch.is_grad_enabled() and (x.requires_grad or w.requires_grad)
    return _HeadLoss.apply(x, w, bias, onehot, weights, _gate_of(x) if x_single_use else None, defer)


class _BatchNormTrain(torch.autograd.Function):
    """y = act(BN(x)) with batch statistics on a bf16 activation (tf.layers.batch_normalization(training=True) between a
    conv and its ReLU); updates the moving statistics in place.  gamma / beta / statistics f32."""

    @staticmethod
    def forward(ctx, x, gamma, beta, moving_mean, moving_var, eps, momentum, act):
        mean, var = ob.bn_stats(x)
        scale, shift = ops.bn_fold(gamma, beta, mean, var, eps)
        y = ob.bn_apply(x, scale, shift, act)
        ops.bn_update_moving_(moving_mean, moving_var, mean, var, x.numel() // x.shape[-1], momentum)
        ctx.save_for_backward(x, y, mean, var, gamma)
        ctx.eps, ctx.act = eps, act
        ctx.sinks = (grad_sink(gamma), grad_sink(beta))
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, y, mean, var, gamma = ctx.saved_tensors
        dx, dgamma, dbeta = ob.bn_bwd(x, dy.contiguous(), y, ctx.act, mean, var, gamma, ctx.eps)
        sg, sb = ctx.sinks
        if sg is not None:
            sg.copy_(dgamma)
        if sb is not None:
            sb.copy_(dbeta)
        return dx, (None if sg is not None else dgamma), (None if sb is not None else dbeta), None, None, None, None, None


def batch_norm_train(x, gamma, beta, moving_mean, moving_var, eps=ops.BN_EPS, momentum=ops.BN_MOMENTUM, act=None):
    return _BatchNormTrain.apply(x, gamma, beta, moving_mean, moving_var, float(eps), float(momentum), act)
