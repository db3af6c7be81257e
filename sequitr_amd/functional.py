"""Differentiable wrappers: torch.autograd is the TAPE only (plumbing) -- every forward and every
gradient is one of the hand-written HIP kernels in ops.py.  These let the reference's hook-style
graph code (UNet.build, sequitr/networks/unet.py:224-322; generator_network /
discriminator_network, sequitr/networks/gan.py:149-316) be trained as written.

The convolution / activation / pixel-norm / pooling / blend functions form a set that is CLOSED
under differentiation: each backward is itself a composition of these functions, so
``torch.autograd.grad(..., create_graph=True)`` works through them.  That is what the WGAN-GP
penalty needs -- it differentiates the discriminator's input gradient w.r.t. the discriminator's
weights (gan.py:719-729, ``tf.gradients`` inside the loss).
"""
import torch

from . import ops


# ---------------------------------------------------------------------------------------------
# convolution: ConvFwd / ConvDgrad / ConvWgrad are each other's derivatives
# ---------------------------------------------------------------------------------------------
def grad_sink(param):
    """Gradient destination a trainer attached to a parameter (``param._sq_grad_sink``: a float32 view of
    its flat gradient bucket), or None.  Each U-Net parameter feeds exactly one op per step, so that op's
    gradient kernel writes the view directly and hands autograd ``None`` -- no AccumulateGrad add kernel
    per parameter, no zeroing dependency.  Only first-order backward passes use it."""
    return getattr(param, '_sq_grad_sink', None) if param is not None else None


# Which PARAMETER gradients the running backward pass is for.  A custom Function only knows that a weight
# "requires grad" (ctx.needs_input_grad), not whether THIS torch.autograd.grad call asked for it, so without a hint
# every pass through the discriminator computes the weight and bias gradients of all its layers: the WGAN-GP
# penalty's inner pass (grad of D(mix) w.r.t. mix only, gan.py:721) and the generator step (which differentiates
# THROUGH the discriminator, gan.py:650) threw away a full set of wgrad + finish + bias-sum launches each.
# `with grads_wanted(params):` restricts the parameter gradients computed inside to `params` (an empty list: none);
# gradients of everything else come back as None, which autograd.grad discards anyway.
_WANTED = None


class grads_wanted(object):
    def __init__(self, params):
        self.ids = None if params is None else frozenset(id(p) for p in params)

    def __enter__(self):
        global _WANTED
        self.prev, _WANTED = _WANTED, self.ids
        return self

    def __exit__(self, *exc):
        global _WANTED
        _WANTED = self.prev
        return False


# Gradient sinks of a first-order pass through graphs in which a parameter is used SEVERAL times (the GAN's discriminator step:
# D(Gz | X), D(mix) and the penalty's second-order terms all contribute to every discriminator weight).  While
# `with grad_sinks({id(param): float32 view}):` and an ops_bf16.WgradQueue are active, a bf16 3x3 / 1x1 conv's weight-gradient
# work is queued with that view as its destination (first contribution writes, later ones accumulate) and autograd gets None:
# the step's ~50 weight-gradient launches + finishes + ~40 framework adds become a few grouped launches (WgradQueue.flush).
_SINKS = None


class grad_sinks(object):
    def __init__(self, mapping):
        self.map, self.touched = mapping, set()

    def __enter__(self):
        global _SINKS
        self.prev, _SINKS = _SINKS, self
        return self

    def __exit__(self, *exc):
        global _SINKS
        _SINKS = self.prev
        return False


def _sink_wgrad(x, dy, K, wscale, w_id, bias_id=None, want_bias=False):
    """queue dW (+ db) of a conv for the grouped launch, destination = the parameters' sinks; False: not applicable here"""
    sk = _SINKS
    if sk is None or torch.is_grad_enabled() or x.dim() != 4:
        return False
    if x.dtype == torch.float32 and dy.dtype == torch.float32:
        return _sink_dense_wgrad(sk, x, dy, K, wscale, w_id, bias_id, want_bias)
    if x.dtype != torch.bfloat16 or dy.dtype != torch.bfloat16:
        return False
    from . import ops_bf16 as ob
    q = ob._QUEUE[0]
    sw = sk.map.get(w_id)
    if q is None or q.max_elems <= 0 or sw is None:
        return False
    N, H, W, Cin = x.shape
    Cout = dy.shape[-1]
    if Cin % 8 or Cout % 8:
        return False
    mosaic = None
    if ops.USE_MOSAIC and W < 16 and N * H > 1:                 # small-image levels: as one mosaic (3x3), or their own launches
        mosaic = ops._mosaic_plan(N, H, W) if (K == 3 and Cin % 16 == 0 and Cout % 16 == 0) else None
        if mosaic is None:
            return False
    sb = None
    if want_bias:
        sb = sk.map.get(bias_id)
        if sb is None:
            return False
    q.push(x.contiguous(), dy.contiguous(), K, sw, sb, dw_scale=wscale, mosaic=mosaic)
    sk.touched.add(w_id)
    if sb is not None:
        sk.touched.add(bias_id)
    return True


def _sink_dense_wgrad(sk, x, dy, K, wscale, w_id, bias_id, want_bias):
    """a dense layer in row form (F.dense: x (1,1,M,Kin), dY (1,1,M,N), few rows, long reduction -- the discriminator's
    8208 -> 512 layer gets three contributions per step): its weight-gradient kernel writes the parameter's sink directly, the
    first contribution of a pass overwriting and later ones accumulating (sq_dense_wgrad_f32's accumulate bits), instead of
    three (Kin, N) tensors summed by framework adds"""
    if K != 1 or x.shape[0] != 1 or x.shape[1] != 1 or not ops.USE_DENSE:
        return False
    M, Kin, N = x.shape[2], x.shape[3], dy.shape[3]
    sw = sk.map.get(w_id)
    if M > 128 or Kin * N < (1 << 16) or sw is None or sw.numel() != Kin * N:
        return False
    sb = sk.map.get(bias_id) if want_bias else None
    if want_bias and (sb is None or sb.numel() != N):
        return False
    acc = (1 if w_id in sk.touched else 0) | (2 if (want_bias and bias_id in sk.touched) else 0)
    ops.dense_wgrad(x.reshape(M, Kin), dy.reshape(M, N).contiguous(), want_bias=want_bias, dw_scale=wscale, dw_out=sw.view(Kin, N),
                    db_out=sb.view(N) if sb is not None else None, accumulate=acc)
    sk.touched.add(w_id)
    if sb is not None:
        sk.touched.add(bias_id)
    return True


def _pid(t):
    """identity of the PARAMETER behind t: layers hand views of their variables to the ops (bias.view(-1), the
    (1,1,Cin,Cout) form of a dense kernel); the id is taken at forward time, when the variable is certainly alive"""
    if t is None:
        return None
    base = t._base if t._base is not None else t
    return id(base)


def _want(pid):
    return pid is not None and (_WANTED is None or pid in _WANTED)


def convT_param_grads(dwp, dbp, Cin, Cout, sinks):
    """(dW (2,2,Cout,Cin), db (Cout)) of the 2x2/s2 transpose conv from the 1x1 wgrad of its
    space-to-depth form (dwp (1,1,Cin,4Cout), dbp (4Cout) or None); written into the sinks when set."""
    sw, sb = sinks
    dw = dwp.reshape(Cin, 2, 2, Cout).permute(1, 2, 3, 0)
    if sw is not None:
        sw.copy_(dw)
        dw = None
    else:
        dw = dw.contiguous()
    db = None
    if dbp is not None:
        if sb is not None:
            torch.sum(dbp.reshape(4, Cout), 0, out=sb)
        else:
            db = dbp.reshape(4, Cout).sum(0)
    return dw, db


class _ConvFwd(torch.autograd.Function):
    """y = conv2d(x, w * wscale), no bias, no activation (linear in x and in w)."""

    @staticmethod
    def forward(ctx, x, w, wscale):
        ctx.wscale, ctx.w_id = wscale, _pid(w)
        ctx.save_for_backward(x, w)
        return ops.conv2d(x, w, None, act=None, wscale=wscale)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx = _ConvDgrad.apply(dy, w, ctx.wscale) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1] and _want(ctx.w_id) and not _sink_wgrad(x, dy, w.shape[0], ctx.wscale, ctx.w_id):
            dw = _ConvWgrad.apply(x, dy, w.shape[0], ctx.wscale)
        return dx, dw, None


class _ConvDgrad(torch.autograd.Function):
    """dx = A_w^T dy, the adjoint of ConvFwd in x."""

    @staticmethod
    def forward(ctx, dy, w, wscale):
        ctx.wscale, ctx.w_id = wscale, _pid(w)
        ctx.save_for_backward(dy, w)
        return ops.conv_dgrad_raw(dy.contiguous(), w, wscale)

    @staticmethod
    def backward(ctx, ddx):
        dy, w = ctx.saved_tensors
        d_dy = _ConvFwd.apply(ddx.contiguous(), w, ctx.wscale) if ctx.needs_input_grad[0] else None
        d_w = None
        if ctx.needs_input_grad[1] and _want(ctx.w_id) and not _sink_wgrad(ddx, dy, w.shape[0], ctx.wscale, ctx.w_id):
            d_w = _ConvWgrad.apply(ddx.contiguous(), dy, w.shape[0], ctx.wscale)
        return d_dy, d_w, None


class _ConvWgrad(torch.autograd.Function):
    """dw[tap,ci,co] = wscale * sum_p x[p+tap,ci] dy[p,co]  (bilinear in x and dy)."""

    @staticmethod
    def forward(ctx, x, dy, K, wscale):
        ctx.K, ctx.wscale = K, wscale
        ctx.save_for_backward(x, dy)
        dw, _ = ops.conv_wgrad_raw(x, dy.contiguous(), K, want_bias=False, dw_scale=wscale)   # w' = w * wscale (gan.py:79)
        return dw

    @staticmethod
    def backward(ctx, ddw):
        x, dy = ctx.saved_tensors
        ddw = ddw.contiguous()
        d_x = _ConvDgrad.apply(dy, ddw, ctx.wscale) if ctx.needs_input_grad[0] else None
        d_dy = _ConvFwd.apply(x, ddw, ctx.wscale) if ctx.needs_input_grad[1] else None
        return d_x, d_dy, None, None


class _ActBwd(torch.autograd.Function):
    """dpre = dy * act'(pre), decided from the activation output y (piecewise constant in y)."""

    @staticmethod
    def forward(ctx, dy, y, act):
        ctx.act = act
        ctx.save_for_backward(y)
        return ops.act_bwd(dy.contiguous(), y, act)

    @staticmethod
    def backward(ctx, ddpre):
        (y,) = ctx.saved_tensors
        return _ActBwd.apply(ddpre, y, ctx.act), None, None


class _Act(torch.autograd.Function):
    """stand-alone activation (k_leaky_relu_alpha, gan.py:44-46)."""

    @staticmethod
    def forward(ctx, x, act):
        y = ops.act_fwd(x.contiguous(), act)
        ctx.act = act
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return _ActBwd.apply(dy, y, ctx.act), None


def act(x, kind):
    return _Act.apply(x, kind) if ops.ACT[kind] else x


class ActGate(object):
    """Backward of a convolution's fused activation as an epilogue of the kernel that produces the gradient.
    conv2d() hangs one of these on its output y (when FUSE_ACT_GATES is on); an op that is y's ONLY differentiable
    consumer and whose backward kernel can apply act'(y) for free -- pixel_norm (y is its own input), the 2x2
    average pool (y is read for 4 bytes per element) and the next conv's dgrad -- takes it, applies the gate and sets
    `applied` (pixel_norm: first-order passes only; pool and dgrad also in the penalty's create_graph pass, as the
    differentiable _BcastGated / _ConvDgradGated);
    _Conv2d.backward then skips its own act_bwd pass (12 % of the GAN iteration).  Same multiply, same bits.
    The single-consumer premise holds for the reference's generator / discriminator wiring (gan.py:149-316: a conv's
    activation output feeds pixel_norm, the next conv or the pool, never two of them); it is a switch, default off,
    that GenerativeAdverserialNetwork turns on around its solver steps."""

    def __init__(self, act):
        self.act, self.applied = act, False
        self.shared = False         # set when the gated tensor gains a second consumer in a gradient graph (_PixelNorm.backward)


FUSE_ACT_GATES = False


class fuse_act_gates(object):
    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        global FUSE_ACT_GATES
        self.prev, FUSE_ACT_GATES = FUSE_ACT_GATES, self.on

    def __exit__(self, *exc):
        global FUSE_ACT_GATES
        FUSE_ACT_GATES = self.prev
        return False


def _gate_of(t):
    g = getattr(t, '_sq_act_gate', None)
    return g if isinstance(g, ActGate) else None


class _ConvDgradGated(torch.autograd.Function):
    """act_bwd(ConvDgrad(dy, w), gate, act) as ONE differentiable op: the gated dgrad kernel (ops.conv_dgrad_actgate) in passes
    that are differentiated again -- the WGAN-GP penalty's create_graph pass through D(mix) used to run the dgrad and a
    separate, differentiable act_bwd launch per conv -> leaky -> conv pair.  The gate is piecewise constant: no gradient
    flows to it, and d/d(dy) = ConvFwd(act_bwd(ddx, gate)), d/dw = Wgrad(act_bwd(ddx, gate), dy) -- the kernels the unfused
    pair's backward runs.  `out` = [the result, computed by the caller to find out whether the fused form exists]."""

    @staticmethod
    def forward(ctx, dy, w, wscale, gate, act, out):
        ctx.wscale, ctx.act, ctx.w_id = wscale, act, _pid(w)
        ctx.save_for_backward(dy, w, gate)
        return out[0]

    @staticmethod
    def backward(ctx, ddx):
        dy, w, gate = ctx.saved_tensors
        g = _ActBwd.apply(ddx.contiguous(), gate, ctx.act)
        d_dy = _ConvFwd.apply(g, w, ctx.wscale) if ctx.needs_input_grad[0] else None
        d_w = None
        if ctx.needs_input_grad[1] and _want(ctx.w_id) and not _sink_wgrad(g, dy, w.shape[0], ctx.wscale, ctx.w_id):
            d_w = _ConvWgrad.apply(g, dy, w.shape[0], ctx.wscale)
        return d_dy, d_w, None, None, None, None


class _ChannelSum(torch.autograd.Function):
    """db[c] = sum over pixels of t[..., c]."""

    @staticmethod
    def forward(ctx, t):
        ctx.shape = t.shape
        C = t.shape[-1]
        npix = t.numel() // C
        t = t.contiguous()
        if C % 4 == 0:
            return ops.wgrad1x1_small(ops._ones(npix, 1, t.device), t).view(C)
        return ops.wgrad1x1_small(t, ops._ones(npix, 4, t.device))[:, 0].contiguous()

    @staticmethod
    def backward(ctx, ddb):
        return ddb.expand(ctx.shape).contiguous()             # never on the GAN / U-Net training paths


class _Conv2d(torch.autograd.Function):
    """Fused forward kernel: act(conv2d(x, w*wscale) + bias)."""

    @staticmethod
    def forward(ctx, x, w, bias, act, wscale, gate, pn=None):
        ctx.gate = gate
        # x = the activation output of the conv in front (it hangs its ActGate on x): this conv is x's only consumer
        # (gan.py:149-316 wiring), so in first-order passes its dgrad kernel applies act'(x) in the epilogue
        ctx.in_gate = _gate_of(x) if gate is not None or FUSE_ACT_GATES else None
        if pn is not None and ops.conv2d_pixelnorm_takes(x, w):
            # the pixel norm that follows (weighted_conv2d's norm=True) comes out of the same kernel; it is handed to
            # pixel_norm() through `pn` (a one-slot list the tape does not see), which returns it instead of launching
            y, pn[1] = ops.conv2d_pixelnorm(x, w, bias, act=act, wscale=wscale, eps=pn[0])
        else:
            y = ops.conv2d(x, w, bias, act=act, wscale=wscale)
        ctx.act, ctx.wscale, ctx.has_bias = act, wscale, bias is not None
        ctx.w_id, ctx.bias_id = _pid(w), _pid(bias)
        ctx.sinks = (grad_sink(w), grad_sink(bias)) if wscale == 1.0 else (None, None)
        ctx.save_for_backward(x, w, y if ops.ACT[act] else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        gated = ctx.gate is not None and ctx.gate.applied       # the producer of dy applied act'(y) already (ActGate)
        if ctx.gate is not None:
            ctx.gate.applied = False
        dpre = _ActBwd.apply(dy, y, ctx.act) if (y is not None and not gated) else dy.contiguous()
        return _conv_backward(ctx, x, w, dpre) + (None, None, None, None)


def _conv_backward(ctx, x, w, dpre):
    """(dx, dW, db) of pre = conv2d(x, w * wscale) + bias from d(pre): shared by _Conv2d and _Conv2dPool"""
    dx = dw = db = None
    if ctx.needs_input_grad[0]:
        ig = ctx.in_gate
        if ig is not None and not ig.applied:
            dx = ops.conv_dgrad_actgate(dpre, w, ctx.wscale, x, ig.act)
            if dx is not None:
                ig.applied = True                           # the conv that produced x skips its act_bwd pass
                if torch.is_grad_enabled():                 # a pass that is differentiated again: the same kernel, on the tape
                    dx = _ConvDgradGated.apply(dpre, w, ctx.wscale, x, ig.act, [dx])
        if dx is None:
            dx = _ConvDgrad.apply(dpre, w, ctx.wscale)
    want_w = ctx.needs_input_grad[1] and _want(ctx.w_id)
    need_b = ctx.has_bias and ctx.needs_input_grad[2] and _want(ctx.bias_id)
    if not want_w and not need_b:
        return dx, None, None
    if want_w and _sink_wgrad(x, dpre, w.shape[0], ctx.wscale, ctx.w_id, ctx.bias_id, need_b):
        return dx, None, None                               # queued: dW / db land in the parameters' sinks at the flush
    if want_w and not torch.is_grad_enabled():
        # first-order fast path: dW and db from ONE pass of the wgrad kernel
        sw, sb = ctx.sinks
        Cin, Cout = w.shape[2], w.shape[3]
        if sw is None or (Cout % 4) or not (Cin % 8 == 0 or Cin == 1):   # sinks only on the MFMA wgrad kernels
            sw = sb = None
        dw, db = ops.conv_wgrad_raw(x, dpre, w.shape[0], want_bias=need_b, dw_out=sw,
                                    db_out=sb if need_b else None, dw_scale=ctx.wscale)   # w' = w * wscale (gan.py:79)
        if sw is not None:
            dw = None
            db = None if sb is not None else db
    else:
        if want_w:
            dw = _ConvWgrad.apply(x, dpre, w.shape[0], ctx.wscale)
        if need_b:
            db = _ChannelSum.apply(dpre)
    return dx, dw, db


class _Conv2dPool(torch.autograd.Function):
    """avgpool2x2(act(conv2d(x, w * wscale) + bias)): the pooled tensor is written from the conv's epilogue (ops.conv2d_avgpool),
    y stays for the backward -- the up-sampling of d(pool) through the activation's backward (the gated broadcast kernel; in
    passes that are differentiated again its tape form _BcastGated), then the conv's own backward."""

    @staticmethod
    def forward(ctx, x, w, bias, act, wscale):
        ctx.in_gate = _gate_of(x) if FUSE_ACT_GATES else None
        y, p = ops.conv2d_avgpool(x, w, bias, act=act, wscale=wscale)
        ctx.act, ctx.wscale, ctx.has_bias = act, wscale, bias is not None
        ctx.w_id, ctx.bias_id = _pid(w), _pid(bias)
        ctx.sinks = (None, None)
        ctx.save_for_backward(x, w, y)
        return p

    @staticmethod
    def backward(ctx, dp):
        x, w, y = ctx.saved_tensors
        if not ops.ACT[ctx.act]:
            dpre = _Bcast2x2.apply(dp, 0.25)
        elif torch.is_grad_enabled():
            dpre = _BcastGated.apply(dp, y, 0.25, ctx.act)
        else:
            dpre = ops.broadcast2x2_act_bwd(dp.contiguous(), y, 0.25, ctx.act)
        return _conv_backward(ctx, x, w, dpre) + (None, None)


def conv2d_avgpool(x, w, bias=None, act=None, wscale=1.0):
    """avgpool2x2(conv2d(...)) as one tape entry where the fused kernel exists (bf16 features, 3x3, even sides >= 16), else the two ops"""
    if ops._conv2d_avgpool_takes(x, w):
        return _Conv2dPool.apply(x, w, bias, act, float(wscale))
    return avgpool2x2(conv2d(x, w, bias, act, wscale))


def conv2d(x, w, bias=None, act=None, wscale=1.0, pixelnorm_eps=None):
    """pixelnorm_eps: the caller will apply pixel_norm(y, eps) next (weighted_conv2d's norm=True, gan.py:96-97): where the fused
    kernel exists the normalised tensor is produced with y and pixel_norm(y) returns it without a launch"""
    gate = ActGate(act) if (FUSE_ACT_GATES and ops.ACT[act]) else None
    pn = [float(pixelnorm_eps), None] if pixelnorm_eps is not None else None
    y = _Conv2d.apply(x, w, bias, act, float(wscale), gate, pn)
    if gate is not None:
        y._sq_act_gate = gate
    if pn is not None and pn[1] is not None:
        y._sq_pn = (pn[0], pn[1])
    return y


def dense(x, w, bias=None, act=None):
    """tf.layers.dense on (N, Cin) with w (Cin, Cout): a 1x1 convolution over a row of N "pixels"."""
    N, Cin = x.shape
    y = conv2d(x.reshape(1, 1, N, Cin), w.reshape(1, 1, Cin, w.shape[1]), bias, act=act)
    return y.reshape(N, w.shape[1])


class _Head(torch.autograd.Function):
    """1x1 conv to <= 4 channels (U-Net to_image), no activation; first-order only."""

    @staticmethod
    def forward(ctx, x, w, bias):
        y = ops.conv2d(x, w, bias, act=None)
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        ctx.sinks = (grad_sink(w), grad_sink(bias))
        return y

    @staticmethod
    def backward(ctx, dz):
        x, w = ctx.saved_tensors
        sw, sb = ctx.sinks
        dx, dw, db = ops.conv1x1_small_bwd(x, w, dz.contiguous(), want_dx=ctx.needs_input_grad[0], dw_out=sw,
                                           db_out=sb if ctx.has_bias else None)
        return dx, (None if sw is not None else dw), (db if ctx.has_bias and sb is None else None)


def conv1x1_head(x, w, bias=None):
    return _Head.apply(x, w, bias)


class _Cast(torch.autograd.Function):
    """storage cast float32 <-> bfloat16 (linear: its derivative is the cast back, to any order)"""

    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src = x.dtype
        return ops.cast(x.contiguous(), dtype)

    @staticmethod
    def backward(ctx, dy):
        return _Cast.apply(dy, ctx.src), None


def cast(x, dtype):
    return x if x.dtype == dtype else _Cast.apply(x, dtype)


# ---------------------------------------------------------------------------------------------
# pixel norm (gan.py:49-51), up to second order
# ---------------------------------------------------------------------------------------------
class _PixelNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps, gate, ready=None):
        ctx.eps, ctx.gate = eps, gate
        ctx.save_for_backward(x)
        if ready is not None:                                   # [tensor]: the conv that produced x normalised it in its epilogue
            return ready[0]
        return ops.pixelnorm(x, eps)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        if ctx.gate is not None:
            if torch.is_grad_enabled():
                # A pass that is differentiated again puts _PixelNormBwd(x, dy) on the tape: x = act(conv) now has a SECOND
                # consumer, and the pass that differentiates it sends x two gradients -- this node's and _PixelNormBwd's
                # dL/dx.  Only their sum may go through act'(x), so the gate must stay with the conv (round 4: with the
                # gate taken here the conv skipped act' for the whole sum; every from_image gradient of a discriminator
                # step whose penalty was active was wrong -- found by oracle/gan_bf16_ref.py).
                ctx.gate.shared = True
            elif not ctx.gate.shared:
                ctx.gate.applied = True                         # x = act(conv): act'(x) rides in this kernel (ActGate)
                return ops.pixelnorm_bwd(x, dy.contiguous(), ctx.eps, act=ctx.gate.act), None, None, None
        return _PixelNormBwd.apply(x, dy, ctx.eps), None, None, None


class _PixelNormBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g, eps):
        ctx.eps = eps
        g = g.contiguous()
        ctx.save_for_backward(x, g)
        return ops.pixelnorm_bwd(x, g, eps)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, v):
        x, g = ctx.saved_tensors
        dg, dx2 = ops.pixelnorm_bwd2(x, g, v.contiguous(), ctx.eps)
        return dx2, dg, None


def pixel_norm(x, epsilon=1e-8):
    # the conv that produced x may have normalised it in its epilogue (conv2d(pixelnorm_eps=)): the result rides on x ONCE --
    # it is taken off here, or x -> result -> PixelNorm node -> saved x would be a reference cycle the collector cannot see
    pn = x.__dict__.pop('_sq_pn', None) if hasattr(x, '__dict__') else None
    ready = [pn[1]] if (pn is not None and pn[0] == float(epsilon)) else None
    return _PixelNorm.apply(x, float(epsilon), _gate_of(x), ready)


# ---------------------------------------------------------------------------------------------
# 2x2 pooling / broadcasting (avg-pool, double_size) -- each other's derivatives
# ---------------------------------------------------------------------------------------------
class _Pool2x2(torch.autograd.Function):
    """scale * (sum of each 2x2 patch); scale 0.25 = tf.layers.average_pooling2d (gan.py:189-192)."""

    @staticmethod
    def forward(ctx, x, scale, gate=None):
        ctx.scale, ctx.gate = scale, gate
        if gate is not None:
            ctx.save_for_backward(x)
        if scale == 0.25 and x.shape[-1] % 4 == 0:
            return ops.avgpool2x2(x)                          # ((a+b)+(c+d))*0.25, the oracle's order
        return ops.sumpool2x2(x.contiguous(), scale)

    @staticmethod
    def backward(ctx, dy):
        if ctx.gate is not None and ctx.saved_tensors[0].shape[-1] % 4 == 0:
            (x,) = ctx.saved_tensors                            # x = act(conv): act'(x) rides in the up-sampling (ActGate)
            ctx.gate.applied = True
            if torch.is_grad_enabled():                         # a pass that is differentiated again: the same kernel, on the tape
                return _BcastGated.apply(dy, x, ctx.scale, ctx.gate.act), None, None
            return ops.broadcast2x2_act_bwd(dy.contiguous(), x, ctx.scale, ctx.gate.act), None, None
        return _Bcast2x2.apply(dy, ctx.scale), None, None


class _BcastGated(torch.autograd.Function):
    """act_bwd(scale * up-sampling of dy, gate) as one differentiable op (the gated form of _Pool2x2's backward in passes that
    are differentiated again); d/d(dy) = Pool(act_bwd(dd, gate)), nothing flows to the gate"""

    @staticmethod
    def forward(ctx, dy, gate, scale, act):
        ctx.scale, ctx.act = scale, act
        ctx.save_for_backward(gate)
        return ops.broadcast2x2_act_bwd(dy.contiguous(), gate, scale, act)

    @staticmethod
    def backward(ctx, dd):
        (gate,) = ctx.saved_tensors
        return _Pool2x2.apply(_ActBwd.apply(dd.contiguous(), gate, ctx.act), ctx.scale, None), None, None, None


class _Bcast2x2(torch.autograd.Function):
    """scale * nearest-neighbour 2x up-sampling; scale 1 = double_size (gan.py:133-136)."""

    @staticmethod
    def forward(ctx, x, scale):
        ctx.scale = scale
        return ops.broadcast2x2(x.contiguous(), scale)

    @staticmethod
    def backward(ctx, dy):
        return _Pool2x2.apply(dy.contiguous(), ctx.scale, None), None


def avgpool2x2(x):
    return _Pool2x2.apply(x, 0.25, _gate_of(x))


def double_size(x):
    return _Bcast2x2.apply(x, 1.0)


class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.maxpool2x2(x)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.maxpool2x2_bwd(x, dy.contiguous())


def maxpool2x2(x):
    return _MaxPool.apply(x)


# ---------------------------------------------------------------------------------------------
# per-sample scaling / dot products and blends (fade-in, real/fake interpolation, penalty norm)
# ---------------------------------------------------------------------------------------------
class _ScalePerSample(torch.autograd.Function):
    """y[n] = s[n] * x[n]  (or (1 - s[n]) * x[n])."""

    @staticmethod
    def forward(ctx, x, s, one_minus):
        ctx.one_minus = one_minus
        ctx.save_for_backward(x, s)
        return ops.scale(x.contiguous(), s, one_minus=one_minus)

    @staticmethod
    def backward(ctx, dy):
        x, s = ctx.saved_tensors
        dy = dy.contiguous()
        dx = _ScalePerSample.apply(dy, s, ctx.one_minus) if ctx.needs_input_grad[0] else None
        ds = None
        if ctx.needs_input_grad[1]:
            ds = _DotPerSample.apply(dy, x)
            if ctx.one_minus:
                ds = -ds
        return dx, ds, None


class _DotPerSample(torch.autograd.Function):
    """out[n] = sum_i a[n,i] * b[n,i]."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return ops.dot_per_sample(a.contiguous(), b.contiguous())

    @staticmethod
    def backward(ctx, dout):
        a, b = ctx.saved_tensors
        dout = dout.contiguous()
        da = _ScalePerSample.apply(b, dout, False) if ctx.needs_input_grad[0] else None
        db = _ScalePerSample.apply(a, dout, False) if ctx.needs_input_grad[1] else None
        return da, db


def scale_per_sample(x, s, one_minus=False):
    return _ScalePerSample.apply(x, s, one_minus)


class _SqNormPerSample(torch.autograd.Function):
    """out[n] = sum_i a[n,i]^2: dot_per_sample(a, a) as a ONE-input op -- with two inputs the tape sends `a` two equal gradients
    (two per-sample scalings of an image-sized tensor and the framework's add of them); here one scaling by 2 dout"""

    @staticmethod
    def forward(ctx, a):
        a = a.contiguous()
        ctx.save_for_backward(a)
        return ops.dot_per_sample(a, a)

    @staticmethod
    def backward(ctx, dout):
        (a,) = ctx.saved_tensors
        return _ScalePerSample.apply(a, dout + dout, False)


def dot_per_sample(a, b):
    if a is b:
        return _SqNormPerSample.apply(a)
    return _DotPerSample.apply(a, b)


def lerp(a, b, alpha, out=None):
    """alpha*a + (1-alpha)*b with scalar alpha (fade-in, gan.py:687-694) or a per-sample (N,)
    tensor (r of gan.py:709-714).  One fused kernel forward; gradients are per-sample scalings.
    out: destination the result is written into (a contiguous view, e.g. the first half of a stacked batch)."""
    if not isinstance(alpha, torch.Tensor):
        alpha = torch.full((a.shape[0],), float(alpha), dtype=torch.float32, device=a.device)
    return _Lerp.apply(a, b, alpha, None if out is None else [out])


class _Lerp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, alpha, out=None):
        ctx.save_for_backward(alpha)
        if out is None:
            return ops.lerp(a.contiguous(), b.contiguous(), alpha)
        # `out` ([tensor], opaque to the tape): the caller's buffer is written, and what the tape sees is a NEW tensor over
        # the same memory -- not a view of the buffer, so no in-place-on-a-view bookkeeping (CopySlices) enters the graph
        dst = out[0]
        ops.lerp(a.contiguous(), b.contiguous(), alpha, out=dst)
        return torch.empty(0, dtype=dst.dtype, device=dst.device).set_(dst.untyped_storage(), dst.storage_offset(), dst.shape,
                                                                       dst.stride())

    @staticmethod
    def backward(ctx, dy):
        (alpha,) = ctx.saved_tensors
        da = _ScalePerSample.apply(dy, alpha, False) if ctx.needs_input_grad[0] else None
        db = _ScalePerSample.apply(dy, alpha, True) if ctx.needs_input_grad[1] else None
        return da, db, None, None


# ---------------------------------------------------------------------------------------------
# U-Net-only pieces (first order)
# ---------------------------------------------------------------------------------------------
class _ConvT(torch.autograd.Function):
    """2x2/s2 transpose conv + bias.  Backward = space-to-depth, then a 1x1 dgrad and a 1x1 wgrad:
    convT(x) == depth_to_space(conv1x1(x, W')) with W'[c][(2a+b)*Cout + o] = W[a,b,o,c]."""

    @staticmethod
    def forward(ctx, x, w, bias):
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        ctx.sinks = (grad_sink(w), grad_sink(bias))
        return ops.convT2x2s2(x, w, bias)

    @staticmethod
    @torch.autograd.function.once_differentiable
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
            dw, db = convT_param_grads(dwp, dbp if ctx.has_bias else None, Cin, Cout, ctx.sinks)
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
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        a, b = ctx.saved_tensors if ctx.kind == 'eltwise_mul' else (None, None)
        da, db = ops.bridge_bwd(dy.contiguous(), a, b, ctx.kind)
        return da, db, None


def bridge(a, b, kind):
    return _Bridge.apply(a, b, kind)


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rate, seed, mask, step_dev):
        y, m = ops.dropout_fwd(x, rate, seed=seed, mask=mask, step_dev=step_dev)
        ctx.rate = rate
        ctx.save_for_backward(m)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        (m,) = ctx.saved_tensors
        return ops.dropout_bwd(dy.contiguous(), m, ctx.rate), None, None, None, None


def dropout(x, rate, seed=0, mask=None, step_dev=None):
    if rate <= 0.0:
        return x
    return _Dropout.apply(x, float(rate), int(seed), mask, step_dev)


class _WeightedSoftmaxCE(torch.autograd.Function):
    """loss (0-d float32 tensor) with the fused forward+backward kernel; d loss/d logits is
    computed in the same pass and scaled by the incoming gradient in backward."""

    @staticmethod
    def forward(ctx, logits, onehot, weights):
        loss, dz = ops.wsoftmax_ce(logits, onehot, weights, want_grad=True)
        ctx.save_for_backward(dz)
        return loss.to(torch.float32)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dloss):
        (dz,) = ctx.saved_tensors
        return dz * dloss, None, None


def weighted_softmax_cross_entropy(logits, onehot, weights):
    return _WeightedSoftmaxCE.apply(logits, onehot, weights)


class _BatchNormTrain(torch.autograd.Function):
    """y = act(BN(x)) with batch statistics; updates the moving statistics in place (first order only)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, moving_mean, moving_var, eps, momentum, act):
        mean, var = ops.bn_stats(x)
        scale, shift = ops.bn_fold(gamma, beta, mean, var, eps)
        y = ops.bn_apply(x, scale, shift, act)
        ops.bn_update_moving_(moving_mean, moving_var, mean, var, x.numel() // x.shape[-1], momentum)
        ctx.save_for_backward(x, y, mean, var, gamma)
        ctx.eps, ctx.act = eps, act
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, mean, var, gamma = ctx.saved_tensors
        dx, dgamma, dbeta = ops.bn_bwd(x, dy.contiguous(), y, ctx.act, mean, var, gamma, ctx.eps)
        return dx, dgamma, dbeta, None, None, None, None, None


def batch_norm_train(x, gamma, beta, moving_mean, moving_var, eps=ops.BN_EPS, momentum=ops.BN_MOMENTUM, act=None):
    return _BatchNormTrain.apply(x, gamma, beta, moving_mean, moving_var, float(eps), float(momentum), act)


class _ConvT3x3(torch.autograd.Function):
    """3x3 / stride-2 SAME transpose conv (first order): zero insertion + SAME conv with the rotated,
    transposed kernel.  Backward: dX = odd samples of conv(dY, w as HWIO) (the strided conv this op is the
    adjoint of), dW = transform(wgrad(u, dY))."""

    @staticmethod
    def forward(ctx, x, w, bias):
        u = ops.zero_insert2x(x)
        y = ops.conv2d(u, ops.conv_weight_transform(w), bias, act=None)
        ctx.save_for_backward(u, w)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        u, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = ops.gather_odd2x(ops.conv2d(dy, w, None, act=None)) if ctx.needs_input_grad[0] else None
        dwt, db = ops.conv2d_wgrad(u, dy, 3, want_bias=ctx.has_bias)
        return dx, ops.conv_weight_transform(dwt), db


def convT3x3s2(x, w, bias=None):
    return _ConvT3x3.apply(x, w, bias)


# ---------------------------------------------------------------------------------------------
# minibatch-stdev feature map (gan.py:204-212), up to second order; WGAN-GP loss algebra (gan.py:715-729)
# ---------------------------------------------------------------------------------------------
class _MbStdMap(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, groups, cells):
        x = x.contiguous()
        ctx.groups = groups
        ctx.save_for_backward(x)
        return ops.mbstd_map(x, groups, cells)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return _MbStdMapBwd.apply(x, dy.contiguous(), ctx.groups), None, None


class _MbStdMapBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dy, groups):
        ctx.groups = groups
        ctx.save_for_backward(x, dy)
        return ops.mbstd_map_bwd(x, dy, groups)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, v):
        x, dy = ctx.saved_tensors
        ddy, dx2 = ops.mbstd_map_bwd2(x, dy, v.contiguous(), ctx.groups)
        return dx2, ddy, None


def mbstd_map(x, groups=1, cells=16):
    """(N, cells) map holding the minibatch statistic of each of `groups` stacked minibatches; differentiable twice"""
    return _MbStdMap.apply(x, int(groups), int(cells))


class _HeadConcat(torch.autograd.Function):
    """concat([float(conv), mb], -1) flattened per sample (the discriminator's output block, gan.py:213-226) on bf16 features:
    one kernel instead of cast + cat, and ONE adjoint kernel (_HeadSplit) instead of two slice copies, a cast and the slices'
    zero-filled gradients.  Linear: the two are each other's derivatives."""

    @staticmethod
    def forward(ctx, conv, mb):
        ctx.shapes = (tuple(conv.shape), tuple(mb.shape))
        return ops._gb().head_concat(conv.contiguous(), mb.contiguous())

    @staticmethod
    def backward(ctx, dflat):
        return _HeadSplit.apply(dflat, ctx.shapes[0], ctx.shapes[1])


class _HeadSplit(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dflat, conv_shape, mb_shape):
        return ops._gb().head_split(dflat.contiguous(), conv_shape, mb_shape)

    @staticmethod
    def backward(ctx, ddconv, ddmb):
        return _HeadConcat.apply(ddconv, ddmb), None, None


def head_concat(conv, mb):
    """(N, P (C + 1)) float32 rows of the discriminator's dense head from the bf16 conv output (N,h,w,C) and the (N, P) f32
    minibatch-stdev map"""
    return _HeadConcat.apply(conv, mb)


class _WganLosses(torch.autograd.Function):
    """stacked_n > 0: `Dz` is the (2n,) output of ONE discriminator pass over the stacked [generated; real] minibatches
    and Dx is None -- the two halves are read in place and the backward returns ONE (2n,) gradient (two sliced views
    cost the tape two zero fills, two copies and an add)."""

    @staticmethod
    def forward(ctx, Dz, Dx, gn2, stacked_n):
        ctx.set_materialize_grads(False)                        # an unused loss's gradient arrives as None, not as a filled zero
        Dz = Dz.contiguous()
        ctx.stacked_n = stacked_n
        if stacked_n:
            Dz, Dx = Dz[:stacked_n], Dz[stacked_n:]
        Dx = Dx.contiguous() if Dx is not None else None
        gn2 = gn2.contiguous() if gn2 is not None else None
        ctx.save_for_backward(Dz, Dx, gn2)
        out = ops.wgan_losses(Dz, Dx, gn2)
        return out[0], out[1]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gd, gg):
        Dz, Dx, gn2 = ctx.saved_tensors
        gd = gd.contiguous() if gd is not None else None
        gg = gg.contiguous() if gg is not None else None
        if ctx.stacked_n:
            n = ctx.stacked_n
            dall = torch.empty((2 * n,), dtype=Dz.dtype, device=Dz.device)
            _, _, dgn2 = ops.wgan_losses_bwd(Dz, Dx, gn2, gd, gg, out_dz=dall[:n], out_dx=dall[n:])
            return dall, None, dgn2, None
        dDz, dDx, dgn2 = ops.wgan_losses_bwd(Dz, Dx, gn2, gd, gg)
        return dDz, dDx, dgn2, None


def wgan_losses(Dz, Dx=None, gn2=None):
    """(d_loss, g_loss) of gan.py:715-729 from the discriminator outputs and the per-sample squared gradient norm of
    D(mix); with Dz alone: (unused, g_loss = mean(-Dz))"""
    return _WganLosses.apply(Dz, Dx, gn2, 0)


def wgan_losses_stacked(Dzx, gn2):
    """the same from the (2n,) output of one discriminator pass over [generated; real] stacked along the batch axis"""
    return _WganLosses.apply(Dzx, None, gn2, Dzx.shape[0] // 2)
