"""Rounding-point emulation of the progressive WGAN-GP in BASELINE config 5's own dtype -- TEST INFRASTRUCTURE ONLY
(imported by tests/ alone; the product never loads anything under oracle/).

What it restates: sequitr/networks/gan.py:44-136 (leaf ops), :149-240 (discriminator), :246-316 (generator),
:665-732 (fade-in, interpolated sample, one-sided WGAN-GP penalty, drift term), the same graph as
oracle/torch_gan_ref.py.  What it adds: every value the HIP path STORES as bfloat16 -- feature maps and feature-map
gradients, in the forward pass, the backward pass and the penalty's second-order pass -- and every operand the HIP
path ROUNDS to bfloat16 on its way into a matrix core is rounded here at the same point (round-to-nearest-even,
f64 -> f32 -> bf16: the kernels compute in f32 and round once per stored value); everything between two rounding
points is float64 arithmetic on torch-CPU.  So

    fp64 reference      vs this emulation : the effect of the roundings alone (what "it is rounding" must mean), and
    this emulation      vs the HIP path   : everything else -- f32 accumulation order and any defect.

Parity unpinned against the reference itself (TensorFlow is not importable here; the reference holds no GAN
fixtures): the emulation with rounding switched off is pinned to oracle/torch_gan_ref.py's plain autograd
(tests/test_gan_bf16_emulation.py, CPU), which is pinned to the reference by source text only.

Storage rule restated (sequitr_amd/ops_gan_bf16.py header): a tensor with >= 8 channels between the generator's first
feature map and the discriminator's last convolution is bf16 ('b' below), images (<= 4 channels), the generator's
latent block up to its pixel norm, the discriminator's output block from the minibatch statistic on, parameters,
parameter gradients and losses are f32 ('f').  Operand rule of the convolution family (ops.conv2d / conv_dgrad_raw /
conv_wgrad_raw dispatch) is `conv_policy` below.

The derivative structure is the closed operator set of sequitr_amd/functional.py written out again on torch-CPU
(ConvFwd / ConvDgrad / ConvWgrad are each other's derivatives, pixel norm up to second order, pool <-> broadcast),
because WHERE a gradient is rounded is a property of that structure; each leaf is evaluated with torch's own
float64 kernels (F.conv2d, torch.nn.grad.conv2d_weight, autograd for the pixel-norm second order), none of it shared
with the product.
"""
import numpy as np
import torch
import torch.nn.functional as TF
from torch.autograd import Function

DT = torch.float64
ROUND = True                 # False: no rounding anywhere -- must equal oracle/torch_gan_ref.py's autograd
# OBSERVER(kind, inputs: dict, output): called by every leaf evaluation, in every pass (forward, the create_graph
# backward, the second-order backward), with the operands it read, their storage classes and the value it stored --
# tests replay each one through the HIP operator of the same name on the SAME operands (a comparison that does not
# compound roundings from layer to layer, see tests/test_gpu_gan_bf16.py)
OBSERVER = None


def _obs(kind, out, **inputs):
    if OBSERVER is not None:
        OBSERVER(kind, inputs, out)
    return out


def q(t, s='b'):
    """the value `t` takes when stored in class `s`: 'b' = bfloat16 (RNE through f32), 'f' = left as computed"""
    if s != 'b' or not ROUND:
        return t
    return t.to(torch.float32).to(torch.bfloat16).to(DT)


# ---- convolution family -------------------------------------------------------------------------------------------
def conv_policy(kind, K, ci, co, npix, s_in, s_out):
    """(round the tensor operand(s), round the filter operand, class of the result) of one convolution-family
    launch, restating the product's dispatch.  kind 'conv': a forward-shaped launch from `ci` to `co` channels (the
    forward conv, or a dgrad seen as the conv it is run as); 'wgrad': x has ci channels (class s_in), dy co (s_out).
      * a bf16 tensor on either side (ops_gan_bf16.conv2d / conv_wgrad): image -> features (f32 in, 1x1) and features ->
        image (<= 4 channels out, 1x1) multiply in f32 with the f32 filter; feature -> feature convs take the packed
        bf16 filter; weight gradients take both tensors as stored.
      * f32 tensors on both sides (the dense layers in row form): the split-reduction f32 kernels where the
        reduction is long (>= 1024 inputs forward; Cin*Cout >= 65536 for the weight gradient) and <= 128 rows,
        else the "mixed" kernels (operands rounded on the way into the matrix core) where their channel rules
        hold (conv: ci % 8 == 0 and co % 4 == 0; wgrad: both % 16), else exact f32."""
    if kind == 'conv':
        if s_in == 'b' or s_out == 'b':
            if s_in == 'f' or co <= 4:
                return False, False, s_out
            return False, True, s_out                            # the tensor operand is bf16 already
        if K == 1 and npix <= 128 and ci >= 1024 and ci % 4 == 0:
            return False, False, 'f'
        if ci % 8 == 0 and co % 4 == 0:
            return True, True, 'f'
        return False, False, 'f'
    if s_in == 'b' or s_out == 'b':
        return False, False, 'f'
    if K == 1 and npix <= 128 and ci * co >= (1 << 16):
        return False, False, 'f'
    if ci % 16 == 0 and co % 16 == 0:
        return True, False, 'f'
    return False, False, 'f'


def _nchw(t):
    return t.permute(0, 3, 1, 2)


def _nhwc(t):
    return t.permute(0, 2, 3, 1)


def _conv(x, w_hwio, K):
    return _nhwc(TF.conv2d(_nchw(x), w_hwio.permute(3, 2, 0, 1), None, padding=K // 2))


def _conv_adj(dy, w_hwio, K):
    """adjoint in x of _conv(., w)"""
    return _nhwc(TF.conv_transpose2d(_nchw(dy), w_hwio.permute(3, 2, 0, 1), None, padding=K // 2))


def _conv_wg(x, dy, K):
    cin, cout = x.shape[-1], dy.shape[-1]
    g = torch.nn.grad.conv2d_weight(_nchw(x).contiguous(), (cout, cin, K, K), _nchw(dy).contiguous(), padding=K // 2)
    return g.permute(2, 3, 1, 0)


def _npix(t):
    return t.numel() // t.shape[-1]


class ConvFwd(Function):
    """y = conv(x, w * ws): linear in x and w; sx / so: storage classes of x and y"""

    @staticmethod
    def forward(ctx, x, w, ws, sx, so):
        K, _, ci, co = w.shape
        rt, rw, sr = conv_policy('conv', K, ci, co, _npix(x), sx, so)
        ctx.a = (ws, sx, so)
        ctx.save_for_backward(x, w)
        return _obs('conv', q(_conv(q(x) if rt else x, q(w * ws) if rw else w * ws, K), sr), x=x, w=w, ws=ws, sx=sx, so=so)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        ws, sx, so = ctx.a
        dx = ConvDgrad.apply(dy, w, ws, sx, so) if ctx.needs_input_grad[0] else None
        dw = ConvWgrad.apply(x, dy, w.shape[0], ws, sx, so) if ctx.needs_input_grad[1] else None
        return dx, dw, None, None, None


class ConvDgrad(Function):
    """dx = adjoint of ConvFwd in x, run by the product as a forward-shaped conv from co to ci channels"""

    @staticmethod
    def forward(ctx, dy, w, ws, sx, so):
        K, _, ci, co = w.shape
        rt, rw, sr = conv_policy('conv', K, co, ci, _npix(dy), so, sx)
        ctx.a = (ws, sx, so)
        ctx.save_for_backward(dy, w)
        return _obs('dgrad', q(_conv_adj(q(dy) if rt else dy, q(w * ws) if rw else w * ws, K), sr), dy=dy, w=w, ws=ws, sx=sx, so=so)

    @staticmethod
    def backward(ctx, ddx):
        dy, w = ctx.saved_tensors
        ws, sx, so = ctx.a
        d_dy = ConvFwd.apply(ddx, w, ws, sx, so) if ctx.needs_input_grad[0] else None
        d_w = ConvWgrad.apply(ddx, dy, w.shape[0], ws, sx, so) if ctx.needs_input_grad[1] else None
        return d_dy, d_w, None, None, None


class ConvWgrad(Function):
    """dw = ws * sum_p x[p + tap] (x) dy[p]: bilinear in x and dy, f32 result"""

    @staticmethod
    def forward(ctx, x, dy, K, ws, sx, so):
        rt, _, _ = conv_policy('wgrad', K, x.shape[-1], dy.shape[-1], _npix(x), sx, so)
        ctx.a = (K, ws, sx, so)
        ctx.save_for_backward(x, dy)
        return _obs('wgrad', _conv_wg(q(x) if rt else x, q(dy) if rt else dy, K) * ws, x=x, dy=dy, K=K, ws=ws, sx=sx, so=so)

    @staticmethod
    def backward(ctx, ddw):
        x, dy = ctx.saved_tensors
        K, ws, sx, so = ctx.a
        d_x = ConvDgrad.apply(dy, ddw, ws, sx, so) if ctx.needs_input_grad[0] else None
        d_dy = ConvFwd.apply(x, ddw, ws, sx, so) if ctx.needs_input_grad[1] else None
        return d_x, d_dy, None, None, None, None


def _slope(y, act):
    return torch.where(y > 0, torch.ones_like(y), torch.full_like(y, 0.2)) if act else None


class ActBwd(Function):
    """dpre = dy * leaky'(pre), decided from the stored activation OUTPUT y; a stored pass of its own (class s)"""

    @staticmethod
    def forward(ctx, dy, y, s):
        ctx.s = s
        ctx.save_for_backward(y)
        return _obs('act_bwd', q(dy * _slope(y, True), s), dy=dy, y=y, s=s)

    @staticmethod
    def backward(ctx, dd):
        (y,) = ctx.saved_tensors
        return ActBwd.apply(dd, y, ctx.s), None, None


class ConvAct(Function):
    """weighted_conv2d's fused forward (gan.py:86-95): y = store(act(conv(x, w * ws) + bias)) -- ONE rounding, at the
    end; backward = ActBwd (its own stored pass) then the linear conv's derivatives, bias gradient = channel sums of
    the stored d(pre)"""

    @staticmethod
    def forward(ctx, x, w, b, ws, act, sx, so):
        K, _, ci, co = w.shape
        rt, rw, sr = conv_policy('conv', K, ci, co, _npix(x), sx, so)
        pre = _conv(q(x) if rt else x, q(w * ws) if rw else w * ws, K) + b.reshape(-1)
        y = q(TF.leaky_relu(pre, 0.2) if act else pre, sr)
        ctx.a = (ws, act, sx, so, tuple(b.shape))
        ctx.save_for_backward(x, w, y)
        return _obs('conv_act', y, x=x, w=w, b=b, ws=ws, act=act, sx=sx, so=so)

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        ws, act, sx, so, bshape = ctx.a
        dpre = ActBwd.apply(dy, y, so) if act else dy
        dx = ConvDgrad.apply(dpre, w, ws, sx, so) if ctx.needs_input_grad[0] else None
        dw = ConvWgrad.apply(x, dpre, w.shape[0], ws, sx, so) if ctx.needs_input_grad[1] else None
        db = None
        if ctx.needs_input_grad[2]:
            db = _obs('bias_grad', dpre.sum(tuple(range(dpre.dim() - 1))).reshape(bshape), x=x, dpre=dpre, K=w.shape[0], sx=sx, so=so)
        return dx, dw, db, None, None, None, None


# ---- pixel norm, up to second order (gan.py:49-51) -------------------------------------------------------------------
def _pn(x, eps):
    return x * torch.rsqrt(torch.mean(x * x, dim=-1, keepdim=True) + eps)


class PixelNorm(Function):
    @staticmethod
    def forward(ctx, x, eps, s):
        ctx.a = (eps, s)
        ctx.save_for_backward(x)
        return _obs('pixelnorm', q(_pn(x, eps), s), x=x, eps=eps, s=s)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return PixelNormBwd.apply(x, g, *ctx.a), None, None


def _pn_vjp(x, g, eps):
    with torch.enable_grad():
        x_ = x.detach().requires_grad_(True)
        (dx,) = torch.autograd.grad(_pn(x_, eps), x_, g.detach())
    return dx


class PixelNormBwd(Function):
    """dx of pixel norm as a stored pass; its own derivative (the penalty's second-order pass) by autograd on the
    float64 expression, both results stored"""

    @staticmethod
    def forward(ctx, x, g, eps, s):
        ctx.a = (eps, s)
        ctx.save_for_backward(x, g)
        return _obs('pixelnorm_bwd', q(_pn_vjp(x, g, eps), s), x=x, g=g, eps=eps, s=s)

    @staticmethod
    def backward(ctx, v):
        x, g = ctx.saved_tensors
        eps, s = ctx.a
        with torch.enable_grad():
            g_ = g.detach().requires_grad_(True)
            x_ = x.detach().requires_grad_(True)
            (dx,) = torch.autograd.grad(_pn(x_, eps), x_, g_, create_graph=True)
            dx2, dg = torch.autograd.grad(dx, (x_, g_), v.detach())
        dx2, dg = _obs('pixelnorm_bwd2', (q(dx2, s), q(dg, s)), x=x, g=g, v=v, eps=eps, s=s)
        return dx2, dg, None, None


# ---- 2x2 pooling <-> broadcasting (gan.py:133-136, 189-192) ----------------------------------------------------------
class Pool(Function):
    @staticmethod
    def forward(ctx, x, scale, s):
        ctx.a = (scale, s)
        a, b = x[:, 0::2, 0::2], x[:, 0::2, 1::2]
        c, d = x[:, 1::2, 0::2], x[:, 1::2, 1::2]
        return _obs('pool', q(scale * ((a + b) + (c + d)), s), x=x, scale=scale, s=s)

    @staticmethod
    def backward(ctx, dy):
        return Bcast.apply(dy, *ctx.a), None, None


class Bcast(Function):
    @staticmethod
    def forward(ctx, x, scale, s):
        ctx.a = (scale, s)
        return _obs('bcast', q(scale * x, s).repeat_interleave(2, 1).repeat_interleave(2, 2), x=x, scale=scale, s=s)

    @staticmethod
    def backward(ctx, dy):
        return Pool.apply(dy, *ctx.a), None, None


class Cast(Function):
    """storage cast f32 <-> bf16: linear, its derivative is the cast back"""

    @staticmethod
    def forward(ctx, x, s_from, s_to):
        ctx.a = (s_from, s_to)
        return _obs('cast', q(x, s_to), x=x, s_from=s_from, s_to=s_to)

    @staticmethod
    def backward(ctx, g):
        s_from, s_to = ctx.a
        return Cast.apply(g, s_to, s_from), None, None


class GradStore(Function):
    """identity on a stored tensor whose gradient arrives from SEVERAL nodes of the gradient graph (the conv output in front
    of a pixel norm: in the second-order pass both the norm's backward and the norm's second backward send it a
    gradient): the framework adds the bf16 contributions and stores the sum as bf16"""

    @staticmethod
    def forward(ctx, x, s):
        ctx.s = s
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        return _obs('grad_sum', q(g, ctx.s), g=g, s=ctx.s), None


class Fork(Function):
    """a stored bf16 tensor with TWO differentiable consumers: the framework adds the two bf16 gradients (f32 add,
    stored as bf16)"""

    @staticmethod
    def forward(ctx, x, s):
        ctx.s = s
        return x.clone(), x.clone()

    @staticmethod
    def backward(ctx, g1, g2):
        return _Add.apply(g1, g2, ctx.s), None


class _Add(Function):
    @staticmethod
    def forward(ctx, a, b, s):
        return q(a + b, s)

    @staticmethod
    def backward(ctx, g):
        return g, g, None


# ---- layers ---------------------------------------------------------------------------------------------------------
def wscale(k):
    kh, kw, _, cout = k.shape
    return float(np.float32(np.sqrt(np.float32(2.0 / float(kh * kw * cout)))))


def wconv(x, W, name, sx, so, act=True, norm=True):
    """weighted_conv2d (gan.py:61-99): fused conv + bias + activation, then pixel norm as a stored pass"""
    k = W[name + '/filter']
    y = ConvAct.apply(x, k, W[name + '/bias'], wscale(k), act, sx, so)
    return PixelNorm.apply(GradStore.apply(y, so), 1e-8, so) if norm else y


def dense(x, W, name, act=False):
    """tf.layers.dense on (N, Cin): the 1x1 conv over a row of N pixels the product runs it as (f32 both sides)"""
    k = W[name + '/kernel']
    n, cin = x.shape
    y = ConvAct.apply(x.reshape(1, 1, n, cin), k.reshape(1, 1, cin, k.shape[1]), W[name + '/bias'], 1.0, act, 'f', 'f')
    return y.reshape(n, k.shape[1])


def half_size(x):
    """resize_nearest_neighbor(align_corners=True) to H/2 (gan.py:128-131)"""
    n, h, w, c = x.shape
    ho, wo = h // 2, w // 2
    ys = [min(int(np.round(np.float32(i) * np.float32(h - 1) / np.float32(ho - 1))) if ho > 1 else 0, h - 1) for i in range(ho)]
    xs = [min(int(np.round(np.float32(i) * np.float32(w - 1) / np.float32(wo - 1))) if wo > 1 else 0, w - 1) for i in range(wo)]
    return x[:, ys][:, :, xs]


def up2_image(x):
    return x.repeat_interleave(2, 1).repeat_interleave(2, 2)


def generator(z, W, filters, want=(-2, -1)):
    """gan.py:246-316; returns {level index: image} for the levels in `want` (negative = from the end): only the
    to_image ops the losses read are evaluated, as a TF-1 session would"""
    p = 'GAN/generator/'
    nl = len(filters)
    want = {(l if l >= 0 else nl + l) for l in want if -nl <= l < nl}
    d = dense(PixelNorm.apply(z.reshape(z.shape[0], -1), 1e-8, 'f'), W, p + 'latent/dense1', act=True)
    x = PixelNorm.apply(d.reshape(-1, 4, 4, filters[0]), 1e-8, 'f')
    x = Cast.apply(x, 'f', 'b')                                  # bf16 storage starts at the first feature map
    feats = [wconv(x, W, p + 'latent/conv', 'b', 'b')]
    for l, f in enumerate(filters[1:]):
        src = feats[-1]
        if l in want:                                            # level l's features feed its to_image AND the next block
            src, feats[-1] = Fork.apply(src, 'b')
        u = Bcast.apply(src, 1.0, 'b')
        c1 = wconv(u, W, p + 'layer_%d/conv1' % l, 'b', 'b')
        feats.append(wconv(c1, W, p + 'layer_%d/conv2' % l, 'b', 'b'))
    return {l: wconv(feats[l], W, p + 'to_image/to_image%d' % l, 'b', 'f', act=False, norm=False) for l in sorted(want)}


def discriminator(x, W, filters):
    """gan.py:149-240 on one minibatch; returns logits (N,)"""
    p = 'GAN/discriminator/'
    nl = len(filters)
    x = wconv(x, W, p + 'from_image/from_image%d' % (nl - 1), 'f', 'b')
    for l, f in enumerate(filters[1:]):
        s = p + 'layer_%d/' % (nl - l - 1)
        c1 = wconv(x, W, s + 'conv1', 'b', 'b', norm=False)
        x = Pool.apply(wconv(c1, W, s + 'conv2', 'b', 'b', norm=False), 0.25, 'b')
    xa, xb = Fork.apply(x, 'b')
    xf = Cast.apply(xa, 'b', 'f')                                # the statistic is taken of the f32 copy
    var = xf.var(dim=0, unbiased=False).mean()
    mb = torch.ones((x.shape[0], 4, 4, 1), dtype=DT) * torch.sqrt(var)
    c = Cast.apply(wconv(xb, W, p + 'output/conv', 'b', 'b', norm=False), 'b', 'f')
    flat = torch.cat([c, mb], -1).reshape(-1, 16 * (filters[-1] + 1))
    h = dense(flat, W, p + 'output/dense', act=True)
    return dense(h, W, p + 'output/logits').reshape(-1)


def losses(X, Z, alpha, r, W, filters, level, details=None):
    """(Gz_raw, d_loss, g_loss) of gan.py:665-732 at `level`; W: {name: float64 tensor}.  details: a dict that receives
    'grad_norm' (N,), the per-sample |d D(mix) / d mix| -- the one-sided penalty is active only where it exceeds 1"""
    f = filters[:level + 1]
    imgs = generator(Z, W, f)
    Gz_raw = imgs[level]
    Xr = X
    if level > 0:
        Gz = alpha * Gz_raw + (1. - alpha) * up2_image(imgs[level - 1])
        Xr = alpha * X + (1. - alpha) * up2_image(half_size(X))
    else:
        Gz = Gz_raw
    df = f[::-1]
    Dz, Dx = discriminator(Gz, W, df), discriminator(Xr, W, df)
    rr = r.reshape(-1, 1, 1, 1)
    mix = (rr * Xr + (1 - rr) * Gz).detach().requires_grad_(True)
    Dmix = discriminator(mix, W, df)
    grad = torch.autograd.grad(Dmix.sum(), mix, create_graph=True)[0]
    gn = torch.sqrt((grad * grad).sum((1, 2, 3)))
    if details is not None:
        details['grad_norm'] = gn.detach().clone()
    pen = 10.0 * torch.square(torch.clamp(gn - 1.0, min=0.0))
    g_loss = torch.mean(-Dz)
    d_loss = torch.mean(-Dx + Dz + pen + 0.001 * torch.square(Dx))
    return Gz_raw, d_loss, g_loss


def generator_loss(X, Z, alpha, W, filters, level):
    """g_loss alone, as the generator step evaluates it (gan.py:650: through the discriminator, Gz attached)"""
    f = filters[:level + 1]
    imgs = generator(Z, W, f)
    Gz = imgs[level]
    if level > 0:
        Gz = alpha * Gz + (1. - alpha) * up2_image(imgs[level - 1])
    return torch.mean(-discriminator(Gz, W, f[::-1]))


def variable_shapes(filters, level):
    """{name: shape} of every variable the level-`level` losses touch (SURVEY A.2 / A.7 scopes; gan.py:163-168 block
    naming: discriminator blocks are layer_L .. layer_1), for tests that need weights without the product"""
    f = list(filters[:level + 1])
    out = {}
    g, d = 'GAN/generator/', 'GAN/discriminator/'

    def conv(name, k, ci, co):
        out[name + '/filter'], out[name + '/bias'] = (k, k, ci, co), (1, 1, 1, co)
    out[g + 'latent/dense1/kernel'], out[g + 'latent/dense1/bias'] = (512, 16 * f[0]), (16 * f[0],)
    conv(g + 'latent/conv', 3, f[0], f[0])
    for l in range(level):
        conv(g + 'layer_%d/conv1' % l, 3, f[l], f[l + 1])
        conv(g + 'layer_%d/conv2' % l, 3, f[l + 1], f[l + 1])
    for l in range(max(level - 1, 0), level + 1):
        conv(g + 'to_image/to_image%d' % l, 1, f[l], 2)
    df = f[::-1]
    nl = len(df)
    conv(d + 'from_image/from_image%d' % (nl - 1), 1, 2, df[0])
    for l in range(nl - 1):
        conv(d + 'layer_%d/conv1' % (nl - l - 1), 3, df[l], df[l + 1])
        conv(d + 'layer_%d/conv2' % (nl - l - 1), 3, df[l + 1], df[l + 1])
    conv(d + 'output/conv', 3, df[-1], df[-1])
    out[d + 'output/dense/kernel'], out[d + 'output/dense/bias'] = (16 * (df[-1] + 1), df[-1]), (df[-1],)
    out[d + 'output/logits/kernel'], out[d + 'output/logits/bias'] = (df[-1], 1), (1,)
    return out


def to_torch(weights, requires_grad=True):
    return {k: torch.as_tensor(np.asarray(v)).to(DT).requires_grad_(requires_grad) for k, v in weights.items()}


class rounding(object):
    """`with rounding(False):` -- the emulation without its roundings (== oracle/torch_gan_ref.py)"""

    def __init__(self, on):
        self.on = bool(on)

    def __enter__(self):
        global ROUND
        self.prev, ROUND = ROUND, self.on

    def __exit__(self, *exc):
        global ROUND
        ROUND = self.prev
        return False
