"""fp64 torch-CPU restatement of the progressive WGAN-GP of sequitr/networks/gan.py -- TEST
INFRASTRUCTURE ONLY.  Leaf ops: gan.py:44-136; networks: gan.py:149-316; losses: gan.py:665-732
(one-sided gradient penalty, lambda 10, eps drift 0.001 Dx^2); variable names as the reference's
scopes.  Uses torch autograd with create_graph=True for the penalty, as tf.gradients does.
"""
import numpy as np
import torch
import torch.nn.functional as F

DT = torch.float64


def pixel_norm(x, eps=1e-8):                               # NHWC
    return x * torch.rsqrt(torch.mean(x * x, dim=-1, keepdim=True) + eps)


def lrelu(x):
    return F.leaky_relu(x, 0.2)


def wconv(x, W, name, act=True, norm=True):
    k, b = W[name + '/filter'], W[name + '/bias'].reshape(-1)
    kh, kw, cin, cout = k.shape
    ws = torch.tensor(np.float32(np.sqrt(np.float32(2.0 / float(kh * kw * cout)))), dtype=DT)
    wk = k * ws
    y = F.conv2d(x.permute(0, 3, 1, 2), wk.permute(3, 2, 0, 1), b, padding=kh // 2).permute(0, 2, 3, 1)
    if act:
        y = lrelu(y)
    if norm:
        y = pixel_norm(y)
    return y


def dense(x, W, name, act=False):
    y = x @ W[name + '/kernel'] + W[name + '/bias']
    return lrelu(y) if act else y


def up2(x):
    return x.repeat_interleave(2, 1).repeat_interleave(2, 2)


def avgpool(x):
    return F.avg_pool2d(x.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)


def half_size(x):
    """resize_nearest_neighbor(align_corners=True) to H/2 (gan.py:128-131)."""
    n, h, w, c = x.shape
    ho, wo = h // 2, w // 2
    ys = [min(int(np.round(np.float32(i) * np.float32(h - 1) / np.float32(ho - 1))) if ho > 1 else 0, h - 1) for i in range(ho)]
    xs = [min(int(np.round(np.float32(i) * np.float32(w - 1) / np.float32(wo - 1))) if wo > 1 else 0, w - 1) for i in range(wo)]
    return x[:, ys][:, :, xs]


def generator(z, W, filters):
    p = 'GAN/generator/'
    d = dense(pixel_norm(z), W, p + 'latent/dense1', act=True)
    x = pixel_norm(d.reshape(-1, 4, 4, filters[0]))
    layers = [wconv(x, W, p + 'latent/conv')]
    for l, f in enumerate(filters[1:]):
        u = up2(layers[-1])
        c1 = wconv(u, W, p + 'layer_%d/conv1' % l)
        layers.append(wconv(c1, W, p + 'layer_%d/conv2' % l))
    outs = [wconv(c, W, p + 'to_image/to_image%d' % l, act=False, norm=False) for l, c in enumerate(layers)]
    return outs, outs[-1]


def discriminator(x, W, filters):
    p = 'GAN/discriminator/'
    nl = len(filters)
    x = wconv(x, W, p + 'from_image/from_image%d' % (nl - 1))
    for l, f in enumerate(filters[1:]):
        s = p + 'layer_%d/' % (nl - l - 1)
        x = avgpool(wconv(wconv(x, W, s + 'conv1', norm=False), W, s + 'conv2', norm=False))
    var = x.var(dim=0, unbiased=False).mean()
    mb = torch.ones((x.shape[0], 4, 4, 1), dtype=x.dtype) * torch.sqrt(var)
    c = wconv(x, W, p + 'output/conv', norm=False)
    flat = torch.cat([c, mb], -1).reshape(-1, 16 * (filters[-1] + 1))
    h = dense(flat, W, p + 'output/dense', act=True)
    return dense(h, W, p + 'output/logits').reshape(-1)


def losses(X, Z, alpha, r, W, filters, level):
    """(Gz_raw, d_loss, g_loss) of gan.py:665-732 at `level`; W: {name: fp64 tensor}."""
    f = filters[:level + 1]
    outs, Gz_raw = generator(Z, W, f)
    Xr = X
    if level > 0:
        Gz = alpha * Gz_raw + (1. - alpha) * up2(outs[-2])
        Xr = alpha * X + (1. - alpha) * up2(half_size(X))
    else:
        Gz = Gz_raw
    df = f[::-1]
    Dz, Dx = discriminator(Gz, W, df), discriminator(Xr, W, df)
    rr = r.reshape(-1, 1, 1, 1)
    mix = (rr * Xr + (1 - rr) * Gz).detach().requires_grad_(True)
    Dmix = discriminator(mix, W, df)
    grad = torch.autograd.grad(Dmix.sum(), mix, create_graph=True)[0]
    gn = torch.sqrt((grad * grad).sum((1, 2, 3)))
    pen = 10.0 * torch.square(torch.clamp(gn - 1.0, min=0.0))
    g_loss = torch.mean(-Dz)
    d_loss = torch.mean(-Dx + Dz + pen + 0.001 * torch.square(Dx))
    return Gz_raw, d_loss, g_loss


def to_torch(weights, requires_grad=True):
    return {k: torch.as_tensor(np.asarray(v)).to(DT).requires_grad_(requires_grad) for k, v in weights.items()}
