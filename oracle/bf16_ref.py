"""CPU emulation of the bf16 U-Net training graph -- TEST INFRASTRUCTURE ONLY.

fp64 arithmetic on values that are rounded to bfloat16 at exactly the points where the HIP path
stores a tensor as bf16: every activation between layers, the bf16 copies
of the conv / transpose-conv weights, and every activation-gradient tensor on the way back.
Weight gradients, biases, logits and the loss stay in full precision.  Wiring:
sequitr/networks/unet.py:224-322.
"""
import numpy as np
import torch
import torch.nn.functional as F

DT = torch.float64


def _r(t):
    return t.to(torch.float32).to(torch.bfloat16).to(DT)


class _Rnd(torch.autograd.Function):
    """round to bf16 in the forward pass (straight-through gradient)."""

    @staticmethod
    def forward(ctx, x):
        return _r(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _RndGrad(torch.autograd.Function):
    """identity forward; rounds the gradient flowing back to bf16 (a stored bf16 gradient tensor)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _r(g)


rnd, rnd_grad = _Rnd.apply, _RndGrad.apply


def _forward(x_nhwc, W, params):
    """(per-level outputs NCHW, logits NHWC) of the bf16 graph; W: {name: fp64 tensor}."""
    filters = tuple(params.get("filters", (16, 32, 64, 128, 256)))
    bridge = params.get("bridge", "eltwise_mul")
    x = torch.as_tensor(np.asarray(x_nhwc)).to(DT).permute(0, 3, 1, 2)
    bn = bool(params.get("batch_norm", False))                   # oracle/torch_ref.py unet_loss_and_grads, same branch
    eps = float(params.get("bn_epsilon", 1e-3))

    def conv(t, s, first=False):
        w, b = W[s + "/kernel"], W[s + "/bias"]
        if not first:
            t, w = rnd_grad(t), rnd(w)               # dX is stored as bf16; the filter copy is bf16
        z = F.conv2d(t, w.permute(3, 2, 0, 1), b, padding=1)
        if not bn:
            return rnd(F.relu(z))
        # batch_norm: the conv stores z as bf16, the statistics are taken of those bf16 values, act(BN(z)) is rounded
        # once; the gradient wrt z is stored as bf16 (sq_bn_bwd_bf16)
        z = rnd_grad(rnd(z))
        if params.get("bn_moving", False):                         # inference form: the saved moving statistics
            mu, var = W[s + "/moving_mean"].view(1, -1, 1, 1), W[s + "/moving_variance"].view(1, -1, 1, 1)
        else:
            mu = z.mean((0, 2, 3), keepdim=True)
            var = ((z - mu) ** 2).mean((0, 2, 3), keepdim=True)
        g, be = W[s + "/gamma"].view(1, -1, 1, 1), W[s + "/beta"].view(1, -1, 1, 1)
        return rnd(F.relu(g * (z - mu) / torch.sqrt(var + eps) + be))

    def block(t, s, first=False):
        t = conv(t, s + "/conv1", first)
        # a block output feeds two consumers (pool / transpose-conv and the bridge): autograd adds the two
        # bf16 gradient tensors and stores the sum as bf16
        return rnd_grad(conv(t, s + "/conv2"))

    net = [block(x, "UNet/down0", first=True)]
    for i in range(1, len(filters)):
        net.append(block(F.max_pool2d(rnd_grad(net[-1]), 2, 2), "UNet/down%d" % i))
    for i in reversed(range(len(filters) - 1)):
        s = "UNet/up%d" % i
        up = rnd(F.conv_transpose2d(rnd_grad(net[-1]), rnd(W[s + "/upscale/kernel"]).permute(3, 2, 0, 1),
                                    W[s + "/upscale/bias"], stride=2))
        u, k = rnd_grad(up), rnd_grad(net[i])
        merged = u * k if bridge == "eltwise_mul" else (u + k if bridge == "eltwise_add" else u - k)
        net.append(block(rnd(merged), s))
    h = rnd_grad(net[-1]).permute(0, 2, 3, 1)
    logits = h @ W["UNet/to_image/kernel"].reshape(filters[0], -1) + W["UNet/to_image/bias"]
    return net, logits


def unet_logits_bf16(x_nhwc, weights, params=None):
    """Forward only (inference: pass params['bn_moving']=True with batch_norm): logits (N,H,W,classes) ndarray."""
    with torch.no_grad():
        W = {k: torch.as_tensor(np.asarray(v)).to(DT) for k, v in weights.items()}
        return _forward(x_nhwc, W, params or {})[1].numpy()


def unet_loss_and_grads_bf16(x_nhwc, onehot, wmap, weights, params=None):
    W = {k: torch.as_tensor(np.asarray(v)).to(DT).requires_grad_(True) for k, v in weights.items()}
    net, logits = _forward(x_nhwc, W, params or {})
    y = torch.as_tensor(np.asarray(onehot)).to(DT)
    wm = torch.as_tensor(np.asarray(wmap)).to(DT).reshape(logits.shape[:-1])
    loss = (wm * -(y * F.log_softmax(logits, -1)).sum(-1)).sum() / wm.numel()
    for t in net:
        t.retain_grad()
    loss.backward()
    unet_loss_and_grads_bf16.last_net = [(t.detach().permute(0, 2, 3, 1).numpy(), t.grad.permute(0, 2, 3, 1).numpy())
                                         for t in net]
    return float(loss.detach()), {k: v.grad.numpy() for k, v in W.items() if v.grad is not None}, logits.detach().numpy()
