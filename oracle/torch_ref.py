"""Independent torch-CPU restatement of the same network -- TEST INFRASTRUCTURE ONLY.

Two uses:
  * fp64 cross-check that pins oracle/sq_oracle.c (tests/test_oracle.py): a
    different code path (ATen conv, different summation order, wider type);
  * bench.py's ``cpu_baseline`` leg: the fp32 oneDNN forward on every host
    core stands in for the reference's TF-CPU path, which cannot run here
    (TensorFlow 1.x is not installable).

Wiring: sequitr/networks/unet.py:224-322.  Leaf-op defaults: SURVEY.md A.1.
"""
import numpy as np
import torch
import torch.nn.functional as F


def _t(a, dtype):
    return torch.as_tensor(np.asarray(a)).to(dtype)


def to_nchw(x, dtype=torch.float64):
    return _t(x, dtype).permute(0, 3, 1, 2).contiguous()


def to_nhwc_np(t):
    return t.permute(0, 2, 3, 1).contiguous().numpy()


def conv2d(x_nchw, w_hwio, bias, act=None, wscale=1.0):
    dt = x_nchw.dtype
    w = (_t(w_hwio, torch.float32) * np.float32(wscale)).to(dt).permute(3, 2, 0, 1).contiguous()
    b = _t(bias, dt) if bias is not None else None
    y = F.conv2d(x_nchw, w, b, padding=w.shape[-1] // 2)
    if act == "relu":
        y = F.relu(y)
    elif act == "leaky":
        y = F.leaky_relu(y, 0.2)
    return y


def convT2x2s2(x_nchw, w_tf, bias):
    """w_tf: (2,2,Cout,Cin) TF conv2d_transpose layout -> torch (Cin,Cout,2,2)."""
    dt = x_nchw.dtype
    w = _t(w_tf, dt).permute(3, 2, 0, 1).contiguous()
    b = _t(bias, dt) if bias is not None else None
    return F.conv_transpose2d(x_nchw, w, b, stride=2)


def bridge_op(up, skip, bridge):                # sequitr/networks/unet.py:190-200
    if bridge == "eltwise_add":
        return up + skip
    if bridge == "eltwise_mul":
        return up * skip
    if bridge == "eltwise_sub":
        return up - skip
    if bridge == "concat":
        return torch.cat([up, skip], 1)          # upscale first, bridge second
    return up


def unet_forward(x_nhwc, weights, params=None, dtype=torch.float64, return_net=False):
    params = params or {}
    filters = tuple(params.get("filters", (16, 32, 64, 128, 256)))
    bridge = params.get("bridge", "eltwise_mul")
    x = to_nchw(x_nhwc, dtype)

    def block(t, scope):
        for k in ("conv1", "conv2"):
            t = conv2d(t, weights[scope + "/" + k + "/kernel"], weights[scope + "/" + k + "/bias"],
                       act="relu")
        return t

    net = [block(x, "UNet/down0")]
    for i in range(1, len(filters)):
        net.append(block(F.max_pool2d(net[-1], 2, 2), "UNet/down%d" % i))
    for i in reversed(range(len(filters) - 1)):
        s = "UNet/up%d" % i
        up = convT2x2s2(net[-1], weights[s + "/upscale/kernel"], weights[s + "/upscale/bias"])
        net.append(block(bridge_op(up, net[i], bridge), s))
    logits = conv2d(net[-1], weights["UNet/to_image/kernel"], weights["UNet/to_image/bias"])
    net.append(logits)
    if return_net:
        return to_nhwc_np(logits), [to_nhwc_np(t) for t in net]
    return to_nhwc_np(logits)


class TorchCpuUNet(object):
    """fp32 channels_last oneDNN forward with pre-converted weights (timing leg)."""

    def __init__(self, weights, params=None, threads=None):
        self.params = params or {}
        self.filters = tuple(self.params.get("filters", (16, 32, 64, 128, 256)))
        self.bridge = self.params.get("bridge", "eltwise_mul")
        if threads:
            torch.set_num_threads(int(threads))
        self.threads = torch.get_num_threads()
        cl = torch.channels_last
        self.w = {}
        for k, v in weights.items():
            t = torch.as_tensor(np.asarray(v), dtype=torch.float32)
            if k.endswith("kernel"):
                t = t.permute(3, 2, 0, 1).contiguous()      # HWIO->OIHW ; (2,2,O,I)->(I,O,2,2)
                if "upscale" not in k:
                    t = t.contiguous(memory_format=cl)
            self.w[k] = t

    @torch.no_grad()
    def __call__(self, x_nhwc):
        x = torch.as_tensor(np.asarray(x_nhwc), dtype=torch.float32).permute(0, 3, 1, 2)
        x = x.contiguous(memory_format=torch.channels_last)
        w = self.w

        def block(t, s):
            for k in ("conv1", "conv2"):
                t = F.relu(F.conv2d(t, w[s + "/" + k + "/kernel"], w[s + "/" + k + "/bias"], padding=1))
            return t

        net = [block(x, "UNet/down0")]
        for i in range(1, len(self.filters)):
            net.append(block(F.max_pool2d(net[-1], 2, 2), "UNet/down%d" % i))
        for i in reversed(range(len(self.filters) - 1)):
            s = "UNet/up%d" % i
            up = F.conv_transpose2d(net[-1], w[s + "/upscale/kernel"], w[s + "/upscale/bias"], stride=2)
            net.append(block(bridge_op(up, net[i], self.bridge), s))
        logits = F.conv2d(net[-1], w["UNet/to_image/kernel"], w["UNet/to_image/bias"])
        return logits.permute(0, 2, 3, 1).contiguous().numpy()


def wsoftmax_ce(logits, onehot, weights):
    """fp64 loss + dlogits via autograd."""
    z = torch.as_tensor(np.asarray(logits), dtype=torch.float64).requires_grad_(True)
    y = torch.as_tensor(np.asarray(onehot)).to(torch.float64)
    w = torch.as_tensor(np.asarray(weights), dtype=torch.float64).reshape(z.shape[:-1])
    ce = -(y * F.log_softmax(z, -1)).sum(-1)
    loss = (w * ce).sum() / w.numel()
    loss.backward()
    return float(loss.detach()), z.grad.numpy()


def unet_loss_and_grads(x_nhwc, onehot, wmap, weights, params=None, dropout_masks=None, dtype=torch.float64):
    """fp64 autograd reference of one training forward/backward (tests/test_gpu_train.py):
    returns (loss, {variable name: gradient ndarray}, logits).  Dropout uses the SUPPLIED masks
    (list of NHWC uint8 arrays in call order) and rate params['dropout']."""
    params = params or {}
    filters = tuple(params.get("filters", (16, 32, 64, 128, 256)))
    bridge = params.get("bridge", "eltwise_mul")
    rate = float(params.get("dropout", 0.0)) if dropout_masks is not None else 0.0
    masks = list(dropout_masks) if dropout_masks is not None else None
    W = {k: torch.as_tensor(np.asarray(v)).to(dtype).requires_grad_(True) for k, v in weights.items()}
    x = to_nchw(x_nhwc, dtype)

    def conv(t, w_hwio, b, act):
        y = F.conv2d(t, w_hwio.permute(3, 2, 0, 1), b, padding=w_hwio.shape[0] // 2)
        return F.relu(y) if act else y

    bn = bool(params.get("batch_norm", False))
    eps = float(params.get("bn_epsilon", 1e-3))

    def block(t, s):
        for k in ("conv1", "conv2"):
            if bn:                                              # training form: batch statistics
                z = conv(t, W[s + "/" + k + "/kernel"], W[s + "/" + k + "/bias"], False)
                mu = z.mean((0, 2, 3), keepdim=True)
                var = ((z - mu) ** 2).mean((0, 2, 3), keepdim=True)
                g, b = W[s + "/" + k + "/gamma"].view(1, -1, 1, 1), W[s + "/" + k + "/beta"].view(1, -1, 1, 1)
                t = F.relu(g * (z - mu) / torch.sqrt(var + eps) + b)
            else:
                t = conv(t, W[s + "/" + k + "/kernel"], W[s + "/" + k + "/bias"], True)
        if masks is not None and rate > 0:
            m = torch.as_tensor(np.asarray(masks.pop(0))).to(dtype).permute(0, 3, 1, 2)
            t = t * m / (1.0 - rate)
        return t

    net = [block(x, "UNet/down0")]
    for i in range(1, len(filters)):
        net.append(block(F.max_pool2d(net[-1], 2, 2), "UNet/down%d" % i))
    for i in reversed(range(len(filters) - 1)):
        s = "UNet/up%d" % i
        up = F.conv_transpose2d(net[-1], W[s + "/upscale/kernel"].permute(3, 2, 0, 1), W[s + "/upscale/bias"], stride=2)
        if W[s + "/upscale/kernel"].shape[0] == 3:              # k=3, s=2, TF SAME: keep the first 2H x 2W
            up = up[:, :, :2 * net[-1].shape[2], :2 * net[-1].shape[3]]
        net.append(block(bridge_op(up, net[i], bridge), s))
    logits = conv(net[-1], W["UNet/to_image/kernel"], W["UNet/to_image/bias"], False).permute(0, 2, 3, 1)
    y = torch.as_tensor(np.asarray(onehot)).to(dtype)
    wm = torch.as_tensor(np.asarray(wmap)).to(dtype).reshape(logits.shape[:-1])
    loss = (wm * -(y * F.log_softmax(logits, -1)).sum(-1)).sum() / wm.numel()
    loss.backward()
    return float(loss.detach()), {k: v.grad.numpy() for k, v in W.items()}, logits.detach().numpy()
