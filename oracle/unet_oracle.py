"""U-Net forward on the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Wiring restated from sequitr/networks/unet.py:224-322 (UNet.build, conv_block,
down_layer, up_layer); leaf ops are the build defaults of SURVEY.md A.1 because
the reference leaves them abstract (unet.py:326-343).  State-dict keys mirror
the reference's variable scopes (unet.py:234,268-271,294,312-318,252).
"""
import numpy as np

from . import c_oracle as co

DEFAULT_FILTERS = (16, 32, 64, 128, 256)      # sequitr/networks/unet.py:40


def unet_forward(x, weights, params=None, return_net=False):
    """x: (N,H,W,num_inputs) float32; weights: {scope/kernel|bias: ndarray}.

    Inference mode: dropout (unet.py:274-276) is the identity.
    Returns logits (N,H,W,num_outputs) [, list of per-layer activations].
    """
    params = params or {}
    filters = tuple(params.get("filters", DEFAULT_FILTERS))
    bridge = params.get("bridge", "eltwise_mul")               # unet.py:138

    bn = bool(params.get("batch_norm", False))                 # SURVEY A.1 optional BN, inference form
    eps = float(params.get("bn_epsilon", 1e-3))

    def conv_block(t, scope):                                  # unet.py:265-277
        for k in ("conv1", "conv2"):
            s = scope + "/" + k
            if not bn:
                t = co.conv2d(t, weights[s + "/kernel"], weights[s + "/bias"], act="relu")
                continue
            z = co.conv2d(t, weights[s + "/kernel"], weights[s + "/bias"], act=None)
            ones, zeros = np.ones(z.shape[-1], np.float32), np.zeros(z.shape[-1], np.float32)
            scale, shift = co.bn_fold(weights[s + "/gamma"], weights[s + "/beta"],
                                      weights.get(s + "/moving_mean", zeros),
                                      weights.get(s + "/moving_variance", ones), eps)
            t = co.bn_apply(z, scale, shift, act="relu")
        return t

    x = np.ascontiguousarray(x, np.float32)
    net = [conv_block(x, "UNet/down0")]                        # unet.py:238
    for i in range(1, len(filters)):                           # unet.py:241-243
        net.append(conv_block(co.maxpool2x2(net[-1]), "UNet/down%d" % i))
    for i in reversed(range(len(filters) - 1)):                # unet.py:246-249
        s = "UNet/up%d" % i
        if tuple(params.get("up_kernel", (2, 2))) == (3, 3):
            up = convT3x3s2(net[-1], weights[s + "/upscale/kernel"], weights[s + "/upscale/bias"])
            up = {"eltwise_add": up + net[i], "eltwise_mul": up * net[i], "eltwise_sub": up - net[i],
                  None: up}[bridge]
        elif bridge == "concat":                                # unet.py:196-197: tf.concat([upscale, skip], -1)
            up = co.convT2x2s2(net[-1], weights[s + "/upscale/kernel"], weights[s + "/upscale/bias"])
            up = np.ascontiguousarray(np.concatenate([up, net[i]], axis=-1))
        else:
            up = co.convT2x2s2(net[-1], weights[s + "/upscale/kernel"], weights[s + "/upscale/bias"],
                               skip=net[i], bridge=bridge)     # unet.py:312-319
        net.append(conv_block(up, s))                          # unet.py:321
    logits = co.conv2d(net[-1], weights["UNet/to_image/kernel"],
                       weights["UNet/to_image/bias"], act=None)  # unet.py:252-253
    net.append(logits)
    return (logits, net) if return_net else logits


def convT3x3s2(x, w_tf, bias):
    """SURVEY A.1 `up_kernel` = (3,3): tf conv2d_transpose(k=3, s=2, SAME).  Restated as the build
    defines it: zero insertion (x at the odd positions) + SAME 3x3 conv with the 180-degree rotated,
    in/out-transposed kernel, each output one fmaf chain in the conv oracle's order."""
    N, H, W, C = x.shape
    u = np.zeros((N, 2 * H, 2 * W, C), np.float32)
    u[:, 1::2, 1::2, :] = x
    w = np.ascontiguousarray(np.transpose(w_tf[::-1, ::-1], (0, 1, 3, 2)))       # (3,3,Cout,Cin) -> HWIO rotated
    return co.conv2d(u, w, bias, act=None)


def predict_mask(logits):
    return co.argmax_u8(logits)
