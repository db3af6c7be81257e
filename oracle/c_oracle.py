"""ctypes binding of oracle/sq_oracle.c (numpy in, numpy out).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Build with
``make -C oracle`` (done by ``__graft_entry__.build()``).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libsq_oracle.so")

ACT = {None: 0, "none": 0, "relu": 1, "leaky": 2}
BRIDGE = {None: 0, "eltwise_add": 1, "eltwise_mul": 2, "eltwise_sub": 3}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(
                "oracle library missing: run `make -C oracle` (or __graft_entry__.build())")
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _check(rc, name):
    if rc != 0:
        raise RuntimeError("%s failed with code %d" % (name, rc))


def conv2d(x, w, bias=None, act=None, wscale=1.0):
    """x (N,H,W,Cin), w (K,K,Cin,Cout) -> (N,H,W,Cout)."""
    x, w = _f32(x), _f32(w)
    N, H, W, Cin = x.shape
    K, K2, Ci, Cout = w.shape
    assert K == K2 and Ci == Cin
    b = _f32(bias) if bias is not None else None
    y = np.empty((N, H, W, Cout), np.float32)
    _check(lib().oracle_conv2d_nhwc_f32(_p(x), _p(w), _p(b), _p(y), N, H, W, Cin, Cout, K,
                                       ctypes.c_float(wscale), ACT[act]), "conv2d")
    return y


def pixelnorm(x, eps=1e-8):
    x = _f32(x)
    y = np.empty_like(x)
    C = x.shape[-1]
    _check(lib().oracle_pixelnorm_f32(_p(x), _p(y), ctypes.c_long(x.size // C), C,
                                      ctypes.c_float(eps)), "pixelnorm")
    return y


def maxpool2x2(x):
    x = _f32(x)
    N, H, W, C = x.shape
    y = np.empty((N, H // 2, W // 2, C), np.float32)
    _check(lib().oracle_maxpool2x2_f32(_p(x), _p(y), N, H, W, C), "maxpool")
    return y


def avgpool2x2(x):
    x = _f32(x)
    N, H, W, C = x.shape
    y = np.empty((N, H // 2, W // 2, C), np.float32)
    _check(lib().oracle_avgpool2x2_f32(_p(x), _p(y), N, H, W, C), "avgpool")
    return y


def convT2x2s2(x, w, bias=None, skip=None, bridge=None):
    """x (N,H,W,Cin), w (2,2,Cout,Cin) TF layout -> (N,2H,2W,Cout), then bridge."""
    x, w = _f32(x), _f32(w)
    N, H, W, Cin = x.shape
    assert w.shape[:2] == (2, 2) and w.shape[3] == Cin
    Cout = w.shape[2]
    b = _f32(bias) if bias is not None else None
    s = _f32(skip) if skip is not None else None
    if BRIDGE[bridge] != 0:
        assert s is not None and s.shape == (N, 2 * H, 2 * W, Cout)
    y = np.empty((N, 2 * H, 2 * W, Cout), np.float32)
    _check(lib().oracle_convT2x2s2_nhwc_f32(_p(x), _p(w), _p(b), _p(s), _p(y), N, H, W, Cin, Cout,
                                           BRIDGE[bridge]), "convT")
    return y


def argmax_u8(logits):
    z = _f32(logits)
    C = z.shape[-1]
    m = np.empty(z.shape[:-1], np.uint8)
    _check(lib().oracle_argmax_u8(_p(z), _p(m), ctypes.c_long(z.size // C), C), "argmax")
    return m


def upsample_nn2x(x):
    x = _f32(x)
    N, H, W, C = x.shape
    y = np.empty((N, 2 * H, 2 * W, C), np.float32)
    _check(lib().oracle_upsample_nn2x_f32(_p(x), _p(y), N, H, W, C), "upsample")
    return y


def wsoftmax_ce(logits, onehot, weights, want_grad=True):
    """returns (loss float64, dlogits float32 or None)."""
    z = _f32(logits)
    y = np.ascontiguousarray(onehot, dtype=np.uint8)
    w = _f32(weights)
    C = z.shape[-1]
    npix = z.size // C
    assert y.shape == z.shape and w.size == npix
    loss = ctypes.c_double(0.0)
    dz = np.empty_like(z) if want_grad else None
    _check(lib().oracle_wsoftmax_ce_f32(_p(z), _p(y), _p(w), ctypes.c_long(npix), C,
                                       ctypes.byref(loss), _p(dz)), "wsoftmax_ce")
    return loss.value, dz


def bn_stats(x):
    x = _f32(x)
    C = x.shape[-1]
    mean, var = np.empty(C, np.float32), np.empty(C, np.float32)
    _check(lib().oracle_bn_stats_f32(_p(x), _p(mean), _p(var), ctypes.c_long(x.size // C), C), "bn_stats")
    return mean, var


def bn_fold(gamma, beta, mean, var, eps=1e-3):
    gamma, beta, mean, var = _f32(gamma), _f32(beta), _f32(mean), _f32(var)
    scale, shift = np.empty_like(gamma), np.empty_like(gamma)
    _check(lib().oracle_bn_fold_f32(_p(gamma), _p(beta), _p(mean), _p(var), ctypes.c_float(eps), _p(scale),
                                    _p(shift), gamma.size), "bn_fold")
    return scale, shift


def bn_apply(x, scale, shift, act=None):
    x = _f32(x)
    C = x.shape[-1]
    y = np.empty_like(x)
    _check(lib().oracle_bn_apply_f32(_p(x), _p(_f32(scale)), _p(_f32(shift)), _p(y), ctypes.c_long(x.size // C), C,
                                     ACT[act]), "bn_apply")
    return y
