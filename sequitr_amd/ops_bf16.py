"""bf16 operator layer (BASELINE configs 3-5): torch.bfloat16 activations in HBM, fp32 master
weights / biases / accumulation; every op is a hand-written HIP kernel of libsequitr_hip.so."""
import torch

from . import _lib
from .ops import ACT, BRIDGE, _ptr, _stream, _workspace  # noqa: F401

BF16 = torch.bfloat16


def _chk(t, name, dtype=BF16, ndim=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.SequitrHipError("%s must be a GPU tensor (no CPU fallback exists)" % name)
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous (NHWC)" % name)
    if ndim is not None and t.dim() != ndim:
        raise ValueError("%s must have %d dims, got %s" % (name, ndim, tuple(t.shape)))
    return t


def pack_weights(w, transform=False, wscale=1.0):
    """f32 HWIO (K,K,Cin,Cout) -> packed bf16 filter.  transform=True packs the dgrad filter (the
    result convolves Cout channels to Cin)."""
    _chk(w, "w", dtype=torch.float32, ndim=4)
    K, _, Cin, Cout = w.shape
    ci, co = (Cout, Cin) if transform else (Cin, Cout)
    lib = _lib.load()
    n = lib.sq_conv_packed_weights_elems_bf16(K, ci, co)
    if n < 0:
        raise _lib.SequitrHipError("pack_weights: unsupported K=%d Cin=%d Cout=%d" % (K, ci, co))
    wp = torch.empty((n,), dtype=BF16, device=w.device)
    _lib.check(lib.sq_conv_pack_weights_bf16(_ptr(w), _ptr(wp), K, ci, co, float(wscale), 1 if transform else 0,
                                            _stream()), "sq_conv_pack_weights_bf16")
    return wp


def conv2d(x, wp, bias, K, Cout, act=None):
    """x (N,H,W,Cin) bf16, wp packed filter -> (N,H,W,Cout) bf16."""
    _chk(x, "x", ndim=4), _chk(wp, "wp")
    N, H, W, Cin = x.shape
    if bias is not None:
        _chk(bias, "bias", dtype=torch.float32)
    y = torch.empty((N, H, W, Cout), dtype=BF16, device=x.device)
    lib = _lib.load()
    _lib.check(lib.sq_conv2d_nhwc_fwd_bf16(_ptr(x), _ptr(wp), _ptr(bias), _ptr(y), N, H, W, Cin, Cout, K, ACT[act],
                                          _stream()), "sq_conv2d_nhwc_fwd_bf16")
    return y


def conv3x3_first(x, w, bias, act="relu"):
    """f32 (N,H,W,1) image, f32 (3,3,1,Cout) filter -> bf16 activation."""
    _chk(x, "x", dtype=torch.float32, ndim=4), _chk(w, "w", dtype=torch.float32, ndim=4)
    N, H, W, _ = x.shape
    Cout = w.shape[3]
    y = torch.empty((N, H, W, Cout), dtype=BF16, device=x.device)
    lib = _lib.load()
    _lib.check(lib.sq_conv3x3_first_fwd_bf16(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), N, H, W, Cout, ACT[act], _stream()),
               "sq_conv3x3_first_fwd_bf16")
    return y


def conv2d_wgrad(x, dy, K, want_bias=True):
    """(dW (K,K,Cin,Cout) f32, db f32 or None) from bf16 X and bf16 dY."""
    _chk(x, "x", ndim=4), _chk(dy, "dy", ndim=4)
    N, H, W, Cin = x.shape
    Cout = dy.shape[3]
    lib = _lib.load()
    nbytes = lib.sq_conv2d_nhwc_wgrad_workspace_bf16(N, H, W, Cin, Cout, K)
    if nbytes < 0:
        raise _lib.SequitrHipError("conv2d_wgrad(bf16): unsupported Cin=%d Cout=%d K=%d" % (Cin, Cout, K))
    ws = _workspace(nbytes, x.device)
    dw = torch.empty((K, K, Cin, Cout), dtype=torch.float32, device=x.device)
    db = torch.empty((Cout,), dtype=torch.float32, device=x.device) if want_bias else None
    _lib.check(lib.sq_conv2d_nhwc_wgrad_bf16(_ptr(x), _ptr(dy), _ptr(dw), _ptr(db), _ptr(ws), N, H, W, Cin, Cout, K,
                                            _stream()), "sq_conv2d_nhwc_wgrad_bf16")
    return dw, db
