"""Operator layer: torch tensors in HBM -> C-ABI launches (include/sequitr_hip.h).

These functions are the leaf operators behind the reference's hooks
(sequitr/networks/unet.py:326-343) and GAN helpers (sequitr/networks/gan.py:44-136).
torch is plumbing only: it owns device memory and the HIP stream.  Every op
validates device / dtype / contiguity / shape on the host before launching, and
there is no CPU path: a CPU tensor is an error.
"""
import torch

from . import _lib

ACT = {None: 0, "none": 0, "relu": 1, "leaky": 2}
BRIDGE = {None: 0, "none": 0, "eltwise_add": 1, "eltwise_mul": 2, "eltwise_sub": 3}


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(t, name, dtype=torch.float32, ndim=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise _lib.SequitrHipError("%s must live in GPU memory (no CPU fallback exists)" % name)
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous (NHWC)" % name)
    if ndim is not None and t.dim() != ndim:
        raise ValueError("%s must have %d dims, got %s" % (name, ndim, tuple(t.shape)))
    return t


def _ptr(t):
    return t.data_ptr() if t is not None else None


def _out(out, shape, like, dtype=torch.float32):
    if out is None:
        return torch.empty(shape, dtype=dtype, device=like.device)
    _chk(out, "out", dtype=dtype)
    if tuple(out.shape) != tuple(shape):
        raise ValueError("out has shape %s, expected %s" % (tuple(out.shape), tuple(shape)))
    return out


def conv2d(x, w, bias=None, act=None, wscale=1.0, out=None):
    """KxK SAME conv + bias + activation.  x (N,H,W,Cin), w (K,K,Cin,Cout) HWIO."""
    _chk(x, "x", ndim=4), _chk(w, "w", ndim=4)
    N, H, W, Cin = x.shape
    K, K2, Ci, Cout = w.shape
    if K != K2 or Ci != Cin:
        raise ValueError("weight shape %s does not match input channels %d" % (tuple(w.shape), Cin))
    if bias is not None:
        _chk(bias, "bias")
        if bias.numel() != Cout:
            raise ValueError("bias must have %d elements" % Cout)
    y = _out(out, (N, H, W, Cout), x)
    lib = _lib.load()
    _lib.check(lib.sq_conv2d_nhwc_fwd_f32(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), N, H, W, Cin, Cout, K,
                                         float(wscale), ACT[act], _stream()), "sq_conv2d_nhwc_fwd_f32")
    return y


def _pool(x, fn_name, out):
    _chk(x, "x", ndim=4)
    N, H, W, C = x.shape
    y = _out(out, (N, H // 2, W // 2, C), x)
    lib = _lib.load()
    _lib.check(getattr(lib, fn_name)(_ptr(x), _ptr(y), N, H, W, C, _stream()), fn_name)
    return y


def maxpool2x2(x, out=None):
    return _pool(x, "sq_maxpool2x2_fwd_f32", out)


def avgpool2x2(x, out=None):
    return _pool(x, "sq_avgpool2x2_fwd_f32", out)


def convT2x2s2(x, w, bias=None, skip=None, bridge=None, out=None):
    """2x2/s2 transpose conv (+bias) fused with bridge(upscale, skip).  w (2,2,Cout,Cin)."""
    _chk(x, "x", ndim=4), _chk(w, "w", ndim=4)
    N, H, W, Cin = x.shape
    if tuple(w.shape[:2]) != (2, 2) or w.shape[3] != Cin:
        raise ValueError("transpose-conv weight must be (2,2,Cout,%d), got %s" % (Cin, tuple(w.shape)))
    Cout = w.shape[2]
    b = BRIDGE[bridge]
    if b:
        if skip is None:
            raise ValueError("bridge %r needs a skip tensor" % bridge)
        _chk(skip, "skip", ndim=4)
        if tuple(skip.shape) != (N, 2 * H, 2 * W, Cout):
            raise ValueError("skip has shape %s, expected %s" % (tuple(skip.shape), (N, 2 * H, 2 * W, Cout)))
    if bias is not None:
        _chk(bias, "bias")
    y = _out(out, (N, 2 * H, 2 * W, Cout), x)
    lib = _lib.load()
    _lib.check(lib.sq_convT2x2s2_nhwc_fwd_f32(_ptr(x), _ptr(w), _ptr(bias), _ptr(skip) if b else None,
                                             _ptr(y), N, H, W, Cin, Cout, b, _stream()),
               "sq_convT2x2s2_nhwc_fwd_f32")
    return y


def bridge(a, b, kind, out=None):
    _chk(a, "a"), _chk(b, "b")
    if a.shape != b.shape:
        raise ValueError("bridge operands differ in shape: %s vs %s" % (tuple(a.shape), tuple(b.shape)))
    y = _out(out, a.shape, a)
    lib = _lib.load()
    _lib.check(lib.sq_bridge_fwd_f32(_ptr(a), _ptr(b), _ptr(y), a.numel(), BRIDGE[kind], _stream()),
               "sq_bridge_fwd_f32")
    return y


def conv1x1_argmax(x, w, bias=None, want_mask=True):
    """to_image head + prediction: returns (logits f32 (N,H,W,Cout), mask u8 (N,H,W) or None)."""
    _chk(x, "x", ndim=4), _chk(w, "w", ndim=4)
    N, H, W, Cin = x.shape
    if tuple(w.shape[:3]) != (1, 1, Cin):
        raise ValueError("1x1 weight must be (1,1,%d,Cout), got %s" % (Cin, tuple(w.shape)))
    Cout = w.shape[3]
    if bias is not None:
        _chk(bias, "bias")
    logits = torch.empty((N, H, W, Cout), dtype=torch.float32, device=x.device)
    mask = torch.empty((N, H, W), dtype=torch.uint8, device=x.device) if want_mask else None
    lib = _lib.load()
    _lib.check(lib.sq_conv1x1_argmax_fwd_f32(_ptr(x), _ptr(w), _ptr(bias), _ptr(logits), _ptr(mask),
                                            N, H, W, Cin, Cout, _stream()), "sq_conv1x1_argmax_fwd_f32")
    return logits, mask


def argmax_u8(logits):
    _chk(logits, "logits")
    C = logits.shape[-1]
    mask = torch.empty(logits.shape[:-1], dtype=torch.uint8, device=logits.device)
    lib = _lib.load()
    _lib.check(lib.sq_argmax_u8(_ptr(logits), _ptr(mask), logits.numel() // C, C, _stream()), "sq_argmax_u8")
    return mask


def pixelnorm(x, eps=1e-8, out=None):
    _chk(x, "x")
    C = x.shape[-1]
    y = _out(out, x.shape, x)
    lib = _lib.load()
    _lib.check(lib.sq_pixelnorm_fwd_f32(_ptr(x), _ptr(y), x.numel() // C, C, float(eps), _stream()),
               "sq_pixelnorm_fwd_f32")
    return y


def upsample_nn2x(x, out=None):
    _chk(x, "x", ndim=4)
    N, H, W, C = x.shape
    y = _out(out, (N, 2 * H, 2 * W, C), x)
    lib = _lib.load()
    _lib.check(lib.sq_upsample_nn2x_f32(_ptr(x), _ptr(y), N, H, W, C, _stream()), "sq_upsample_nn2x_f32")
    return y


def wsoftmax_ce(logits, onehot, weights, want_grad=True, grad_scale=1.0):
    """Weighted softmax-CE: returns (loss: 0-d float64 device tensor, dlogits or None)."""
    _chk(logits, "logits"), _chk(onehot, "onehot", dtype=torch.uint8), _chk(weights, "weights")
    C = logits.shape[-1]
    npix = logits.numel() // C
    if onehot.shape != logits.shape or weights.numel() != npix:
        raise ValueError("label / weight shapes do not match the logits")
    lib = _lib.load()
    nparts = lib.sq_wsoftmax_ce_partials(npix)
    ws = torch.empty(nparts + 1, dtype=torch.float64, device=logits.device)
    dz = torch.empty_like(logits) if want_grad else None
    _lib.check(lib.sq_wsoftmax_ce_fwd_bwd_f32(_ptr(logits), _ptr(onehot), _ptr(weights), npix, C,
                                             float(grad_scale), ws.data_ptr(), ws.data_ptr() + 8 * nparts,
                                             _ptr(dz), _stream()), "sq_wsoftmax_ce_fwd_bwd_f32")
    return ws[nparts], dz
