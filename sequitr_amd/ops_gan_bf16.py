"""Operators of the progressive GAN on bf16 FEATURE tensors (BASELINE config 5 with bf16 storage).

`ops.py` dispatches here on the tensor dtype: once the generator's / discriminator's first feature tensor is bf16 (the casts
and the image-side convolutions under ``ops.mixed_precision(store_bf16=True)``), every op of sequitr_amd.functional -- forward,
backward and the WGAN-GP penalty's second-order pass -- stays on bf16 tensors.  The convention is by channel count:
tensors with >= 8 channels ("features", C % 8 == 0) are bf16, tensors with <= 4 channels ("images": real / generated data,
their gradients) are float32, as are parameters, their gradients, the discriminator's (N,) outputs and the losses.
Reference: sequitr/networks/gan.py:44-136 (ops), 149-316 (networks).  Kernels: csrc/sq_gan_bf16.hip, sq_conv_bf16.hip,
sq_conv_wgrad_bf16.hip.
"""
import torch

from . import _lib
from . import ops
from .ops import ACT, _ptr, _stream, _workspace, _grad_out

BF16 = torch.bfloat16
F32 = torch.float32


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


def _feat(t, name):
    _chk(t, name)
    if t.shape[-1] % 8:
        raise _lib.SequitrHipError("%s: bf16 feature tensors need a channel count that is a multiple of 8, got %d"
                                   % (name, t.shape[-1]))
    return t


# ---- streaming ops ------------------------------------------------------------------------------------------------------
def pixelnorm(x, eps=1e-8):
    _feat(x, "x")
    C = x.shape[-1]
    y = torch.empty_like(x)
    _lib.check(_lib.load().sq_pixelnorm_fwd_bf16(_ptr(x), _ptr(y), x.numel() // C, C, float(eps), _stream()),
               "sq_pixelnorm_fwd_bf16")
    return y


def pixelnorm_bwd(x, dy, eps=1e-8, act=None):
    _feat(x, "x"), _chk(dy, "dy")
    C = x.shape[-1]
    dx = torch.empty_like(x)
    _lib.check(_lib.load().sq_pixelnorm_bwd_bf16(_ptr(x), _ptr(dy), _ptr(dx), x.numel() // C, C, float(eps), ACT[act],
                                                _stream()), "sq_pixelnorm_bwd_bf16")
    return dx


def pixelnorm_bwd2(x, g, v, eps=1e-8):
    _feat(x, "x"), _chk(g, "g"), _chk(v, "v")
    C = x.shape[-1]
    dg, dx2 = torch.empty_like(x), torch.empty_like(x)
    _lib.check(_lib.load().sq_pixelnorm_bwd2_bf16(_ptr(x), _ptr(g), _ptr(v), _ptr(dg), _ptr(dx2), x.numel() // C, C,
                                                 float(eps), _stream()), "sq_pixelnorm_bwd2_bf16")
    return dg, dx2


def sumpool2x2(x, scale=1.0):
    _feat(x, "x")
    if x.dim() != 4:
        raise ValueError("x must be NHWC")
    N, H, W, C = x.shape
    y = torch.empty((N, H // 2, W // 2, C), dtype=BF16, device=x.device)
    _lib.check(_lib.load().sq_sumpool2x2_bf16(_ptr(x), _ptr(y), N, H, W, C, float(scale), _stream()), "sq_sumpool2x2_bf16")
    return y


def broadcast2x2(src, scale=1.0):
    _feat(src, "src")
    N, h, w, C = src.shape
    dst = torch.empty((N, 2 * h, 2 * w, C), dtype=BF16, device=src.device)
    _lib.check(_lib.load().sq_broadcast2x2_bf16(_ptr(src), _ptr(dst), N, 2 * h, 2 * w, C, float(scale), _stream()),
               "sq_broadcast2x2_bf16")
    return dst


def broadcast2x2_act_bwd(src, gate, scale, act):
    _feat(src, "src"), _chk(gate, "gate", ndim=4)
    N, h, w, C = src.shape
    if tuple(gate.shape) != (N, 2 * h, 2 * w, C):
        raise ValueError("gate %s does not match the up-sampled %s" % (tuple(gate.shape), (N, 2 * h, 2 * w, C)))
    dst = torch.empty_like(gate)
    _lib.check(_lib.load().sq_broadcast2x2_act_bwd_bf16(_ptr(src), _ptr(gate), _ptr(dst), N, 2 * h, 2 * w, C, float(scale),
                                                       ACT[act], _stream()), "sq_broadcast2x2_act_bwd_bf16")
    return dst


def act_fwd(x, act):
    _chk(x, "x")
    if ACT[act] == 0:
        return x
    y = torch.empty_like(x)
    _lib.check(_lib.load().sq_act_fwd_bf16(_ptr(x), _ptr(y), x.numel(), ACT[act], _stream()), "sq_act_fwd_bf16")
    return y


def act_bwd(dy, y, act):
    _chk(dy, "dy"), _chk(y, "y")
    if ACT[act] == 0:
        return dy
    dx = torch.empty_like(dy)
    _lib.check(_lib.load().sq_act_bwd_bf16(_ptr(dy), _ptr(y), _ptr(dx), dy.numel(), ACT[act], _stream()), "sq_act_bwd_bf16")
    return dx


def cast(x, dtype):
    """f32 <-> bf16 copy of a contiguous tensor (the network's two storage boundaries: the generator's latent block and
    the discriminator's output block, both (N,4,4,512))."""
    if x.dtype == dtype:
        return x
    lib = _lib.load()
    y = torch.empty(x.shape, dtype=dtype, device=x.device)
    if dtype == BF16:
        _chk(x, "x", dtype=F32)
        _lib.check(lib.sq_cast_f32_to_bf16(_ptr(x), _ptr(y), x.numel(), _stream()), "sq_cast_f32_to_bf16")
    elif dtype == F32:
        _chk(x, "x")
        _lib.check(lib.sq_cast_bf16_to_f32(_ptr(x), _ptr(y), x.numel(), _stream()), "sq_cast_bf16_to_f32")
    else:
        raise TypeError("cast: float32 <-> bfloat16 only, got %s" % (dtype,))
    return y


# ---- image-side 1x1 convolutions ----------------------------------------------------------------------------------------
def wgrad1x1_small(a, b, scale=1.0, want_asum=False):
    """(Ca, C) f32 = scale * sum_p a[p,:]^T b[p,:]; a f32 (..., Ca <= 4) or None (ones, Ca = 1), b bf16 (..., C % 8 == 0).
    want_asum: returns (m, (Ca,) per-channel sums of a) -- to_image's weight and bias gradient from one pass."""
    _feat(b, "b")
    C = b.shape[-1]
    npix = b.numel() // C
    Ca = 1
    if a is not None:
        _chk(a, "a", dtype=F32)
        Ca = a.shape[-1]
        if a.numel() // Ca != npix:
            raise ValueError("wgrad1x1_small: operands cover different pixel counts")
    lib = _lib.load()
    nbytes = lib.sq_wgrad1x1_small_workspace_bf16(npix, Ca, C)
    if nbytes < 0:
        raise _lib.SequitrHipError("wgrad1x1_small(bf16): unsupported Ca=%d C=%d" % (Ca, C))
    ws = _workspace(nbytes, b.device)
    m = torch.empty((Ca, C), dtype=F32, device=b.device)
    asum = torch.empty((Ca,), dtype=F32, device=b.device) if want_asum else None
    _lib.check(lib.sq_wgrad1x1_small_bf16(_ptr(a), _ptr(b), _ptr(m), _ptr(asum), _ptr(ws), npix, Ca, C, float(scale), _stream()),
               "sq_wgrad1x1_small_bf16")
    return (m, asum) if want_asum else m


def _matrix(w, Cin, Cout, dgrad):
    """(Cin, Cout) row-major f32 matrix of a 1x1 filter; dgrad: `w` is the forward filter (1,1,Cout,Cin)"""
    return w.reshape(Cout, Cin).t().contiguous() if dgrad else w.reshape(Cin, Cout)


def takes(x, w, dgrad=False):
    """does conv2d(x, w) belong to the bf16-storage forms?  bf16 features in, or (under store_bf16) an f32 image into features"""
    if x.dtype == BF16:
        return True
    if not ops.STORE_BF16 or x.dtype != F32 or x.dim() != 4 or w.dim() != 4:
        return False
    K, Cin = w.shape[0], x.shape[-1]
    Cout = w.shape[2] if dgrad else w.shape[3]
    # an IMAGE: a spatial map (the smallest level is 4 x 4); the (1,1,N,C) row form of a dense layer is not one, whatever its
    # channel count (the discriminator's 1-unit logits layer would otherwise have its input gradient rounded to bf16)
    return K == 1 and Cin <= 4 and Cout % 8 == 0 and x.shape[1] >= 2 and x.shape[2] >= 2


SPLITK_SLICES = 8          # room for this many f32 slices of a small-image conv's output (sq_conv2d_nhwc_mosaic_bf16 picks S <= 8)


def _splitk_workspace(npix, Cout, device):
    nbytes = SPLITK_SLICES * npix * Cout * 4
    if nbytes > (64 << 20):                                     # only the small levels (4x4 / 8x8 images) are ever split
        return None, 0
    return _workspace(nbytes, device), nbytes


def conv2d(x, w, bias=None, act=None, wscale=1.0, dgrad=False):
    """weighted_conv2d on bf16 features / to_image / from_image (gan.py:61-125): KxK SAME conv + bias + activation.
    x (N,H,W,Cin); w (K,K,Cin,Cout) f32 HWIO, or with dgrad the filter (K,K,Cout,Cin) of the forward conv whose input
    gradient this is.  Output dtype by channel count: >= 8 channels bf16, <= 4 channels f32."""
    if x.dim() != 4 or w.dim() != 4:
        raise ValueError("conv2d: x must be NHWC and w HWIO")
    N, H, W, Cin = x.shape
    K = w.shape[0]
    wi, wo = (w.shape[3], w.shape[2]) if dgrad else (w.shape[2], w.shape[3])
    Cout = wo
    if w.shape[1] != K or wi != Cin:
        raise ValueError("weight shape %s does not match input channels %d" % (tuple(w.shape), Cin))
    _chk(w, "w", dtype=F32)
    if bias is not None:
        _chk(bias, "bias", dtype=F32)
        if bias.numel() != Cout:
            raise ValueError("bias must have %d elements" % Cout)
    lib = _lib.load()
    npix = N * H * W
    if x.dtype == F32:                                          # image -> features
        _chk(x, "x", dtype=F32)
        if not (K == 1 and Cin <= 4 and Cout % 8 == 0):
            raise _lib.SequitrHipError("conv2d(bf16 storage): image-side convs are 1x1, <= 4 channels -> C %% 8 == 0")
        y = torch.empty((N, H, W, Cout), dtype=BF16, device=x.device)
        _lib.check(lib.sq_conv1x1_smallin_fwd_bf16(_ptr(x), _ptr(_matrix(w, Cin, Cout, dgrad)), _ptr(bias), _ptr(y), npix, Cin,
                                                  Cout, float(wscale), ACT[act], _stream()), "sq_conv1x1_smallin_fwd_bf16")
        return y
    _feat(x, "x")
    if Cout <= 4:                                               # features -> image
        if K != 1:
            raise _lib.SequitrHipError("conv2d(bf16 storage): a conv to %d channels must be 1x1" % Cout)
        y = torch.empty((N, H, W, Cout), dtype=F32, device=x.device)
        _lib.check(lib.sq_conv1x1_smallout_fwd_bf16(_ptr(x), _ptr(_matrix(w, Cin, Cout, dgrad)), _ptr(bias), _ptr(y), npix, Cin,
                                                   Cout, float(wscale), ACT[act], _stream()), "sq_conv1x1_smallout_fwd_bf16")
        return y
    if Cout % 8:
        raise _lib.SequitrHipError("conv2d(bf16 storage): Cout=%d must be a multiple of 8 (or <= 4)" % Cout)
    if ops.USE_MOSAIC and W < 16 and N * H > 1:
        if K == 1 and npix % 16 == 0:                           # pixels are independent: a free view
            return conv2d(x.view(1, npix // 16, 16, Cin), w, bias, act, wscale, dgrad).view(N, H, W, Cout)
        plan = ops._mosaic_plan(N, H, W) if K == 3 else None
        if plan is not None:
            wp = ops._packed_filter(w, K, Cin, Cout, wscale, dgrad)
            y = torch.empty((N, H, W, Cout), dtype=BF16, device=x.device)
            ws, nbytes = _splitk_workspace(npix, Cout, x.device)
            _lib.check(lib.sq_conv2d_nhwc_mosaic_bf16(_ptr(x), _ptr(wp), _ptr(bias), None, _ptr(y), N, H, W, Cin, Cout,
                                                     ACT[act], plan[0], plan[1], _ptr(ws), nbytes, _stream()),
                       "sq_conv2d_nhwc_mosaic_bf16")
            return y
    wp = ops._packed_filter(w, K, Cin, Cout, wscale, dgrad)
    y = torch.empty((N, H, W, Cout), dtype=BF16, device=x.device)
    _lib.check(lib.sq_conv2d_nhwc_fwd_bf16(_ptr(x), _ptr(wp), _ptr(bias), _ptr(y), N, H, W, Cin, Cout, K, ACT[act], _stream()),
               "sq_conv2d_nhwc_fwd_bf16")
    return y


def conv_dgrad_actgate(dy, w, wscale, gate, act):
    """dgrad of conv2d(., w) followed by act_bwd(., gate, act) in one kernel (same two roundings), or None where that form
    does not exist (image-side convs, the flat 1x1 small-image view)."""
    K, _, Cin, Cout = w.shape                                   # forward filter: the dgrad maps Cout -> Cin channels
    if dy.dim() != 4 or gate.dtype != BF16 or not ACT[act] or Cin % 8 or Cout % 8 or dy.shape[-1] != Cout:
        return None
    N, H, W, _ = dy.shape
    if tuple(gate.shape) != (N, H, W, Cin):
        return None
    _chk(dy, "dy"), _chk(gate, "gate")
    lib = _lib.load()
    if ops.USE_MOSAIC and W < 16 and N * H > 1:
        plan = ops._mosaic_plan(N, H, W) if K == 3 else None
        if plan is None:
            return None
        wp = ops._packed_filter(w, K, Cout, Cin, wscale, True)
        dx = torch.empty((N, H, W, Cin), dtype=BF16, device=dy.device)
        ws, nbytes = _splitk_workspace(N * H * W, Cin, dy.device)
        _lib.check(lib.sq_conv2d_nhwc_mosaic_bf16(_ptr(dy), _ptr(wp), None, _ptr(gate), _ptr(dx), N, H, W, Cout, Cin, ACT[act],
                                                 plan[0], plan[1], _ptr(ws), nbytes, _stream()), "sq_conv2d_nhwc_mosaic_bf16")
        return dx
    wp = ops._packed_filter(w, K, Cout, Cin, wscale, True)
    dx = torch.empty((N, H, W, Cin), dtype=BF16, device=dy.device)
    _lib.check(lib.sq_conv2d_nhwc_dgrad_actgate_bf16(_ptr(dy), _ptr(wp), _ptr(gate), ACT[act], _ptr(dx), N, H, W, Cout, Cin, K,
                                                    _stream()), "sq_conv2d_nhwc_dgrad_actgate_bf16")
    return dx


def conv_wgrad(x, dy, K, want_bias=False, dw_out=None, db_out=None, dw_scale=1.0):
    """(dW (K,K,Cin,Cout) * dw_scale, db or None), f32, from x (N,H,W,Cin) and dY (N,H,W,Cout) where at least one side is a
    bf16 feature tensor; an f32 side is an image (<= 4 channels, K = 1)."""
    if x.dim() != 4 or dy.dim() != 4 or tuple(x.shape[:3]) != tuple(dy.shape[:3]):
        raise ValueError("x %s and dy %s differ in N,H,W" % (tuple(x.shape), tuple(dy.shape)))
    N, H, W, Cin = x.shape
    Cout = dy.shape[-1]
    npix = N * H * W
    lib = _lib.load()
    if x.dtype == F32 or dy.dtype == F32:
        if K != 1 or dw_out is not None or db_out is not None:
            raise _lib.SequitrHipError("conv wgrad (bf16 storage): an image-side conv is 1x1 and has no gradient sinks")
        if x.dtype == F32:                                      # from_image: the image is the input
            dw = wgrad1x1_small(x, dy, dw_scale).view(1, 1, Cin, Cout)
            db = wgrad1x1_small(None, dy).view(Cout) if want_bias else None
            return dw, db
        m, db = wgrad1x1_small(dy, x, dw_scale, want_asum=True)     # to_image: the image is the output; db from the same pass
        return m.t().contiguous().view(1, 1, Cin, Cout), (db if want_bias else None)
    _feat(x, "x"), _feat(dy, "dy")
    if ops.USE_MOSAIC and W < 16 and N * H > 1:
        if K == 1 and npix % 16 == 0:
            return conv_wgrad(x.view(1, npix // 16, 16, Cin), dy.view(1, npix // 16, 16, Cout), K, want_bias, dw_out, db_out,
                              dw_scale)
        plan = ops._mosaic_plan(N, H, W) if (K == 3 and Cin % 16 == 0 and Cout % 16 == 0) else None
        if plan is not None:
            nbytes = lib.sq_conv2d_nhwc_wgrad_workspace_bf16(1, plan[0] * (H + 1), plan[1] * (W + 1), Cin, Cout, K)
            if nbytes < 0:
                raise _lib.SequitrHipError("conv wgrad (bf16 mosaic): unsupported shape Cin=%d Cout=%d" % (Cin, Cout))
            ws = _workspace(nbytes, x.device)
            dw = _grad_out(dw_out, (K, K, Cin, Cout), x.device)
            db = _grad_out(db_out, (Cout,), x.device) if want_bias else None
            _lib.check(lib.sq_conv2d_nhwc_wgrad_mosaic_bf16(_ptr(x), _ptr(dy), _ptr(dw), _ptr(db), _ptr(ws), N, H, W, Cin, Cout,
                                                           plan[0], plan[1], float(dw_scale), _stream()),
                       "sq_conv2d_nhwc_wgrad_mosaic_bf16")
            return dw, db
    nbytes = lib.sq_conv2d_nhwc_wgrad_workspace_bf16(N, H, W, Cin, Cout, K)
    if nbytes < 0:
        raise _lib.SequitrHipError("conv wgrad (bf16 storage): unsupported shape Cin=%d Cout=%d K=%d" % (Cin, Cout, K))
    ws = _workspace(nbytes, x.device)
    dw = _grad_out(dw_out, (K, K, Cin, Cout), x.device)
    db = _grad_out(db_out, (Cout,), x.device) if want_bias else None
    _lib.check(lib.sq_conv2d_nhwc_wgrad_scaled_bf16(_ptr(x), _ptr(dy), _ptr(dw), _ptr(db), _ptr(ws), N, H, W, Cin, Cout, K,
                                                   float(dw_scale), _stream()), "sq_conv2d_nhwc_wgrad_scaled_bf16")
    return dw, db


def conv2d_avgpool_takes(x, w):
    """can conv2d_avgpool fuse this block?  bf16 features, 3x3, even image sides of at least 16 (the small-image levels keep
    their mosaic conv + a pool launch)"""
    return (x.dtype == BF16 and x.dim() == 4 and w.dim() == 4 and w.shape[0] == 3 and w.shape[1] == 3 and x.shape[1] % 2 == 0
            and x.shape[2] % 2 == 0 and x.shape[2] >= 16 and x.shape[3] % 8 == 0 and w.shape[3] % 8 == 0 and w.shape[2] == x.shape[3])


def conv2d_avgpool(x, w, bias=None, act=None, wscale=1.0):
    """(y, avgpool2x2(y)) with y = act(conv3x3(x, w * wscale) + bias): the second conv of a discriminator block and the pool that
    follows it (gan.py:171-192) from one kernel; the pooled tensor equals sumpool2x2(y, 0.25) bit for bit."""
    if not conv2d_avgpool_takes(x, w):
        raise _lib.SequitrHipError("conv2d_avgpool: needs bf16 features, a 3x3 filter and even sides >= 16")
    _feat(x, "x"), _chk(w, "w", dtype=F32)
    if bias is not None:
        _chk(bias, "bias", dtype=F32)
    N, H, W, Cin = x.shape
    Cout = w.shape[3]
    wp = ops._packed_filter(w, 3, Cin, Cout, wscale, False)
    y = torch.empty((N, H, W, Cout), dtype=BF16, device=x.device)
    p = torch.empty((N, H // 2, W // 2, Cout), dtype=BF16, device=x.device)
    _lib.check(_lib.load().sq_conv2d_nhwc_fwd_avgpool_bf16(_ptr(x), _ptr(wp), _ptr(bias), _ptr(y), _ptr(p), N, H, W, Cin, Cout,
                                                          ACT[act], _stream()), "sq_conv2d_nhwc_fwd_avgpool_bf16")
    return y, p


USE_CONV_PN = __import__('os').environ.get("SQ_CONV_PN", "1") != "0"      # A/B switch: 0 = conv and pixel norm as two launches


def conv2d_pixelnorm_takes(x, w):
    """can conv2d_pixelnorm fuse this layer?  bf16 features, a 3x3 filter, all output channels in one block (Cout <= 64) and
    images the plain (non-mosaic) kernel takes"""
    return (USE_CONV_PN and x.dtype == BF16 and x.dim() == 4 and w.dim() == 4 and w.shape[0] == 3 and w.shape[1] == 3
            and x.shape[3] % 8 == 0 and w.shape[3] % 8 == 0 and w.shape[3] <= 64 and w.shape[2] == x.shape[3]
            and not (ops.USE_MOSAIC and x.shape[2] < 16))


def conv2d_pixelnorm(x, w, bias=None, act=None, wscale=1.0, eps=1e-8, want_y=True):
    """(y or None, pixel_norm(y)) with y = act(conv3x3(x, w * wscale) + bias): weighted_conv2d(norm=True) (gan.py:86-97) from one
    kernel; want_y=False skips the store of y (nobody will differentiate through the norm)."""
    if not conv2d_pixelnorm_takes(x, w):
        raise _lib.SequitrHipError("conv2d_pixelnorm: needs bf16 features, a 3x3 filter, Cout <= 64 and image sides >= 16")
    _feat(x, "x"), _chk(w, "w", dtype=F32)
    if bias is not None:
        _chk(bias, "bias", dtype=F32)
    N, H, W, Cin = x.shape
    Cout = w.shape[3]
    wp = ops._packed_filter(w, 3, Cin, Cout, wscale, False)
    y = torch.empty((N, H, W, Cout), dtype=BF16, device=x.device) if want_y else None
    yn = torch.empty((N, H, W, Cout), dtype=BF16, device=x.device)
    _lib.check(_lib.load().sq_conv2d_nhwc_fwd_pixelnorm_bf16(_ptr(x), _ptr(wp), _ptr(bias), _ptr(y), _ptr(yn), N, H, W, Cin, Cout,
                                                            ACT[act], float(eps), _stream()), "sq_conv2d_nhwc_fwd_pixelnorm_bf16")
    return y, yn


def head_concat(conv, mb):
    """flat (N, P (C + 1)) f32 = concat([float(conv (N,...,C) bf16), mb (N, P) f32 as one more channel], -1) flattened per sample"""
    _feat(conv, "conv"), _chk(mb, "mb", dtype=F32)
    C = conv.shape[-1]
    npix = conv.numel() // C
    if mb.numel() != npix:
        raise ValueError("head_concat: %d map values for %d pixels" % (mb.numel(), npix))
    N = conv.shape[0]
    flat = torch.empty((N, (npix // N) * (C + 1)), dtype=F32, device=conv.device)
    _lib.check(_lib.load().sq_head_concat_fwd_bf16(_ptr(conv), _ptr(mb), _ptr(flat), npix, C, _stream()), "sq_head_concat_fwd_bf16")
    return flat


def head_split(dflat, conv_shape, mb_shape):
    """the adjoint of head_concat: (dconv bf16 of conv_shape, dmb f32 of mb_shape) from dflat (N, P (C + 1)) f32"""
    _chk(dflat, "dflat", dtype=F32)
    C = conv_shape[-1]
    npix = 1
    for d in conv_shape[:-1]:
        npix *= int(d)
    if dflat.numel() != npix * (C + 1):
        raise ValueError("head_split: dflat %s does not fit %s" % (tuple(dflat.shape), tuple(conv_shape)))
    dconv = torch.empty(tuple(conv_shape), dtype=BF16, device=dflat.device)
    dmb = torch.empty(tuple(mb_shape), dtype=F32, device=dflat.device)
    _lib.check(_lib.load().sq_head_concat_bwd_bf16(_ptr(dflat), _ptr(dconv), _ptr(dmb), npix, C, _stream()), "sq_head_concat_bwd_bf16")
    return dconv, dmb
