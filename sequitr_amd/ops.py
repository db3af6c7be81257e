"""Operator layer: torch tensors in HBM -> C-ABI launches (include/sequitr_hip.h).

These functions are the leaf operators behind the reference's hooks
(sequitr/networks/unet.py:326-343) and GAN helpers (sequitr/networks/gan.py:44-136).
torch is plumbing only: it owns device memory and the HIP stream.  Every op
validates device / dtype / contiguity / shape on the host before launching, and
there is no CPU path: a CPU tensor is an error.
"""
import os

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


USE_MOSAIC = os.environ.get("SQ_MOSAIC", "1") != "0"      # A/B switch for the small-image batching below


def _mosaic_plan(N, H, W):
    """(R, Cc) cell grid that packs N small images into one image with the fewest 16x16 conv tiles
    (sq_mosaic_pack_f32), or None when the batch is better left as it is."""
    if not USE_MOSAIC or N < 2 or max(H, W) > 8:
        return None
    best = None
    for cc in range(1, N + 1):
        r = -(-N // cc)
        tiles = -(-(r * (H + 1)) // 16) * -(-(cc * (W + 1)) // 16)
        if best is None or tiles < best[0]:
            best = (tiles, r, cc)
    return None if best[0] >= N else (best[1], best[2])


def mosaic_pack(x, R, Cc):
    _chk(x, "x", ndim=4)
    N, H, W, C = x.shape
    m = torch.empty((1, R * (H + 1), Cc * (W + 1), C), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    _lib.check(lib.sq_mosaic_pack_f32(_ptr(x), _ptr(m), N, H, W, C, R, Cc, _stream()), "sq_mosaic_pack_f32")
    return m


def mosaic_unpack(m, N, H, W, R, Cc):
    _chk(m, "m", ndim=4)
    C = m.shape[3]
    y = torch.empty((N, H, W, C), dtype=torch.float32, device=m.device)
    lib = _lib.load()
    _lib.check(lib.sq_mosaic_unpack_f32(_ptr(m), _ptr(y), N, H, W, C, R, Cc, _stream()), "sq_mosaic_unpack_f32")
    return y


USE_DENSE = os.environ.get("SQ_DENSE", "1") != "0"        # A/B switch for the split-reduction dense kernel
MOSAIC_IN_KERNEL = os.environ.get("SQ_MOSAIC_IN_KERNEL", "1") != "0"   # A/B: mosaic addressing inside the mixed conv

# "mixed" convolutions: f32 tensors, bf16 multiply, f32 accumulate (sq_conv2d_nhwc_{fwd,wgrad}_mixed_f32).  While the
# flag is set, conv2d / conv2d_dgrad / conv2d_wgrad route every layer the mixed kernels take (Cin % 8, Cout % 4;
# wgrad: both % 16) to them; image-side 1x1 convs (to_image / from_image / class heads) and dense stay f32.
MIXED = False
# bf16 STORAGE on top of the bf16 multiplies (the GAN's dtype 'bf16'): feature tensors (C % 8 == 0) are bf16 in HBM, image
# tensors (<= 4 channels) stay f32.  The flag only decides what an image-side 1x1 conv emits; everything downstream
# dispatches on the tensor dtype (ops_gan_bf16.py).
STORE_BF16 = False
_BF16 = torch.bfloat16


def _gb():
    from . import ops_gan_bf16
    return ops_gan_bf16


class mixed_precision:
    """`with ops.mixed_precision():` -- bf16-multiply convolutions for everything launched inside, including a
    backward pass run inside the block (the GAN's `dtype='mixed'`); store_bf16=True additionally keeps the feature
    tensors in bf16 (`dtype='bf16'`, BASELINE config 5)."""

    def __init__(self, on=True, store_bf16=False):
        self.on = bool(on)
        self.store = bool(store_bf16) and self.on

    def __enter__(self):
        global MIXED, STORE_BF16
        self.prev, MIXED = (MIXED, STORE_BF16), self.on
        STORE_BF16 = self.store
        return self

    def __exit__(self, *exc):
        global MIXED, STORE_BF16
        MIXED, STORE_BF16 = self.prev
        if not MIXED:
            _PACKS.clear()
        return False


# packed bf16 filters of PARAMETERS (leaf tensors that require grad), valid until the weights change: a solver step
# evaluates the discriminator three times and differentiates it twice with the same weights, and the optimiser only
# writes them as the step's last act.  Invalidation is tied to the weight UPDATE, not to the nesting depth of the
# precision contexts: every Adam wrapper below calls invalidate_packs() (the kernels update the weights in place
# through raw pointers, which no tensor version counter sees), the GAN's solver steps clear on the way out, and
# leaving the outermost mixed_precision block clears as before.  Entries hold the parameter itself, so its storage
# cannot be recycled under the key.
_PACKS = {}


_PLANS = __import__('weakref').WeakSet()     # live FilterPackPlans: their packs go stale with every weight update too


_INVALIDATIONS = [0]


def invalidate_packs():
    """drop every cached bf16 filter pack (call after anything that writes parameters in place)"""
    _PACKS.clear()
    _TRANSFORMS.clear()
    _INVALIDATIONS[0] += 1
    for p in list(_PLANS):
        p.fresh = False


def invalidation_epoch():
    """number of invalidate_packs() calls so far: an owner of several FilterPackPlans that knows WHICH weights its own steps
    move (the GAN: a solver step updates one network) compares it with the value it saw last to learn whether anybody else
    wrote parameters in between"""
    return _INVALIDATIONS[0]


class FilterPackPlan(object):
    """Every bf16 filter pack a solver step of the mixed-precision GAN needs -- the forward and the dgrad form of each
    equalised-LR conv kernel, factor folded in -- as ONE launch over the flat parameter buffer
    (sq_conv_pack_weights_multi_scaled_bf16) instead of one pack launch per use (97 per iteration at level 6).
    `named` = [(name, leaf view of `flat`, float offset, wscale[, forms])].  run() packs and marks the plan fresh; any weight
    update (invalidate_packs) marks it stale, after which _packed_filter falls back to packing on demand."""

    def __init__(self, flat, named):
        lib = _lib.load()
        rows, scales, dst, item = [], [], 0, 0
        self.views = {}
        for entry in named:
            name, leaf, off, wscale = entry[:4]
            forms = entry[4] if len(entry) > 4 else 'NT'        # which packs: N = forward, T = dgrad (transform)
            if leaf.dim() == 2:                                 # a dense kernel (Cin, Cout): the 1x1 conv F.dense runs it as
                K, (Cin, Cout) = 1, leaf.shape
            elif leaf.dim() == 4 and leaf.shape[0] == leaf.shape[1]:
                K, _, Cin, Cout = leaf.shape
            else:
                continue
            for transform, (ci, co) in ((0, (Cin, Cout)), (1, (Cout, Cin))):
                if ci % 8 or co % 4 or ('T' if transform else 'N') not in forms:
                    continue
                n = lib.sq_conv_packed_weights_elems_bf16(K, ci, co)
                if n <= 0:
                    continue
                if n % 8:                                       # the pack kernel writes 8 elements (16 bytes) per thread
                    raise _lib.SequitrHipError("FilterPackPlan: %d elements in a pack (multiple of 8 needed)" % n)
                rows.append([off, dst, K, ci, co, transform, item, 0])
                scales.append(float(wscale))
                self.views[(id(leaf), bool(transform))] = (dst, n, float(wscale), K, ci, co)
                dst += (n + 7) // 8 * 8
                item += n
        if len(rows) > 128:
            raise _lib.SequitrHipError("FilterPackPlan: %d packs exceed the 128-entry table" % len(rows))
        self.flat, self.n, self.total, self.fresh = flat, len(rows), item, False
        dev = flat.device
        self.table = torch.tensor(rows if rows else [[0] * 8], dtype=torch.int32, device=dev).contiguous()
        self.scales = torch.tensor(scales if scales else [1.0], dtype=torch.float32, device=dev)
        self.out = torch.zeros(max(dst, 8), dtype=torch.bfloat16, device=dev)
        self.leaves = [entry[1] for entry in named]             # keep the ids valid
        _PLANS.add(self)

    def run(self):
        if self.n:
            _lib.check(_lib.load().sq_conv_pack_weights_multi_scaled_bf16(
                _ptr(self.flat), _ptr(self.out), _ptr(self.table), _ptr(self.scales), self.n, self.total, _stream()),
                "sq_conv_pack_weights_multi_scaled_bf16")
        self.fresh = True

    def lookup(self, w, K, Cin, Cout, wscale, transform):
        base = w._base if w._base is not None else w            # F.dense hands the (1,1,Cin,Cout) view of its kernel
        e = self.views.get((id(base), bool(transform)))
        if e is None or not self.fresh:
            return None
        d0, n, ws, k, ci, co = e
        if (k, ci, co) != (K, Cin, Cout) or ws != float(wscale):
            return None
        return self.out[d0:d0 + n]


def _packed_filter(w, K, Cin, Cout, wscale, transform):
    """bf16 pack of the conv Cin -> Cout; transform: `w` is the FORWARD filter (K,K,Cout,Cin) of which this conv
    is the dgrad (taps rotated, channel roles swapped inside the pack kernel -- no separate transform pass)."""
    for plan in list(_PLANS):                                    # the step's one-launch packs, while they are fresh
        wp = plan.lookup(w, K, Cin, Cout, wscale, transform)
        if wp is not None:
            return wp
    cacheable = w.is_leaf and w.requires_grad
    key = (w.data_ptr(), K, Cin, Cout, float(wscale), bool(transform))
    if cacheable and key in _PACKS:
        return _PACKS[key][1]
    lib = _lib.load()
    n = lib.sq_conv_packed_weights_elems_bf16(K, Cin, Cout)
    wp = torch.empty((n,), dtype=torch.bfloat16, device=w.device)
    _lib.check(lib.sq_conv_pack_weights_bf16(_ptr(w), _ptr(wp), K, Cin, Cout, float(wscale), 1 if transform else 0,
                                            _stream()), "sq_conv_pack_weights_bf16")
    if cacheable:
        _PACKS[key] = (w, wp)
    return wp


def _conv2d_mixed(x, w, bias, act, wscale, y, transform):
    N, H, W, Cin = x.shape
    K, Cout = w.shape[0], (w.shape[2] if transform else w.shape[3])
    lib = _lib.load()
    wp = _packed_filter(w, K, Cin, Cout, wscale, transform)
    _lib.check(lib.sq_conv2d_nhwc_fwd_mixed_f32(_ptr(x), _ptr(wp), _ptr(bias), _ptr(y), N, H, W, Cin, Cout, K, ACT[act],
                                               _stream()), "sq_conv2d_nhwc_fwd_mixed_f32")
    return y


def dense(x, w, bias=None, act=None, wscale=1.0):
    """y (M,N) = act(x (M,K) @ w (K,N) * wscale + bias): few rows, long reduction (sq_dense_fwd_f32)."""
    _chk(x, "x", ndim=2), _chk(w, "w", ndim=2)
    M, K = x.shape
    N = w.shape[1]
    if w.shape[0] != K:
        raise ValueError("dense: weight %s does not match %d inputs" % (tuple(w.shape), K))
    if bias is not None:
        _chk(bias, "bias")
    lib = _lib.load()
    ws = _workspace(lib.sq_dense_workspace_f32(M, K, N), x.device)
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    _lib.check(lib.sq_dense_fwd_f32(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), _ptr(ws), M, K, N, float(wscale), ACT[act],
                                   _stream()), "sq_dense_fwd_f32")
    return y


def dense_wgrad(x, dy, want_bias=True, dw_scale=1.0, shape4=False, dw_out=None, db_out=None, accumulate=0):
    """(dW (K,N) * dw_scale, db (N) or None) of y = x @ w + b for a few rows: x (M,K), dY (M,N), M <= 128 (sq_dense_wgrad_f32);
    shape4: dW as the (1,1,K,N) filter of the 1x1 conv F.dense runs the layer as.  dw_out / db_out: contiguous float32
    destinations (a parameter's gradient sink); accumulate: bit 0 -- dW is ADDED to dw_out's contents, bit 1 -- db to db_out's."""
    _chk(x, "x", ndim=2), _chk(dy, "dy", ndim=2)
    M, K = x.shape
    N = dy.shape[1]
    if dy.shape[0] != M:
        raise ValueError("dense_wgrad: x %s and dy %s differ in rows" % (tuple(x.shape), tuple(dy.shape)))
    dw = _grad_out(dw_out, (K, N), x.device) if dw_out is not None else torch.empty(
        (1, 1, K, N) if shape4 else (K, N), dtype=torch.float32, device=x.device)
    db = (_grad_out(db_out, (N,), x.device) if db_out is not None else torch.empty((N,), dtype=torch.float32, device=x.device)) \
        if want_bias else None
    _lib.check(_lib.load().sq_dense_wgrad_f32(_ptr(x), _ptr(dy), _ptr(dw), _ptr(db), M, K, N, float(dw_scale), int(accumulate),
                                             _stream()), "sq_dense_wgrad_f32")
    return dw, db


def conv2d(x, w, bias=None, act=None, wscale=1.0, out=None, _dgrad=False):
    """KxK SAME conv + bias + activation.  x (N,H,W,Cin), w (K,K,Cin,Cout) HWIO.
    Batches of small images (H, W <= 8) run as one mosaic image (3x3) or as a flat pixel strip (1x1):
    same fmaf chain per output, far fewer and fuller 16x16 tiles.
    _dgrad (internal): `w` is the filter (K,K,Cout,Cin) of the forward conv whose input gradient this is."""
    if out is None and (x.dtype == _BF16 or STORE_BF16) and _gb().takes(x, w, _dgrad):
        return _gb().conv2d(x, w, bias, act, wscale, _dgrad)
    _chk(x, "x", ndim=4), _chk(w, "w", ndim=4)
    N, H, W, Cin = x.shape
    if _dgrad:
        takes_dense = out is None and w.shape[0] == 1 and N * H * W <= 128 and Cin >= 1024 and Cin % 4 == 0 and USE_DENSE
        if takes_dense or not (MIXED and Cin % 8 == 0 and w.shape[2] % 4 == 0):
            w, _dgrad = conv_weight_transform(w), False        # the f32 kernels take the transformed filter
    K, K2, Ci, Cout = w.shape
    if _dgrad:
        Ci, Cout = Cout, Ci
    if K != K2 or Ci != Cin:
        raise ValueError("weight shape %s does not match input channels %d" % (tuple(w.shape), Cin))
    if bias is not None:
        _chk(bias, "bias")
        if bias.numel() != Cout:
            raise ValueError("bias must have %d elements" % Cout)
    if out is None and K == 1 and N * H * W <= 128 and Cin >= 1024 and Cin % 4 == 0 and USE_DENSE:
        return dense(x.view(N * H * W, Cin), w.view(Cin, Cout), bias, act, wscale).view(N, H, W, Cout)
    if out is None and USE_MOSAIC and W < 16 and N * H > 1:
        P = N * H * W
        if K == 1 and P % 16 == 0:                            # pixels are independent: a free view
            return conv2d(x.view(1, P // 16, 16, Cin), w, bias, act, wscale, _dgrad=_dgrad).view(N, H, W, Cout)
        plan = _mosaic_plan(N, H, W) if (K == 3 and Cin % 4 == 0 and Cout % 4 == 0) else None
        if plan is not None and MOSAIC_IN_KERNEL and MIXED and Cin % 8 == 0:
            # the mixed kernel addresses the compact tensors through the mosaic map itself: no pack / unpack launches
            wp = _packed_filter(w, K, Cin, Cout, wscale, _dgrad)
            y = torch.empty((N, H, W, Cout), dtype=torch.float32, device=x.device)
            _lib.check(_lib.load().sq_conv2d_nhwc_mixed_mosaic_f32(_ptr(x), _ptr(wp), _ptr(bias), None, _ptr(y), N, H, W, Cin,
                                                                  Cout, ACT[act], plan[0], plan[1], _stream()),
                       "sq_conv2d_nhwc_mixed_mosaic_f32")
            return y
        if plan is not None:
            ym = conv2d(mosaic_pack(x, *plan), w, bias, act, wscale, _dgrad=_dgrad)
            return mosaic_unpack(ym, N, H, W, *plan)
    y = _out(out, (N, H, W, Cout), x)
    if MIXED and Cin % 8 == 0 and Cout % 4 == 0:
        return _conv2d_mixed(x, w, bias, act, wscale, y, _dgrad)
    lib = _lib.load()
    _lib.check(lib.sq_conv2d_nhwc_fwd_f32(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), N, H, W, Cin, Cout, K,
                                         float(wscale), ACT[act], _stream()), "sq_conv2d_nhwc_fwd_f32")
    return y


def conv2d_concat(xa, xb, w, bias=None, act=None):
    """conv_layer on the concat bridge: act(conv(concat([xa, xb], -1)) + bias) without materialising the concatenation
    (unet.py:196-197: the up-scaled tensor first, the skip tensor second).  xa, xb (N,H,W,Ca), w (K,K,2Ca,Cout)."""
    _chk(xa, "xa", ndim=4), _chk(xb, "xb", ndim=4), _chk(w, "w", ndim=4)
    if tuple(xa.shape) != tuple(xb.shape):
        raise ValueError("conv2d_concat: the two sources differ in shape: %s vs %s" % (tuple(xa.shape), tuple(xb.shape)))
    N, H, W, Ca = xa.shape
    K, K2, Cin, Cout = w.shape
    if K != K2 or Cin != 2 * Ca:
        raise ValueError("conv2d_concat: weight %s does not match 2 x %d input channels" % (tuple(w.shape), Ca))
    if bias is not None:
        _chk(bias, "bias")
    y = torch.empty((N, H, W, Cout), dtype=torch.float32, device=xa.device)
    _lib.check(_lib.load().sq_conv2d_concat_nhwc_fwd_f32(_ptr(xa), _ptr(xb), _ptr(w), _ptr(bias), _ptr(y), N, H, W, Ca, Cout,
                                                        K, ACT[act], _stream()), "sq_conv2d_concat_nhwc_fwd_f32")
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
    if x.dtype == _BF16 and out is None:
        return _gb().sumpool2x2(x, 0.25)
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
    if x.dtype == _BF16 and out is None:
        return _gb().pixelnorm(x, eps)
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


# ----------------------------------------------------------------------------------------------
# training-side operators (gradients of the ops above; include/sequitr_hip.h "Training side")
# ----------------------------------------------------------------------------------------------
class WorkspaceArena(object):
    """Scratch memory for the kernels that take a caller-provided workspace (split reductions, block partials), with an
    OWNER STREAM.  A workspace is one buffer that consecutive launches reuse; that is only sound while all of them are
    ordered on one stream, and a captured hipGraph bakes the buffer's address into its kernel nodes.  Rules:

      * the buffer only grows; a buffer that is replaced is RETIRED, never freed while the arena lives -- a captured
        graph (or a kernel still in flight) may hold its address, and a block allocated during a capture belongs to
        that graph's private pool, which the caching allocator returns to the driver once the graph is gone;
      * outside a capture the arena belongs to the stream that used it last: a launch from another stream first makes
        that stream wait for everything the previous owner enqueued (strict arenas -- SQ_WS_STRICT=1 or strict=True
        -- raise instead, so that an accidental multi-stream schedule is noticed rather than serialised); `hand_over()`
        is the explicit form a trainer calls when it moves between its warm-up, capture and replay streams;
      * inside a capture every launch that uses the arena must come from ONE capturing stream: a stream forked inside
        the capture (a side "lane") needs an arena of its own -- two lanes sharing one buffer is a race the graph does
        not order -- and that always raises.

    The library-wide default arena (one per device) serves callers that own nothing (the GAN, the stand-alone
    operators); UNetTrainer and the captured predictors own theirs and select it with `use_arena`."""

    def __init__(self, name='default', strict=None):
        self.name = name
        self.strict = (os.environ.get('SQ_WS_STRICT', '0') != '0') if strict is None else bool(strict)
        self.buf = None
        self.retired = []
        self.owner = None                  # torch.cuda.Stream (a reference: its handle cannot be recycled under us)
        self.capture_owner = None          # raw handle of the capturing stream seen first in the current capture

    def _own(self, device):
        cur = torch.cuda.current_stream(device)
        if torch.cuda.is_current_stream_capturing():
            if self.capture_owner is None:
                self.capture_owner = cur.cuda_stream
            elif self.capture_owner != cur.cuda_stream:
                raise _lib.SequitrHipError(
                    "workspace arena %r: used from two capturing streams (0x%x, 0x%x) -- a stream forked inside a "
                    "capture needs its own WorkspaceArena" % (self.name, self.capture_owner, cur.cuda_stream))
            return
        self.capture_owner = None
        if self.owner is None:
            self.owner = cur
        elif self.owner.cuda_stream != cur.cuda_stream:
            if self.strict:
                raise _lib.SequitrHipError(
                    "workspace arena %r belongs to stream 0x%x, launch on stream 0x%x: call hand_over() (the new "
                    "stream then waits for the old one) or give the second stream its own arena"
                    % (self.name, self.owner.cuda_stream, cur.cuda_stream))
            cur.wait_stream(self.owner)
            self.owner = cur

    def hand_over(self, stream=None):
        """Make `stream` (default: the current one) the owner; it waits for all work of the previous owner."""
        new = stream if stream is not None else torch.cuda.current_stream()
        if self.owner is not None and self.owner.cuda_stream != new.cuda_stream:
            new.wait_stream(self.owner)
        self.owner = new
        return self

    def acquire(self, nbytes, device):
        self._own(device)
        if self.buf is None or self.buf.numel() * 4 < nbytes or self.buf.device != torch.device(device):
            if self.buf is not None:
                self.retired.append(self.buf)                    # its address may be baked into a captured graph
            self.buf = torch.empty((max(int(nbytes), 1 << 20) + 3) // 4, dtype=torch.float32, device=device)
        return self.buf


_DEFAULT_ARENAS = {}                              # device index -> the library-wide default arena
_ARENA = [None]                                   # the arena of the object launching right now (see use_arena)


class use_arena(object):
    """`with use_arena(trainer.arena): ...` -- every workspace the block's launches ask for comes from that arena.
    A plain module-level slot, not a thread-local: autograd runs the backward of these launches on its own device
    thread, and one host thread drives one GPU (the reference's one-process-per-job model)."""

    def __init__(self, arena):
        self.arena, self.prev = arena, None

    def __enter__(self):
        self.prev, _ARENA[0] = _ARENA[0], self.arena
        return self.arena

    def __exit__(self, *exc):
        _ARENA[0] = self.prev


def _workspace(nbytes, device):
    """Caller-owned scratch for one launch: from the active arena (use_arena) or the device's default arena."""
    arena = _ARENA[0]
    if arena is None:
        arena = _DEFAULT_ARENAS.get(device.index)
        if arena is None:
            arena = _DEFAULT_ARENAS[device.index] = WorkspaceArena('default:%s' % (device.index,))
    return arena.acquire(nbytes, device)


_TRANSFORMS = {}                                  # dgrad filters of PARAMETERS, kept until the next weight update (invalidate_packs)


def conv_weight_transform(w):
    """HWIO (K,K,Cin,Cout) -> dgrad filter (K,K,Cout,Cin), taps rotated by 180 degrees.  A parameter's transform is
    cached until the optimiser writes the weights (the discriminator is differentiated several times per solver step)."""
    _chk(w, "w", ndim=4)
    cacheable = w.is_leaf and w.requires_grad
    key = (w.data_ptr(), tuple(w.shape))
    if cacheable and key in _TRANSFORMS:
        return _TRANSFORMS[key][1]
    K, _, Cin, Cout = w.shape
    wt = torch.empty((K, K, Cout, Cin), dtype=torch.float32, device=w.device)
    lib = _lib.load()
    _lib.check(lib.sq_conv_weight_transform_f32(_ptr(w), _ptr(wt), K, Cin, Cout, _stream()),
               "sq_conv_weight_transform_f32")
    if cacheable:
        _TRANSFORMS[key] = (w, wt)
    return wt


def conv2d_dgrad(dy, w, wscale=1.0):
    """dX of conv2d: a forward convolution of dY with the transformed filter."""
    return conv2d(dy, w, None, act=None, wscale=wscale, _dgrad=True)


def _grad_out(buf, shape, device):
    """A caller-provided gradient destination (e.g. a view of the flat gradient bucket) or a new tensor."""
    if buf is None:
        return torch.empty(shape, dtype=torch.float32, device=device)
    if buf.dtype != torch.float32 or not buf.is_contiguous() or buf.numel() != int(torch.Size(shape).numel()):
        raise ValueError("gradient destination %s does not fit %s" % (tuple(buf.shape), tuple(shape)))
    return buf


def conv2d_wgrad(x, dy, K, want_bias=True, dw_out=None, db_out=None, dw_scale=1.0):
    """(dW (K,K,Cin,Cout), db (Cout) or None) from X (N,H,W,Cin) and dY (N,H,W,Cout).  dw_out / db_out:
    optional float32 destinations the kernel writes straight into.  dw_scale: factor on dW (not db), applied in the
    finish kernel (the equalised-LR factor of weighted_conv2d, gan.py:75-79)."""
    _chk(x, "x", ndim=4), _chk(dy, "dy", ndim=4)
    N, H, W, Cin = x.shape
    Cout = dy.shape[3]
    if tuple(dy.shape[:3]) != (N, H, W):
        raise ValueError("x %s and dy %s differ in N,H,W" % (tuple(x.shape), tuple(dy.shape)))
    if K == 1 and N * H * W <= 128 and Cin * Cout >= (1 << 16) and USE_DENSE and dw_out is None and db_out is None:
        return dense_wgrad(x.view(N * H * W, Cin), dy.view(N * H * W, Cout), want_bias, dw_scale, shape4=True)
    if USE_MOSAIC and W < 16 and N * H > 1 and Cin % 4 == 0 and Cout % 4 == 0:
        P = N * H * W
        if K == 1 and P % 16 == 0:
            return conv2d_wgrad(x.view(1, P // 16, 16, Cin), dy.view(1, P // 16, 16, Cout), K, want_bias,
                                dw_out, db_out, dw_scale)
        plan = _mosaic_plan(N, H, W) if K == 3 else None
        if plan is not None and MOSAIC_IN_KERNEL and MIXED:
            lib = _lib.load()
            MH, MW = plan[0] * (H + 1), plan[1] * (W + 1)
            nbytes = lib.sq_conv2d_nhwc_wgrad_workspace_mixed_f32(1, MH, MW, Cin, Cout, K)
            if nbytes >= 0:                                     # both channel counts % 16: the mixed kernel, mosaic addressed inside
                ws = _workspace(nbytes, x.device)
                dw = _grad_out(dw_out, (K, K, Cin, Cout), x.device)
                db = _grad_out(db_out, (Cout,), x.device) if want_bias else None
                _lib.check(lib.sq_conv2d_nhwc_wgrad_mixed_mosaic_f32(_ptr(x), _ptr(dy), _ptr(dw), _ptr(db), _ptr(ws), N, H, W,
                                                                    Cin, Cout, plan[0], plan[1], float(dw_scale), _stream()),
                           "sq_conv2d_nhwc_wgrad_mixed_mosaic_f32")
                return dw, db
        if plan is not None:                                  # separator cells of dY are zero: they add nothing
            return conv2d_wgrad(mosaic_pack(x, *plan), mosaic_pack(dy, *plan), K, want_bias, dw_out, db_out, dw_scale)
    lib = _lib.load()
    mixed = MIXED and lib.sq_conv2d_nhwc_wgrad_workspace_mixed_f32(N, H, W, Cin, Cout, K) >= 0
    nbytes = (lib.sq_conv2d_nhwc_wgrad_workspace_mixed_f32 if mixed else lib.sq_conv2d_nhwc_wgrad_workspace_f32)(
        N, H, W, Cin, Cout, K)
    if nbytes < 0:
        raise _lib.SequitrHipError("conv2d_wgrad: unsupported shape Cin=%d Cout=%d K=%d" % (Cin, Cout, K))
    ws = _workspace(nbytes, x.device)
    dw = _grad_out(dw_out, (K, K, Cin, Cout), x.device)
    db = _grad_out(db_out, (Cout,), x.device) if want_bias else None
    if Cin <= 7 and dw_scale != 1.0:
        raise _lib.SequitrHipError("conv2d_wgrad: dw_scale needs the MFMA kernels (Cin=%d)" % Cin)
    fn = lib.sq_conv2d_nhwc_wgrad_scaled_mixed_f32 if mixed else lib.sq_conv2d_nhwc_wgrad_scaled_f32
    _lib.check(fn(_ptr(x), _ptr(dy), _ptr(dw), _ptr(db), _ptr(ws), N, H, W, Cin, Cout, K, float(dw_scale), _stream()),
               "sq_conv2d_nhwc_wgrad_scaled_mixed_f32" if mixed else "sq_conv2d_nhwc_wgrad_scaled_f32")
    return dw, db
    _lib.check(lib.sq_conv2d_nhwc_wgrad_f32(_ptr(x), _ptr(dy), _ptr(dw), _ptr(db), _ptr(ws), N, H, W, Cin, Cout,
                                           K, _stream()), "sq_conv2d_nhwc_wgrad_f32")
    return dw, db


def act_bwd(dy, y, act):
    if dy.dtype == _BF16:
        return _gb().act_bwd(dy, y, act)
    _chk(dy, "dy"), _chk(y, "y")
    if ACT[act] == 0:
        return dy
    dx = torch.empty_like(dy)
    lib = _lib.load()
    _lib.check(lib.sq_act_bwd_f32(_ptr(dy), _ptr(y), _ptr(dx), dy.numel(), ACT[act], _stream()), "sq_act_bwd_f32")
    return dx


def maxpool2x2_bwd(x, dy):
    _chk(x, "x", ndim=4), _chk(dy, "dy", ndim=4)
    N, H, W, C = x.shape
    dx = torch.empty_like(x)
    lib = _lib.load()
    _lib.check(lib.sq_maxpool2x2_bwd_f32(_ptr(x), _ptr(dy), _ptr(dx), N, H, W, C, _stream()), "sq_maxpool2x2_bwd_f32")
    return dx


def broadcast2x2(src, scale=1.0):
    """(N,h,w,C) -> (N,2h,2w,C), every source pixel copied (x scale) to its 2x2 patch."""
    if src.dtype == _BF16:
        return _gb().broadcast2x2(src, scale)
    _chk(src, "src", ndim=4)
    N, h, w, C = src.shape
    dst = torch.empty((N, 2 * h, 2 * w, C), dtype=torch.float32, device=src.device)
    lib = _lib.load()
    _lib.check(lib.sq_broadcast2x2_f32(_ptr(src), _ptr(dst), N, 2 * h, 2 * w, C, float(scale), _stream()),
               "sq_broadcast2x2_f32")
    return dst


def broadcast2x2_act_bwd(src, gate, scale, act):
    """scale * 2x nearest up-sampling of src, passed through the backward of the activation whose output is `gate`"""
    if src.dtype == _BF16:
        return _gb().broadcast2x2_act_bwd(src, gate, scale, act)
    _chk(src, "src", ndim=4), _chk(gate, "gate", ndim=4)
    N, h, w, C = src.shape
    if tuple(gate.shape) != (N, 2 * h, 2 * w, C):
        raise ValueError("gate %s does not match the up-sampled %s" % (tuple(gate.shape), (N, 2 * h, 2 * w, C)))
    dst = torch.empty_like(gate)
    _lib.check(_lib.load().sq_broadcast2x2_act_bwd_f32(_ptr(src), _ptr(gate), _ptr(dst), N, 2 * h, 2 * w, C, float(scale),
                                                      ACT[act], _stream()), "sq_broadcast2x2_act_bwd_f32")
    return dst


def sumpool2x2(x, scale=1.0):
    """scale * (sum of every 2x2 patch); any channel count."""
    if x.dtype == _BF16:
        return _gb().sumpool2x2(x, scale)
    _chk(x, "x", ndim=4)
    N, H, W, C = x.shape
    y = torch.empty((N, H // 2, W // 2, C), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    _lib.check(lib.sq_sumpool2x2_f32(_ptr(x), _ptr(y), N, H, W, C, float(scale), _stream()), "sq_sumpool2x2_f32")
    return y


def bridge_bwd(dy, a, b, kind):
    _chk(dy, "dy")
    da, db = torch.empty_like(dy), torch.empty_like(dy)
    lib = _lib.load()
    _lib.check(lib.sq_bridge_bwd_f32(_ptr(dy), _ptr(a), _ptr(b), _ptr(da), _ptr(db), dy.numel(), BRIDGE[kind],
                                    _stream()), "sq_bridge_bwd_f32")
    return da, db


def space_to_depth2(dy):
    """(N,2H,2W,C) -> (N,H,W,4C), channel index (2a+b)*C + c."""
    _chk(dy, "dy", ndim=4)
    N, H2, W2, C = dy.shape
    g = torch.empty((N, H2 // 2, W2 // 2, 4 * C), dtype=torch.float32, device=dy.device)
    lib = _lib.load()
    _lib.check(lib.sq_space_to_depth2_f32(_ptr(dy), _ptr(g), N, H2 // 2, W2 // 2, C, _stream()),
               "sq_space_to_depth2_f32")
    return g


def conv1x1_small_bwd(x, w, dz, want_dx=True, dw_out=None, db_out=None):
    """Backward of the to_image head: returns (dx or None, dw (1,1,Cin,Cout), db (Cout))."""
    _chk(x, "x", ndim=4), _chk(w, "w", ndim=4), _chk(dz, "dz", ndim=4)
    N, H, W, Cin = x.shape
    Cout = w.shape[3]
    npix = N * H * W
    lib = _lib.load()
    ws = _workspace(lib.sq_conv1x1_small_bwd_workspace_f32(npix, Cin, Cout), x.device)
    dx = torch.empty_like(x) if want_dx else None
    dw = _grad_out(dw_out, (1, 1, Cin, Cout), x.device)
    db = _grad_out(db_out, (Cout,), x.device)
    _lib.check(lib.sq_conv1x1_small_bwd_f32(_ptr(x), _ptr(w), _ptr(dz), _ptr(dx), _ptr(dw), _ptr(db), _ptr(ws),
                                           npix, Cin, Cout, _stream()), "sq_conv1x1_small_bwd_f32")
    return dx, dw, db


def dropout_fwd(x, rate, seed=0, mask=None, step_dev=None):
    """returns (y, mask u8).  A supplied mask is used as-is (parity tests)."""
    _chk(x, "x")
    y = torch.empty_like(x)
    given = mask is not None
    if given:
        _chk(mask, "mask", dtype=torch.uint8)
    else:
        mask = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    lib = _lib.load()
    _lib.check(lib.sq_dropout_fwd_f32(_ptr(x), _ptr(y), _ptr(mask), x.numel(), float(rate), int(seed) & 0xFFFFFFFF,
                                     1 if given else 0, _ptr(step_dev), _stream()), "sq_dropout_fwd_f32")
    return y, mask


def dropout_bwd(dy, mask, rate):
    _chk(dy, "dy"), _chk(mask, "mask", dtype=torch.uint8)
    dx = torch.empty_like(dy)
    lib = _lib.load()
    _lib.check(lib.sq_dropout_bwd_f32(_ptr(dy), _ptr(mask), _ptr(dx), dy.numel(), float(rate), _stream()),
               "sq_dropout_bwd_f32")
    return dx


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0):
    """In-place Adam update of the flat parameter buffer p."""
    for t, n in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _chk(t, n)
    if not (p.numel() == g.numel() == m.numel() == v.numel()):
        raise ValueError("adam_step: buffers differ in size")
    lib = _lib.load()
    _lib.check(lib.sq_adam_step_f32(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), float(lr), float(beta1),
                                   float(beta2), float(eps), int(step), float(grad_scale), _stream()),
               "sq_adam_step_f32")
    invalidate_packs()


def axpy_(y, x, alpha=1.0):
    """y += alpha * x in place over flat fp32 buffers (gradient accumulation across micro-batches)."""
    _chk(y, "y"), _chk(x, "x")
    if y.numel() != x.numel():
        raise ValueError("axpy_: buffers differ in size")
    _lib.check(_lib.load().sq_axpy_f32(_ptr(y), _ptr(x), float(alpha), y.numel(), _stream()), "sq_axpy_f32")
    return y


# ----------------------------------------------------------------------------------------------
# GAN-side operators (include/sequitr_hip.h "GAN side")
# ----------------------------------------------------------------------------------------------
def pixelnorm_bwd(x, dy, eps=1e-8, act=None):
    """dx of pixel_norm; act: x is the output of that activation and its backward is applied in the same pass"""
    if x.dtype == _BF16:
        return _gb().pixelnorm_bwd(x, dy, eps, act)
    _chk(x, "x"), _chk(dy, "dy")
    C = x.shape[-1]
    dx = torch.empty_like(x)
    lib = _lib.load()
    if ACT[act]:
        _lib.check(lib.sq_pixelnorm_bwd_act_f32(_ptr(x), _ptr(dy), _ptr(dx), x.numel() // C, C, float(eps), ACT[act],
                                               _stream()), "sq_pixelnorm_bwd_act_f32")
        return dx
    _lib.check(lib.sq_pixelnorm_bwd_f32(_ptr(x), _ptr(dy), _ptr(dx), x.numel() // C, C, float(eps), _stream()),
               "sq_pixelnorm_bwd_f32")
    return dx


def pixelnorm_bwd2(x, g, v, eps=1e-8):
    """second-order: (dL/dg, dL/dx) of dx = pixelnorm_bwd(x, g) given v = dL/d(dx)."""
    if x.dtype == _BF16:
        return _gb().pixelnorm_bwd2(x, g, v, eps)
    _chk(x, "x"), _chk(g, "g"), _chk(v, "v")
    C = x.shape[-1]
    dg, dx2 = torch.empty_like(x), torch.empty_like(x)
    lib = _lib.load()
    _lib.check(lib.sq_pixelnorm_bwd2_f32(_ptr(x), _ptr(g), _ptr(v), _ptr(dg), _ptr(dx2), x.numel() // C, C,
                                        float(eps), _stream()), "sq_pixelnorm_bwd2_f32")
    return dg, dx2


def resize_nearest(x, size):
    """tf.image.resize_nearest_neighbor(x, size, align_corners=True)."""
    _chk(x, "x", ndim=4)
    N, Hi, Wi, C = x.shape
    Ho, Wo = int(size[0]), int(size[1])
    y = torch.empty((N, Ho, Wo, C), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    _lib.check(lib.sq_resize_nearest_f32(_ptr(x), _ptr(y), N, Hi, Wi, Ho, Wo, C, _stream()), "sq_resize_nearest_f32")
    return y


def lerp(a, b, alpha, out=None):
    """alpha*a + (1-alpha)*b; alpha a python float or a per-sample (N,) device tensor.  out: a contiguous float32
    destination of a's shape (e.g. one half of a stacked batch)."""
    _chk(a, "a"), _chk(b, "b")
    if a.shape != b.shape:
        raise ValueError("lerp operands differ in shape")
    if out is None:
        y = torch.empty_like(a)
    else:
        y = _chk(out, "out")
        if y.shape != a.shape:
            raise ValueError("lerp: out %s does not fit %s" % (tuple(y.shape), tuple(a.shape)))
    per = a.numel() // a.shape[0]
    lib = _lib.load()
    if isinstance(alpha, torch.Tensor):
        _chk(alpha, "alpha")
        _lib.check(lib.sq_lerp_f32(_ptr(a), _ptr(b), _ptr(y), a.numel(), per, 0.0, _ptr(alpha), _stream()), "sq_lerp_f32")
    else:
        _lib.check(lib.sq_lerp_f32(_ptr(a), _ptr(b), _ptr(y), a.numel(), per, float(alpha), None, _stream()), "sq_lerp_f32")
    return y


def scale(x, s, one_minus=False):
    """s*x (or (1-s)*x); s a python float or a per-sample (N,) device tensor."""
    _chk(x, "x")
    y = torch.empty_like(x)
    per = x.numel() // x.shape[0]
    lib = _lib.load()
    if isinstance(s, torch.Tensor):
        _chk(s, "s")
        _lib.check(lib.sq_scale_f32(_ptr(x), _ptr(y), x.numel(), per, 0.0, _ptr(s), int(one_minus), _stream()), "sq_scale_f32")
    else:
        _lib.check(lib.sq_scale_f32(_ptr(x), _ptr(y), x.numel(), per, float(s), None, int(one_minus), _stream()), "sq_scale_f32")
    return y


def _conv2d_avgpool_takes(x, w):
    return x.dtype == _BF16 and _gb().conv2d_avgpool_takes(x, w)


def conv2d_avgpool(x, w, bias=None, act=None, wscale=1.0):
    """(y, 2x2 average pool of y), y = act(conv3x3(x, w * wscale) + bias), from one kernel (bf16 feature tensors only)"""
    return _gb().conv2d_avgpool(x, w, bias, act, wscale)


def conv2d_pixelnorm_takes(x, w):
    return x.dtype == _BF16 and _gb().conv2d_pixelnorm_takes(x, w)


def conv2d_pixelnorm(x, w, bias=None, act=None, wscale=1.0, eps=1e-8, want_y=True):
    """(y, pixel_norm(y, eps)), y = act(conv3x3(x, w * wscale) + bias), from one kernel (bf16 features, Cout <= 64)"""
    return _gb().conv2d_pixelnorm(x, w, bias, act, wscale, eps, want_y)


def cast(x, dtype):
    """float32 <-> bfloat16 copy (the GAN's two storage boundaries)"""
    return _gb().cast(x, dtype)


def act_fwd(x, act):
    if x.dtype == _BF16:
        return _gb().act_fwd(x, act)
    _chk(x, "x")
    if ACT[act] == 0:
        return x
    y = torch.empty_like(x)
    lib = _lib.load()
    _lib.check(lib.sq_act_fwd_f32(_ptr(x), _ptr(y), x.numel(), ACT[act], _stream()), "sq_act_fwd_f32")
    return y


def dot_per_sample(a, b):
    """(N,) tensor of sum_i a[n,i]*b[n,i]."""
    _chk(a, "a"), _chk(b, "b")
    N = a.shape[0]
    out = torch.empty((N,), dtype=torch.float32, device=a.device)
    lib = _lib.load()
    ws = _workspace(lib.sq_dot_per_sample_workspace_f32(N), a.device)
    _lib.check(lib.sq_dot_per_sample_f32(_ptr(a), _ptr(b), _ptr(out), _ptr(ws), N, a.numel() // N, _stream()),
               "sq_dot_per_sample_f32")
    return out


def mbstd(x):
    """minibatch stdev scalar (0-d device tensor) of x (N, ...)."""
    _chk(x, "x")
    out = torch.empty((), dtype=torch.float32, device=x.device)
    ws = _workspace(1024, x.device)
    lib = _lib.load()
    _lib.check(lib.sq_mbstd_fwd_f32(_ptr(x), _ptr(out), _ptr(ws), x.shape[0], x.numel() // x.shape[0], _stream()),
               "sq_mbstd_fwd_f32")
    return out


def mbstd_map(x, groups=1, cells=16):
    """minibatch-stdev feature map (gan.py:204-212): x (G*n, ...) f32 or bf16 -> (G*n, cells) f32 filled with each group's
    statistic"""
    bf = x.dtype == _BF16
    _chk(x, "x", dtype=x.dtype if bf else torch.float32)
    N = x.shape[0]
    if N % groups:
        raise ValueError("mbstd_map: batch %d is not %d equal groups" % (N, groups))
    y = torch.empty((N, cells), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    ws = _workspace(lib.sq_mbstd_map_workspace(groups), x.device)
    fn = lib.sq_mbstd_map_fwd_bf16 if bf else lib.sq_mbstd_map_fwd_f32
    _lib.check(fn(_ptr(x), _ptr(y), _ptr(ws), groups, N // groups, x.numel() // N, cells, _stream()), "sq_mbstd_map_fwd")
    return y


def mbstd_map_bwd(x, dy, groups=1):
    """dx (x's dtype: bf16 features get a bf16 gradient, the f32 value rounded once)"""
    bf = x.dtype == _BF16
    _chk(x, "x", dtype=x.dtype if bf else torch.float32), _chk(dy, "dy")
    N = x.shape[0]
    dx = torch.empty_like(x)
    lib = _lib.load()
    ws = _workspace(lib.sq_mbstd_map_workspace(groups), x.device)
    fn = lib.sq_mbstd_map_bwd_bf16 if bf else lib.sq_mbstd_map_bwd_f32
    _lib.check(fn(_ptr(x), _ptr(dy), _ptr(dx), _ptr(ws), groups, N // groups, x.numel() // N, dy.numel() // N, _stream()),
               "sq_mbstd_map_bwd")
    return dx


def mbstd_map_bwd2(x, dy, v, groups=1):
    bf = x.dtype == _BF16
    _chk(x, "x", dtype=x.dtype if bf else torch.float32), _chk(dy, "dy"), _chk(v, "v", dtype=x.dtype if bf else torch.float32)
    N = x.shape[0]
    ddy, dx2 = torch.empty_like(dy), torch.empty_like(x)
    lib = _lib.load()
    ws = _workspace(lib.sq_mbstd_map_workspace(groups), x.device)
    fn = lib.sq_mbstd_map_bwd2_bf16 if bf else lib.sq_mbstd_map_bwd2_f32
    _lib.check(fn(_ptr(x), _ptr(dy), _ptr(v), _ptr(ddy), _ptr(dx2), _ptr(ws), groups, N // groups, x.numel() // N, dy.numel() // N,
                  _stream()), "sq_mbstd_map_bwd2")
    return ddy, dx2


def wgan_losses(Dz, Dx=None, gn2=None):
    """(d_loss, g_loss) as a 2-element device tensor (gan.py:715-729); Dx = gn2 = None: only g_loss is meaningful"""
    _chk(Dz, "Dz")
    if (Dx is None) != (gn2 is None):
        raise ValueError("wgan_losses: Dx and gn2 go together")
    if Dx is not None:
        _chk(Dx, "Dx"), _chk(gn2, "gn2")
    out = torch.empty((2,), dtype=torch.float32, device=Dz.device)
    _lib.check(_lib.load().sq_wgan_losses_fwd_f32(_ptr(Dz), _ptr(Dx), _ptr(gn2), _ptr(out), Dz.numel(), _stream()),
               "sq_wgan_losses_fwd_f32")
    return out


def wgan_losses_bwd(Dz, Dx, gn2, g_dloss, g_gloss, out_dz=None, out_dx=None):
    """out_dz / out_dx: contiguous destinations (the two halves of one stacked gradient tensor)"""
    dDz = torch.empty_like(Dz) if out_dz is None else out_dz
    dDx = (torch.empty_like(Dx) if out_dx is None else out_dx) if Dx is not None else None
    dgn2 = torch.empty_like(gn2) if gn2 is not None else None
    _lib.check(_lib.load().sq_wgan_losses_bwd_f32(_ptr(Dz), _ptr(Dx), _ptr(gn2), _ptr(g_dloss), _ptr(g_gloss), _ptr(dDz),
                                                 _ptr(dDx), _ptr(dgn2), Dz.numel(), _stream()), "sq_wgan_losses_bwd_f32")
    return dDz, dDx, dgn2


def wgrad1x1_small(a, b):
    """(Ca,Cb) = sum_p a[p,:]^T b[p,:]; a (...,Ca<=4), b (...,Cb%4==0) over the same pixels."""
    if b.dtype == _BF16:
        return _gb().wgrad1x1_small(a, b)
    _chk(a, "a"), _chk(b, "b")
    Ca, Cb = a.shape[-1], b.shape[-1]
    npix = a.numel() // Ca
    if b.numel() // Cb != npix:
        raise ValueError("wgrad1x1_small: operands cover different pixel counts")
    lib = _lib.load()
    nbytes = lib.sq_wgrad1x1_small_workspace_f32(npix, Ca, Cb)
    if nbytes < 0:
        raise _lib.SequitrHipError("wgrad1x1_small: unsupported Ca=%d Cb=%d" % (Ca, Cb))
    ws = _workspace(nbytes, a.device)
    m = torch.empty((Ca, Cb), dtype=torch.float32, device=a.device)
    _lib.check(lib.sq_wgrad1x1_small_f32(_ptr(a), _ptr(b), _ptr(m), _ptr(ws), npix, Ca, Cb, _stream()),
               "sq_wgrad1x1_small_f32")
    return m


# ---- shape-dispatching raw gradients of conv2d (used by sequitr_amd.functional) -----------------
_ONES_CACHE = {}


def _ones(npix, c, device):
    key = (device.index, c)
    t = _ONES_CACHE.get(key)
    if t is None or t.shape[0] < npix:
        t = torch.ones((npix, c), dtype=torch.float32, device=device)
        _ONES_CACHE[key] = t
    return t[:npix]


def conv_dgrad_raw(dy, w, wscale=1.0):
    """dX (N,H,W,Cin) of y = conv2d(x, w*wscale) given dY (N,H,W,Cout); any supported channel mix."""
    K, _, Cin, Cout = w.shape
    if Cin % 4 != 0 and not (K == 1 and Cin <= 4):
        raise _lib.SequitrHipError("conv dgrad to %d channels with K=%d is not supported" % (Cin, K))
    return conv2d(dy, w, None, act=None, wscale=wscale, _dgrad=True)


def conv_dgrad_actgate(dy, w, wscale, gate, act):
    """conv_dgrad_raw followed by act_bwd(., gate, act) in one kernel -- or None where that form does not exist (f32
    precision, small-image mosaics, odd channel counts): the caller then runs the two ops."""
    K, _, Cin, Cout = w.shape                                   # forward filter: the dgrad maps Cout -> Cin channels
    if dy.dtype == _BF16:
        return _gb().conv_dgrad_actgate(dy, w, wscale, gate, act)
    N, H, W, C = dy.shape
    if not (MIXED and ACT[act] and C == Cout and Cout % 8 == 0 and Cin % 4 == 0):
        return None
    if K == 1 and N * H * W <= 128 and Cout >= 1024 and USE_DENSE:
        return None
    _chk(dy, "dy", ndim=4), _chk(gate, "gate", ndim=4)
    if tuple(gate.shape) != (N, H, W, Cin):
        return None
    if USE_MOSAIC and W < 16 and N * H > 1:                     # the small-image levels: mosaic addressing, or the two ops
        plan = _mosaic_plan(N, H, W) if (K == 3 and MOSAIC_IN_KERNEL) else None
        if plan is None:
            return None
        wp = _packed_filter(w, K, Cout, Cin, wscale, True)
        dx = torch.empty((N, H, W, Cin), dtype=torch.float32, device=dy.device)
        _lib.check(_lib.load().sq_conv2d_nhwc_mixed_mosaic_f32(_ptr(dy), _ptr(wp), None, _ptr(gate), _ptr(dx), N, H, W, Cout,
                                                              Cin, ACT[act], plan[0], plan[1], _stream()),
                   "sq_conv2d_nhwc_mixed_mosaic_f32")
        return dx
    wp = _packed_filter(w, K, Cout, Cin, wscale, True)
    dx = torch.empty((N, H, W, Cin), dtype=torch.float32, device=dy.device)
    _lib.check(_lib.load().sq_conv2d_nhwc_dgrad_actgate_mixed_f32(_ptr(dy), _ptr(wp), _ptr(gate), ACT[act], _ptr(dx), N, H, W,
                                                                 Cout, Cin, K, _stream()), "sq_conv2d_nhwc_dgrad_actgate_mixed_f32")
    return dx


def conv_wgrad_raw(x, dy, K, want_bias=False, dw_out=None, db_out=None, dw_scale=1.0):
    """(dW (K,K,Cin,Cout) * dw_scale, db or None) for every channel mix the GAN / U-Net graphs use; the factor rides
    in the finish kernel on the MFMA paths and is one extra multiply kernel on the small image-side 1x1 forms."""
    if x.dtype == _BF16 or dy.dtype == _BF16:
        return _gb().conv_wgrad(x, dy, K, want_bias, dw_out, db_out, dw_scale)
    Cin, Cout = x.shape[-1], dy.shape[-1]
    N, H, W = x.shape[0], x.shape[1], x.shape[2]
    lib = _lib.load()
    if Cout % 4 == 0 and lib.sq_conv2d_nhwc_wgrad_workspace_f32(N, H, W, Cin, Cout, K) >= 0:
        if Cin <= 7 and dw_scale != 1.0:
            dw, db = conv2d_wgrad(x, dy, K, want_bias=want_bias, dw_out=dw_out, db_out=db_out)
            return dw * dw_scale, db
        return conv2d_wgrad(x, dy, K, want_bias=want_bias, dw_out=dw_out, db_out=db_out, dw_scale=dw_scale)
    if dw_scale != 1.0:
        dw, db = conv_wgrad_raw(x, dy, K, want_bias=want_bias, dw_out=dw_out, db_out=db_out)
        return dw * dw_scale, db
    if dw_out is not None or db_out is not None:
        raise _lib.SequitrHipError("conv wgrad: gradient destinations need the MFMA kernel (Cin=%d Cout=%d)" % (Cin, Cout))
    npix = N * H * W
    if K == 1 and Cin <= 7 and Cout % 4 == 0:                      # from_image: image side is the input
        dw = wgrad1x1_small(x, dy).view(1, 1, Cin, Cout)
        db = wgrad1x1_small(_ones(npix, 1, x.device), dy).view(Cout) if want_bias else None
        return dw, db
    if K == 1 and Cout <= 7 and Cin % 4 == 0:                      # to_image / class heads: image side is the output
        dw = wgrad1x1_small(dy, x).t().contiguous().view(1, 1, Cin, Cout)
        db = wgrad1x1_small(dy, _ones(npix, 4, x.device))[:, 0].contiguous() if want_bias else None
        return dw, db
    raise _lib.SequitrHipError("conv wgrad: unsupported Cin=%d Cout=%d K=%d" % (Cin, Cout, K))


# ----------------------------------------------------------------------------------------------
# fused inference variants (include/sequitr_hip.h "Fused inference variants")
# ----------------------------------------------------------------------------------------------
def conv3x3_pool(x, w, bias, act="relu"):
    """conv + bias + act -> (y, maxpool2x2(y)) from one kernel."""
    _chk(x, "x", ndim=4), _chk(w, "w", ndim=4)
    N, H, W, Cin = x.shape
    Cout = w.shape[3]
    if tuple(w.shape[:3]) != (3, 3, Cin):
        raise ValueError("conv3x3_pool: weight %s does not match %d input channels" % (tuple(w.shape), Cin))
    if bias is not None:
        _chk(bias, "bias")
    y = torch.empty((N, H, W, Cout), dtype=torch.float32, device=x.device)
    p = torch.empty((N, H // 2, W // 2, Cout), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    _lib.check(lib.sq_conv3x3_pool_fwd_f32(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), _ptr(p), N, H, W, Cin, Cout,
                                          ACT[act], _stream()), "sq_conv3x3_pool_fwd_f32")
    return y, p


def conv3x3_head(x, w, bias, head_w, head_b, act="relu", want_mask=True):
    """conv (Cin -> 16) + bias + act, then the 1x1 head and argmax: returns (logits, mask)."""
    _chk(x, "x", ndim=4), _chk(w, "w", ndim=4), _chk(head_w, "head_w", ndim=4)
    N, H, W, Cin = x.shape
    if tuple(w.shape) != (3, 3, Cin, 16) or tuple(head_w.shape[:3]) != (1, 1, 16):
        raise ValueError("conv3x3_head needs w (3,3,%d,16) and head_w (1,1,16,C)" % Cin)
    hc = head_w.shape[3]
    logits = torch.empty((N, H, W, hc), dtype=torch.float32, device=x.device)
    mask = torch.empty((N, H, W), dtype=torch.uint8, device=x.device) if want_mask else None
    lib = _lib.load()
    _lib.check(lib.sq_conv3x3_head_fwd_f32(_ptr(x), _ptr(w), _ptr(bias), _ptr(head_w), _ptr(head_b), _ptr(logits),
                                          _ptr(mask), N, H, W, Cin, hc, ACT[act], _stream()), "sq_conv3x3_head_fwd_f32")
    return logits, mask


def convT_conv3x3(x_low, wt, bt, skip, bridge, w, bias, act="relu"):
    """up0 of the U-Net in one kernel: act(conv3x3(bridge(convT2x2s2(x_low) + bt, skip)) + bias) for the
    level-0 shape (x_low (N,H/2,W/2,32), wt (2,2,16,32), skip (N,H,W,16), w (3,3,16,16))."""
    _chk(x_low, "x_low", ndim=4), _chk(wt, "wt", ndim=4), _chk(skip, "skip", ndim=4), _chk(w, "w", ndim=4)
    N, H, W, C = skip.shape
    if C != 16 or tuple(x_low.shape) != (N, H // 2, W // 2, 32) or tuple(wt.shape) != (2, 2, 16, 32) \
            or tuple(w.shape) != (3, 3, 16, 16) or H % 2 or W % 2:
        raise ValueError("convT_conv3x3 is the level-0 block: x_low (N,H/2,W/2,32), skip (N,H,W,16)")
    if bt is not None:
        _chk(bt, "bt")
    if bias is not None:
        _chk(bias, "bias")
    y = torch.empty((N, H, W, 16), dtype=torch.float32, device=skip.device)
    lib = _lib.load()
    _lib.check(lib.sq_convT_conv3x3_fwd_f32(_ptr(x_low), _ptr(wt), _ptr(bt), _ptr(skip), BRIDGE[bridge], _ptr(w),
                                           _ptr(bias), _ptr(y), N, H, W, ACT[act], _stream()),
               "sq_convT_conv3x3_fwd_f32")
    return y


def conv3x3_first_block(x, w1, b1, w2, b2, want_pool=True):
    """down0 conv_block for a 1-channel input: relu(conv2(relu(conv1(x)))) -> (y, pooled or None)."""
    _chk(x, "x", ndim=4), _chk(w1, "w1", ndim=4), _chk(w2, "w2", ndim=4), _chk(b1, "b1"), _chk(b2, "b2")
    N, H, W, Cin = x.shape
    if Cin != 1 or tuple(w1.shape) != (3, 3, 1, 16) or tuple(w2.shape) != (3, 3, 16, 16):
        raise ValueError("conv3x3_first_block is the 1 -> 16 -> 16 level-0 block")
    y = torch.empty((N, H, W, 16), dtype=torch.float32, device=x.device)
    p = torch.empty((N, H // 2, W // 2, 16), dtype=torch.float32, device=x.device) if want_pool else None
    lib = _lib.load()
    _lib.check(lib.sq_conv3x3_first_block_fwd_f32(_ptr(x), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), _ptr(y), _ptr(p),
                                                 N, H, W, _stream()), "sq_conv3x3_first_block_fwd_f32")
    return y, p


def adam_step_dev(p, g, m, v, lr, beta1, beta2, eps, state, grad_scale=1.0):
    """hipGraph-safe Adam: `state` = int32[2] device tensor {step, lr_t bits}; incremented on the device."""
    for t, n in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _chk(t, n)
    _chk(state, "state", dtype=torch.int32)
    lib = _lib.load()
    _lib.check(lib.sq_adam_step_dev_f32(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), float(lr), float(beta1),
                                       float(beta2), float(eps), _ptr(state), float(grad_scale), _stream()),
               "sq_adam_step_dev_f32")
    invalidate_packs()


def adam_advance_dev(state, lr, beta1, beta2):
    """one minimize(): step += 1 and lr_t on the device (`state` int32[2]); hipGraph-safe."""
    _chk(state, "state", dtype=torch.int32)
    _lib.check(_lib.load().sq_adam_advance_dev(_ptr(state), float(lr), float(beta1), float(beta2), _stream()),
               "sq_adam_advance_dev")


def adam_advance_warmup_dev(state, lr, beta1, beta2, warmup_steps):
    """adam_advance_dev with the linear warm-up lr * min(1, t / warmup_steps) evaluated on the device."""
    _chk(state, "state", dtype=torch.int32)
    _lib.check(_lib.load().sq_adam_advance_warmup_dev(_ptr(state), float(lr), float(beta1), float(beta2),
                                                     int(warmup_steps), _stream()), "sq_adam_advance_warmup_dev")


def adam_apply_dev(p, g, m, v, beta1, beta2, eps, state, grad_scale=1.0):
    """Adam update of one tensor with the {step, lr_t} of the last adam_advance_dev."""
    for t, n in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _chk(t, n)
    _chk(state, "state", dtype=torch.int32)
    _lib.check(_lib.load().sq_adam_apply_dev_f32(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), float(beta1), float(beta2),
                                                float(eps), _ptr(state), float(grad_scale), _stream()),
               "sq_adam_apply_dev_f32")
    invalidate_packs()


def adam_table(params, grads, ms, vs):
    """Device table for adam_apply_multi_dev: one row {p, g, m, v, count, first chunk} per tensor.  Built on the host and
    uploaded (a synchronous copy: call it outside any stream capture); the tensors must stay where they are."""
    rows, chunk, first = [], _lib.load().sq_adam_multi_chunk(), 0
    for p, g, m, v in zip(params, grads, ms, vs):
        for t, n in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
            _chk(t, n)
        if not (p.numel() == g.numel() == m.numel() == v.numel()):
            raise ValueError("adam_table: tensor sizes differ")
        rows.append([p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), first])
        first += (p.numel() + chunk - 1) // chunk
    table = torch.tensor(rows, dtype=torch.int64).to(params[0].device)
    table._sq_chunks = first
    return table


def adam_apply_multi_dev(table, beta1, beta2, eps, state, grad_scale=1.0):
    """adam_apply_dev over every row of an adam_table() in one launch."""
    _chk(table, "table", dtype=torch.int64), _chk(state, "state", dtype=torch.int32)
    _lib.check(_lib.load().sq_adam_apply_multi_dev_f32(_ptr(table), int(table.shape[0]), int(table._sq_chunks), float(beta1),
                                                      float(beta2), float(eps), _ptr(state), float(grad_scale), _stream()),
               "sq_adam_apply_multi_dev_f32")
    invalidate_packs()


# ----------------------------------------------------------------------------------------------
# batch normalisation (include/sequitr_hip.h "Batch normalisation"; SURVEY.md A.1 `batch_norm`)
# ----------------------------------------------------------------------------------------------
BN_EPS, BN_MOMENTUM = 1e-3, 0.99                  # tf.layers.batch_normalization defaults


def _bn_shape(x):
    _chk(x, "x")
    C = x.shape[-1]
    return x.numel() // C, C


def bn_stats(x):
    """(mean, population variance) per channel over every leading axis of the NHWC tensor x."""
    npix, C = _bn_shape(x)
    lib = _lib.load()
    nbytes = lib.sq_bn_workspace_f32(npix, C)
    if nbytes < 0:
        raise _lib.SequitrHipError("bn_stats: unsupported channel count %d" % C)
    ws = _workspace(nbytes, x.device)
    mean = torch.empty((C,), dtype=torch.float32, device=x.device)
    var = torch.empty((C,), dtype=torch.float32, device=x.device)
    _lib.check(lib.sq_bn_stats_f32(_ptr(x), _ptr(mean), _ptr(var), _ptr(ws), npix, C, _stream()), "sq_bn_stats_f32")
    return mean, var


def bn_fold(gamma, beta, mean, var, eps=BN_EPS):
    for t, n in ((gamma, "gamma"), (beta, "beta"), (mean, "mean"), (var, "var")):
        _chk(t, n)
    C = gamma.numel()
    scale, shift = torch.empty_like(gamma), torch.empty_like(gamma)
    lib = _lib.load()
    _lib.check(lib.sq_bn_fold_f32(_ptr(gamma), _ptr(beta), _ptr(mean), _ptr(var), float(eps), _ptr(scale),
                                 _ptr(shift), C, _stream()), "sq_bn_fold_f32")
    return scale, shift


def bn_apply(x, scale, shift, act=None):
    npix, C = _bn_shape(x)
    _chk(scale, "scale"), _chk(shift, "shift")
    y = torch.empty_like(x)
    lib = _lib.load()
    _lib.check(lib.sq_bn_apply_f32(_ptr(x), _ptr(scale), _ptr(shift), _ptr(y), npix, C, ACT[act], _stream()),
               "sq_bn_apply_f32")
    return y


def bn_update_moving_(moving_mean, moving_var, mean, var, npix, momentum=BN_MOMENTUM):
    lib = _lib.load()
    _lib.check(lib.sq_bn_update_moving_f32(_ptr(moving_mean), _ptr(moving_var), _ptr(mean), _ptr(var),
                                          float(momentum), int(npix), moving_mean.numel(), _stream()),
               "sq_bn_update_moving_f32")


def bn_inference(x, gamma, beta, moving_mean, moving_var, eps=BN_EPS, act=None):
    """y = act(BN(x)) with the moving statistics (training == False)."""
    scale, shift = bn_fold(gamma, beta, moving_mean, moving_var, eps)
    return bn_apply(x, scale, shift, act)


def bn_bwd(x, dy, y, act, mean, var, gamma, eps=BN_EPS):
    """(dx, dgamma, dbeta) of y = act(BN_batchstats(x)); y is needed only when act is not None."""
    npix, C = _bn_shape(x)
    _chk(dy, "dy")
    lib = _lib.load()
    ws = _workspace(lib.sq_bn_workspace_f32(npix, C), x.device)
    dx = torch.empty_like(x)
    dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(gamma)
    _lib.check(lib.sq_bn_bwd_f32(_ptr(x), _ptr(dy), _ptr(y) if ACT[act] else None, ACT[act], _ptr(mean), _ptr(var),
                                _ptr(gamma), float(eps), _ptr(dx), _ptr(dgamma), _ptr(dbeta), _ptr(ws), npix, C,
                                _stream()), "sq_bn_bwd_f32")
    return dx, dgamma, dbeta


# ----------------------------------------------------------------------------------------------
# 3x3 / stride-2 transpose conv (`up_kernel` = (3,3)) = zero insertion + SAME 3x3 conv
# ----------------------------------------------------------------------------------------------
def zero_insert2x(x):
    _chk(x, "x", ndim=4)
    N, H, W, C = x.shape
    u = torch.empty((N, 2 * H, 2 * W, C), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    _lib.check(lib.sq_zero_insert2x_f32(_ptr(x), _ptr(u), N, H, W, C, _stream()), "sq_zero_insert2x_f32")
    return u


def gather_odd2x(du):
    _chk(du, "du", ndim=4)
    N, H2, W2, C = du.shape
    dx = torch.empty((N, H2 // 2, W2 // 2, C), dtype=torch.float32, device=du.device)
    lib = _lib.load()
    _lib.check(lib.sq_gather_odd2x_f32(_ptr(du), _ptr(dx), N, H2 // 2, W2 // 2, C, _stream()), "sq_gather_odd2x_f32")
    return dx


def convT3x3s2(x, w, bias=None):
    """TF conv2d_transpose(kernel 3, stride 2, SAME): x (N,H,W,Cin), w (3,3,Cout,Cin) -> (N,2H,2W,Cout)."""
    _chk(w, "w", ndim=4)
    if tuple(w.shape[:2]) != (3, 3) or w.shape[3] != x.shape[3]:
        raise ValueError("convT3x3s2: weight %s does not match %d input channels" % (tuple(w.shape), x.shape[3]))
    return conv2d(zero_insert2x(x), conv_weight_transform(w), bias, act=None)


# ----------------------------------------------------------------------------------------------
# EDT weight maps on the device (include/sequitr_hip.h "EDT weight maps"; pipeline.py:475-479)
# ----------------------------------------------------------------------------------------------
def _wm_args(img):
    _chk(img, "img")
    if img.dim() == 4 and img.shape[-1] == 1:
        img = img.reshape(img.shape[:3])
    if img.dim() != 3:
        raise ValueError("img must be (N,H,W) or (N,H,W,1), got %s" % (tuple(img.shape),))
    N, H, W = img.shape
    lib = _lib.load()
    nbytes = lib.sq_weightmap_workspace(N, H, W)
    if nbytes < 0:
        raise ValueError("weight map batch %s is too large for one call" % (tuple(img.shape),))
    return img, N, H, W, lib, _workspace(nbytes, img.device)


def edt_squared(img):
    """Exact squared Euclidean distance (int32) of every pixel to the nearest pixel with 1 - img == 0."""
    img, N, H, W, lib, ws = _wm_args(img)
    d2 = torch.empty((N, H, W), dtype=torch.int32, device=img.device)
    _lib.check(lib.sq_edt_sq_f32(_ptr(img), _ptr(d2), _ptr(ws), N, H, W, _stream()), "sq_edt_sq_f32")
    return d2


def weightmap_edt(img, w0=10.0, sigma=5.0, dtype=torch.float32):
    """ImageWeightMap (pipeline.py:455-479) of a batch of binary label images, on the device.
    dtype float64 = the reference's own precision; float32 = the loss kernel's `weights` operand."""
    img, N, H, W, lib, ws = _wm_args(img)
    out = torch.empty((N, H, W), dtype=dtype, device=img.device)
    o64, o32 = (_ptr(out), None) if dtype == torch.float64 else (None, _ptr(out))
    if dtype not in (torch.float64, torch.float32):
        raise TypeError("weightmap_edt: dtype must be float32 or float64")
    _lib.check(lib.sq_weightmap_edt_f32(_ptr(img), o64, o32, _ptr(ws), N, H, W, float(w0), float(sigma), _stream()),
               "sq_weightmap_edt_f32")
    return out


def wm2_boundary_points(img):
    """ImageWeightMap2's boundary-point mask (pipeline.py:516-528) of (N,H,W) binary f32 labels: uint8 (N,H,W)."""
    _chk(img, "img", ndim=3)
    N, H, W = img.shape
    pts = torch.empty((N, H, W), dtype=torch.uint8, device=img.device)
    _lib.check(_lib.load().sq_wm2_boundary_points_u8(_ptr(img), _ptr(pts), N, H, W, _stream()), "sq_wm2_boundary_points_u8")
    return pts


_DELAUNAY_OUT = {}


def delaunay2d_batch(xy, offsets, compact=False):
    """HOST: exact Delaunay triangulation of each tile's integer points (sq_delaunay2d_batch_i32, native, threaded).
    xy (P,2) int32 CPU tensor (row, column), offsets (T+1) int64 CPU tensor.  Returns (simplices (S,7) int32, longest (S)
    float64) as views of a pinned staging buffer ready for the upload -- overwritten by the next call: upload (or
    clone) them first.  S = 2 P: tile s owns rows 2 offsets[s] .. 2 offsets[s+1] - 1 and the few it does not need are
    padding with tile = -1 (weightmap_delaunay skips them); compact=True returns copies without the padding rows."""
    if xy.device.type != "cpu" or offsets.device.type != "cpu" or xy.dtype != torch.int32 or offsets.dtype != torch.int64:
        raise TypeError("delaunay2d_batch takes CPU tensors: xy int32 (P,2), offsets int64 (T+1)")
    xy, offsets = xy.contiguous(), offsets.contiguous()
    nsets = int(offsets.numel()) - 1
    cap = max(2 * int(offsets[-1]), 1)
    buf = _DELAUNAY_OUT.get("buf")
    if buf is None or buf[0].shape[0] < cap:                    # pinned staging, kept across calls (grow-only)
        pin = torch.cuda.is_available()
        room = cap + cap // 4
        buf = _DELAUNAY_OUT["buf"] = (torch.empty((room, 7), dtype=torch.int32, pin_memory=pin),
                                      torch.empty((room,), dtype=torch.float64, pin_memory=pin))
    simp, lng = buf
    n = _lib.load().sq_delaunay2d_batch_i32(xy.data_ptr(), offsets.data_ptr(), nsets, simp.data_ptr(), lng.data_ptr(), cap)
    if n < 0:
        _lib.check(int(n), "sq_delaunay2d_batch_i32")
    if compact:
        keep = simp[:n, 0] >= 0
        return simp[:n][keep].clone(), lng[:n][keep].clone()
    return simp[:n], lng[:n]


def weightmap_delaunay(img, simplices, longest, w0=10.0, sigma=5.0, dtype=torch.float32):
    """The per-pixel part of ImageWeightMap2 (pipeline.py:514-566) on the device: img (N,H,W) binary f32, simplices
    (S,7) int32 {tile, x0,y0, x1,y1, x2,y2}, longest (S) float64 -- see sq_weightmap2_delaunay_f32."""
    _chk(img, "img")
    if img.dim() == 4 and img.shape[-1] == 1:
        img = img.reshape(img.shape[:3])
    if img.dim() != 3:
        raise ValueError("img must be (N,H,W) or (N,H,W,1), got %s" % (tuple(img.shape),))
    _chk(simplices, "simplices", dtype=torch.int32, ndim=2), _chk(longest, "longest", dtype=torch.float64, ndim=1)
    if simplices.shape[1] != 7 or simplices.shape[0] != longest.shape[0] or simplices.shape[0] == 0:
        raise ValueError("simplices must be (S,7) int32 with S = len(longest) > 0")
    if dtype not in (torch.float64, torch.float32):
        raise TypeError("weightmap_delaunay: dtype must be float32 or float64")
    N, H, W = img.shape
    lib = _lib.load()
    nbytes = lib.sq_weightmap2_workspace(N, H, W)
    if nbytes < 0:
        raise ValueError("weight map batch %s is too large for one call" % (tuple(img.shape),))
    ws = _workspace(nbytes, img.device)
    out = torch.empty((N, H, W), dtype=dtype, device=img.device)
    o64, o32 = (_ptr(out), None) if dtype == torch.float64 else (None, _ptr(out))
    _lib.check(lib.sq_weightmap2_delaunay_f32(_ptr(img), _ptr(simplices), _ptr(longest), int(simplices.shape[0]), o64, o32,
                                             _ptr(ws), N, H, W, float(w0), float(sigma), _stream()),
               "sq_weightmap2_delaunay_f32")
    return out
