"""bf16 operator layer (BASELINE configs 3-5): torch.bfloat16 activations in HBM, fp32 master
weights / biases / accumulation; every op is a hand-written HIP kernel of libsequitr_hip.so."""
import torch

from . import _lib
from .ops import ACT, BRIDGE, _ptr, _stream, _workspace, _grad_out  # noqa: F401

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
    packs = getattr(w, "_sq_packs", None)                     # filled once per step by PackPlan.run()
    if packs is not None and wscale == 1.0:
        hit = packs.get("T" if transform else "N")
        if hit is not None:
            return hit
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


def conv2d_dropout(x, wp, bias, K, Cout, act, rate, seed=0, step_dev=None):
    """dropout(act(conv(x))) in one kernel; the mask is sq_dropout_fwd_bf16's hash and is not stored."""
    _chk(x, "x", ndim=4), _chk(wp, "wp")
    N, H, W, Cin = x.shape
    if bias is not None:
        _chk(bias, "bias", dtype=torch.float32)
    y = torch.empty((N, H, W, Cout), dtype=BF16, device=x.device)
    lib = _lib.load()
    _lib.check(lib.sq_conv2d_nhwc_fwd_dropout_bf16(_ptr(x), _ptr(wp), _ptr(bias), _ptr(y), N, H, W, Cin, Cout, K,
                                                  ACT[act], float(rate), int(seed) & 0xFFFFFFFF, _ptr(step_dev),
                                                  _stream()), "sq_conv2d_nhwc_fwd_dropout_bf16")
    return y


def sign_mask_like(y):
    """storage for the sign mask of a (N,H,W,C) ReLU output: N*H*W*C/8 bytes"""
    N, H, W, C = y.shape
    if C % 16:
        raise ValueError("sign mask: C=%d must be a multiple of 16" % C)
    return torch.empty((N, H, W, C // 8), dtype=torch.uint8, device=y.device)


def conv2d_mask(x, wp, bias, K, Cout, act="relu"):
    """(y, sign mask of y) of act(conv(x)): the mask leaves the conv's epilogue beside y."""
    _chk(x, "x", ndim=4), _chk(wp, "wp")
    N, H, W, Cin = x.shape
    if bias is not None:
        _chk(bias, "bias", dtype=torch.float32)
    y = torch.empty((N, H, W, Cout), dtype=BF16, device=x.device)
    m = sign_mask_like(y)
    _lib.check(_lib.load().sq_conv2d_nhwc_fwd_mask_bf16(_ptr(x), _ptr(wp), _ptr(bias), _ptr(y), _ptr(m), N, H, W, Cin, Cout, K,
                                                       ACT[act], _stream()), "sq_conv2d_nhwc_fwd_mask_bf16")
    return y, m


def conv3x3_first_mask(x, w, bias, act="relu"):
    """conv3x3_first with the sign mask of its output."""
    _chk(x, "x", dtype=torch.float32, ndim=4), _chk(w, "w", dtype=torch.float32, ndim=4)
    N, H, W, Cin = x.shape
    Cout = w.shape[3]
    y = torch.empty((N, H, W, Cout), dtype=BF16, device=x.device)
    m = sign_mask_like(y)
    _lib.check(_lib.load().sq_conv3x3_first_fwd_mask_bf16(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), _ptr(m), N, H, W, Cin, Cout,
                                                         ACT[act], _stream()), "sq_conv3x3_first_fwd_mask_bf16")
    return y, m


def conv2d_dgrad_mask(dy, wp_t, mask, K, Cout, scale=1.0):
    """conv2d_dgrad_relu with the gate read from a sign mask (conv2d_mask / conv3x3_first_mask) instead of the tensor."""
    _chk(dy, "dy", ndim=4), _chk(wp_t, "wp_t"), _chk(mask, "mask", dtype=torch.uint8)
    N, H, W, Cin = dy.shape
    if mask.numel() != N * H * W * Cout // 8:
        raise ValueError("mask of %d bytes does not fit (%d,%d,%d,%d)" % (mask.numel(), N, H, W, Cout))
    dx = torch.empty((N, H, W, Cout), dtype=BF16, device=dy.device)
    _lib.check(_lib.load().sq_conv2d_nhwc_dgrad_maskgate_bf16(_ptr(dy), _ptr(wp_t), _ptr(mask), float(scale), _ptr(dx), N, H,
                                                             W, Cin, Cout, K, _stream()), "sq_conv2d_nhwc_dgrad_maskgate_bf16")
    return dx


def conv2d_dropout_pool(x, wp, bias, K, Cout, act, rate, seed=0, step_dev=None):
    """(y, maxpool2x2(y)) with y = dropout(act(conv(x))) (rate 0: no dropout), one kernel: the pooled tensor is written from
    the conv's epilogue instead of being re-read from y by a pooling pass."""
    _chk(x, "x", ndim=4), _chk(wp, "wp")
    N, H, W, Cin = x.shape
    if bias is not None:
        _chk(bias, "bias", dtype=torch.float32)
    y = torch.empty((N, H, W, Cout), dtype=BF16, device=x.device)
    yp = torch.empty((N, H // 2, W // 2, Cout), dtype=BF16, device=x.device)
    lib = _lib.load()
    _lib.check(lib.sq_conv2d_nhwc_fwd_dropout_pool_bf16(_ptr(x), _ptr(wp), _ptr(bias), _ptr(y), _ptr(yp), N, H, W, Cin, Cout,
                                                       K, ACT[act], float(rate), int(seed) & 0xFFFFFFFF, _ptr(step_dev),
                                                       _stream()), "sq_conv2d_nhwc_fwd_dropout_pool_bf16")
    return y, yp


def conv_first_block_takes(x, w1, w2):
    """the one-launch first block: a single-channel f32 image, 16 filters, even H and W."""
    return (x.dtype == torch.float32 and x.dim() == 4 and x.shape[3] == 1 and tuple(w1.shape) == (3, 3, 1, 16) and
            tuple(w2.shape) == (3, 3, 16, 16) and x.shape[1] % 2 == 0 and x.shape[2] % 2 == 0)


def conv_first_block_dropout_pool(x, w1, b1, wp2, b2, rate, seed=0, step_dev=None):
    """(y1, mask1, y, maxpool2x2(y)): conv3x3_first_mask(x, w1, b1) followed by conv2d_dropout_pool(y1, wp2, b2, 3, 16, 'relu',
    rate) as ONE kernel -- y1 is made per tile in the block and only written (for conv2's weight gradient), never re-read."""
    _chk(x, "x", dtype=torch.float32, ndim=4), _chk(w1, "w1", dtype=torch.float32, ndim=4), _chk(wp2, "wp2")
    N, H, W, _ = x.shape
    y1 = torch.empty((N, H, W, 16), dtype=BF16, device=x.device)
    m1 = sign_mask_like(y1)
    y = torch.empty((N, H, W, 16), dtype=BF16, device=x.device)
    yp = torch.empty((N, H // 2, W // 2, 16), dtype=BF16, device=x.device)
    _lib.check(_lib.load().sq_conv3x3_first_block_dropout_pool_bf16(
        _ptr(x), _ptr(w1), _ptr(b1), _ptr(y1), _ptr(m1), _ptr(wp2), _ptr(b2), _ptr(y), _ptr(yp), N, H, W, float(rate),
        int(seed) & 0xFFFFFFFF, _ptr(step_dev), _stream()), "sq_conv3x3_first_block_dropout_pool_bf16")
    return y1, m1, y, yp


def relu_scale_bwd(dy, y, scale):
    """dx = y > 0 ? dy * scale : 0 -- backward of dropout(relu(.)) from its output alone."""
    _chk(dy, "dy"), _chk(y, "y")
    dx = torch.empty_like(dy)
    lib = _lib.load()
    _lib.check(lib.sq_relu_scale_bwd_bf16(_ptr(dy), _ptr(y), _ptr(dx), dy.numel(), float(scale), _stream()),
               "sq_relu_scale_bwd_bf16")
    return dx


def conv2d_dgrad_relu(dy, wp_t, gate, K, scale=1.0):
    """dX of a conv whose input was the ReLU output `gate`: conv(dy, wp_t) passed where gate > 0; scale != 1: the
    input was dropout(ReLU(.)) = `gate` and what passes is multiplied by scale = 1 / (1 - rate)."""
    _chk(dy, "dy", ndim=4), _chk(wp_t, "wp_t"), _chk(gate, "gate", ndim=4)
    N, H, W, Cin = dy.shape
    Cout = gate.shape[3]
    if tuple(gate.shape[:3]) != (N, H, W):
        raise ValueError("gate %s does not match dy %s" % (tuple(gate.shape), tuple(dy.shape)))
    dx = torch.empty((N, H, W, Cout), dtype=BF16, device=dy.device)
    lib = _lib.load()
    if scale == 1.0:
        _lib.check(lib.sq_conv2d_nhwc_dgrad_relu_bf16(_ptr(dy), _ptr(wp_t), _ptr(gate), _ptr(dx), N, H, W, Cin, Cout, K,
                                                     _stream()), "sq_conv2d_nhwc_dgrad_relu_bf16")
    else:
        _lib.check(lib.sq_conv2d_nhwc_dgrad_gate_bf16(_ptr(dy), _ptr(wp_t), _ptr(gate), float(scale), _ptr(dx), N, H, W,
                                                     Cin, Cout, K, _stream()), "sq_conv2d_nhwc_dgrad_gate_bf16")
    return dx


def conv2d_dgrad_junction(dy, wp_t, up, skip, kind, K, Cout):
    """dgrad conv (dy -> d merged, Cout channels) with the decoder junction's backward in the epilogue: returns
    (g (N,H/2,W/2,4Cout) = d_up in space-to-depth layout, dskip (N,H,W,Cout)); d merged is never written."""
    _chk(dy, "dy", ndim=4), _chk(wp_t, "wp_t")
    N, H, W, Cin = dy.shape
    if kind == 'eltwise_mul':
        _chk(up, "up", ndim=4), _chk(skip, "skip", ndim=4)
        if tuple(up.shape) != (N, H, W, Cout) or tuple(skip.shape) != (N, H, W, Cout):
            raise ValueError("conv2d_dgrad_junction: up / skip must be %s" % ((N, H, W, Cout),))
    g = torch.empty((N, H // 2, W // 2, 4 * Cout), dtype=BF16, device=dy.device)
    dskip = torch.empty((N, H, W, Cout), dtype=BF16, device=dy.device)
    lib = _lib.load()
    _lib.check(lib.sq_conv2d_nhwc_dgrad_junction_bf16(_ptr(dy), _ptr(wp_t), _ptr(up), _ptr(skip), _ptr(g), _ptr(dskip), N, H,
                                                     W, Cin, Cout, K, BRIDGE[kind], _stream()),
               "sq_conv2d_nhwc_dgrad_junction_bf16")
    return g, dskip


def conv3x3_first(x, w, bias, act="relu"):
    """f32 (N,H,W,Cin) image (Cin 1..7), f32 (3,3,Cin,Cout) filter -> bf16 activation."""
    _chk(x, "x", dtype=torch.float32, ndim=4), _chk(w, "w", dtype=torch.float32, ndim=4)
    N, H, W, Cin = x.shape
    Cout = w.shape[3]
    if tuple(w.shape[:3]) != (3, 3, Cin):
        raise ValueError("conv3x3_first: filter %s does not match %d input channels" % (tuple(w.shape), Cin))
    y = torch.empty((N, H, W, Cout), dtype=BF16, device=x.device)
    lib = _lib.load()
    _lib.check(lib.sq_conv3x3_first_fwd_bf16(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), N, H, W, Cin, Cout, ACT[act],
                                             _stream()), "sq_conv3x3_first_fwd_bf16")
    return y


# ---- weight gradients of several layers in one launch -------------------------------------------------------------------
import os as _os

WGRAD_GROUP_MAX_ELEMS = int(_os.environ.get("SQ_WGRAD_GROUP", str(1 << 31)))    # 0: off; layers with fewer X elements defer (default: all)
_QUEUE = [None]                 # a module slot, not thread-local: the autograd engine runs backward on its own thread


class WgradQueue(object):
    """Collects the weight-gradient work of a backward pass -- (x, dY, destination sinks) of every layer small enough that
    x.numel() < WGRAD_GROUP_MAX_ELEMS (default: every layer; deep-only measured 0.5 % slower) --
    and runs it as ONE grouped launch + one grouped finish launch per kernel shape (sq_conv2d_nhwc_wgrad_group_bf16).
    Same kernel body per layer; each layer is cut into fewer, longer blocks than when launched alone, so the f32 sums are
    grouped differently (equal to f32 rounding, run-to-run identical).  Only layers whose gradients go straight to a sink
    are deferred."""

    def __init__(self, max_elems=None):
        self.max_elems = WGRAD_GROUP_MAX_ELEMS if max_elems is None else int(max_elems)
        self.items, self.keep = [], []
        self.seen_dw, self.seen_db = set(), set()               # destinations already written by an item of this queue

    def takes(self, x, Cin, Cout, dw_out):
        return (self.max_elems > 0 and dw_out is not None and Cin % 16 == 0 and Cout % 16 == 0
                and x.numel() < self.max_elems)

    def push(self, x, dy, K, dw, db, convT_cout=0, dw_scale=1.0, mosaic=None):
        """a second item for the same destination ACCUMULATES (a parameter that several passes of one step use, e.g. the GAN's
        discriminator: the stacked D(Gz | X) pass and the penalty's pass through D(mix)); the first one writes"""
        N, H, W, Cin = x.shape
        pw, pb = _ptr(dw), _ptr(db)
        acc = (1 if pw in self.seen_dw else 0) | (2 if (pb is not None and pb in self.seen_db) else 0)
        self.seen_dw.add(pw)
        if pb is not None:
            self.seen_db.add(pb)
        R, Cc = mosaic if mosaic is not None else (0, 0)          # small images taken as one mosaic (ops._mosaic_plan)
        self.items.append((_ptr(x), _ptr(dy), pw, pb, N, H, W, Cin, dy.shape[3], K, convT_cout, float(dw_scale), acc, R, Cc))
        self.keep.append((x, dy, dw, db))                       # alive until the launch has been enqueued

    def flush(self):
        n = len(self.items)
        if not n:
            return
        arr = (_lib.WgradItem * n)()
        for i, it in enumerate(self.items):
            (arr[i].x, arr[i].dy, arr[i].dw, arr[i].db, arr[i].N, arr[i].H, arr[i].W, arr[i].Cin, arr[i].Cout, arr[i].K,
             arr[i].convT_cout, arr[i].dw_scale, arr[i].accumulate, arr[i].mosaic_R, arr[i].mosaic_Cc) = it
        lib = _lib.load()
        nbytes = lib.sq_conv2d_nhwc_wgrad_group_workspace_bf16(arr, n)
        if nbytes < 0:
            raise _lib.SequitrHipError("WgradQueue: an item the grouped kernel does not take")
        ws = _workspace(nbytes, self.keep[0][0].device)
        _lib.check(lib.sq_conv2d_nhwc_wgrad_group_bf16(arr, n, _ptr(ws), _stream()), "sq_conv2d_nhwc_wgrad_group_bf16")
        self.items, self.keep = [], []                          # seen_* stay: a later flush of this queue still accumulates


class deferred_wgrads(object):
    """`with ob.deferred_wgrads(): loss.backward()` -- the queue is flushed on the way out (inside a capture: captured)"""

    def __init__(self, max_elems=None):
        self.q = WgradQueue(max_elems)

    def __enter__(self):
        self.prev, _QUEUE[0] = _QUEUE[0], self.q
        return self.q

    def __exit__(self, et, ev, tb):
        _QUEUE[0] = self.prev
        if et is None:
            self.q.flush()
        return False


def conv2d_wgrad(x, dy, K, want_bias=True, dw_out=None, db_out=None):
    """(dW (K,K,Cin,Cout) f32, db f32 or None) from bf16 X and bf16 dY."""
    _chk(x, "x", ndim=4), _chk(dy, "dy", ndim=4)
    N, H, W, Cin = x.shape
    Cout = dy.shape[3]
    q = _QUEUE[0]
    if q is not None and q.takes(x, Cin, Cout, dw_out) and (db_out is not None or not want_bias):
        q.push(x, dy, K, dw_out, db_out if want_bias else None)
        return dw_out, (db_out if want_bias else None)
    lib = _lib.load()
    nbytes = lib.sq_conv2d_nhwc_wgrad_workspace_bf16(N, H, W, Cin, Cout, K)
    if nbytes < 0:
        raise _lib.SequitrHipError("conv2d_wgrad(bf16): unsupported Cin=%d Cout=%d K=%d" % (Cin, Cout, K))
    ws = _workspace(nbytes, x.device)
    dw = _grad_out(dw_out, (K, K, Cin, Cout), x.device)
    db = _grad_out(db_out, (Cout,), x.device) if want_bias else None
    _lib.check(lib.sq_conv2d_nhwc_wgrad_bf16(_ptr(x), _ptr(dy), _ptr(dw), _ptr(db), _ptr(ws), N, H, W, Cin, Cout, K,
                                            _stream()), "sq_conv2d_nhwc_wgrad_bf16")
    return dw, db


def convT_wgrad(x, g, Cout, want_bias=True, dw_out=None, db_out=None):
    """(dW (2,2,Cout,Cin) f32, db (Cout) f32 or None) of the 2x2/s2 transpose conv from its bf16 input x and its
    output gradient in space-to-depth form g (N,H,W,4*Cout)."""
    _chk(x, "x", ndim=4), _chk(g, "g", ndim=4)
    N, H, W, Cin = x.shape
    if tuple(g.shape) != (N, H, W, 4 * Cout):
        raise ValueError("convT_wgrad: g must be %s" % ((N, H, W, 4 * Cout),))
    q = _QUEUE[0]
    if q is not None and q.takes(x, Cin, 4 * Cout, dw_out) and (db_out is not None or not want_bias):
        q.push(x, g, 1, dw_out, db_out if want_bias else None, convT_cout=Cout)
        return dw_out, (db_out if want_bias else None)
    lib = _lib.load()
    nbytes = lib.sq_conv2d_nhwc_wgrad_workspace_bf16(N, H, W, Cin, 4 * Cout, 1)
    if nbytes < 0:
        raise _lib.SequitrHipError("convT_wgrad(bf16): unsupported Cin=%d Cout=%d" % (Cin, Cout))
    ws = _workspace(nbytes, x.device)
    dw = _grad_out(dw_out, (2, 2, Cout, Cin), x.device)
    db = _grad_out(db_out, (Cout,), x.device) if want_bias else None
    _lib.check(lib.sq_convT2x2s2_wgrad_bf16(_ptr(x), _ptr(g), _ptr(dw), _ptr(db), _ptr(ws), N, H, W, Cin, Cout, _stream()),
               "sq_convT2x2s2_wgrad_bf16")
    return dw, db


class PackPlan(object):
    """All bf16 filter packs of one training step as ONE launch (sq_conv_pack_weights_multi_bf16).
    `flat` is the flat fp32 parameter buffer, `leaves` {name: (leaf tensor, float offset)}; every 3x3 / 1x1
    kernel with Cin % 16 == 0 gets its forward pack ("N") and its dgrad pack ("T"), every (2,2,Cout,Cin)
    transpose-conv kernel a plain bf16 copy ("cast") and the pack of its 1x1 dgrad form ("convT_dgrad").
    The packed views hang on the leaves as `_sq_packs`; pack_weights()/to_bf16() pick them up."""

    def __init__(self, flat, leaves):
        lib = _lib.load()
        rows, dst, item = [], 0, 0

        def add(src, K, ci, co, transform, kind, count):
            nonlocal dst, item
            if count % 8:                                       # the pack kernel writes 8 elements (16 bytes) per thread
                raise _lib.SequitrHipError("PackPlan: %d elements in a pack (multiple of 8 needed)" % count)
            rows.append([src, dst, K, ci, co, transform, item, kind])
            view = (dst, count)
            dst += (count + 7) // 8 * 8                         # keep every pack 16-byte aligned
            item += count
            return view

        views = {}
        for name, (leaf, off) in leaves.items():
            if leaf.dim() != 4:
                continue
            K, K2, A, B = leaf.shape
            if 'upscale' in name and K == 2:                    # (2,2,Cout,Cin) transpose-conv kernel
                Cout, Cin = A, B
                if Cin % 16 or (4 * Cout) % 16:
                    continue
                views[name] = {'cast': add(off, 2, Cin, Cout, 0, 1, leaf.numel())}
                n = lib.sq_conv_packed_weights_elems_bf16(1, 4 * Cout, Cin)
                if n > 0:                                       # HWIO view (1,1,4Cout,Cin) -> conv 4Cout -> Cin
                    views[name]['convT_dgrad'] = add(off, 1, 4 * Cout, Cin, 0, 0, n)
                continue
            Cin, Cout = A, B
            v = {}
            n = lib.sq_conv_packed_weights_elems_bf16(K, Cin, Cout)
            if K == K2 and n > 0:
                v['N'] = add(off, K, Cin, Cout, 0, 0, n)
            n = lib.sq_conv_packed_weights_elems_bf16(K, Cout, Cin)
            if K == K2 and n > 0:
                v['T'] = add(off, K, Cout, Cin, 1, 0, n)
            if v:
                views[name] = v
        if len(rows) > 128:
            raise _lib.SequitrHipError("PackPlan: %d packs exceed the 128-entry table" % len(rows))
        self.flat, self.total, self.n = flat, item, len(rows)
        self.table = torch.tensor(rows, dtype=torch.int32, device=flat.device).contiguous()
        self.out = torch.zeros(max(dst, 8), dtype=BF16, device=flat.device)
        for name, v in views.items():
            leaf = leaves[name][0]
            leaf._sq_packs = {}
            for key, (d0, cnt) in v.items():
                t = self.out[d0:d0 + cnt]
                leaf._sq_packs[key] = t.view(leaf.shape) if key == 'cast' else t

    def run(self):
        if self.n == 0:
            return
        lib = _lib.load()
        _lib.check(lib.sq_conv_pack_weights_multi_bf16(_ptr(self.flat), _ptr(self.out), _ptr(self.table), self.n,
                                                      self.total, _stream()), "sq_conv_pack_weights_multi_bf16")


def to_bf16(x):
    packs = getattr(x, "_sq_packs", None)
    if packs is not None and packs.get("cast") is not None:
        return packs["cast"]
    _chk(x, "x", dtype=torch.float32)
    y = torch.empty(x.shape, dtype=BF16, device=x.device)
    lib = _lib.load()
    _lib.check(lib.sq_cast_f32_to_bf16(_ptr(x), _ptr(y), x.numel(), _stream()), "sq_cast_f32_to_bf16")
    return y


def to_f32(x):
    _chk(x, "x")
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    lib = _lib.load()
    _lib.check(lib.sq_cast_bf16_to_f32(_ptr(x), _ptr(y), x.numel(), _stream()), "sq_cast_bf16_to_f32")
    return y


def maxpool2x2(x):
    _chk(x, "x", ndim=4)
    N, H, W, C = x.shape
    y = torch.empty((N, H // 2, W // 2, C), dtype=BF16, device=x.device)
    lib = _lib.load()
    _lib.check(lib.sq_maxpool2x2_fwd_bf16(_ptr(x), _ptr(y), N, H, W, C, _stream()), "sq_maxpool2x2_fwd_bf16")
    return y


def maxpool2x2_bwd(x, dy):
    _chk(x, "x", ndim=4), _chk(dy, "dy", ndim=4)
    N, H, W, C = x.shape
    dx = torch.empty_like(x)
    lib = _lib.load()
    _lib.check(lib.sq_maxpool2x2_bwd_bf16(_ptr(x), _ptr(dy), _ptr(dx), N, H, W, C, _stream()), "sq_maxpool2x2_bwd_bf16")
    return dx


def act_bwd(dy, y, act):
    _chk(dy, "dy"), _chk(y, "y")
    if ACT[act] == 0:
        return dy
    dx = torch.empty_like(dy)
    lib = _lib.load()
    _lib.check(lib.sq_act_bwd_bf16(_ptr(dy), _ptr(y), _ptr(dx), dy.numel(), ACT[act], _stream()), "sq_act_bwd_bf16")
    return dx


def bridge(a, b, kind):
    _chk(a, "a"), _chk(b, "b")
    y = torch.empty_like(a)
    lib = _lib.load()
    _lib.check(lib.sq_bridge_fwd_bf16(_ptr(a), _ptr(b), _ptr(y), a.numel(), BRIDGE[kind], _stream()), "sq_bridge_fwd_bf16")
    return y


def bridge_bwd(dy, a, b, kind):
    _chk(dy, "dy")
    da, db = torch.empty_like(dy), torch.empty_like(dy)
    lib = _lib.load()
    _lib.check(lib.sq_bridge_bwd_bf16(_ptr(dy), _ptr(a), _ptr(b), _ptr(da), _ptr(db), dy.numel(), BRIDGE[kind], _stream()),
               "sq_bridge_bwd_bf16")
    return da, db


def dropout_fwd(x, rate, seed=0, mask=None, step_dev=None):
    _chk(x, "x")
    y = torch.empty_like(x)
    given = mask is not None
    if not given:
        mask = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    lib = _lib.load()
    _lib.check(lib.sq_dropout_fwd_bf16(_ptr(x), _ptr(y), _ptr(mask), x.numel(), float(rate), int(seed) & 0xFFFFFFFF,
                                      1 if given else 0, _ptr(step_dev), _stream()), "sq_dropout_fwd_bf16")
    return y, mask


def dropout_bwd(dy, mask, rate):
    _chk(dy, "dy")
    dx = torch.empty_like(dy)
    lib = _lib.load()
    _lib.check(lib.sq_dropout_bwd_bf16(_ptr(dy), _ptr(mask), _ptr(dx), dy.numel(), float(rate), _stream()), "sq_dropout_bwd_bf16")
    return dx


def act_dropout_bwd(dy, mask, y, rate, act):
    """dropout backward + the backward of the activation whose output y entered the dropout, one pass."""
    _chk(dy, "dy"), _chk(y, "y"), _chk(mask, "mask", dtype=torch.uint8)
    dx = torch.empty_like(dy)
    lib = _lib.load()
    _lib.check(lib.sq_act_dropout_bwd_bf16(_ptr(dy), _ptr(mask), _ptr(y), _ptr(dx), dy.numel(), float(rate), ACT[act],
                                          _stream()), "sq_act_dropout_bwd_bf16")
    return dx


def convT2x2s2(x, w_bf16, bias, skip=None, bridge_kind=None):
    """x (N,H,W,Cin) bf16, w_bf16 = bf16 copy of the (2,2,Cout,Cin) kernel, bias f32."""
    _chk(x, "x", ndim=4), _chk(w_bf16, "w", ndim=4)
    N, H, W, Cin = x.shape
    Cout = w_bf16.shape[2]
    b = BRIDGE[bridge_kind]
    if b:
        _chk(skip, "skip", ndim=4)
    y = torch.empty((N, 2 * H, 2 * W, Cout), dtype=BF16, device=x.device)
    lib = _lib.load()
    _lib.check(lib.sq_convT2x2s2_nhwc_fwd_bf16(_ptr(x), _ptr(w_bf16), _ptr(bias), _ptr(skip) if b else None, _ptr(y),
                                              N, H, W, Cin, Cout, b, _stream()), "sq_convT2x2s2_nhwc_fwd_bf16")
    return y


def convT2x2s2_bridge_both(x, w_bf16, bias, skip, bridge_kind):
    """(up, merged = bridge(up, skip)) of a decoder junction from one pass; same bits as convT2x2s2 + bridge."""
    _chk(x, "x", ndim=4), _chk(w_bf16, "w", ndim=4), _chk(skip, "skip", ndim=4)
    N, H, W, Cin = x.shape
    Cout = w_bf16.shape[2]
    if tuple(skip.shape) != (N, 2 * H, 2 * W, Cout):
        raise ValueError("skip has shape %s, expected %s" % (tuple(skip.shape), (N, 2 * H, 2 * W, Cout)))
    up = torch.empty((N, 2 * H, 2 * W, Cout), dtype=BF16, device=x.device)
    merged = torch.empty_like(up)
    lib = _lib.load()
    _lib.check(lib.sq_convT2x2s2_bridge_both_fwd_bf16(_ptr(x), _ptr(w_bf16), _ptr(bias), _ptr(skip), _ptr(up), _ptr(merged),
                                                     N, H, W, Cin, Cout, BRIDGE[bridge_kind], _stream()),
               "sq_convT2x2s2_bridge_both_fwd_bf16")
    return up, merged


def bridge_bwd_s2d(dy, up, skip, kind):
    """(g (N,H,W,4C) = d_up in space-to-depth layout, dskip) of merged = bridge(up, skip); dy (N,2H,2W,C)."""
    _chk(dy, "dy", ndim=4)
    N, H2, W2, C = dy.shape
    g = torch.empty((N, H2 // 2, W2 // 2, 4 * C), dtype=BF16, device=dy.device)
    dskip = torch.empty_like(dy)
    lib = _lib.load()
    _lib.check(lib.sq_bridge_bwd_s2d_bf16(_ptr(dy), _ptr(up), _ptr(skip), _ptr(g), _ptr(dskip), N, H2 // 2, W2 // 2, C,
                                         BRIDGE[kind], _stream()), "sq_bridge_bwd_s2d_bf16")
    return g, dskip


def maxpool2x2_bwd_add(x, dy, add, gate_scale=0.0):
    """max-pool backward plus a second gradient of x (same shape as x), one pass.  gate_scale > 0: x is a
    dropout(ReLU(.)) block output and the sum leaves through that gate (x > 0 ? sum * gate_scale : 0)."""
    _chk(x, "x", ndim=4), _chk(dy, "dy", ndim=4), _chk(add, "add", ndim=4)
    N, H, W, C = x.shape
    dx = torch.empty_like(x)
    lib = _lib.load()
    _lib.check(lib.sq_maxpool2x2_bwd_add_gate_bf16(_ptr(x), _ptr(dy), _ptr(add), _ptr(dx), N, H, W, C, float(gate_scale),
                                                  _stream()), "sq_maxpool2x2_bwd_add_gate_bf16")
    return dx


def space_to_depth2(dy):
    """(N,2H,2W,C) bf16 -> (N,H,W,4C): a pure permutation, run by the f32 kernel on bf16 pairs."""
    _chk(dy, "dy", ndim=4)
    N, H2, W2, C = dy.shape
    g = torch.empty((N, H2 // 2, W2 // 2, 4 * C), dtype=BF16, device=dy.device)
    lib = _lib.load()
    _lib.check(lib.sq_space_to_depth2_f32(_ptr(dy), _ptr(g), N, H2 // 2, W2 // 2, C // 2, _stream()), "sq_space_to_depth2_f32")
    return g


def head_fwd(x, w, bias, want_mask=True):
    """bf16 (N,H,W,Cin) -> f32 logits (N,H,W,Cout), uint8 mask."""
    _chk(x, "x", ndim=4), _chk(w, "w", dtype=torch.float32, ndim=4)
    N, H, W, Cin = x.shape
    Cout = w.shape[3]
    logits = torch.empty((N, H, W, Cout), dtype=torch.float32, device=x.device)
    mask = torch.empty((N, H, W), dtype=torch.uint8, device=x.device) if want_mask else None
    lib = _lib.load()
    _lib.check(lib.sq_conv1x1_head_fwd_bf16(_ptr(x), _ptr(w), _ptr(bias), _ptr(logits), _ptr(mask), N * H * W, Cin, Cout,
                                           _stream()), "sq_conv1x1_head_fwd_bf16")
    return logits, mask


def head_bwd(x, w, dz, want_dx=True, dw_out=None, db_out=None, gate_scale=0.0):
    _chk(x, "x", ndim=4), _chk(dz, "dz", dtype=torch.float32)
    N, H, W, Cin = x.shape
    Cout = w.shape[3]
    npix = N * H * W
    lib = _lib.load()
    ws = _workspace(lib.sq_conv1x1_head_bwd_workspace_bf16(npix, Cin, Cout), x.device)
    dx = torch.empty_like(x) if want_dx else None
    dw = _grad_out(dw_out, (1, 1, Cin, Cout), x.device)
    db = _grad_out(db_out, (Cout,), x.device)
    _lib.check(lib.sq_conv1x1_head_bwd_gate_bf16(_ptr(x), _ptr(w), _ptr(dz), _ptr(dx), _ptr(dw), _ptr(db), _ptr(ws), npix,
                                                Cin, Cout, float(gate_scale), _stream()), "sq_conv1x1_head_bwd_gate_bf16")
    return dx, dw, db


def _head_wce_check(x, w, onehot, weights):
    _chk(x, "x", ndim=4), _chk(w, "w", dtype=torch.float32, ndim=4)
    _chk(onehot, "onehot", dtype=torch.uint8), _chk(weights, "weights", dtype=torch.float32)
    N, H, W, Cin = x.shape
    Cout = w.shape[3]
    if tuple(onehot.shape) != (N, H, W, Cout) or weights.numel() != N * H * W:
        raise ValueError("label / weight shapes do not match the head's output")
    return N * H * W, Cin, Cout


def head_wce_fwd(x, w, bias, onehot, weights):
    """to_image head + weighted softmax-CE, forward: 0-d f32 loss; the logits stay in registers."""
    npix, Cin, Cout = _head_wce_check(x, w, onehot, weights)
    lib = _lib.load()
    parts = torch.empty(lib.sq_wsoftmax_ce_partials(npix), dtype=torch.float64, device=x.device)
    loss = torch.empty((), dtype=torch.float32, device=x.device)
    _lib.check(lib.sq_conv1x1_head_wce_fwd_bf16(_ptr(x), _ptr(w), _ptr(bias), _ptr(onehot), _ptr(weights), _ptr(parts),
                                               _ptr(loss), npix, Cin, Cout, _stream()), "sq_conv1x1_head_wce_fwd_bf16")
    return loss


def head_wce_bwd(x, w, bias, onehot, weights, dloss, want_dx=True, dw_out=None, db_out=None, gate_scale=0.0, loss_out=None):
    """backward of head_wce_fwd from the 0-d f32 gradient arriving at the loss: (dx, dW, db).  loss_out (0-d f32): the same
    pass also writes the loss there -- the forward's value, bit for bit (a deferred loss, functional_bf16.deferred_loss)."""
    npix, Cin, Cout = _head_wce_check(x, w, onehot, weights)
    _chk(dloss, "dloss", dtype=torch.float32)
    lib = _lib.load()
    ws = _workspace(lib.sq_conv1x1_head_bwd_workspace_bf16(npix, Cin, Cout), x.device)
    dx = torch.empty_like(x) if want_dx else None
    dw = _grad_out(dw_out, (1, 1, Cin, Cout), x.device)
    db = _grad_out(db_out, (Cout,), x.device)
    if loss_out is not None:
        _chk(loss_out, "loss_out", dtype=torch.float32)
        parts = torch.empty(lib.sq_wsoftmax_ce_partials(npix), dtype=torch.float64, device=x.device)
        _lib.check(lib.sq_conv1x1_head_wce_bwd_loss_bf16(_ptr(x), _ptr(w), _ptr(bias), _ptr(onehot), _ptr(weights), _ptr(dloss),
                                                        _ptr(dx), _ptr(dw), _ptr(db), _ptr(ws), _ptr(parts), _ptr(loss_out),
                                                        npix, Cin, Cout, float(gate_scale), _stream()),
                   "sq_conv1x1_head_wce_bwd_loss_bf16")
        return dx, dw, db
    _lib.check(lib.sq_conv1x1_head_wce_bwd_bf16(_ptr(x), _ptr(w), _ptr(bias), _ptr(onehot), _ptr(weights), _ptr(dloss),
                                               _ptr(dx), _ptr(dw), _ptr(db), _ptr(ws), npix, Cin, Cout, float(gate_scale),
                                               _stream()), "sq_conv1x1_head_wce_bwd_bf16")
    return dx, dw, db


def conv3x3_first_wgrad(x, dy, dw_out=None, db_out=None):
    """x f32 (N,H,W,Cin), dy bf16 (N,H,W,Cout) -> (dW (3,3,Cin,Cout) f32, db f32)."""
    _chk(x, "x", dtype=torch.float32, ndim=4), _chk(dy, "dy", ndim=4)
    N, H, W, Cin = x.shape
    Cout = dy.shape[3]
    lib = _lib.load()
    ws = _workspace(lib.sq_conv3x3_first_wgrad_workspace_bf16(N, H, W, Cin, Cout), x.device)
    dw = _grad_out(dw_out, (3, 3, Cin, Cout), x.device)
    db = _grad_out(db_out, (Cout,), x.device)
    _lib.check(lib.sq_conv3x3_first_wgrad_bf16(_ptr(x), _ptr(dy), _ptr(dw), _ptr(db), _ptr(ws), N, H, W, Cin, Cout,
                                               _stream()), "sq_conv3x3_first_wgrad_bf16")
    return dw, db


# ---- batch normalisation on bf16 activations (optional `batch_norm` of conv_layer) ----------------------
def _bn_shape(x):
    _chk(x, "x")
    C = x.shape[-1]
    return x.numel() // C, C


def bn_stats(x):
    """(mean, population variance) per channel of the bf16 NHWC tensor x: f32 results, f64 accumulation."""
    npix, C = _bn_shape(x)
    lib = _lib.load()
    nbytes = lib.sq_bn_workspace_f32(npix, C)
    if nbytes < 0:
        raise _lib.SequitrHipError("bn_stats(bf16): unsupported channel count %d" % C)
    ws = _workspace(nbytes, x.device)
    mean = torch.empty((C,), dtype=torch.float32, device=x.device)
    var = torch.empty((C,), dtype=torch.float32, device=x.device)
    _lib.check(lib.sq_bn_stats_bf16(_ptr(x), _ptr(mean), _ptr(var), _ptr(ws), npix, C, _stream()), "sq_bn_stats_bf16")
    return mean, var


def bn_apply(x, scale, shift, act=None):
    """y = act(fmaf(x, scale[c], shift[c])) rounded to bf16 (scale / shift f32: ops.bn_fold)."""
    npix, C = _bn_shape(x)
    _chk(scale, "scale", dtype=torch.float32), _chk(shift, "shift", dtype=torch.float32)
    y = torch.empty_like(x)
    _lib.check(_lib.load().sq_bn_apply_bf16(_ptr(x), _ptr(scale), _ptr(shift), _ptr(y), npix, C, ACT[act], _stream()),
               "sq_bn_apply_bf16")
    return y


def bn_inference(x, gamma, beta, moving_mean, moving_var, eps, act=None):
    from . import ops
    scale, shift = ops.bn_fold(gamma, beta, moving_mean, moving_var, eps)
    return bn_apply(x, scale, shift, act)


def bn_bwd(x, dy, y, act, mean, var, gamma, eps):
    """(dx bf16, dgamma f32, dbeta f32) of y = act(BN_batchstats(x)); y is needed only when act is not None."""
    npix, C = _bn_shape(x)
    _chk(dy, "dy")
    lib = _lib.load()
    ws = _workspace(lib.sq_bn_workspace_f32(npix, C), x.device)
    dx = torch.empty_like(x)
    dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(gamma)
    _lib.check(lib.sq_bn_bwd_bf16(_ptr(x), _ptr(dy), _ptr(y) if ACT[act] else None, ACT[act], _ptr(mean), _ptr(var),
                                 _ptr(gamma), float(eps), _ptr(dx), _ptr(dgamma), _ptr(dbeta), _ptr(ws), npix, C,
                                 _stream()), "sq_bn_bwd_bf16")
    return dx, dgamma, dbeta
