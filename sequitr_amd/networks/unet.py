"""U-Net on the MI355X HIP back end, behind the reference's operator API.

Mirrors sequitr/networks/unet.py: ``UNet(params, mode)`` with the same ``params``
keys and attributes (unet.py:126-216), ``build(features) -> logits`` with the same
wiring (unet.py:224-262), ``conv_block`` / ``down_layer`` / ``up_layer``
(unet.py:265-322) and the overridable leaf hooks ``conv_layer``,
``conv_layer_1x1``, ``conv_transpose_layer``, ``pool_layer`` (unet.py:326-343) --
plus ``max_pool_layer``, the name ``build`` actually calls (unet.py:242).

The reference's base class is abstract and its ``UNet2D`` subclass is not in the
tree; ``UNet2D`` here supplies the leaf ops with the documented
defaults of SURVEY.md A.1, every one a hand-written HIP kernel reached through
the C-ABI (include/sequitr_hip.h).  TF variable scopes are mirrored by a scope
stack, so ``state_dict()`` keys equal the reference's variable names
(``UNet/down0/conv1/kernel`` ...).
"""
import contextlib
import logging

import numpy as np
import torch

from .. import functional as F
from .. import functional_bf16 as FB
from .. import ops

DEFAULT_FILTERS = (16, 32, 64, 128, 256)                       # unet.py:40
DEFAULT_DROPOUT = 0.4                                          # unet.py:41
BRIDGE_TYPES = ('eltwise_add', 'eltwise_mul', 'eltwise_sub', 'concat', None)   # unet.py:42

TRAIN, EVAL, PREDICT = 'train', 'eval', 'infer'                # tf.estimator.ModeKeys values

logger = logging.getLogger('worker_process')                   # unet.py:46


def variance_scaling(shape, rng, scale=1.0):
    """TF-1.x ``tf.initializers.variance_scaling()`` defaults (unet.py:143):
    mode fan_in, truncated normal at +-2 sigma, stddev = sqrt(scale/fan_in)/0.8796...
    fan_in = shape[-2] * prod(shape[:-2]) as TF's _compute_fans does (for a
    conv2d_transpose kernel (kh,kw,Cout,Cin) that is Cout*kh*kw)."""
    fan_in = float(shape[-2] * int(np.prod(shape[:-2])))
    std = np.sqrt(scale / max(1.0, fan_in)) / .87962566103423978
    out = rng.standard_normal(size=shape)
    bad = np.abs(out) > 2.0
    while bad.any():                                           # TF re-draws outliers
        out[bad] = rng.standard_normal(size=int(bad.sum()))
        bad = np.abs(out) > 2.0
    return (out * std).astype(np.float32)


def unet_variable_shapes(params):
    """(key, shape) of every U-Net variable in creation (= build) order; keys are the
    reference's variable-scope names (unet.py:234,252,268-271,294,312-318)."""
    f = list(params.get('filters', DEFAULT_FILTERS))
    k = tuple(params.get('kernel', (3, 3)))
    cin = params.get('num_inputs', 1)
    nout = params.get('num_outputs', 2)
    concat = params.get('bridge', 'eltwise_mul') == 'concat'
    uk = tuple(params.get('up_kernel', (2, 2)))
    bn = bool(params.get('batch_norm', False))                # trainable BN variables follow their conv's
    out = []

    def conv_vars(s, ci, fo):
        v = [(s + '/kernel', k + (ci, fo)), (s + '/bias', (fo,))]
        return v + ([(s + '/gamma', (fo,)), (s + '/beta', (fo,))] if bn else [])

    for i, fo in enumerate(f):
        for j, ci in enumerate((cin, fo)):
            out += conv_vars('UNet/down%d/conv%d' % (i, j + 1), ci, fo)
        cin = fo
    for i in reversed(range(len(f) - 1)):
        s = 'UNet/up%d' % i
        out += [(s + '/upscale/kernel', uk + (f[i], f[i + 1])), (s + '/upscale/bias', (f[i],))]
        for j, ci in enumerate((2 * f[i] if concat else f[i], f[i])):
            out += conv_vars(s + '/conv%d' % (j + 1), ci, f[i])
    out += [('UNet/to_image/kernel', (1, 1, f[0], nout)), ('UNet/to_image/bias', (nout,))]
    return out


def init_unet_weights(params, seed=0):
    """Host-side initial weights {key: float32 ndarray}: variance_scaling kernels, zero
    biases.  Needs no GPU; UNet2D.initialize() uploads exactly these."""
    rng = np.random.default_rng(seed)
    w = {}
    for key, shape in unet_variable_shapes(params):
        if key.endswith('kernel'):
            w[key] = variance_scaling(shape, rng)
        else:
            w[key] = (np.ones if key.endswith('gamma') else np.zeros)(shape, np.float32)
    return w


class UNet(object):
    """Base class: wiring only, leaf hooks abstract (as sequitr/networks/unet.py:53-343)."""

    def __init__(self, params, mode):
        self._mode = mode
        self.name = params.get('name', 'UNet2d_test')          # unet.py:132-139
        self.filters = params.get('filters', DEFAULT_FILTERS)
        self.dropout = params.get('dropout', DEFAULT_DROPOUT)
        self.n_inputs = params.get('num_inputs', 1)
        self.n_outputs = params.get('num_outputs', 2)
        self.shape = params.get('shape', (1024, 1024))
        self.bridge_type = params.get('bridge', 'eltwise_mul')
        self.kernel = params.get('kernel', (3, 3))
        self._net = None
        self._scopes = []

    # -- geometry properties, unet.py:148-167 --------------------------------------
    @property
    def width(self):
        return self.shape[0]

    @property
    def height(self):
        return self.shape[1]

    @property
    def slices(self):
        if self.ndim < 3:
            return 0
        return self.shape[2]

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def training(self):                                        # unet.py:170-172
        return self._mode == TRAIN

    @property
    def btype(self):
        raise DeprecationWarning("Use @bridge_type")

    @property
    def bridge_type(self):
        return self._bridge_type

    @bridge_type.setter
    def bridge_type(self, bridge):                             # unet.py:182-202
        if bridge not in BRIDGE_TYPES:
            raise ValueError('Bridge type not recognized')
        if bridge in ('eltwise_add', 'eltwise_mul', 'eltwise_sub'):
            self.bridge = lambda x, y, _k=bridge: (F.bridge(x, y, _k) if (x.requires_grad or y.requires_grad)
                                                   else ops.bridge(x, y, _k))
        elif bridge == 'concat':
            self.bridge = lambda x, y: torch.cat([x, y], -1)   # upscale first (unet.py:197)
        else:
            logger.warning('Bridge function in UNet not recognized')
            self.bridge = lambda x, y: x
        self._default_bridge = self.bridge
        self._bridge_type = bridge

    # -- scope stack standing in for tf.variable_scope ------------------------------
    @contextlib.contextmanager
    def variable_scope(self, name):
        self._scopes.append(name)
        try:
            yield
        finally:
            self._scopes.pop()

    @property
    def scope(self):
        return '/'.join(self._scopes)

    def reshape_input(self, features):                         # unet.py:205-216
        full_shape = [-1, self.slices, self.width, self.height, self.n_inputs]
        input_shape = [d for d in full_shape if d != 0]
        return features.reshape(input_shape)

    def logits(self):                                          # unet.py:220-222
        return self._net[-1]

    def build(self, features):
        """unet.py:224-262: returns the un-normalised logits (N,W,H,num_outputs)."""
        logger.info('Building UNet ({0:s})...'.format(self.__class__.__name__))
        with self.variable_scope('UNet'):
            input_layer = self.reshape_input(features)
            self._net = [self.down_layer(input_layer, self.filters[0], name=0)]
            for i, f in enumerate(self.filters[1:]):
                prev_layer = self.max_pool_layer(self._net[-1])
                self._net.append(self.down_layer(prev_layer, f, name=i + 1))
            for i, f in reversed(list(enumerate(self.filters[:-1]))):
                prev_layer = self._net[-1]
                bridge = self._net[i]
                self._net.append(self.up_layer(prev_layer, f, bridge, name=i))
            with self.variable_scope('to_image'):
                logits = self.conv_layer_1x1(self._net[-1], self.n_outputs)
        logger.info('Output layer -> shape {0:s}'.format(str(tuple(logits.shape))))
        self._net.append(logits)
        logger.info('...Done')
        return logits

    def conv_block(self, x, filters):                          # unet.py:265-277
        with self.variable_scope('conv1'):
            conv1 = self.conv_layer(x, filters)
        with self.variable_scope('conv2'):
            conv2 = self.conv_layer(conv1, filters)
        return self.dropout_layer(conv2)

    def dropout_layer(self, x):
        """tf.layers.dropout(rate=self.dropout, training=self.training), unet.py:274-276."""
        return x

    def down_layer(self, x, filters, name=None):               # unet.py:282-296
        logger.info('Down layer -> shape {0:s}'.format(str(tuple(x.shape))))
        with self.variable_scope('down{0:d}'.format(name)):
            out = self.conv_block(x, filters)
        return out

    def up_layer(self, x, filters, bridge, name=None):         # unet.py:299-322
        logger.info('Up layer -> shape {0:s} (bridge: {1:s})'.format(str(tuple(x.shape)),
                                                                      str(self.bridge_type)))
        with self.variable_scope('up{0:d}'.format(name)):
            with self.variable_scope('upscale'):
                upscale = self.conv_transpose_layer(x, filters)
            with self.variable_scope('bridge'):
                bridge = self.bridge(upscale, bridge)
            out = self.conv_block(bridge, filters)
        return out

    # -- leaf hooks, unet.py:326-343 -------------------------------------------------
    def conv_layer(self, x, filters):
        """ Convolution layer, conv-relu with padding """
        raise NotImplementedError

    def conv_layer_1x1(self, x, filters):
        """ Return a 1x1 convolution layer """
        raise NotImplementedError

    def conv_transpose_layer(self, x, filters):
        """ Transpose convolution (aka deconvolution) layer """
        raise NotImplementedError

    def pool_layer(self, x):
        """ Max pool operation """
        raise NotImplementedError

    def max_pool_layer(self, x):
        """ the name build() calls (unet.py:242, UNet_LEGACY unet.py:727) """
        return self.pool_layer(x)


class UNet2D(UNet):
    """2-D U-Net whose leaf ops are the HIP kernels of libsequitr_hip.so.

    Defaults: conv_layer = 3x3 SAME conv + bias + ReLU;
    pool = 2x2/s2 max; conv_transpose_layer = 2x2/s2 transpose conv + bias, no
    activation; conv_layer_1x1 = 1x1 conv + bias, no activation.  Extra params
    keys: ``device`` (torch device, default cuda:current), ``seed`` (weight init).
    """

    def __init__(self, params, mode=PREDICT):
        UNet.__init__(self, params, mode)
        if mode not in (TRAIN, EVAL, PREDICT):
            raise ValueError("mode must be 'train', 'eval' or 'infer'")
        dev = params.get('device', None)
        self.device = torch.device(dev) if dev is not None else torch.device('cuda', torch.cuda.current_device())
        if self.device.type != 'cuda':
            raise RuntimeError('UNet2D runs on the HIP back end only (device=%s)' % self.device)
        if tuple(self.kernel) not in ((3, 3), (1, 1)):
            raise ValueError('kernel %s unsupported: the HIP conv kernels are 3x3 or 1x1' % (self.kernel,))
        self._params = dict(params)
        self._seed = params.get('seed', 0)
        self._rng = np.random.default_rng(self._seed)
        self._vars = {}                                        # scope/name -> device tensor
        self._loaded, self._creatable = False, set()           # set by a strict load_state_dict
        self._dropout_calls = 0                               # dropout layer index within one build()
        self._builds = 0                                      # host-side step salt when no device counter is set
        self.dropout_masks = None
        self.fuse = bool(params.get('fuse', True))          # fused inference kernels (same bits)
        self.fuse_up = bool(params.get('fuse_up', True))    # ... incl. convT+bridge inside up0's first conv
        # optional BN between conv and ReLU (tf.layers.batch_normalization defaults)
        self.up_kernel = tuple(params.get('up_kernel', (2, 2)))   # transpose-conv kernel: (2,2) default or (3,3)
        if self.up_kernel not in ((2, 2), (3, 3)):
            raise ValueError('up_kernel %s unsupported: (2,2) or (3,3)' % (self.up_kernel,))
        self.batch_norm = bool(params.get('batch_norm', False))
        self.bn_eps = float(params.get('bn_epsilon', ops.BN_EPS))
        self.bn_momentum = float(params.get('bn_momentum', ops.BN_MOMENTUM))
        self._mask = None

    # -- variables ---------------------------------------------------------------------
    def get_variable(self, name, shape, init):
        key = self.scope + '/' + name
        v = self._vars.get(key)
        if v is None:
            if self._loaded and key not in self._creatable:
                # a model was loaded: build() must find every variable in it, never fill gaps with fresh random draws
                raise KeyError('variable %s is not in the loaded model (load_state_dict(strict=False) to allow '
                               'missing variables to be initialised)' % key)
            v = torch.from_numpy(init(shape)).to(self.device)
            self._vars[key] = v
        elif tuple(v.shape) != tuple(shape):
            raise ValueError('variable %s has shape %s, wanted %s' % (key, tuple(v.shape), tuple(shape)))
        return v

    def _kernel(self, shape):
        return self.get_variable('kernel', shape, lambda s: variance_scaling(s, self._rng))

    def _bias(self, n):
        return self.get_variable('bias', (n,), lambda s: np.zeros(s, np.float32))

    def state_dict(self):
        return {k: v.detach().cpu().numpy() for k, v in self._vars.items()}

    def expected_variables(self):
        """({key: shape} build() will ask for, {key: shape} of optional non-trainable state): the trainable set is
        unet_variable_shapes(params); with batch_norm the moving statistics of every conv are optional on load
        (absent -> tf.layers' initial values 0 / 1)."""
        req = dict(unet_variable_shapes(self._params))
        opt = {}
        if self.batch_norm:
            for k, shp in list(req.items()):
                if k.endswith('/gamma'):
                    sc = k.rsplit('/', 1)[0]
                    opt[sc + '/moving_mean'] = opt[sc + '/moving_variance'] = shp
        return req, opt

    def load_state_dict(self, weights, strict=True):
        """Copy a model into the net.  strict (default): the keys must be exactly the variables this configuration
        builds (plus optional BN moving statistics) with the right shapes -- a UNet_LEGACY checkpoint, another depth or
        filter schedule, or a BN checkpoint in a non-BN net raise here instead of segmenting with partly random weights;
        afterwards build() refuses to create variables the model did not hold.  strict=False: deliberate partial load,
        missing variables are initialised by build() as in a fresh net."""
        req, opt = self.expected_variables()
        if strict:
            missing = [k for k in req if k not in weights]
            unexpected = [k for k in weights if k not in req and k not in opt]
            bad = [(k, tuple(np.shape(weights[k])), tuple(s)) for k, s in list(req.items()) + list(opt.items())
                   if k in weights and tuple(np.shape(weights[k])) != tuple(s)]
            if missing or unexpected or bad:
                raise ValueError('load_state_dict: model does not fit this %s configuration -- missing %s; unexpected %s; '
                                 'shape mismatches (key, got, wanted) %s' % (self.__class__.__name__, missing[:6],
                                                                             unexpected[:6], bad[:6]))
            self._loaded, self._creatable = True, set(opt)
        for k, v in weights.items():
            self._vars[k] = torch.as_tensor(np.ascontiguousarray(v, dtype=np.float32)).to(self.device)

    def initialize(self):
        """Create every variable without running a tile (same draws as a first build)."""
        self.load_state_dict(self._initial_weights(), strict=False)      # a fresh net: hooks may still add variables
        return self

    def _initial_weights(self):
        return init_unet_weights(self._params, self._seed)

    # -- input ---------------------------------------------------------------------------
    def reshape_input(self, features):
        if isinstance(features, np.ndarray):
            features = torch.from_numpy(np.require(features, np.float32, ['C', 'W']))
        if not isinstance(features, torch.Tensor):
            raise TypeError('features must be a numpy array or a torch tensor')
        features = features.to(self.device, dtype=torch.float32, non_blocking=True)
        return UNet.reshape_input(self, features).contiguous()

    # -- leaf hooks ------------------------------------------------------------------------
    # inference: raw fused kernels; training (mode == 'train'): the differentiable wrappers of
    # sequitr_amd.functional (same forward kernels + hand-written gradient kernels)
    def conv_layer(self, x, filters):
        k = tuple(self.kernel)
        w, b = self._kernel(k + (x.shape[-1], filters)), self._bias(filters)
        if self.batch_norm:
            z = F.conv2d(x, w, b, act=None) if self.training else ops.conv2d(x, w, b, act=None)
            return self.batch_norm_layer(z, act='relu')
        if self.training:
            return F.conv2d(x, w, b, act='relu')
        return ops.conv2d(x, w, b, act='relu')

    def batch_norm_layer(self, z, act=None):
        """tf.layers.batch_normalization(training=self.training) + activation: batch statistics (and a
        moving-average update) when training, the moving statistics otherwise.  Variables gamma, beta,
        moving_mean, moving_variance live in the conv's scope."""
        n = z.shape[-1]
        gamma = self.get_variable('gamma', (n,), lambda s: np.ones(s, np.float32))
        beta = self.get_variable('beta', (n,), lambda s: np.zeros(s, np.float32))
        mmean = self.get_variable('moving_mean', (n,), lambda s: np.zeros(s, np.float32))
        mvar = self.get_variable('moving_variance', (n,), lambda s: np.ones(s, np.float32))
        if self.training:
            return F.batch_norm_train(z, gamma, beta, mmean, mvar, self.bn_eps, self.bn_momentum, act=act)
        return ops.bn_inference(z, gamma, beta, mmean, mvar, self.bn_eps, act=act)

    def conv_layer_1x1(self, x, filters):
        w, b = self._kernel((1, 1, x.shape[-1], filters)), self._bias(filters)
        small = filters <= 7 and x.shape[-1] % 4 == 0
        if self.training:
            if small and filters <= 4 and x.shape[-1] in (8, 16, 32):
                return F.conv1x1_head(x, w, b)
            return F.conv2d(x, w, b, act=None)
        if small:
            logits, self._mask = ops.conv1x1_argmax(x, w, b)   # logits + prediction in one pass
            return logits
        return ops.conv2d(x, w, b, act=None)

    def conv_transpose_layer(self, x, filters):
        if self.up_kernel == (3, 3):                           # SURVEY A.1 alternative: k=3, s=2, SAME
            w, b = self._kernel((3, 3, filters, x.shape[-1])), self._bias(filters)
            return F.convT3x3s2(x, w, b) if self.training else ops.convT3x3s2(x, w, b)
        w, b = self._kernel((2, 2, filters, x.shape[-1])), self._bias(filters)
        if self.training:
            return F.convT2x2s2(x, w, b)
        return ops.convT2x2s2(x, w, b)

    def pool_layer(self, x):
        return F.maxpool2x2(x) if self.training else ops.maxpool2x2(x)

    def dropout_layer(self, x):
        """tf.layers.dropout(rate, training) after conv2 of every block (unet.py:274-276):
        identity at inference; Bernoulli(1-rate) mask / (1-rate) in training.  ``dropout_masks``
        (list of uint8 tensors, consumed in call order) pins the masks for parity tests."""
        if not self.training or self.dropout <= 0.0:
            return x
        mask = self.dropout_masks.pop(0) if getattr(self, 'dropout_masks', None) else None
        step_dev = None if mask is not None else getattr(self, '_step_dev', None)
        return F.dropout(x, self.dropout, seed=self._dropout_seed(step_dev), mask=mask, step_dev=step_dev)

    def up_layer(self, x, filters, bridge, name=None):
        """Same wiring as the base class, but when neither conv_transpose_layer nor the
        bridge has been overridden the two run as ONE kernel at inference (convT epilogue
        applies the bridge), saving a write + two reads of the up-scaled tensor."""
        fused = (not self.training and self.up_kernel == (2, 2)
                 and type(self).conv_transpose_layer is UNet2D.conv_transpose_layer
                 and self.bridge is self._default_bridge
                 and self.bridge_type in ('eltwise_add', 'eltwise_mul', 'eltwise_sub'))
        if (not self.training and self.bridge_type == 'concat' and self.bridge is self._default_bridge
                and type(self).conv_layer is UNet2D.conv_layer and type(self).conv_block is UNet2D.conv_block
                and not self.batch_norm and filters % 16 == 0 and tuple(self.kernel) == (3, 3)):
            # concat bridge (unet.py:196-197) without the concatenated tensor: conv1 of the block takes its first
            # `filters` input channels from the up-scaled tensor and the rest from the skip tensor (sq_conv2d_concat_*)
            with self.variable_scope('up{0:d}'.format(name)):
                with self.variable_scope('upscale'):
                    upscale = self.conv_transpose_layer(x, filters)
                with self.variable_scope('conv1'):
                    w1, b1 = self._kernel((3, 3, 2 * filters, filters)), self._bias(filters)
                    conv1 = ops.conv2d_concat(upscale, bridge, w1, b1, act='relu')
                with self.variable_scope('conv2'):
                    conv2 = self.conv_layer(conv1, filters)
                out = self.dropout_layer(conv2)
            return out
        if not fused:
            return UNet.up_layer(self, x, filters, bridge, name=name)
        with self.variable_scope('up{0:d}'.format(name)):
            with self.variable_scope('upscale'):
                w, b = self._kernel((2, 2, filters, x.shape[-1])), self._bias(filters)
            merged = ops.convT2x2s2(x, w, b, skip=bridge, bridge=self.bridge_type)
            out = self.conv_block(merged, filters)
        return out

    def _dropout_seed(self, step_dev):
        """Seed of this dropout layer: (net seed, layer index) - and, when no device step counter salts
        it inside the kernel, the host build count folded in the same way (seed + step * 0x9E3779B9)."""
        self._dropout_calls += 1
        s = self._seed * 1000003 + self._dropout_calls
        if step_dev is None:
            s += (self._builds - 1) * 0x9E3779B9
        return s & 0xFFFFFFFF

    def build(self, features):
        self._mask = None
        self._dropout_calls = 0
        self._builds += 1
        if self.fuse and self._fusable() and self._fits_fused(features):
            return self._build_fused(features)
        return UNet.build(self, features)

    def _fits_fused(self, features):
        """the fused kernels address with 32-bit buffer offsets: the largest activation must stay below 2 GiB
        (bigger batches run hook by hook, where conv2d switches to its 64-bit kernel)"""
        n = int(np.prod(features.shape)) // max(1, self.width * self.height * self.n_inputs)
        return n * self.width * self.height * max(self.filters[0], 2 * self.n_outputs) * 4 < (1 << 31)

    # -- fused inference graph ---------------------------------------------------------------
    _HOOKS = ('conv_layer', 'conv_layer_1x1', 'conv_transpose_layer', 'pool_layer', 'max_pool_layer',
              'conv_block', 'down_layer', 'up_layer', 'dropout_layer', 'reshape_input')

    def _fusable(self):
        """The fused kernels replace whole hook sequences, so they are used only when no hook (and
        not the bridge) has been overridden and the graph is the plain inference graph."""
        if self.training or tuple(self.kernel) != (3, 3) or self.batch_norm or self.up_kernel != (2, 2):
            return False
        if self.bridge is not self._default_bridge or self.bridge_type not in ('eltwise_add', 'eltwise_mul', 'eltwise_sub'):
            return False
        if any(f % 16 for f in self.filters) or len(self.filters) < 2:
            return False
        return all(getattr(type(self), h) is getattr(UNet2D, h) for h in self._HOOKS)

    def _v(self, path, kind, shape):
        scopes = path.split('/')
        self._scopes.extend(scopes)
        try:
            return self._kernel(shape) if kind == 'kernel' else self._bias(shape[0])
        finally:
            del self._scopes[-len(scopes):]

    def _build_fused(self, features):
        """Same graph and the same bits as UNet.build (unet.py:224-262), with the fused kernels:
        down0 conv1+conv2(+pool) when the input has one channel, conv2+max-pool in every encoder
        block, transpose-conv+bridge in every decoder block, last conv + 1x1 head + argmax.
        Variables are created in the same order as the unfused build."""
        f = list(self.filters)
        L = len(f)
        x = self.reshape_input(features)
        net, pooled = [], None
        for i in range(L):                                         # encoder
            s = 'UNet/down%d' % i
            cin = x.shape[-1] if i == 0 else f[i - 1]
            w1, b1 = self._v(s + '/conv1', 'kernel', (3, 3, cin, f[i])), self._v(s + '/conv1', 'bias', (f[i],))
            w2, b2 = self._v(s + '/conv2', 'kernel', (3, 3, f[i], f[i])), self._v(s + '/conv2', 'bias', (f[i],))
            src = x if i == 0 else pooled
            last = i == L - 1
            if i == 0 and cin == 1 and f[0] == 16:
                y, pooled = ops.conv3x3_first_block(src, w1, b1, w2, b2, want_pool=not last)
            else:
                c1 = ops.conv2d(src, w1, b1, act='relu')
                if last:
                    y, pooled = ops.conv2d(c1, w2, b2, act='relu'), None
                else:
                    y, pooled = ops.conv3x3_pool(c1, w2, b2, act='relu')
            net.append(y)
        head_fused = f[0] == 16 and self.n_outputs <= 4
        logits = None
        for i in reversed(range(L - 1)):                           # decoder
            s = 'UNet/up%d' % i
            wt, bt = self._v(s + '/upscale', 'kernel', (2, 2, f[i], f[i + 1])), self._v(s + '/upscale', 'bias', (f[i],))
            w1, b1 = self._v(s + '/conv1', 'kernel', (3, 3, f[i], f[i])), self._v(s + '/conv1', 'bias', (f[i],))
            w2, b2 = self._v(s + '/conv2', 'kernel', (3, 3, f[i], f[i])), self._v(s + '/conv2', 'bias', (f[i],))
            if f[i] == 16 and f[i + 1] == 32 and self.fuse_up:
                # level 0: transpose conv + bridge are computed in the staging of conv1 (never reach HBM)
                c1 = ops.convT_conv3x3(net[-1], wt, bt, net[i], self.bridge_type, w1, b1, act='relu')
            else:
                merged = ops.convT2x2s2(net[-1], wt, bt, skip=net[i], bridge=self.bridge_type)
                c1 = ops.conv2d(merged, w1, b1, act='relu')
            if i == 0 and head_fused:
                wh = self._v('UNet/to_image', 'kernel', (1, 1, f[0], self.n_outputs))
                bh = self._v('UNet/to_image', 'bias', (self.n_outputs,))
                logits, self._mask = ops.conv3x3_head(c1, w2, b2, wh, bh, act='relu')
                net.append(None)                                   # up0's activation is never materialised
            else:
                net.append(ops.conv2d(c1, w2, b2, act='relu'))
        if logits is None:
            wh = self._v('UNet/to_image', 'kernel', (1, 1, f[0], self.n_outputs))
            bh = self._v('UNet/to_image', 'bias', (self.n_outputs,))
            if self.n_outputs <= 7:
                logits, self._mask = ops.conv1x1_argmax(net[-1], wh, bh)
            else:
                logits = ops.conv2d(net[-1], wh, bh, act=None)
        net.append(logits)
        self._net = net
        return logits

    # -- prediction ------------------------------------------------------------------------
    def predict(self, features):
        """build + argmax: returns the uint8 class mask (N,W,H), ties -> lowest class."""
        logits = self.build(features)
        if self._mask is None or self._mask.shape != logits.shape[:-1]:
            self._mask = ops.argmax_u8(logits)
        return self._mask

    def predict_stream(self, tiles, batch=32, want_logits=False, pipe=None, on_batch=None):
        """predict() over a whole stack of host tiles with upload, network and download of consecutive batches
        overlapped (frontend.TileStreamer): returns (masks, logits-or-None) as host arrays, the same bits as
        predict() batch by batch."""
        from ..frontend import segment_tiles
        return segment_tiles(self, tiles, batch=batch, want_logits=want_logits, pipe=pipe, on_batch=on_batch)


class UNet_LEGACY(UNet2D):
    """The reference's older wiring (unet.py:445-729): identical arithmetic, but no variable scopes -- the layer
    names ``L{i}d`` / ``L{i}u`` are passed to down_layer / up_layer and never used (unet.py:626-639, 655, 678), so
    under TF-1.x the variables carry ``tf.layers``' automatic names: ``conv2d``, ``conv2d_1`` ... in creation order
    (the 1x1 head is the last ``conv2d_<n>``) and ``conv2d_transpose``, ``conv2d_transpose_1`` ...  ``state_dict()``
    has exactly those keys, so a legacy checkpoint exported to ``.npz`` loads by name.  The hook it calls for pooling
    is ``max_pool_layer`` (unet.py:727).  Runs the per-layer path (the fused kernels key on the scoped names)."""

    def __init__(self, params, mode=PREDICT):
        UNet2D.__init__(self, dict(params, fuse=False), mode)
        self._auto = {}

    def expected_variables(self):
        req, opt = UNet2D.expected_variables(self)
        _, back = legacy_state_dict({k: None for k in req}, self._params)
        fwd = {v: k for k, v in back.items()}                  # scoped name -> tf.layers automatic name

        def rename(d):
            return {fwd[k.rsplit('/', 1)[0]] + '/' + k.rsplit('/', 1)[1]: s for k, s in d.items()}
        return rename(req), rename(opt)

    def _initial_weights(self):                                # the same draws, under the automatic names
        return legacy_state_dict(init_unet_weights(self._params, self._seed), self._params)[0]

    def _auto_scope(self, kind):
        n = self._auto.get(kind, 0)
        self._auto[kind] = n + 1
        return self.variable_scope(kind if n == 0 else '{0:s}_{1:d}'.format(kind, n))

    def build(self, features):                                 # unet.py:613-647
        logger.info('Building UNet ({0:s})...'.format(self.__class__.__name__))
        self._auto = {}
        input_layer = self.reshape_input(features)
        self._net = [self.down_layer(input_layer, self.filters[0], name='L0d')]
        for i, f in enumerate(self.filters[1:]):
            prev_layer = self.max_pool_layer(self._net[-1])
            self._net.append(self.down_layer(prev_layer, f, name='L{0:d}d'.format(i + 1)))
        for i, f in reversed(list(enumerate(self.filters[:-1]))):
            self._net.append(self.up_layer(self._net[-1], f, self._net[i], name='L{0:d}u'.format(i)))
        logits = self.conv_layer_1x1(self._net[-1], self.n_outputs)
        self._net.append(logits)
        logger.info('Output layer -> shape {0:s}'.format(str(tuple(logits.shape))))
        return logits

    def down_layer(self, x, filters, name=None):               # unet.py:655-675: the name is not used
        return self.dropout_layer(self.conv_layer(self.conv_layer(x, filters), filters))

    def up_layer(self, x, filters, bridge, name=None):         # unet.py:678-705
        merged = self.bridge(self.conv_transpose_layer(x, filters), bridge)
        return self.dropout_layer(self.conv_layer(self.conv_layer(merged, filters), filters))

    def conv_layer(self, x, filters):
        with self._auto_scope('conv2d'):
            return UNet2D.conv_layer(self, x, filters)

    def conv_layer_1x1(self, x, filters):
        with self._auto_scope('conv2d'):
            return UNet2D.conv_layer_1x1(self, x, filters)

    def conv_transpose_layer(self, x, filters):
        with self._auto_scope('conv2d_transpose'):
            return UNet2D.conv_transpose_layer(self, x, filters)


def legacy_state_dict(weights, params):
    """Scoped U-Net weights (``UNet/down0/conv1/kernel`` ...) under UNet_LEGACY's automatic names, in creation
    order -- and the inverse mapping, for moving checkpoints between the two wirings."""
    out, back, nc, nt = {}, {}, 0, 0
    for key, _shape in unet_variable_shapes(params):
        scope, var = key.rsplit('/', 1)
        if scope not in back:
            if scope.endswith('/upscale'):
                back[scope] = 'conv2d_transpose' if nt == 0 else 'conv2d_transpose_%d' % nt
                nt += 1
            else:
                back[scope] = 'conv2d' if nc == 0 else 'conv2d_%d' % nc
                nc += 1
        out[back[scope] + '/' + var] = weights[key]
    return out, {v: k for k, v in back.items()}


class GraphedPredict(object):
    """UNet2D.predict captured as one hipGraph for a fixed batch shape (inference is GPU-bound, but one graph
    launch per batch instead of ~25 kernel launches makes the host side immune to scheduling jitter: a
    loaded host was observed to starve the queue).  Same kernels, same bits as the eager path."""

    def __init__(self, net, batch_shape, warmup=2):
        if net.training:
            raise RuntimeError('GraphedPredict captures the inference graph (mode infer / eval)')
        self.net = net
        self.x = torch.zeros(tuple(batch_shape), dtype=torch.float32, device=net.device)
        side = torch.cuda.Stream(device=net.device)
        side.wait_stream(torch.cuda.current_stream(net.device))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                net.predict(self.x)
        torch.cuda.current_stream(net.device).wait_stream(side)
        torch.cuda.synchronize(net.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.mask = net.predict(self.x)
            self.logits = net.logits()

    def __call__(self, features):
        """features: array / tensor of the captured shape; returns (mask, logits) static device tensors
        (overwritten by the next call)."""
        if isinstance(features, np.ndarray):
            features = torch.from_numpy(np.require(features, np.float32, ['C', 'W']))
        if tuple(features.shape) != tuple(self.x.shape):
            raise ValueError('captured for %s, got %s' % (tuple(self.x.shape), tuple(features.shape)))
        self.x.copy_(features, non_blocking=True)
        self.graph.replay()
        return self.mask, self.logits


class UNet2DBf16(UNet2D):
    """The same graph with bf16 activations in HBM (BASELINE configs 3-5: bf16 compute, fp32 master
    weights and fp32 accumulation).  The image enters as f32 (one channel), every activation between
    the layers is bfloat16, the logits leave as f32.  Leaf ops: sequitr_amd.functional_bf16 /
    ops_bf16 (bf16 MFMA convolutions, transposing-LDS-read weight gradients)."""

    def __init__(self, params, mode=PREDICT):
        UNet2D.__init__(self, dict(params, fuse=False), mode)
        self.fuse_block = bool(params.get('fuse_block', True))   # conv_block / decoder junction as single tape entries
        self._skip_boxes = {}
        if not 1 <= self.n_inputs <= 7:
            raise ValueError('the bf16 graph takes an f32 image of 1..7 channels (num_inputs)')
        if any(f % 16 for f in self.filters) or self.bridge_type == 'concat':
            raise ValueError('the bf16 graph needs filter counts that are multiples of 16 and an eltwise bridge')
        k = self.bridge_type
        self.bridge = (lambda x, y, _k=k: FB.bridge(x, y, _k)) if k else (lambda x, y: x)
        self._default_bridge = self.bridge

    def conv_layer(self, x, filters):
        w, b = self._kernel((3, 3, x.shape[-1], filters)), self._bias(filters)
        act = None if self.batch_norm else 'relu'              # BN sits between the conv and its ReLU
        z = FB.conv3x3_first(x, w, b, act=act) if x.dtype == torch.float32 else FB.conv2d(x, w, b, act=act)
        return self.batch_norm_layer(z, act='relu') if self.batch_norm else z

    def batch_norm_layer(self, z, act=None):
        """UNet2D.batch_norm_layer on a bf16 activation: same variables (gamma, beta, moving_mean, moving_variance in the
        conv's scope, all f32), statistics in f32 / f64, the result rounded to bf16 once (sq_bn_*_bf16)."""
        from .. import ops_bf16 as ob
        n = z.shape[-1]
        gamma = self.get_variable('gamma', (n,), lambda s: np.ones(s, np.float32))
        beta = self.get_variable('beta', (n,), lambda s: np.zeros(s, np.float32))
        mmean = self.get_variable('moving_mean', (n,), lambda s: np.zeros(s, np.float32))
        mvar = self.get_variable('moving_variance', (n,), lambda s: np.ones(s, np.float32))
        if self.training:
            return FB.batch_norm_train(z, gamma, beta, mmean, mvar, self.bn_eps, self.bn_momentum, act=act)
        return ob.bn_inference(z, gamma, beta, mmean, mvar, self.bn_eps, act=act)

    _WIRING = ('build', 'build_loss', 'down_layer', 'up_layer')

    def _plain_wiring(self):
        """The backward fusions that skip a gradient pass (BlockGate in the head / head + loss / up_junction / the
        pool + skip pair, JunctionHandoff, the pool box) rest on the reference's wiring: a block output feeds the pool
        and its skip junction, or the next up_layer, or the head, and nothing else (unet.py:241-253).  A subclass that
        overrides the wiring (deep supervision, an extra skip) may consume net[-1] or `merged` twice; one gated and
        one ungated contribution would then be summed and the producer would skip its activation backward.  So the
        single-consumer promises are made only while build / build_loss / down_layer / up_layer are this class's own
        (ADVICE r2); otherwise every tensor takes the ordinary tape entries -- same forward bits, full backward."""
        return all(getattr(type(self), h) is getattr(UNet2DBf16, h) for h in self._WIRING)

    def conv_block(self, x, filters):
        """unet.py:265-277.  While neither conv_layer nor dropout_layer is overridden the block is one tape
        entry (FB.conv_block): same forward kernels, backward with the two ReLU gradients fused away."""
        if not (self.training and self.fuse_block and not self.batch_norm and type(self).conv_layer is UNet2DBf16.conv_layer
                and type(self).dropout_layer is UNet2DBf16.dropout_layer):
            return UNet2D.conv_block(self, x, filters)
        with self.variable_scope('conv1'):
            w1, b1 = self._kernel((3, 3, x.shape[-1], filters)), self._bias(filters)
        with self.variable_scope('conv2'):
            w2, b2 = self._kernel((3, 3, filters, filters)), self._bias(filters)
        rate = self.dropout if self.dropout > 0.0 else 0.0
        mask = step_dev = None
        seed = 0
        if rate > 0.0:
            mask = self.dropout_masks.pop(0) if getattr(self, 'dropout_masks', None) else None
            step_dev = None if mask is not None else getattr(self, '_step_dev', None)
            seed = self._dropout_seed(step_dev)
        pool_follows, self._pool_next = getattr(self, '_pool_next', False), False
        return FB.conv_block(x, w1, b1, w2, b2, rate, seed, mask, step_dev, pool_follows=pool_follows)

    def down_layer(self, x, filters, name=None):
        """unet.py:282-296.  Every encoder level but the last is followed by the max pool (unet.py:241-243): its block
        then writes the pooled tensor from conv2's epilogue (FB.conv_block(pool_follows=True)) and pool_layer picks it up."""
        self._pool_next = (self.training and self.fuse_block and not self.batch_norm and isinstance(name, int)
                           and name < len(self.filters) - 1
                           and type(self).pool_layer is UNet2DBf16.pool_layer and self._plain_wiring())
        try:
            return UNet.down_layer(self, x, filters, name=name)
        finally:
            self._pool_next = False

    def conv_layer_1x1(self, x, filters):
        w, b = self._kernel((1, 1, x.shape[-1], filters)), self._bias(filters)
        if self.training:
            # build() hands the last block's output to the head and to nothing else (unet.py:252-253)
            labels = getattr(self, '_loss_inputs', None)
            if labels is not None and x.shape[-1] in (16, 32) and filters <= 5 and self._plain_wiring():
                self._loss = FB.conv1x1_head_loss(x, w, b, labels[0], labels[1], x_single_use=self._only_use(x))
                self._head_wb = (w, b)
                return self._loss                               # build_loss(): the loss stands in for the logits
            return FB.conv1x1_head(x, w, b, x_single_use=self._only_use(x))
        from .. import ops_bf16 as ob
        logits, self._mask = ob.head_fwd(x, w, b)
        return logits

    def _only_use(self, x):
        """x is the last block's output and the caller is its only differentiable consumer (the reference's wiring)"""
        return bool(self._net) and x is self._net[-1] and self._plain_wiring()

    def logits(self):
        """unet.py:220-222.  After build_loss() the logits were never stored: they are evaluated here, on demand,
        from the last block's output (inspection / tests; not part of the training step)."""
        if getattr(self, '_loss', None) is not None and self._net[-1] is self._loss:
            from .. import ops_bf16 as ob
            with torch.no_grad():
                return ob.head_fwd(self._net[-2].detach(), self._head_wb[0].detach(),
                                   None if self._head_wb[1] is None else self._head_wb[1].detach(), want_mask=False)[0]
        return self._net[-1]

    def build_loss(self, features, onehot, weights):
        """Training graph ending in the weighted softmax cross-entropy instead of the logits: the
        head and the loss run as one tape entry (FB.conv1x1_head_loss) and the logits are never stored.  Falls back
        to build() + the stand-alone loss where that entry does not apply (an overridden head, > 5 classes)."""
        from .. import functional as F32
        self._loss_inputs, self._loss = (onehot, weights), None
        try:
            out = self.build(features)
        finally:
            self._loss_inputs = None
        if self._loss is not None and out is self._loss:
            return out
        return F32.weighted_softmax_cross_entropy(out, onehot, weights)

    def conv_transpose_layer(self, x, filters):
        return FB.convT2x2s2(x, self._kernel((2, 2, filters, x.shape[-1])), self._bias(filters))

    def pool_layer(self, x):
        if self.training and self.fuse_block and self._plain_wiring():
            # the same tensor is the skip operand of a decoder junction later: share a box with it so the
            # junction's skip gradient is added inside this pool's backward (FB._MaxPool)
            box = {}
            self._skip_boxes[id(x)] = (x, box)
            return FB.maxpool2x2(x, box)
        return FB.maxpool2x2(x)

    def dropout_layer(self, x):
        if not self.training or self.dropout <= 0.0:
            return x
        mask = self.dropout_masks.pop(0) if getattr(self, 'dropout_masks', None) else None
        step_dev = None if mask is not None else getattr(self, '_step_dev', None)
        return FB.dropout(x, self.dropout, seed=self._dropout_seed(step_dev), mask=mask, step_dev=step_dev)

    def up_layer(self, x, filters, bridge, name=None):
        fused = (self.training and self.fuse_block and self.up_kernel == (2, 2)
                 and type(self).conv_transpose_layer is UNet2DBf16.conv_transpose_layer
                 and self.bridge is self._default_bridge and self.bridge_type in ('eltwise_add', 'eltwise_mul', 'eltwise_sub'))
        if not fused:
            return UNet.up_layer(self, x, filters, bridge, name=name)
        with self.variable_scope('up{0:d}'.format(name)):
            with self.variable_scope('upscale'):
                w, b = self._kernel((2, 2, filters, x.shape[-1])), self._bias(filters)
            entry = self._skip_boxes.get(id(bridge))
            box = entry[1] if entry is not None and entry[0] is bridge else None
            # build() feeds net[-1] to this up_layer only (unet.py:248): its block gate can ride in the dgrad epilogue
            merged = FB.up_junction(x, w, b, bridge, self.bridge_type, box, x_single_use=self._only_use(x),
                                    merged_single_use=type(self).conv_block is UNet2DBf16.conv_block)   # merged goes straight into the block below
            out = self.conv_block(merged, filters)
        return out

    def build(self, features):
        self._skip_boxes = {}
        return UNet2D.build(self, features)
