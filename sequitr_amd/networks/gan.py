"""Progressive WGAN-GP on the MI355X HIP back end -- mirror of sequitr/networks/gan.py.

Same module-level leaf functions (``k_leaky_relu_alpha``, ``pixel_norm``, ``weighted_conv2d``,
``to_image``, ``from_image``, ``half_size``, ``double_size``; gan.py:44-136), the same
``discriminator_network(x, filters)`` / ``generator_network(z, filters)`` wiring (gan.py:149-316),
``GAN2DConfiguration`` (gan.py:418-440) and ``GenerativeAdverserialNetwork`` with ``build`` /
``train`` / ``predict`` (gan.py:443-924).  TensorFlow's graph, session and variable scopes are
replaced by eager HIP kernel launches and ``scope.VariableStore``; variable names are the
reference's (``GAN/generator/latent/conv/filter`` ...).

What differs on purpose (HISTORY.md section 8): the data source is a ``.npy`` stack or synthetic
tiles instead of a TFRecord (TFRecord IO is out of scope); checkpoints are ``model_(HxW).npz``;
``predict`` uses its ``latent`` argument (the reference feeds an undefined global, gan.py:923);
every forward and gradient op, including the minibatch-stdev statistic and the (N,)-sized loss
algebra, is a hand-written HIP kernel (torch supplies the tape, the concatenation and the copies).
"""
import json
import logging
import os

import numpy as np
import torch

from .. import functional as F
from .. import ops
from .. import utils
from . import scope
from .scope import variable_scope

logger = logging.getLogger('worker_process')

TRAIN = 'train'


# ---- handy functions (gan.py:44-51) ---------------------------------------------------------------
def k_leaky_relu_alpha(features, **kwargs):
    """ leaky ReLU with alpha of 0.2 (stand-alone form; convolutions fuse it) """
    return F.act(features, 'leaky')


def pixel_norm(x, epsilon=1e-8):
    """ x * rsqrt(mean(x^2, axis=-1) + eps), gan.py:49-51 """
    return F.pixel_norm(x.contiguous(), epsilon)


def _act_name(activation):
    if activation is None:
        return None
    if activation in ('leaky', k_leaky_relu_alpha):
        return 'leaky'
    if activation == 'relu':
        return 'relu'
    raise ValueError('unsupported activation %r (None, k_leaky_relu_alpha, "leaky", "relu")' % (activation,))


def weighted_conv2d(inputs=None, filters=None, kernel_size=[3, 3], padding="same",
                    activation=k_leaky_relu_alpha, name='conv', reuse=None, norm=True, pool=False):
    """Equalised-learning-rate convolution (gan.py:61-99): kernel ~ N(0,1) scaled at run time by
    sqrt(2 / (kh*kw*filters)), + bias (1,1,1,filters), activation, optional pixel norm.
    pool (not in the reference's signature): the 2x2 average pool of the discriminator's blocks (gan.py:189-192) applied to
    the result -- written from the conv kernel's epilogue where that form exists, else F.avgpool2x2 of the result."""
    if padding.lower() != 'same':
        raise ValueError('only SAME padding is implemented')
    kh, kw = int(kernel_size[0]), int(kernel_size[1])
    cin = int(inputs.shape[-1])
    with variable_scope(name):
        kernels = scope.get_variable('filter', (kh, kw, cin, filters), scope.random_normal)
        bias = scope.get_variable('bias', (1, 1, 1, filters), scope.zeros)
    wscale = float(np.sqrt(np.float32(2.0 / float(kh * kw * filters))))
    if pool and not norm:
        return F.conv2d_avgpool(inputs, kernels, bias.view(-1), act=_act_name(activation), wscale=wscale)
    # norm: the pixel norm leaves the conv's own epilogue where that kernel exists (bf16 features, Cout <= 64); pixel_norm() then
    # returns the tensor it is handed instead of launching
    out = F.conv2d(inputs, kernels, bias.view(-1), act=_act_name(activation), wscale=wscale,
                   pixelnorm_eps=1e-8 if norm else None)
    if norm:
        out = pixel_norm(out)
    return F.avgpool2x2(out) if pool else out


def to_image(X, filters=2, n=None):
    """ 1x1 convolution to an image, no activation, no norm (gan.py:102-113) """
    return weighted_conv2d(inputs=X, filters=filters, kernel_size=[1, 1], activation=None,
                           name='to_image{}'.format(n), norm=False)


def from_image(X, filters=2, n=None):
    """ 1x1 convolution from an image, leaky + pixel norm (gan.py:115-125) """
    return weighted_conv2d(inputs=X, filters=filters, kernel_size=[1, 1], activation=k_leaky_relu_alpha,
                           name='from_image{}'.format(n), norm=True)


def half_size(X):
    """ resize_nearest_neighbor(align_corners=True) to half the size (gan.py:128-131) """
    return ops.resize_nearest(X.contiguous(), (X.shape[1] // 2, X.shape[2] // 2))


def double_size(X):
    """ resize_nearest_neighbor(align_corners=True) to double the size (gan.py:133-136) """
    return F.double_size(X)


def dense(inputs, units, activation=None, name='dense'):
    """tf.layers.dense on the last axis: glorot-uniform kernel (in, units), zero bias."""
    cin = int(inputs.shape[-1])
    with variable_scope(name):
        kernel = scope.get_variable('kernel', (cin, units), scope.glorot_uniform)
        bias = scope.get_variable('bias', (units,), scope.zeros)
    lead = inputs.shape[:-1]
    y = F.dense(inputs.reshape(-1, cin), kernel, bias, act=_act_name(activation))
    return y.reshape(tuple(lead) + (units,))


def minibatch_stdev(x, groups=1):
    """gan.py:204-212: sqrt(mean over (h,w,c) of the population variance over the batch), as a
    constant feature map (N,4,4,1): one workgroup per minibatch (sq_mbstd_map_*_f32, differentiable twice).
    groups > 1: x is `groups` minibatches stacked along the batch axis, each gets its own statistic
    (the reference evaluates the discriminator once per minibatch; see _build_network)."""
    # bf16 storage: the kernels read the bf16 features themselves (f32 arithmetic) and hand back bf16 gradients -- what the
    # f32 copy and the casts of its gradients gave, bit for bit, without their launches
    return F.mbstd_map(x, groups, 16).reshape(x.shape[0], 4, 4, 1)


def discriminator_network(x, filters, groups=1):
    """gan.py:149-240.  Returns (conv_layers, logits (N,)).  groups: number of minibatches stacked in x (every
    layer but the minibatch statistic is per sample)."""
    num_layers = len(filters)
    with variable_scope("from_image"):
        x = from_image(x, filters=filters[0], n=num_layers - 1)
        conv_layers = [x]
    for l, f in enumerate(filters[1:]):
        with variable_scope("layer_{0:d}".format(num_layers - l - 1)):
            conv1 = weighted_conv2d(inputs=conv_layers[-1], filters=f, kernel_size=[3, 3],
                                    activation=k_leaky_relu_alpha, name='conv1', norm=False)
            # conv2 and the average pool that follows it: one kernel where the fused form exists (weighted_conv2d(pool=True))
            conv_layers.append(weighted_conv2d(inputs=conv1, filters=f, kernel_size=[3, 3],
                                               activation=k_leaky_relu_alpha, name='conv2', norm=False, pool=True))
    x = conv_layers[-1]
    with variable_scope('output'):
        mbstd = minibatch_stdev(x, groups)
        conv = weighted_conv2d(inputs=x, filters=filters[-1], kernel_size=[3, 3],
                               activation=k_leaky_relu_alpha, name='conv', norm=False)
        if conv.dtype == torch.bfloat16:                        # bf16 storage ends here: cast + concat + flatten in one pass
            pool_flat = F.head_concat(conv, mbstd.reshape(mbstd.shape[0], 16))
        else:
            conv = torch.cat([F.cast(conv, torch.float32), mbstd], dim=-1)   # the dense head is f32
            pool_flat = conv.reshape(-1, 4 * 4 * (filters[-1] + 1))
        hidden = dense(pool_flat, filters[-1], activation=k_leaky_relu_alpha, name='dense')
        logits = dense(hidden, 1, name='logits')
    return conv_layers, logits.reshape(-1)


def generator_network(z, filters, start_shape=(4, 4), levels=None):
    """gan.py:246-316.  Returns (list of per-level images, last image).  levels: the entries of that list the caller will
    read (None = all): a TF-1 session evaluates only the to_image ops a fetch depends on, so the training step, which
    reads the last two images (gan.py:665-694), never ran the other five; here the others are None."""
    with variable_scope('latent'):
        initial_shape = tuple(start_shape) + (filters[0],)
        num_units = int(np.prod(initial_shape))
        d = dense(pixel_norm(z), num_units, activation=k_leaky_relu_alpha, name='dense1')
        reshaped = pixel_norm(d.reshape((-1,) + initial_shape))
        if ops.STORE_BF16:                                      # bf16 storage starts at the first feature map
            reshaped = F.cast(reshaped, torch.bfloat16)
        conv0 = weighted_conv2d(inputs=reshaped, filters=filters[0], kernel_size=[3, 3],
                                activation=k_leaky_relu_alpha, name='conv', norm=True)
    conv_layers = [conv0]
    for l, f in enumerate(filters[1:]):
        with variable_scope('layer_{0:d}'.format(l)):
            upscale = double_size(conv_layers[-1])
            conv1 = weighted_conv2d(inputs=upscale, filters=f, kernel_size=[3, 3],
                                    activation=k_leaky_relu_alpha, name='conv1', norm=True)
            conv2 = weighted_conv2d(inputs=conv1, filters=f, kernel_size=[3, 3],
                                    activation=k_leaky_relu_alpha, name='conv2', norm=True)
            conv_layers.append(conv2)
    outputs = []
    with variable_scope("to_image"):
        last = len(conv_layers) - 1
        for l, conv in enumerate(conv_layers):
            wanted = levels is None or l == last or l in levels or (l - len(conv_layers)) in levels
            output = to_image(conv, filters=2, n=l) if wanted else None
            outputs.append(output)
    return outputs, output


class GAN2DConfiguration(utils.NetConfiguration):
    """ defaults of gan.py:418-440 """

    def __init__(self, params=None):
        utils.NetConfiguration.__init__(self)
        self.name = 'GAN_competition'
        self.batch_size = 32
        self.repeat_batch = 4
        self.num_outputs = 2
        self.num_levels = 7
        self.num_epochs_per_level = 1
        self.start_size = (4, 4)
        self.learning_rate = 1e-3
        self.warm_start = False
        self.path = ''
        self.training_data = 'train_GAN.tfrecord'


class _Adam(object):
    """tf.train.AdamOptimizer(lr, beta1=0, beta2=0.99) (gan.py:736-751): per-variable slots, ONE
    shared beta-power counter per optimiser, advanced by every minimize() that runs."""

    def __init__(self, lr, beta1=0.0, beta2=0.99, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, eps
        self.t = 0                  # host mirror of the device counter
        self.state = None           # int32[2] {step, lr_t bits} on the device: a captured step replays correctly
        self.slots = {}

    def _live(self, named_vars, grads):
        live = [(name, v, g) for (name, v), g in zip(named_vars, grads) if g is not None]
        for name, v, _ in live:
            if name not in self.slots:
                self.slots[name] = (torch.zeros_like(v), torch.zeros_like(v))
        if live and self.state is None:
            self.state = torch.zeros(2, dtype=torch.int32, device=live[0][1].device)
        return live

    def table(self, named_vars, grads):
        """Pointer table for apply(..., table=): every live variable's {weights, gradient, slots} in device memory, so
        that a minimize() is two launches (advance + one update over the list).  The gradients must be the static
        tensors of a captured gradient graph (the table holds their addresses); None if one is not contiguous."""
        live = self._live(named_vars, grads)
        if not live or not all(g.is_contiguous() and v.is_contiguous() for _, v, g in live):
            return None
        return ops.adam_table([v.detach().view(-1) for _, v, _ in live], [g.view(-1) for _, _, g in live],
                              [self.slots[n][0].view(-1) for n, _, _ in live], [self.slots[n][1].view(-1) for n, _, _ in live])

    def apply(self, named_vars, grads, grad_scale=1.0, table=None):
        live = self._live(named_vars, grads)
        if not live:
            return
        self.t += 1
        ops.adam_advance_dev(self.state, self.lr, self.b1, self.b2)       # ONE shared beta-power step per minimize()
        if table is not None:
            ops.adam_apply_multi_dev(table, self.b1, self.b2, self.eps, self.state, grad_scale=grad_scale)
            return
        for name, v, g in live:
            m, s = self.slots[name]
            ops.adam_apply_dev(v.detach().view(-1), g.contiguous().view(-1), m.view(-1), s.view(-1), self.b1, self.b2,
                               self.eps, self.state, grad_scale=grad_scale)


class GenerativeAdverserialNetwork(object):
    """ProGAN-style WGAN-GP (gan.py:443-924).

    params: num_outputs, batch_size, repeat_batch, num_levels, num_epochs_per_level, start_size,
    training_data (``.npy`` (N,H,W,C) stack; None -> synthetic tiles), learning_rate; new keys:
    device, seed, num_batches_per_epoch (when the data is synthetic), dtype ('f32' default: exact-f32 MFMA
    convolutions; 'bf16': BASELINE config 5 -- bf16-multiply / f32-accumulate convolutions AND bf16 storage of every
    feature map and feature-map gradient, f32 parameters / images / losses; 'mixed': the bf16 multiplies behind f32 tensors),
    graph (default False: True replays the solver steps as hipGraphs), batch_d (default True: the discriminator sees
    the generated and the real minibatch as one stacked batch, each with its own minibatch statistic)."""

    def __init__(self, params, mode=None, discriminator_fn=discriminator_network,
                 generator_fn=generator_network):
        self.__discriminator_fn = discriminator_fn
        self.__generator_fn = generator_fn
        self.initialized = False
        self.__num_channels = params.get('num_outputs', 2)
        self.batch_size = params.get('batch_size', 32)
        self.repeat_batch = params.get('repeat_batch', 1)
        self.num_levels = params.get('num_levels', 3)
        self.num_epochs_per_level = params.get('num_epochs_per_level', 10)
        self.start_size = params.get('start_size', (4, 4))
        self.training_data_filename = params.get('training_data', None)
        self.learning_rate = params.get('learning_rate', 1e-3)
        self.batch_d = bool(params.get('batch_d', True))     # D(Gz) and D(X) in one stacked pass (default discriminator only)
        self._default_d = discriminator_fn is discriminator_network
        self._default_g = generator_fn is generator_network   # ActGate fusion needs the reference's wiring (single consumers)
        self.use_graph = bool(params.get('graph', False))    # replay the solver steps as hipGraphs
        self._graphs = {}
        self._graph_alpha = {}                                  # last fade-in weight written into each graph's static tensor
        self._plan = None
        self._pack_epoch = -1
        self._gflat, self._gsinks = None, None                  # flat gradient buffer + per-parameter sinks (dtype 'bf16')
        self._capture_stream = None
        self._ones_cache = {}
        # every workspace this network's launches use (split reductions, grouped weight-gradient partials) comes from ITS
        # arena: captured solver graphs bake the buffer's address, so it must not be the library-wide default that any other
        # object may grow or use from another stream (ops.WorkspaceArena)
        self.arena = ops.WorkspaceArena('GAN')
        self.dtype = params.get('dtype', 'f32')
        if self.dtype not in ('f32', 'bf16', 'mixed'):
            raise ValueError("dtype must be 'f32', 'bf16' or 'mixed', got %r" % (self.dtype,))
        dev = params.get('device', None)
        self.device = torch.device(dev) if dev is not None else torch.device('cuda', torch.cuda.current_device())
        if self.device.type != 'cuda':
            raise RuntimeError('GenerativeAdverserialNetwork runs on the HIP back end only')
        self.seed = params.get('seed', 0)
        self.store = scope.VariableStore(self.device, seed=self.seed, trainable=True)
        self._torch_rng = torch.Generator(device=self.device)
        self._torch_rng.manual_seed(self.seed)
        self.group = params.get('process_group', None)

        self.dataset = None
        self.num_batches_per_epoch = params.get('num_batches_per_epoch', 1)
        if mode == TRAIN and self.training_data_filename:
            fn = self.training_data_filename
            if not (isinstance(fn, str) and fn.endswith('.npy')):
                raise IOError('training_data must be a .npy stack (TFRecord IO is out of scope): %r' % (fn,))
            self.dataset = np.load(fn, mmap_mode='r', allow_pickle=False)
            self.num_batches_per_epoch = max(1, int(len(self.dataset) / self.batch_size))

        self.restore = False
        self.networks = []
        self.mode = mode
        self.__params = params
        self.__level = 0
        self.global_step = 0
        self.output_dir = params.get('output', None)
        self.filters = utils.filter_doubling(start_filters=8, num_layers=self.num_levels,
                                             max_filters=512, reverse=True)
        self.d_opt = self.g_opt = None
        self._last_losses = None

    # -- properties / helpers, gan.py:542-568 ---------------------------------------------------------
    @property
    def num_channels(self):
        return self.__num_channels

    @property
    def current_level(self):
        return self.__level

    @property
    def num_iterations_this_level(self):
        return self.num_epochs_per_level * self.num_batches_per_epoch

    def get_size(self, level):
        return tuple([s * (2 ** level) for s in self.start_size])

    @property
    def current_size(self):
        return self.get_size(self.current_level)

    def expand(self):
        self.__level += 1
        assert (self.__level <= self.num_levels)

    def set_level(self, level):
        assert 0 <= level < self.num_levels
        self.__level = level

    def generator(self, Z, filters, **kwargs):
        with self.store, variable_scope('GAN'), variable_scope('generator'):
            return self.__generator_fn(Z, filters, **kwargs)

    def discriminator(self, X, filters, **kwargs):
        with self.store, variable_scope('GAN'), variable_scope('discriminator'):
            return self.__discriminator_fn(X, filters, **kwargs)

    # -- variable lists, gan.py:586-614 (including the layer-naming quirk) -----------
    def get_training_variables(self, current_layer):
        tv = self.store.trainable_variables
        d_vars = self.discriminator_training_variables(current_layer)
        g_vars = self.generator_training_variables(current_layer)
        d_vars = d_vars + tv('GAN/discriminator/from_image/from_image{0:d}'.format(current_layer))
        g_vars = g_vars + tv('GAN/generator/to_image/to_image{0:d}'.format(current_layer))
        return d_vars, g_vars

    def generator_training_variables(self, current_layer):
        tv = self.store.trainable_variables
        out = []
        for layer in range(current_layer):
            out += tv('GAN/generator/layer_{0:d}'.format(layer))
        return out + tv('GAN/generator/latent')

    def discriminator_training_variables(self, current_layer):
        tv = self.store.trainable_variables
        out = []
        for layer in range(current_layer):
            out += tv('GAN/discriminator/layer_{0:d}'.format(layer))
        return out + tv('GAN/discriminator/output')

    # -- build, gan.py:619-657 -------------------------------------------------------------------------
    def build(self):
        """Create every level's variables (one probe batch per level) and the two optimisers."""
        self.d_opt, self.g_opt = self._build_optimizers()
        self.networks = []
        self.__level = 0
        with torch.no_grad():
            for i in range(self.num_levels):
                z = torch.zeros((1, 1, 1, 512), dtype=torch.float32, device=self.device)
                x = torch.zeros((1,) + self.current_size + (self.num_channels,), dtype=torch.float32, device=self.device)
                filters = self.filters[:(i + 1)]
                self.generator(z, filters)
                self.discriminator(x, filters[::-1])
                self.networks.append((i, 'd_loss', 'g_loss', 'd_solver', 'g_solver'))
                self.expand()
        self.__level = 0
        self.store.flatten()                                    # one contiguous parameter buffer (views keep their names)
        self._plan = None
        self._pack_epoch = -1
        self.initialized = True

    def _pack_filters(self):
        """dtype 'bf16' / 'mixed': the bf16 filter packs of a solver step (forward + dgrad form of each equalised-LR conv
        kernel), one launch per NETWORK whose weights have moved since they were packed (ops.FilterPackPlan).  A solver step
        updates one network: d_solver leaves the generator's packs valid and g_solver the discriminator's, so in the
        alternating loop every weight is packed once per iteration, not once per step (2 x 73 us -> 2 x ~37 us at level 6).
        Anybody else writing parameters (checkpoint load, assign, another optimiser) shows in ops.invalidation_epoch() and
        stales both."""
        if self.dtype == 'f32' or self.store.flat is None:
            return
        if self._plan is None:
            named = {'d': [], 'g': []}
            for name, v in self.store.vars.items():
                which = 'd' if name.startswith('GAN/discriminator/') else 'g'
                self._pack_named(named[which], name, v)
            self._plan = {k: ops.FilterPackPlan(self.store.flat, named[k]) for k in ('d', 'g')}
            self._pack_ok = {'d': False, 'g': False}
        if ops.invalidation_epoch() != self._pack_epoch:
            self._pack_ok = {'d': False, 'g': False}
        for k in ('d', 'g'):
            if self._pack_ok[k]:
                self._plan[k].fresh = True                      # its weights have not moved since it was packed
            else:
                self._plan[k].run()
                self._pack_ok[k] = True
        self._pack_epoch = ops.invalidation_epoch()

    def _weights_moved(self, kind):
        """the end of a solver step: `kind`'s weights were updated -- every cached pack goes, the OTHER network's plan
        comes back on the next _pack_filters()"""
        ops.invalidate_packs()
        if self._plan is not None:
            self._pack_ok[kind] = False
            self._pack_epoch = ops.invalidation_epoch()

    def _pack_named(self, named, name, v):
        if name.endswith('/filter') and v.dim() == 4 and name in self.store.offsets:
            kh, kw, _, cout = v.shape
            named.append((name, v, self.store.offsets[name], float(np.sqrt(np.float32(2.0 / float(kh * kw * cout))))))
        elif name.endswith('/kernel') and v.dim() == 2 and name in self.store.offsets:
            # dense layers that run as bf16-multiply 1x1 convs -- forward where the reduction is short (the generator's
            # 512 -> 8192 dense1; >= 1024 inputs take the split-reduction f32 kernel), dgrad where the OUTPUT side is
            # short (the discriminator's 8208 -> 512 dense): packed with everything else instead of once per use
            forms = ('N' if v.shape[0] < 1024 else '') + ('T' if v.shape[1] < 1024 else '')
            if forms:
                named.append((name, v, self.store.offsets[name], 1.0, forms))

    def _build_optimizers(self, level=0):
        return _Adam(self.learning_rate, 0.0, 0.99), _Adam(self.learning_rate, 0.0, 0.99)

    def _ones(self, n):
        t = self._ones_cache.get(n)
        if t is None:
            t = self._ones_cache[n] = torch.ones((n,), dtype=torch.float32, device=self.device)
        return t

    def _mixing_r(self, n):
        return torch.rand((n,), generator=self._torch_rng, dtype=torch.float32, device=self.device)

    def _stacks(self):
        """the discriminator step evaluates D on [generated; real] as one stacked batch: at levels > 0 both halves are the
        outputs of blend kernels, which then write straight into the halves of ONE buffer (no concatenation pass)"""
        return self.batch_d and self._default_d and self.current_level > 0

    def _generated(self, Z, alpha):
        """generator half of gan.py:665-694: (Gz_raw, Gz, stacked) -- the current level's image, its fade-in blend with
        the up-sampled image of the level below (the generator's tape attached), and the (2n, H, W, C) buffer whose first
        half Gz occupies (None when the step does not stack)"""
        num_layers = self.current_level
        filters = self.filters[:(num_layers + 1)]
        g_layers, Gz_raw = self.generator(Z, filters, **({'levels': (-2, -1)} if self._default_g else {}))
        if num_layers == 0:
            return Gz_raw, Gz_raw, None
        stacked = None
        if self._stacks():
            n = Gz_raw.shape[0]
            stacked = torch.empty((2 * n,) + tuple(Gz_raw.shape[1:]), dtype=torch.float32, device=self.device)
        Gz = F.lerp(Gz_raw, double_size(g_layers[-2]), alpha, out=None if stacked is None else stacked[:Gz_raw.shape[0]])
        return Gz_raw, Gz, stacked

    def _real(self, X, alpha, out=None):
        """real half: X faded with its own half-resolution copy (gan.py:682-691)"""
        if tuple(X.shape[1:3]) != tuple(self.current_size):
            raise ValueError('X must already be at the current size %s (bilinear resize of the real '
                             'data is host-side IO, gan.py:682-684)' % (self.current_size,))
        if self.current_level == 0:
            return X
        return ops.lerp(X, ops.broadcast2x2(half_size(X), 1.0), alpha, out=out)

    def _prepare(self, X, Z, alpha, need_g_graph, generated=None, want_real=True):
        """Forward of gan.py:665-714 up to the three discriminator inputs.  generated: a _generated() result evaluated
        earlier with the same Z, alpha and generator weights (iteration() shares one generator pass between the two
        solver steps); want_real=False: the generator step never reads X.  Returns (reversed filters, Gz_raw, Gz,
        X_resized, stacked-or-None)."""
        filters = self.filters[:(self.current_level + 1)]
        Gz_raw, Gz, stacked = generated if generated is not None else self._generated(Z, alpha)
        X_resized = None
        if want_real:
            X_resized = self._real(X, alpha, out=None if stacked is None else stacked[Gz.shape[0]:])
        if not need_g_graph:
            Gz = Gz.detach()
        return filters[::-1], Gz_raw, Gz, X_resized, (stacked if want_real else None)

    def _build_network(self, X, Z, alpha, r=None, need_g_graph=True, generated=None):
        """Losses of the current level (gan.py:665-732): returns (Gz_raw, d_loss, g_loss) with the
        autograd graph attached (d_loss w.r.t. the discriminator, g_loss w.r.t. the generator
        unless need_g_graph is False -- the discriminator step never needs it)."""
        d_filters, Gz_raw, Gz, X_resized, stacked = self._prepare(X, Z, alpha, need_g_graph=need_g_graph, generated=generated)
        if self.batch_d and self._default_d:
            # D(Gz) and D(X) as ONE pass over the stacked minibatches (own minibatch statistic each): the same
            # per-sample arithmetic, a third fewer launches of the small deep layers in forward and backward
            if stacked is None or need_g_graph:
                stacked = torch.cat([Gz, X_resized], 0)         # level 0 (no blends), or a caller that differentiates Gz here
            _, Dzx = self.discriminator(stacked, d_filters, groups=2)
            Dz = Dx = None
        else:
            Dzx = None
            _, Dz = self.discriminator(Gz, d_filters)
            _, Dx = self.discriminator(X_resized, d_filters)
        if r is None:
            r = self._mixing_r(X.shape[0])
        mix = F.lerp(X_resized, Gz.detach(), r).detach().requires_grad_(True)
        _, Dmix = self.discriminator(mix, d_filters)
        with F.grads_wanted([]):                                # the input gradient only: no weight / bias gradients
            # d(sum Dmix)/dmix with the ones handed in (Dmix.sum() costs a reduction, its backward a fill and an expand)
            grad = torch.autograd.grad(Dmix, mix, grad_outputs=self._ones(Dmix.shape[0]), create_graph=True)[0]
        # one-sided penalty 10 max(|grad| - 1, 0)^2, drift 0.001 Dx^2, the two means: one launch (sq_wgan_losses_*)
        gn2 = F.dot_per_sample(grad, grad)
        d_loss, g_loss = F.wgan_losses_stacked(Dzx, gn2) if Dzx is not None else F.wgan_losses(Dz, Dx, gn2)
        return Gz_raw, d_loss, g_loss

    def _allreduce(self, grads):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1):
            return 1.0
        live = [g for g in grads if g is not None]
        flat = torch.cat([g.reshape(-1) for g in live])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)       # ONE collective per solver step
        o = 0
        for g in live:
            g.copy_(flat[o:o + g.numel()].view_as(g))
            o += g.numel()
        return 1.0 / dist.get_world_size(self.group)

    def precision(self):
        """context for everything this network launches (forward AND backward): `with net.precision(): ...`"""
        return ops.mixed_precision(self.dtype != 'f32', store_bf16=self.dtype == 'bf16')

    @property
    def last_losses(self):
        """(d_loss, g_loss) of the last discriminator step as floats; read lazily so a step never waits for the GPU"""
        if self._last_losses is None:
            return None
        return tuple(float(t) for t in self._last_losses)

    _WIRING = ('generator', 'discriminator', '_prepare', '_build_network', '_d_grads', '_g_grads', '_param_grads')

    def _act_gates(self):
        """ActGate (the activation backward riding in the consumer's dgrad kernel) promises that every gated activation
        has one differentiable consumer.  That is the reference's wiring: the default generator_fn / discriminator_fn
        AND this class's own graph-building methods -- a subclass that overrides one of them may consume an activation
        twice, so it takes the ordinary act_bwd passes."""
        return (self._default_d and self._default_g
                and all(getattr(type(self), h) is getattr(GenerativeAdverserialNetwork, h) for h in self._WIRING))

    def d_solver(self, X, Z, alpha, r=None):
        """d_opt.minimize(d_loss, var_list=d_vars) for the current level (gan.py:649)."""
        try:
            with self.precision(), F.fuse_act_gates(self._act_gates()), ops.use_arena(self.arena):
                if self._graphable(alpha) and r is None:
                    return self._solver_graphed('d', X, Z, alpha)
                return self._d_solver(X, Z, alpha, r)
        finally:
            self._weights_moved('d')    # the step ends with a weight update, also under an outer precision() block

    def g_solver(self, X, Z, alpha):
        """g_opt.minimize(g_loss, var_list=g_vars, global_step) for the current level (gan.py:650-651)."""
        try:
            with self.precision(), F.fuse_act_gates(self._act_gates()), ops.use_arena(self.arena):
                if self._graphable(alpha):
                    return self._solver_graphed('g', X, Z, alpha)
                return self._g_solver(X, Z, alpha)
        finally:
            self._weights_moved('g')

    def iteration(self, X, Z, alpha, r=None):
        """d_solver(X, Z, alpha) followed by g_solver(X, Z, alpha) -- one pass of the reference's inner loop
        (gan.py:848-851: the same feed runs d_solver, then g_solver) -- with ONE generator forward pass: the
        discriminator step moves only discriminator weights, so the generator pass the generator step would repeat is the
        one the discriminator step has just evaluated, bit for bit (TensorFlow evaluates it once per session.run, i.e.
        twice).  The discriminator step reads it detached; the generator step differentiates through the kept tape.  Same
        losses, same weights as the two calls (tests/test_gpu_gan.py); 45 launches and ~0.4 ms fewer at level 6.
        Returns (d_loss, g_loss) device scalars."""
        self._d_marked = False
        try:
            with self.precision(), F.fuse_act_gates(self._act_gates()), ops.use_arena(self.arena):
                if self._graphable(alpha) and r is None:
                    return self._iteration_graphed(X, Z, alpha)
                return self._iteration(X, Z, alpha, r)
        finally:
            # the discriminator's packs were restaged right after ITS update, in the middle of the iteration, and stay valid
            # into the next one; only the generator's weights have moved since
            if not self._d_marked:
                self._weights_moved('d')
            self._weights_moved('g')

    def _iteration(self, X, Z, alpha, r=None):
        self._pack_filters()
        generated = self._generated(Z, alpha)
        d_vars, grads, losses = self._d_grads(X, Z, alpha, r, generated=generated)
        scale = self._allreduce(grads)
        self.d_opt.apply(d_vars, grads, grad_scale=scale)
        self._last_losses = losses
        self._weights_moved('d')
        self._d_marked = True
        self._pack_filters()                                    # the discriminator's packs follow its new weights
        g_vars, ggrads, glosses = self._g_grads(X, Z, alpha, generated=generated)
        scale = self._allreduce(ggrads)
        self.g_opt.apply(g_vars, ggrads, grad_scale=scale)
        self.global_step += 1
        return losses[0], glosses[0]

    def _iteration_graphed(self, X, Z, alpha):
        """iteration() as four hipGraphs: (generator forward + discriminator gradients), (Adam D), (D(Gz) forward + generator
        gradients through the FIRST graph's generator tape), (Adam G); filter packs and the two all-reduces between them,
        outside any capture.  Call 1 is eager (warm-up), call 2 captures, later calls replay."""
        key = ('it', self.current_level, tuple(X.shape), tuple(Z.shape))
        entry = self._graphs.get(key)
        if entry is None:
            self._graphs[key] = 'warm'
            return self._iteration(X, Z, alpha)
        if entry == 'warm':
            sx, sz = X.clone(), Z.clone()
            sa = torch.full((X.shape[0],), float(alpha), dtype=torch.float32, device=self.device)
            self._graph_alpha[key] = float(alpha)
            sr = torch.empty((X.shape[0],), dtype=torch.float32, device=self.device)
            if self._capture_stream is None:
                self._capture_stream = torch.cuda.Stream(device=self.device)
            self._pack_filters()
            torch.cuda.synchronize(self.device)
            self.arena.hand_over(self._capture_stream)
            gd, ad, gg, ag = (torch.cuda.CUDAGraph() for _ in range(4))
            with torch.cuda.graph(gd, stream=self._capture_stream):
                generated = self._generated(sz, sa)
                d_named, d_grads, d_losses = self._d_grads(sx, sz, sa, sr, generated=generated)
            td, tg = self.d_opt.t, self.g_opt.t
            d_table = self.d_opt.table(d_named, d_grads)
            torch.cuda.synchronize(self.device)
            with torch.cuda.graph(ad, stream=self._capture_stream, pool=gd.pool()):
                self.d_opt.apply(d_named, d_grads, grad_scale=1.0 / self._world(), table=d_table)
            # recording Adam marked every pack stale; the generator step must be captured with FRESH plans, or every conv of
            # it bakes an on-demand pack launch into the graph (the replay restages them in front of the graph, as here)
            self._weights_moved('d')
            self._pack_filters()
            torch.cuda.synchronize(self.device)
            with torch.cuda.graph(gg, stream=self._capture_stream, pool=gd.pool()):
                g_named, g_grads, g_losses = self._g_grads(sx, sz, sa, generated=generated)
            g_table = self.g_opt.table(g_named, g_grads)
            torch.cuda.synchronize(self.device)
            with torch.cuda.graph(ag, stream=self._capture_stream, pool=gd.pool()):
                self.g_opt.apply(g_named, g_grads, grad_scale=1.0 / self._world(), table=g_table)
            self.d_opt.t, self.g_opt.t = td, tg                 # the captures executed nothing
            del generated                                       # (the generator step's tape has been consumed by its capture)
            entry = self._graphs[key] = (gd, ad, gg, ag, sx, sz, sa, sr, d_grads, g_grads, d_losses, g_losses, d_table, g_table)
        gd, ad, gg, ag, sx, sz, sa, sr, d_grads, g_grads, d_losses, g_losses = entry[:12]
        sx.copy_(X), sz.copy_(Z)
        if self._graph_alpha.get(key) != float(alpha):
            sa.fill_(float(alpha))
            self._graph_alpha[key] = float(alpha)
        sr.copy_(self._mixing_r(X.shape[0]))
        self._pack_filters()
        self.arena.hand_over()
        gd.replay()
        self._allreduce(d_grads)
        ad.replay()
        self.d_opt.t += 1
        self._last_losses = d_losses
        self._weights_moved('d')
        self._d_marked = True
        self._pack_filters()                                    # the discriminator's packs, restaged in front of the generator step
        gg.replay()
        self._allreduce(g_grads)
        ag.replay()
        self.g_opt.t += 1
        self.global_step += 1
        return d_losses[0], g_losses[0]

    # -- parameter gradients through sinks (bf16 storage): grouped weight-gradient launches, no framework adds -------------
    def _param_grads(self, loss, named):
        """torch.autograd.grad(loss, variables) for a solver step.  dtype 'bf16': the weight gradients of the bf16 feature
        convolutions are not returned through autograd (one launch + finish per layer and pass, summed by framework adds)
        but queued into per-parameter sinks -- views of one flat gradient buffer laid out like the parameter buffer -- and
        run as grouped launches when the pass is over (functional.grad_sinks, ops_bf16.WgradQueue)."""
        variables = [v for _, v in named]
        if self.dtype != 'bf16' or self.store.flat is None:
            with F.grads_wanted(variables):
                return torch.autograd.grad(loss, variables, allow_unused=True)
        from .. import ops_bf16 as ob
        if self._gflat is None:
            self._gflat = torch.zeros_like(self.store.flat)
            self._gsinks = {id(v): self._gflat[o:o + v.numel()].view(v.shape)
                            for name, v in self.store.vars.items() for o in [self.store.offsets.get(name)] if o is not None}
        with F.grads_wanted(variables), F.grad_sinks(self._gsinks) as sk, ob.deferred_wgrads():
            grads = list(torch.autograd.grad(loss, variables, allow_unused=True))
        for i, v in enumerate(variables):                       # after the flush: the sinks hold their sums
            if id(v) in sk.touched:
                sink = self._gsinks[id(v)]
                if grads[i] is not None:                        # a contribution that did not go through the queue
                    sink.add_(grads[i])
                grads[i] = sink
        return tuple(grads)

    # the two halves of a solver step: (losses + gradients) and (Adam); the all-reduce sits between them
    def _d_grads(self, X, Z, alpha, r, generated=None):
        d_vars, _ = self.get_training_variables(self.current_level)
        _, d_loss, g_loss = self._build_network(X, Z, alpha, r=r, need_g_graph=False, generated=generated)
        grads = self._param_grads(d_loss, d_vars)               # not the block d_vars leaves out
        return d_vars, grads, (d_loss.detach(), g_loss.detach())

    def _g_grads(self, X, Z, alpha, generated=None):
        _, g_vars = self.get_training_variables(self.current_level)
        d_filters, _, Gz, _, _ = self._prepare(X, Z, alpha, need_g_graph=True, generated=generated, want_real=False)
        _, Dz = self.discriminator(Gz, d_filters)
        _, g_loss = F.wgan_losses(Dz)
        grads = self._param_grads(g_loss, g_vars)               # the pass runs THROUGH the discriminator: none of its weights
        return g_vars, grads, (g_loss.detach(),)

    def _d_solver(self, X, Z, alpha, r=None):
        self._pack_filters()
        d_vars, grads, losses = self._d_grads(X, Z, alpha, r)
        scale = self._allreduce(grads)
        self.d_opt.apply(d_vars, grads, grad_scale=scale)
        self._last_losses = losses
        return losses[0]

    def _g_solver(self, X, Z, alpha):
        self._pack_filters()
        g_vars, grads, losses = self._g_grads(X, Z, alpha)
        scale = self._allreduce(grads)
        self.g_opt.apply(g_vars, grads, grad_scale=scale)
        self.global_step += 1
        return losses[0]

    # -- hipGraph replay of the solver steps (a step is ~700 short launches: host-bound from Python) ----------------
    def _graphable(self, alpha):
        return self.use_graph                                   # alpha rides in a device tensor: fade phases replay too

    def _solver_graphed(self, kind, X, Z, alpha):
        """call 1 for a (solver, level, batch shape): the ordinary eager step (it also warms every kernel up);
        call 2: capture (gradients) and (Adam) as two hipGraphs and replay them; later calls: copy the inputs into the
        static buffers and replay.  The fade-in alpha is a per-sample device tensor (the form F.lerp / ops.lerp take
        anyway), refilled before each replay, so one pair of graphs serves the fade and the stabilisation phase; the
        mixing tensor r is drawn eagerly before each replay; the gradient all-reduce stays between the two graphs,
        outside any capture, so the RCCL call is an ordinary stream operation."""
        key = (kind, self.current_level, tuple(X.shape), tuple(Z.shape))
        entry = self._graphs.get(key)
        if entry is None:
            self._graphs[key] = 'warm'
            return self._d_solver(X, Z, alpha) if kind == 'd' else self._g_solver(X, Z, alpha)
        opt = self.d_opt if kind == 'd' else self.g_opt
        if entry == 'warm':
            sx, sz = X.clone(), Z.clone()
            sa = torch.full((X.shape[0],), float(alpha), dtype=torch.float32, device=self.device)
            self._graph_alpha[key] = float(alpha)
            sr = torch.empty((X.shape[0],), dtype=torch.float32, device=self.device) if kind == 'd' else None
            if self._capture_stream is None:
                self._capture_stream = torch.cuda.Stream(device=self.device)
            self._pack_filters()                                # the packs are launched in FRONT of the graph, by staleness (below)
            torch.cuda.synchronize(self.device)
            self.arena.hand_over(self._capture_stream)          # eager step's stream -> the capture stream
            g_grad, g_adam = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_grad, stream=self._capture_stream):
                named, grads, losses = self._d_grads(sx, sz, sa, sr) if kind == 'd' else self._g_grads(sx, sz, sa)
            t_host = opt.t
            table = opt.table(named, grads)                     # addresses of the static gradients: one update launch
            torch.cuda.synchronize(self.device)
            with torch.cuda.graph(g_adam, stream=self._capture_stream):
                opt.apply(named, grads, grad_scale=1.0 / self._world(), table=table)
            opt.t = t_host                                      # the capture executed nothing
            entry = self._graphs[key] = (g_grad, g_adam, sx, sz, sa, sr, grads, losses, table)
        g_grad, g_adam, sx, sz, sa, sr, grads, losses = entry[:8]     # entry[8]: the Adam table the graph reads
        sx.copy_(X), sz.copy_(Z)
        if self._graph_alpha.get(key) != float(alpha):          # the fade-in weight is refilled only when it moves
            sa.fill_(float(alpha))
            self._graph_alpha[key] = float(alpha)
        if sr is not None:
            sr.copy_(self._mixing_r(X.shape[0]))
        self._pack_filters()                                    # whichever network's weights moved since its packs were made
        self.arena.hand_over()                                  # the replay's launches use the arena on THIS stream
        g_grad.replay()
        self._allreduce(grads)
        g_adam.replay()
        opt.t += 1
        if kind == 'd':
            self._last_losses = losses
        else:
            self.global_step += 1
        return losses[0]

    def _world(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.group)
        return 1

    # -- data -----------------------------------------------------------------------------------------
    def build_latent(self):
        return torch.randn((self.batch_size, 1, 1, 512), generator=self._torch_rng, dtype=torch.float32,
                           device=self.device)

    def _next_real_batch(self, step):
        size = self.current_size
        if self.dataset is None:
            g = torch.Generator(device=self.device)
            g.manual_seed(self.seed * 7919 + step)
            return torch.randn((self.batch_size,) + size + (self.num_channels,), generator=g,
                               dtype=torch.float32, device=self.device)
        n = len(self.dataset)
        idx = [(step * self.batch_size + k) % n for k in range(self.batch_size)]
        x = torch.from_numpy(np.ascontiguousarray(self.dataset[sorted(idx)], dtype=np.float32)).to(self.device)
        if tuple(x.shape[1:3]) != size:                      # stored at full size: area-free nearest pick
            x = ops.resize_nearest(x, size)
        return x

    # -- train, gan.py:782-870 ---------------------------------------------------------------------------
    def checkpoint_name(self, level):
        """per-level checkpoint, the reference's ``model_(HxW).ckpt`` (gan.py:864-867) as an .npz"""
        return "model_{0:s}.npz".format(str(self.get_size(level)).replace(', ', 'x'))

    def latest_checkpoint(self):
        """(level, path) of the highest level with a checkpoint in output_dir, or (None, None).  The
        reference restores a hard-coded ``model_(1024x1024).ckpt`` / start_network 7 (gan.py:811-816,
        SURVEY A.5); here ``restore = True`` resumes from whatever level was last completed."""
        if not self.output_dir:
            return None, None
        for level in reversed(range(self.num_levels)):
            fn = os.path.join(self.output_dir, self.checkpoint_name(level))
            if os.path.exists(fn):
                return level, fn
        return None, None

    def train(self, max_steps_per_phase=None):
        if not self.initialized:
            raise Exception("Networks have not been initialized. Please run .build()")
        if self.output_dir:
            utils.check_and_makedir(self.output_dir)
        step_id = 0
        start_network = 0
        if self.restore:
            level, fn = self.latest_checkpoint()
            if fn is not None:
                self.load_checkpoint(fn)
                start_network = level + 1
                logger.info('Restored {0:s}; resuming at level {1:d}'.format(fn, start_network))
        for n in range(start_network, len(self.networks)):
            self.set_level(n)
            iters = self.num_iterations_this_level
            if max_steps_per_phase:
                iters = min(iters, max_steps_per_phase)
            for phase in ('fade', 'stabilisation'):
                for step in range(iters):
                    fade = float(step + 1) / iters if phase == 'fade' else 1.0
                    z, x = self.build_latent(), self._next_real_batch(step_id)
                    step_id += 1
                    for _ in range(self.repeat_batch):
                        self.iteration(x, z, fade)              # d_solver then g_solver on the same feed, one generator pass
                    logger.info('{0} {1} {2} {3}'.format(self.global_step, phase, self.get_size(n), fade))
            if self.output_dir:
                np.savez(os.path.join(self.output_dir, self.checkpoint_name(n)), **self.store.state_dict())

    def load_checkpoint(self, filename):
        with np.load(filename, allow_pickle=False) as z:
            self.store.load_state_dict({k: z[k] for k in z.files})
        self.initialized = True

    def convert_checkpoint_to_model(self, level=None):
        """gan.py:874-903: turn a training checkpoint into an inference model.  Restores the checkpoint
        of `level` (default: the highest one present) and writes ``output_dir/export/``: ``weights.npz`` with
        the GENERATOR variables only and ``model.json`` naming the signature the reference's SavedModel has
        (inputs Z, alpha -> output Gz).  Returns the export folder."""
        if not self.initialized:
            self.build()
        if level is None:
            level, filename = self.latest_checkpoint()
            if filename is None:
                raise IOError('No checkpoint found in {0}'.format(self.output_dir))
        else:
            filename = os.path.join(self.output_dir, self.checkpoint_name(level))
        logger.info('Converting {0:s} to model...'.format(filename))
        self.load_checkpoint(filename)
        export_dir = os.path.join(self.output_dir, 'export')
        utils.check_and_makedir(export_dir)
        gen = {k: v for k, v in self.store.state_dict().items() if k.startswith('GAN/generator/')}
        np.savez(os.path.join(export_dir, 'weights.npz'), **gen)
        with open(os.path.join(export_dir, 'model.json'), 'w') as f:
            json.dump({'inputs': {'Z': [None, 1, 1, 512], 'alpha': []},
                       'outputs': {'Gz': [None] + list(self.get_size(level)) + [self.num_channels]},
                       'level': int(level), 'filters': [int(c) for c in self.filters[:level + 1]],
                       'source_checkpoint': os.path.basename(filename)}, f, indent=2)
        return export_dir

    def predict(self, latent=None, level=None, alpha=1.0):
        """Generate images from latent vectors (N,1,1,512) at `level` (default: the last)."""
        if latent is None:
            raise Exception('You must provide latent variables to the model.')
        level = self.num_levels - 1 if level is None else level
        z = torch.as_tensor(np.asarray(latent), dtype=torch.float32).to(self.device).reshape(-1, 1, 1, 512)
        with torch.no_grad(), self.precision():
            _, out = self.generator(z, self.filters[:(level + 1)], **({'levels': (-1,)} if self._default_g else {}))
        return out


# ---------------------------------------------------------------------------------------------------
# command line, gan.py:1040-1125 (the reference's __main__ with its hard-coded paths as arguments)
# ---------------------------------------------------------------------------------------------------
def to_rgb(Gz):
    """Generated (N,H,W,2) images -> uint8 RGB (N,H,W,3): channels [1, 0, 1], ((1 + x/3) * 127.5) clipped
    to 0..255 (gan.py:1117-1125)."""
    Gz = np.asarray(Gz)
    rgb = np.concatenate([Gz[..., 1:2], Gz[..., 0:1], Gz[..., 1:2]], axis=-1)
    return ((1. + rgb / 3.) * 127.5).clip(0, 255).astype('uint8')


def get_run_number(folder):
    """highest <n> of the GAN<n> run folders in `folder`, 0 when there is none (gan.py:1072-1077)"""
    runs = [f for f in os.listdir(folder) if os.path.isdir(os.path.join(folder, f)) and f.startswith('GAN')
            and f[3:].isdigit()]
    return max([int(r[3:]) for r in runs]) if runs else 0


def main(argv=None):
    import argparse
    p = argparse.ArgumentParser(description='Sequitr: Progressive GAN')
    p.add_argument('--workdir', required=True, help='Path to job directory')
    p.add_argument('--num_epochs', type=int, default=100, help='Specify the number of epochs per expansion')
    p.add_argument('--num_levels', type=int, default=8, help='Specify the number of expansions from (4,4) start')
    p.add_argument('--batch_size', type=int, default=32, help='Specify the batch size')
    p.add_argument('--train', action='store_true', help='Train the model if this flag is present')
    p.add_argument('--restore', action='store_true', help='Continue training a model')
    p.add_argument('--training_data', default=None, help='.npy stack (N,H,W,C); synthetic tiles when absent')
    p.add_argument('--samples', type=int, default=512, help='images to export when predicting')
    args = p.parse_args(argv)

    config = GAN2DConfiguration()
    config.num_levels = args.num_levels
    config.num_epochs_per_level = args.num_epochs
    config.batch_size = args.batch_size
    config.training_data = args.training_data
    params = config.to_params()
    if args.train:
        gan = GenerativeAdverserialNetwork(params, TRAIN)
        gan.restore = args.restore
        gan.output_dir = args.workdir if args.restore else os.path.join(
            args.workdir, "GAN{0:d}".format(get_run_number(args.workdir) + 1))
        gan.build()
        gan.train()
        return gan.convert_checkpoint_to_model()
    gan = GenerativeAdverserialNetwork(params, None)
    gan.output_dir = args.workdir
    gan.build()
    with np.load(os.path.join(args.workdir, 'export', 'weights.npz'), allow_pickle=False) as z:
        gan.store.load_state_dict({k: z[k] for k in z.files})
    n = args.samples
    Z = np.zeros((n, 1, 1, 512))
    for i in range(n):                                          # the reference's latent walk (gan.py:1110-1112)
        Z[i, 0, 0, :] = np.sin((i / 256.) + np.arange(512) / 64.)
    from ..weightmap import imsave
    rgb = to_rgb(gan.predict(latent=Z).cpu().numpy())
    for i in range(n):
        imsave(os.path.join(args.workdir, 'export_{}.tif'.format(i)), rgb[i])
    return args.workdir


if __name__ == "__main__":
    main()
