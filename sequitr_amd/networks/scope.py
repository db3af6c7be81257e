"""A small stand-in for TensorFlow-1 variable scopes, so the GAN code of
sequitr/networks/gan.py (module-level functions that call ``tf.variable_scope`` /
``tf.get_variable`` with ``reuse=tf.AUTO_REUSE``) keeps its shape and its variable names.

A ``VariableStore`` owns the device tensors (name -> tensor, creation order preserved); the
active store and the scope stack are module state, exactly as TF's default graph is.
"""
import contextlib

import numpy as np
import torch

_store_stack = []
_scopes = []


class VariableStore(object):
    def __init__(self, device, seed=0, trainable=True):
        self.device = torch.device(device)
        self.rng = np.random.default_rng(seed)
        self.trainable = trainable
        self.vars = {}                                        # insertion ordered
        self.flat, self.offsets = None, {}                    # set by flatten()

    def __enter__(self):
        _store_stack.append(self)
        return self

    def __exit__(self, *a):
        _store_stack.pop()

    def get(self, name, shape, init):
        v = self.vars.get(name)
        if v is None:
            arr = np.ascontiguousarray(init(tuple(int(s) for s in shape), self.rng), dtype=np.float32)
            v = torch.from_numpy(arr).to(self.device)
            if self.trainable:
                v.requires_grad_(True)
            self.vars[name] = v
        elif tuple(v.shape) != tuple(int(s) for s in shape):
            raise ValueError('variable %s exists with shape %s, wanted %s' % (name, tuple(v.shape), tuple(shape)))
        return v

    def flatten(self):
        """Move every variable into ONE contiguous fp32 buffer (each a 16-byte aligned view, a leaf again): what the
        one-launch filter packing (ops.GanPackPlan) and any flat collective need.  Values, names and order are kept;
        call after the variables exist (GenerativeAdverserialNetwork.build) and before anything caches their pointers."""
        offs, total = {}, 0
        for k, v in self.vars.items():
            offs[k] = total
            total += (v.numel() + 3) // 4 * 4
        flat = torch.zeros(max(total, 4), dtype=torch.float32, device=self.device)
        with torch.no_grad():
            for k, v in list(self.vars.items()):
                view = flat[offs[k]:offs[k] + v.numel()].view(v.shape)
                view.copy_(v)
                leaf = view.detach()
                if self.trainable:
                    leaf.requires_grad_(True)
                self.vars[k] = leaf
        self.flat, self.offsets = flat, offs
        return flat

    def trainable_variables(self, scope=''):
        """[(name, tensor)] whose name starts with ``scope`` (tf.trainable_variables(scope=...)):
        TF matches by regex-prefix on the name, so 'GAN/discriminator/layer_1' also matches
        'GAN/discriminator/layer_10' -- reproduced here by plain prefix matching."""
        return [(k, v) for k, v in self.vars.items() if k.startswith(scope)]

    def state_dict(self):
        return {k: v.detach().cpu().numpy() for k, v in self.vars.items()}

    def load_state_dict(self, weights):
        with torch.no_grad():
            for k, arr in weights.items():
                t = torch.as_tensor(np.ascontiguousarray(arr, dtype=np.float32)).to(self.device)
                if k in self.vars:
                    self.vars[k].copy_(t)
                else:
                    self.vars[k] = t.requires_grad_(True) if self.trainable else t
        from .. import ops                                      # the weights changed in place: cached filter packs /
        ops.invalidate_packs()                                  # transforms of the old values must not be served again


def current_store():
    if not _store_stack:
        raise RuntimeError('no active VariableStore: use `with VariableStore(device):`')
    return _store_stack[-1]


@contextlib.contextmanager
def variable_scope(name, **_ignored):
    """``with variable_scope('generator'):`` -- reuse / auxiliary_name_scope kwargs are accepted
    and ignored (every scope behaves as tf.AUTO_REUSE)."""
    _scopes.append(name)
    try:
        yield
    finally:
        _scopes.pop()


def scope_name():
    return '/'.join(_scopes)


def get_variable(name, shape, initializer):
    full = (scope_name() + '/' + name) if _scopes else name
    return current_store().get(full, shape, initializer)


# initialisers: callables (shape, rng) -> ndarray
def random_normal(shape, rng):                      # tf.initializers.random_normal: mean 0, stddev 1
    return rng.standard_normal(shape)


def zeros(shape, rng):
    return np.zeros(shape)


def glorot_uniform(shape, rng):                     # tf.layers.dense default kernel initialiser
    fan_in, fan_out = shape[-2], shape[-1]
    limit = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-limit, limit, shape)
