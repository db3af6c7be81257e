"""Training step of the U-Net hot path (BASELINE configs 3-4): forward, weighted softmax-CE,
backward, one flat-bucket gradient all-reduce (data parallel) and a fused Adam update.

The reference's U-Net model_fn / loss / optimiser are absent from the tree; the
loss is SURVEY.md A.3, the optimiser is Adam in tf.train.AdamOptimizer form with the
reference's `learning_rate` default 0.01 (sequitr/utils.py:289).  Parameters and gradients
live in two flat fp32 buffers (parallel.FlatBucket): one all-reduce, one optimiser launch.
"""
import torch

from . import functional as F
from . import ops
from .networks.unet import UNet2D, UNet2DBf16, unet_variable_shapes, TRAIN
from .parallel import FlatBucket, allreduce_sum_


class UNetTrainer(object):
    def __init__(self, params, learning_rate=0.01, beta1=0.9, beta2=0.999, epsilon=1e-8, group=None,
                 net_cls=None):
        # params['dtype'] == 'bf16': bf16 activations + bf16 MFMA, fp32 master weights / Adam (configs 3-4)
        if net_cls is None:
            net_cls = UNet2DBf16 if str(params.get('dtype', 'f32')).lower() in ('bf16', 'bfloat16') else UNet2D
        self.net = net_cls(params, TRAIN)
        self.lr, self.b1, self.b2, self.eps = learning_rate, beta1, beta2, epsilon
        self.group = group
        self.step_count = 0
        dev = self.net.device
        shapes = unet_variable_shapes(params)
        self.pbucket = FlatBucket(shapes, dev)
        self.gbucket = FlatBucket(shapes, dev)
        self.m = torch.zeros_like(self.pbucket.flat)
        self.v = torch.zeros_like(self.pbucket.flat)
        self.net.initialize()                                   # seeded host init -> device
        for name in self.pbucket.names:
            view = self.pbucket.view(name)
            view.copy_(self.net._vars[name])
            leaf = view.detach().requires_grad_(True)           # shares the flat storage
            leaf.grad = self.gbucket.view(name)                 # autograd accumulates in place
            self.net._vars[name] = leaf
        self.last_loss = None

    def load_state_dict(self, weights):
        with torch.no_grad():
            for name in self.pbucket.names:
                self.pbucket.view(name).copy_(torch.as_tensor(weights[name]).to(self.pbucket.flat.device))

    def state_dict(self):
        return {k: self.pbucket.view(k).detach().cpu().numpy() for k in self.pbucket.names}

    def grads(self):
        return {k: self.gbucket.view(k).detach().cpu().numpy() for k in self.gbucket.names}

    def forward_backward(self, x, onehot, weights):
        """Leaves the (local) gradients in the flat gradient bucket; returns the loss tensor."""
        self.gbucket.flat.zero_()
        logits = self.net.build(x)
        loss = F.weighted_softmax_cross_entropy(logits, onehot, weights)
        loss.backward()
        self.last_loss = loss.detach()
        return self.last_loss

    def step(self, x, onehot, weights):
        """One optimiser step on this rank's shard of the global batch."""
        loss = self.forward_backward(x, onehot, weights)
        world = allreduce_sum_(self.gbucket.flat, self.group)   # ONE collective per step
        self.step_count += 1
        ops.adam_step(self.pbucket.flat, self.gbucket.flat, self.m, self.v, self.lr, self.b1, self.b2, self.eps,
                      self.step_count, grad_scale=1.0 / world)
        return loss
