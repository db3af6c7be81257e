"""Training step of the U-Net hot path (BASELINE configs 3-4): forward, weighted softmax-CE,
backward, one flat-bucket gradient all-reduce (data parallel) and a fused Adam update.

The reference's U-Net model_fn / loss / optimiser are absent from the tree; the
loss is SURVEY.md A.3, the optimiser is Adam in tf.train.AdamOptimizer form; its default learning
rate and warm-up are this module's DEFAULT_* (the reference's `learning_rate` 0.01, sequitr/utils.py:289,
diverges on this net -- see below).  Parameters and gradients
live in two flat fp32 buffers (parallel.FlatBucket): one all-reduce, one optimiser launch.
"""
import numpy as np
import torch

from . import functional as F
from . import ops
from .networks.unet import UNet2D, UNet2DBf16, unet_variable_shapes, TRAIN
from .parallel import FlatBucket, allreduce_sum_


# Defaults from the round-3 probe on the BASELINE config-3 net (tools/r03_lr_probe.py, profiles/r03_lr_probe.txt,
# HISTORY.md section 8): Adam's first update is +-lr on EVERY weight whatever the gradient's size; at the reference's
# learning_rate 0.01 (sequitr/utils.py:289 -- a NetConfiguration field whose optimiser is absent upstream) that is 40 %
# of a deep-layer weight (sigma 0.024) and the multiplicative bridges carry it to a loss of 1.8e15 on step 2; lr 0.003
# still jumps to 4e5, lr 0.001 to 11, and a 20-step ramp to 7 (f32, step 15).  With lr 0.003 ramped linearly over the
# first 40 steps the loss stayed within 1.03 x its initial value in every run (f32 / bf16, two weight + data seeds) and
# the masks reach IoU ~0.9 with the labels between steps 60 and 100.  Both stay `params` keys.
DEFAULT_LEARNING_RATE = 0.003
DEFAULT_WARMUP_STEPS = 40


class UNetTrainer(object):
    def __init__(self, params, learning_rate=None, beta1=0.9, beta2=0.999, epsilon=1e-8, group=None,
                 net_cls=None, direct_grads=True, warmup_steps=None):
        # params['dtype'] == 'bf16': bf16 activations + bf16 MFMA, fp32 master weights / Adam (configs 3-4)
        if net_cls is None:
            net_cls = UNet2DBf16 if str(params.get('dtype', 'f32')).lower() in ('bf16', 'bfloat16') else UNet2D
        self.net = net_cls(params, TRAIN)
        self.fuse_head_loss = bool(params.get('fuse_head_loss', True))      # A/B switch (tests compare both tapes)
        if learning_rate is None:
            learning_rate = params.get('learning_rate', DEFAULT_LEARNING_RATE)
        self.lr, self.b1, self.b2, self.eps = float(learning_rate), beta1, beta2, epsilon
        # linear learning-rate warm-up over the first `warmup_steps` optimiser steps, walked on the device (HISTORY.md section 8)
        self.warmup_steps = int(params.get('warmup_steps', DEFAULT_WARMUP_STEPS) if warmup_steps is None else warmup_steps)
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
            if direct_grads:
                leaf._sq_grad_sink = leaf.grad                  # ... or the gradient kernel writes it directly
            self.net._vars[name] = leaf
        self.pack_plan = None
        if net_cls is UNet2DBf16 or isinstance(self.net, UNet2DBf16):
            from .ops_bf16 import PackPlan
            self.pack_plan = PackPlan(self.pbucket.flat, {n: (self.net._vars[n], self.pbucket.offsets[n][0])
                                                         for n in self.pbucket.names})
        self.last_loss = None
        # Every workspace this trainer's launches use (split-K partials of the weight gradients, head / loss partials)
        # comes from ITS arena: one buffer, one owner stream, never freed while a captured graph may hold its address
        # (ops.WorkspaceArena; HISTORY.md 4b "the round-2 memory access fault").
        self.arena = ops.WorkspaceArena('UNetTrainer')
        # Adam's step counter lives on the device ({step, lr_t bits}) so a captured step replays correctly.
        self.step_state = torch.zeros(2, dtype=torch.int32, device=dev)
        # The dropout salt is its own device counter: it advances once per forward/backward PASS (inside the captured
        # graph), not once per optimiser step, and starts at a rank-dependent offset -- so the micro-batches of one
        # accumulated step and the ranks of a data-parallel step all draw different masks (config 4's global batch
        # of 128 sees 128 mask sets, not 16: ADVICE r2).
        self.drop_salt = torch.full((1,), (self._rank() * 0x3C6EF35F) & 0x7FFFFFFF, dtype=torch.int32, device=dev)
        self.net._step_dev = self.drop_salt
        self._graphs = None
        # a list here makes every step append a pair of timing events round its gradient all-reduce (bench.py's
        # `allreduce_ms`); None: no events
        self.allreduce_events = None

    def _allreduce(self):
        """ONE collective per optimiser step over the flat gradient bucket; returns the world size"""
        if self.allreduce_events is None:
            return allreduce_sum_(self.gbucket.flat, self.group)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        world = allreduce_sum_(self.gbucket.flat, self.group)
        e1.record()
        self.allreduce_events.append((e0, e1))
        return world

    def load_state_dict(self, weights):
        """strict: every trainable variable of this configuration must be present with its shape; anything else must
        be one of the net's optional non-trainable variables (BN moving statistics)."""
        req, opt = self.net.expected_variables()
        missing = [k for k in self.pbucket.names if k not in weights]
        unexpected = [k for k in weights if k not in req and k not in opt]
        bad = [k for k in weights if k in self.pbucket.shapes and tuple(np.shape(weights[k])) != self.pbucket.shapes[k]]
        if missing or unexpected or bad:
            raise ValueError('UNetTrainer.load_state_dict: model does not fit this configuration -- missing %s; '
                             'unexpected %s; wrong shape %s' % (missing[:6], unexpected[:6], bad[:6]))
        with torch.no_grad():
            for name in self.pbucket.names:
                self.pbucket.view(name).copy_(torch.as_tensor(weights[name]).to(self.pbucket.flat.device))
            for k, v in weights.items():                        # non-trainable state (BN moving statistics)
                if k not in self.pbucket.shapes:
                    self.net._vars[k] = torch.as_tensor(np.ascontiguousarray(v, dtype=np.float32)).to(
                        self.pbucket.flat.device)

    def state_dict(self):
        """Trainable variables (flat bucket) plus the net's non-trainable state (BN moving statistics)."""
        sd = {k: self.pbucket.view(k).detach().cpu().numpy() for k in self.pbucket.names}
        for k, v in self.net._vars.items():
            if k not in sd:
                sd[k] = v.detach().cpu().numpy()
        return sd

    def grads(self):
        return {k: self.gbucket.view(k).detach().cpu().numpy() for k in self.gbucket.names}

    def forward_backward(self, x, onehot, weights):
        """Leaves the (local) gradients in the flat gradient bucket; returns the loss tensor."""
        self.gbucket.flat.zero_()
        with ops.use_arena(self.arena):
            if self.pack_plan is not None:
                self.pack_plan.run()                            # every bf16 filter pack of the step, one launch
            if self.fuse_head_loss and hasattr(self.net, 'build_loss'):
                from . import functional_bf16 as FB
                with FB.deferred_loss():                        # the backward below leaves the loss: no forward head kernel
                    loss = self.net.build_loss(x, onehot, weights)  # bf16 graph: head + loss as one tape entry
            else:
                loss = F.weighted_softmax_cross_entropy(self.net.build(x), onehot, weights)
            if self.pack_plan is not None:                      # the bf16 graph
                from . import ops_bf16 as ob
                with ob.deferred_wgrads():                      # the deep layers' weight gradients: one grouped launch
                    loss.backward()
            else:
                loss.backward()
        self.drop_salt.add_(1)                                  # next pass, next masks (captured with the pass)
        self.last_loss = loss.detach()
        return self.last_loss

    def _rank(self):
        import torch.distributed as dist
        return dist.get_rank(self.group) if dist.is_available() and dist.is_initialized() else 0

    def _world(self):
        import torch.distributed as dist
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def _adam(self, world):
        ops.adam_advance_warmup_dev(self.step_state, self.lr, self.b1, self.b2, self.warmup_steps)
        ops.adam_apply_dev(self.pbucket.flat, self.gbucket.flat, self.m, self.v, self.b1, self.b2, self.eps,
                           self.step_state, grad_scale=1.0 / world)

    def step(self, x, onehot, weights):
        """One optimiser step on this rank's shard of the global batch."""
        if self._graphs is not None:
            return self._step_graphed(x, onehot, weights)
        loss = self.forward_backward(x, onehot, weights)
        world = self._allreduce()                               # ONE collective per step
        self.step_count += 1
        self._adam(world)
        return loss

    def step_accumulate(self, micro_batches):
        """One optimiser step over several equally sized micro-batches [(x, onehot, weights), ...]: this rank's share of
        the global batch when it is larger than one launch batch (BASELINE config 4's global 128 tiles on fewer than 8
        GPUs).  Each micro-batch runs forward/backward (graph replay when captured); its gradient bucket is added,
        scaled by 1/k, into an accumulator (sq_axpy_f32), so after the last one the bucket holds the mean gradient
        of the rank's share; then the usual ONE all-reduce and ONE Adam launch.  Returns the mean loss.  The dropout
        salt advances with every pass (and differs by rank), so every micro-batch draws its own masks."""
        k = len(micro_batches)
        if k == 1:
            return self.step(*micro_batches[0])
        if getattr(self, "_acc", None) is None:
            self._acc = torch.zeros_like(self.gbucket.flat)
        self._acc.zero_()
        total = None
        for x, onehot, weights in micro_batches:
            if self._graphs is not None:
                g_fb, _, sx, so, sw, sloss = self._graphs
                if x.shape != sx.shape:
                    raise ValueError("UNetTrainer: captured for batch shape %s, got %s" % (tuple(sx.shape), tuple(x.shape)))
                if x is not sx:
                    sx.copy_(x), so.copy_(onehot), sw.copy_(weights)
                self.arena.hand_over()                          # the replay's launches use the arena on THIS stream
                g_fb.replay()
                loss = sloss
            else:
                loss = self.forward_backward(x, onehot, weights)
            ops.axpy_(self._acc, self.gbucket.flat, 1.0 / k)
            total = loss.clone() if total is None else total + loss
        self.gbucket.flat.copy_(self._acc)
        world = self._allreduce()                               # still ONE collective per optimiser step
        self.step_count += 1
        if self._graphs is not None:
            self._graphs[1].replay()
        else:
            self._adam(world)
        self.last_loss = total / k
        return self.last_loss

    # ---- hipGraph replay: the step is ~200 short launches, host-bound when issued from Python ----------
    def capture(self, x, onehot, weights, warmup=2):
        """Capture the step for inputs of this shape as two hipGraphs: (zero grads, forward, loss,
        backward) and (Adam).  The gradient all-reduce stays between them, outside any graph, so the
        RCCL call is an ordinary stream operation.  `warmup` eager steps run first on the capture
        stream (they ARE optimiser steps).  Later step() calls copy their inputs into the static
        buffers and replay."""
        if self._graphs is not None:
            raise RuntimeError("UNetTrainer.capture: already captured")
        sx, so, sw = x.clone(), onehot.clone(), weights.clone()
        side = torch.cuda.Stream(device=sx.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, int(warmup))):
                self.step(sx, so, sw)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.arena.hand_over()                                  # warm-up stream -> the stream the replays will run on
        world = self._world()
        warm_loss = self.last_loss.clone()                      # the last warm-up step's loss (capturing runs nothing)
        g_fb, g_opt = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_fb):
            sloss = self.forward_backward(sx, so, sw)
        with torch.cuda.graph(g_opt, pool=g_fb.pool()):
            self._adam(world)
        self._graphs = (g_fb, g_opt, sx, so, sw, sloss)
        self.last_loss = warm_loss
        return self

    def _step_graphed(self, x, onehot, weights):
        g_fb, g_opt, sx, so, sw, sloss = self._graphs
        if x.shape != sx.shape:
            raise ValueError("UNetTrainer: captured for batch shape %s, got %s" % (tuple(sx.shape), tuple(x.shape)))
        if x is not sx:
            sx.copy_(x), so.copy_(onehot), sw.copy_(weights)
        self.arena.hand_over()
        g_fb.replay()
        self._allreduce()
        self.step_count += 1
        g_opt.replay()
        self.last_loss = sloss
        return sloss

    @property
    def static_inputs(self):
        """(x, onehot, weights) buffers a loader may fill in place to skip the copy in step()."""
        return None if self._graphs is None else self._graphs[2:5]
