"""Data parallelism over the 8 GPUs of one MI355X node (new functionality --
the reference has no collective of any kind).

One process per GPU; tiles shard along the batch axis.  Training exchanges gradients with
ONE all-reduce per optimiser step over ONE flat, contiguous bucket: the U-Net's 1.74 M
parameters are 6.98 MB, where xGMI collectives are latency- not bandwidth-bound, so splitting
the bucket only multiplies the latency.  `torch.distributed` backend "nccl" is RCCL on ROCm;
"gloo" is used for the CPU tests of this logic.  Inference needs no collective at all.
"""
import torch


class FlatBucket(object):
    """Packs named tensors into one contiguous buffer; exposes them as views."""

    def __init__(self, named_shapes, device, dtype=torch.float32):
        self.names, self.offsets, self.shapes = [], {}, {}
        total = 0
        for name, shape in named_shapes:
            n = 1
            for d in shape:
                n *= int(d)
            self.names.append(name)
            self.offsets[name] = (total, n)
            self.shapes[name] = tuple(int(d) for d in shape)
            total += (n + 3) // 4 * 4                     # keep every view 16-B aligned
        self.numel = total
        self.flat = torch.zeros(total, dtype=dtype, device=device)

    def view(self, name):
        o, n = self.offsets[name]
        return self.flat[o:o + n].view(self.shapes[name])

    def views(self):
        return {k: self.view(k) for k in self.names}


def shard_range(n_items, rank, world):
    """Contiguous [begin, end) range of items for `rank`; sizes differ by at most one."""
    base, rem = divmod(int(n_items), int(world))
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def epoch_schedule(n_items, batch, world):
    """Data-parallel epoch plan with the SAME number of full-batch steps on every rank (each step is one gradient
    all-reduce: ranks that disagree on the count dead-lock RCCL).  steps = (n_items // world) // batch; one seeded
    permutation of the whole stack per epoch is cut into world x steps x batch indices and the remainder of the
    epoch is dropped (a different remainder every epoch).  Returns (order(epoch, rank) -> steps*batch item indices,
    steps); `order.batch` is the batch size actually used and `order.dropped` the tiles left out per epoch, for the
    caller to log.  A single rank with fewer than `batch` tiles trains on one short batch per epoch (no collective can
    dead-lock and Adam's gradient scale is 1); with world > 1 a rank without one full batch raises -- a short batch
    would be mis-weighted by the 1/world gradient scale."""
    n_items, batch, world = int(n_items), int(batch), int(world)
    if world == 1 and 0 < n_items < batch:
        batch = n_items
    steps = (n_items // world) // batch if batch > 0 else 0
    if steps < 1:
        raise ValueError("data-parallel training needs at least batch_size x world = %d x %d tiles, got %d"
                         % (batch, world, n_items))
    per_rank = steps * batch

    def order(epoch, rank):
        import numpy as np
        perm = np.random.default_rng(int(epoch)).permutation(n_items)
        return perm[rank * per_rank:(rank + 1) * per_rank]
    order.batch = batch
    order.dropped = n_items - per_rank * world
    return order, steps


def allreduce_sum_(flat, group=None):
    """One in-place SUM all-reduce of the flat gradient bucket (no-op without a process group)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        return dist.get_world_size(group)
    return 1
