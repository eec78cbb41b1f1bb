"""
Single-node data parallelism: one process per GPU, the minibatch sharded by sample, ONE all-reduce of a flat fp32
gradient bucket per step over RCCL/xGMI (``torch.distributed`` backend ``nccl`` is RCCL on ROCm; ``gloo`` on CPU for
the tests).  The reference is single-device (main.py:40-43), so this is new; every transform on the hot path is
per-sample independent, the only cross-sample couplings are batch statistics (policy below).

Statistic policy: per-replica batch statistics (standard DDP behaviour).  ``sync_buffers()``
averages the running statistics across ranks for evaluation / checkpointing.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torchrun-style); returns (rank, world,
    local_rank).  No-op for single-process runs."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def shard(batch, rank, world):
    """contiguous per-rank shard of a global batch: y[rank*B/W : (rank+1)*B/W]."""
    B = batch.shape[0]
    if B % world != 0:
        raise ValueError('global batch %d is not divisible by world size %d' % (B, world))
    per = B // world
    return batch[rank * per:(rank + 1) * per]


class GradBucket:
    """All trainable parameters' gradients live in ONE flat fp32 buffer (``p.grad`` are views into it), so the
    gradient exchange is a single collective and zeroing the gradients is a single memset.  With
    ``flatten_params=True`` the parameters themselves are re-homed into one flat buffer too (``p.data`` become views;
    values preserved), which lets the optimizer step be a single fused kernel (FlatAdam).

    Frozen parameters (``requires_grad=False``: P, I, masks, sign_s, int32 pivots of the invertible 1x1 convolution)
    carry no gradient and are skipped.

    Every bucketed parameter is tagged ``_nf_direct_grad = True``: the hand-written backward kernels may then
    accumulate straight into ``p.grad`` (the same ``+=`` autograd's AccumulateGrad would perform) instead of
    returning a temporary that costs one extra add launch per parameter."""

    def __init__(self, params, process_group=None, flatten_params=False):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError('no trainable parameters')
        dev, dt = self.params[0].device, self.params[0].dtype
        self.numel = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(self.numel, dtype=dt, device=dev)
        self.flat_params = torch.empty(self.numel, dtype=dt, device=dev) if flatten_params else None
        self.group = process_group
        o = 0
        self.views = []                                   # the bucket's view for every parameter, in order
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[o:o + n].view_as(p)
            self.views.append(p.grad)
            if flatten_params:
                with torch.no_grad():
                    self.flat_params[o:o + n].copy_(p.data.reshape(-1))
                    p.data = self.flat_params[o:o + n].view_as(p)
            p._nf_direct_grad = True
            o += n

    @property
    def world(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def zero_(self):
        self.flat.zero_()

    def nbytes(self):
        return self.flat.numel() * self.flat.element_size()

    def all_reduce_mean_(self):
        """sum over ranks, then 1/world (the loss is a per-shard mean, main.py:85)."""
        if self.world > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.mul_(1.0 / self.world)
        return self.flat


def broadcast_parameters(module, src=0, group=None):
    """make every replica start from rank ``src``'s parameters and buffers."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src, group=group)


def sync_buffers(module, group=None):
    """average floating-point buffers (running statistics) across ranks."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    w = dist.get_world_size(group)
    with torch.no_grad():
        for b in module.buffers():
            if b.is_floating_point():
                dist.all_reduce(b.data, op=dist.ReduceOp.SUM, group=group)
                b.data.mul_(1.0 / w)
