"""
Single-node data parallelism: one process per GPU, the minibatch sharded by sample, ONE all-reduce of a flat fp32
gradient bucket per step over RCCL/xGMI (``torch.distributed`` backend ``nccl`` is RCCL on ROCm; ``gloo`` on CPU for
the tests).  The reference is single-device (main.py:40-43), so this is new; every transform on the hot path is
per-sample independent, the only cross-sample couplings are batch statistics (policy below).

Statistic policy: per-replica batch statistics by default (standard DDP behaviour; throughput runs);
``sync_buffers()`` averages the running statistics across ranks for evaluation / checkpointing.  The PARITY mode
(``FlowTrainer(sync_stats=True)`` / ``with sync_statistics():``) all-reduces every batch statistic on the path -- flow BatchNorm,
the BatchNorm1d / 2d layers of the conditioners (forward moments and the two sums of their backward), the data-dependent
ActNorm initialisation -- so that W-way data parallelism reproduces the single-process result on the global batch.
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
    # (NF_DP_FORCE_COLLECTIVE=1: a ONE-rank group too -- bench.py / the trainer then run the N > 1 control flow on real RCCL on one GPU)
    if (world > 1 or os.environ.get('NF_DP_FORCE_COLLECTIVE', '0') == '1') and not dist.is_initialized():
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

    @property
    def collective(self):
        """does a step issue the gradient all-reduce?  World size > 1 -- or a ONE-rank group with NF_DP_FORCE_COLLECTIVE=1, which
        sends the trainer down the N > 1 control flow (eager or captured RCCL all-reduce between backward and Adam, start-up
        broadcast) on a single GPU: how the 1-GPU box exercises the RCCL path (tests/test_gpu_rccl.py)."""
        return self.world > 1 or (dist.is_initialized() and os.environ.get('NF_DP_FORCE_COLLECTIVE', '0') == '1')

    def zero_(self):
        if self.flat.is_cuda and self.flat.dtype == torch.float32:
            from . import _native as N
            N.call('nf_zero_fill', self.flat.data_ptr(), self.flat.numel(), N.stream())
        else:
            self.flat.zero_()

    def nbytes(self):
        return self.flat.numel() * self.flat.element_size()

    def all_reduce_mean_(self):
        """sum over ranks, then 1/world (the loss is a per-shard mean, main.py:85)."""
        if self.collective:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            if self.world > 1:
                self.flat.mul_(1.0 / self.world)
        return self.flat


def broadcast_parameters(module, src=0, group=None, force=False):
    """make every replica start from rank ``src``'s parameters and buffers (``force``: also in a one-rank group)."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return 0
    return broadcast_coalesced([t.data for t in list(module.parameters()) + list(module.buffers())], src=src, group=group)


def broadcast_coalesced(tensors, src=0, group=None):
    """one collective per DTYPE instead of one per tensor (the CIFAR Glow holds ~5 000 parameter and buffer tensors: fp32 weights
    and running statistics, int32 pivots, int64 num_batches_tracked): the tensors of a dtype are packed into one flat buffer,
    broadcast, and copied back.  Tensors that already are consecutive views of one flat buffer (GradBucket(flatten_params=True))
    are sent in place, without the pack / unpack copies.  Returns the number of collectives issued."""
    by_dtype = {}
    for t in tensors:
        if t.numel():
            by_dtype.setdefault(t.dtype, []).append(t)
    n = 0
    with torch.no_grad():
        for dt in sorted(by_dtype, key=str):            # the same order on every rank
            ts = by_dtype[dt]
            runs, cur = [], [ts[0]]
            for t in ts[1:]:                            # maximal runs of contiguous tensors that sit back to back in memory
                a = cur[-1]
                if (t.is_contiguous() and a.is_contiguous() and t.device == a.device
                        and t.untyped_storage().data_ptr() == a.untyped_storage().data_ptr()
                        and t.data_ptr() == a.data_ptr() + a.numel() * a.element_size()):
                    cur.append(t)
                else:
                    runs.append(cur)
                    cur = [t]
            runs.append(cur)
            loose = []
            for r in runs:
                if len(r) >= 64:                        # a flat parameter buffer: broadcast the storage range itself
                    total = sum(t.numel() for t in r)
                    flat = r[0].new_empty(0).set_(r[0].untyped_storage(), r[0].storage_offset(), (total, ), (1, ))
                    dist.broadcast(flat, src=src, group=group)
                    n += 1
                else:
                    loose += r
            if loose:
                flat = torch.cat([t.reshape(-1) for t in loose])
                dist.broadcast(flat, src=src, group=group)
                n += 1
                o = 0
                for t in loose:
                    t.copy_(flat[o:o + t.numel()].view_as(t))
                    o += t.numel()
    return n


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


# ---- synchronised batch statistics (parity mode) -----------------------------------------------------------------------------------
_SYNC = {'on': False, 'group': None}


class sync_statistics:
    """context: every batch statistic of the flow is computed over the GLOBAL batch (all ranks of ``group``).  The layers then
    take their layer-by-layer launch paths (the fused persistent kernels compute their statistics in-kernel, per replica) and the
    conditioners their module paths with ``sync_batch_norm``; meant for the parity run, not for throughput.  Works with one
    process too (the collectives are no-ops), which is how the GPU test checks it against the fused path."""

    def __init__(self, group=None):
        self.group = group

    def __enter__(self):
        self._old = dict(_SYNC)
        _SYNC['on'], _SYNC['group'] = True, self.group
        return self

    def __exit__(self, *exc):
        _SYNC.update(self._old)
        return False


def sync_stats_active():
    return _SYNC['on']


def _all_reduce_sum(t, group):
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def global_moments(x, group=None):
    """(mean, biased variance, n) per channel (dim 1) over the batch and pixel dims of ALL ranks: two passes, two all-reduces --
    the squared deviations are taken about the GLOBAL mean, so nothing is computed as E[x^2] - E[x]^2."""
    group = _SYNC['group'] if group is None else group
    dims = [0] + list(range(2, x.dim()))
    n = torch.tensor([float(x.numel() // x.shape[1])], dtype=x.dtype, device=x.device)
    s = x.sum(dim=dims)
    packed = torch.cat([s, n])
    _all_reduce_sum(packed, group)
    n_g = packed[-1]
    mean = packed[:-1] / n_g
    shape = [1, -1] + [1] * (x.dim() - 2)
    m2 = ((x - mean.view(shape)) ** 2).sum(dim=dims)
    _all_reduce_sum(m2, group)
    return mean, m2 / n_g, n_g


class _SyncBatchNorm(torch.autograd.Function):
    """training-mode nn.BatchNorm1d / 2d with statistics over all ranks; the gradient flows through the statistics (the two batch
    sums of the backward are all-reduced), exactly what the single-process layer computes on the concatenated batch."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, group):
        mean, var, n_g = global_moments(x, group)
        invstd = torch.rsqrt(var + eps)
        shape = [1, -1] + [1] * (x.dim() - 2)
        xhat = (x - mean.view(shape)) * invstd.view(shape)
        ctx.save_for_backward(xhat, gamma, invstd, n_g)
        ctx.group = group
        ctx.mark_non_differentiable(mean, var, n_g)
        return xhat * gamma.view(shape) + beta.view(shape), mean, var, n_g

    @staticmethod
    def backward(ctx, g, *_unused):
        xhat, gamma, invstd, n_g = ctx.saved_tensors
        dims = [0] + list(range(2, g.dim()))
        shape = [1, -1] + [1] * (g.dim() - 2)
        sg, sgx = g.sum(dim=dims), (g * xhat).sum(dim=dims)
        packed = torch.cat([sg, sgx])
        _all_reduce_sum(packed, ctx.group)
        C = sg.numel()
        mg, mgx = packed[:C] / n_g, packed[C:] / n_g
        g_x = (gamma * invstd).view(shape) * (g - mg.view(shape) - xhat * mgx.view(shape))
        return g_x, sgx, sg, None, None


def sync_batch_norm(bn, x):
    """``bn``: an nn.BatchNorm1d / 2d module in training mode; same result, running-statistics and num_batches_tracked bookkeeping
    as ``bn(x)`` would give on the concatenation of all ranks' batches."""
    group = _SYNC['group']
    y, mean, var, n_g = _SyncBatchNorm.apply(x, bn.weight, bn.bias, bn.eps, group)   # the forward's own moments: 2 all-reduces, not 4
    with torch.no_grad():
        unb = var * (n_g / torch.clamp(n_g - 1.0, min=1.0))
        bn.num_batches_tracked += 1
        # momentum=None is nn.BatchNorm's cumulative moving average: factor 1 / num_batches_tracked
        m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
        bn.running_mean.mul_(1.0 - m).add_(mean * m)
        bn.running_var.mul_(1.0 - m).add_(unb * m)
    return y
