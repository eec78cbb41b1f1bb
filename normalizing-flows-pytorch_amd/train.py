"""
Training / evaluation harness: the counterpart of ``Model`` in the reference's main.py:38-124 restricted to what the
hot path needs -- ``train_on_batch`` (main.py:78-92), ``sample_y`` (:109-116), ``log_py`` (:121-124) -- plus what the
reference does not have: data parallelism (dist.GradBucket) and whole-step hipGraph capture.

The prior is N(0, I): ``MultivariateNormal(0, I).log_prob(z) = -0.5 |z|^2 - 0.5 D log(2 pi)`` (main.py:49-51); the
reference materialises a D x D covariance (3072 x 3072 for CIFAR), here it is a reduction.
"""
import math

import os

import torch

from . import dist as nfdist


def standard_normal_logprob(z):
    zf = z.reshape(z.shape[0], -1)
    return -0.5 * (zf * zf).sum(dim=1) - 0.5 * zf.shape[1] * math.log(2.0 * math.pi)


def nll_loss(z, log_det_jacobian):
    """loss = -mean(log N(z; 0, I) + log_det_jacobian)   (main.py:85); one HIP reduction on the GPU."""
    if z.is_cuda and z.dtype == torch.float32 and z.shape[0] > 0:
        from . import functional as NF
        return NF.nll_loss(z, log_det_jacobian)
    return -1.0 * torch.mean(standard_normal_logprob(z) + log_det_jacobian)


def bits_per_dim(loss, dims):
    D = 1
    for d in dims:
        D *= int(d)
    return float(loss) / (D * math.log(2.0))


class FlatAdam:
    """torch.optim.Adam semantics (the reference's optimizer, main.py:56-64) as ONE fused HIP launch over the flat
    parameter / gradient buffers of a GradBucket.  ``lr`` and the step counter live on the device, so a captured
    hipGraph keeps working when a scheduler changes the rate (``set_lr``)."""

    def __init__(self, bucket, lr=1.0e-4, betas=(0.9, 0.999), eps=1.0e-8, weight_decay=0.0):
        if bucket.flat_params is None:
            raise ValueError('FlatAdam needs GradBucket(..., flatten_params=True)')
        from . import _native as N
        self._N = N
        self.bucket = bucket
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        dev = bucket.flat.device
        self.exp_avg = torch.zeros_like(bucket.flat)
        self.exp_avg_sq = torch.zeros_like(bucket.flat)
        self.step_count = torch.zeros(1, dtype=torch.int32, device=dev)
        self.lr = torch.full((1, ), float(lr), dtype=torch.float32, device=dev)

    def set_lr(self, lr):
        self.lr.fill_(float(lr))

    def step(self):
        N, b = self._N, self.bucket
        # the parameters must still live in the flat buffer (net.to() / .cpu() after construction re-homes p.data silently)
        p0, p1 = b.params[0], b.params[-1]
        if (p0.data_ptr() != b.flat_params.data_ptr()
                or p1.data_ptr() + p1.numel() * 4 != b.flat_params.data_ptr() + b.numel * 4):
            raise RuntimeError('FlatAdam: the parameters no longer live in the GradBucket\'s flat buffer (was the model moved '
                               'with .to() / .cpu() after the trainer was built?); build a new FlowTrainer')
        N.call('nf_adam_step', N.ptr(b.flat_params), N.ptr(b.flat), N.ptr(self.exp_avg), N.ptr(self.exp_avg_sq),
               N.ptr(self.step_count), N.ptr(self.lr), self.betas[0], self.betas[1], self.eps, self.weight_decay, 1.0,
               b.numel, N.stream())

    def state_dict(self):
        return {'exp_avg': self.exp_avg, 'exp_avg_sq': self.exp_avg_sq, 'step': self.step_count, 'lr': self.lr}

    def load_state_dict(self, sd):
        for k, t in self.state_dict().items():
            t.copy_(sd[k])


class _HipStepGraph:
    """one captured piece of a train step: a torch.cuda.CUDAGraph (= hipGraph on ROCm).  ``FlowTrainer(graph_factory=...)`` takes any
    class with the same two methods; the world-size-2 gloo test passes one that re-runs the closure, so that the capture / replay /
    all-reduce-between-two-graphs control flow of the N > 1 path is exercised without a GPU."""

    def __init__(self):
        self.g = torch.cuda.CUDAGraph()

    def capture(self, fn):
        # thread_local: a collective watchdog / other host thread touching the runtime must not invalidate the capture
        with torch.cuda.graph(self.g, capture_error_mode='thread_local'):
            return fn()

    def replay(self):
        self.g.replay()


class FlowTrainer:
    """Adam (lr 1e-4, betas (0.9, 0.999): configs/default.yaml:13-20) on the NLL, gradients in one flat bucket.

    ``graph=True`` captures zero-grad + forward + backward into one hipGraph and the optimizer step into a second one
    (the gradient all-reduce runs between the two replays; a single process captures both into ONE graph), after
    ``warmup`` eager steps that also perform the data-dependent ActNorm initialisation.  The batch shape is then fixed.
    The call that captures also takes one extra eager step on its batch (allocator warm-up on the capture stream)."""

    TRAIN_MODE_RECHECK = 256        # calls of train_on_batch between two walks over the submodules' training flags

    def __init__(self, net, lr=1.0e-4, betas=(0.9, 0.999), weight_decay=0.0, graph=False, warmup=3, process_group=None,
                 fused_adam=True, sampler=None, sync_stats=False, graph_factory=None, one_graph=None):
        self.net = net
        self.one_graph = (os.environ.get('NF_DP_ONE_GRAPH', '0') == '1') if one_graph is None else bool(one_graph)
        self._g_whole = False
        # parity mode: every batch statistic over the GLOBAL batch -- W-way data parallelism then reproduces
        # the single-process result on the concatenated batch.  Layer-by-layer launches with collectives in between: no hipGraph.
        self.sync_stats = bool(sync_stats)
        if self.sync_stats:
            graph = False
        if graph and any(getattr(m_, 'masks_redrawn_per_call', False) for m_ in net.modules()):
            # MADE re-draws its masks from the host's np.random on every call (flows/maf.py:50,72); for D > 2 the draw varies, and a
            # replayed graph would repeat the masks of the captured step: such models train on eager launches
            graph = False
        self.sampler = sampler      # data.DeviceSampler: train_on_batch() without a batch draws one on the device, inside the graph
        on_gpu = next(net.parameters()).is_cuda
        fused_adam = bool(fused_adam) and on_gpu
        self.bucket = nfdist.GradBucket(net.parameters(), process_group, flatten_params=fused_adam)
        self.graph = bool(graph)
        self._graph_factory = graph_factory or _HipStepGraph
        if fused_adam:
            self.optim = FlatAdam(self.bucket, lr=lr, betas=betas, weight_decay=weight_decay)
        else:
            self.optim = torch.optim.Adam(self.bucket.params, lr=lr, betas=betas, weight_decay=weight_decay,
                                          capturable=self.graph and on_gpu, foreach=True)
        self.warmup = warmup
        self._eager_steps = 0
        self._replicas_synced = False
        self._g_fb = self._g_opt = None
        self._static_y = self._static_z = self._static_loss = None
        # parameters whose gradient arrives from framework autograd (convolutions, BatchNorm2d ...) rather than from a kernel
        # writing into the bucket: found in the first eager step, afterwards their AccumulateGrad adds (one launch per
        # parameter) are replaced by one gather launch per 128 tensors
        self._gather = on_gpu
        self._indirect = None
        self._calls_since_mode_walk = self.TRAIN_MODE_RECHECK          # the first call walks the tree
        self._captured_det = None

    # -- one step, eager --------------------------------------------------------------------------------------------
    def _forward_backward(self, y):
        if y is None:                                   # on-device data: the draw is part of the step (and of its hipGraph)
            y = self.sampler.next()
        if self.sync_stats:
            with nfdist.sync_statistics(self.bucket.group):
                return self._run_step(y.device, lambda: self._forward_loss(y))
        return self._run_step(y.device, lambda: self._forward_loss(y))

    def _forward_loss(self, y):
        z, ld = self.net(y)
        return z, nll_loss(z, ld)

    def _run_step(self, device, forward_loss):
        """zero the bucket, open the step's scratch arena and the deferred-work queues, run ``forward_loss() -> (z, loss)`` and
        ``loss.backward()``, flush.  ``_forward_backward`` is this with the whole model; tests run slices of a model through it."""
        from .workspace import ARENA
        self.bucket.zero_()
        hooks, seen = [], set()
        if self._gather and self._indirect is None:     # first step: watch which parameters autograd itself produces
            for i, p in enumerate(self.bucket.params):
                # (a tensor hook also fires, with None, when a hand-written backward wrote into the bucket and returned nothing)
                hooks.append(p.register_hook(lambda g, i=i: seen.add(i) if g is not None else None))
        elif self._indirect:
            for i in self._indirect:
                self.bucket.params[i].grad = None       # AccumulateGrad then keeps the incoming tensor: no add launch
        from .fused import FPP_DEFER
        from .fused_conv import CONV_DEFER
        ARENA.begin(device)                             # one memset for every zero-initialised accumulator of the step
        FPP_DEFER.begin()                               # the Flow++ steps' slab finalizes: all of them in one go after backward
        from .fused_flowpp_img import FPP_IMG_DEFER
        FPP_IMG_DEFER.begin()                           # the image Flow++ conditioners' weight gradients: sixteen convolutions per launch
        CONV_DEFER.begin()                              # the image conditioners' weight-gradient passes: sixteen layers per launch
        try:
            z, loss = forward_loss()
            loss.backward()
        finally:
            FPP_DEFER.flush()
            FPP_IMG_DEFER.flush()
            CONV_DEFER.flush()
            ARENA.end()
            for h in hooks:
                h.remove()
        if hooks:
            self._indirect = sorted(seen)
        elif self._indirect:
            self._gather_indirect()
        from .workspace import _capturing, in_arena
        if not _capturing() and isinstance(loss, torch.Tensor) and in_arena(loss):
            loss = loss.clone()                         # (the arena is zeroed again when the next step begins; inside a capture the loss is
        return z, loss                                  #  the graph's static output, rewritten by every replay)

    def _gather_indirect(self):
        """copy the framework-produced gradients into their bucket slots (nf_multi_copy) and re-attach the views."""
        import ctypes
        from . import _native as N
        from .fused import CopyDesc
        descs = []
        for i in self._indirect:
            p, view = self.bucket.params[i], self.bucket.views[i]
            g = p.grad
            if g is not None and g.data_ptr() != view.data_ptr():
                g = g.contiguous()
                descs.append((g, view))
            p.grad = view
        for k0 in range(0, len(descs), 128):
            chunk = descs[k0:k0 + 128]
            arr = (CopyDesc * len(chunk))(*[CopyDesc(g.data_ptr(), v.data_ptr(), g.numel()) for g, v in chunk])
            N.call('nf_multi_copy', ctypes.addressof(arr), len(chunk), N.stream())

    def _capture(self, y):
        self._static_y = y.clone() if y is not None else None
        if torch.cuda.is_available() and (y is None or y.is_cuda):
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):               # one more eager step on the side stream (capture etiquette)
                self._forward_backward(self._static_y)
                self.bucket.all_reduce_mean_()
                self.optim.step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
        else:                                           # (a CPU stand-in graph factory: the same extra eager step, no streams)
            self._forward_backward(self._static_y)
            self.bucket.all_reduce_mean_()
            self.optim.step()
        single = not self.bucket.collective            # no all-reduce between backward and Adam: one graph, one replay
        # N > 1 with one_graph: the flat bucket's all-reduce is captured BETWEEN backward and Adam (ProcessGroupNCCL enqueues the RCCL
        # kernel on the capturing stream): one replay per step instead of two graph launches and an eager collective in between.
        # Opt-in (FlowTrainer(one_graph=True) / NF_DP_ONE_GRAPH=1): no multi-GPU node was available to measure it; a capture that
        # fails falls back to the two-graph form below.
        whole = single

        def forward_backward():
            z, loss = self._forward_backward(self._static_y)
            if whole:
                self.bucket.all_reduce_mean_()          # (no-op at world size 1)
                self.optim.step()
            return z.detach(), loss.detach()
        g_fb = None
        if not single and self.one_graph:
            whole = True
            try:
                g_fb = self._graph_factory()
                self._static_z, self._static_loss = g_fb.capture(forward_backward)
            except Exception as e:
                import warnings
                warnings.warn('the data-parallel step could not be captured as ONE hipGraph (%s: %s); two graphs with the all-reduce '
                              'between them' % (type(e).__name__, e))
                g_fb, whole = None, False
                if torch.cuda.is_available():
                    torch.cuda.synchronize()
        if g_fb is None:
            g_fb = self._graph_factory()
            self._static_z, self._static_loss = g_fb.capture(forward_backward)
        g_opt = None
        if not whole:                                   # N > 1: graph A, the flat bucket's all-reduce (eager, RCCL), graph B = Adam
            g_opt = self._graph_factory()
            g_opt.capture(self.optim.step)
        self._g_fb, self._g_opt = g_fb, g_opt
        self._g_whole = whole
        from . import _native as N_
        self._captured_det = N_.deterministic()

    def train_on_batch(self, y=None):
        """returns (z, loss) like main.py:78-92; with graph=True the returned tensors are the graph's static outputs.
        Raises _native.PersistentKernelTimeout as soon as the host sees that a persistent kernel of an EARLIER launch gave up on
        a grid exchange (sticky pinned error word, no synchronisation: at most one step late)."""
        from . import _native as N
        if y is None and self.sampler is None:
            raise ValueError('train_on_batch() without a batch needs FlowTrainer(..., sampler=data.DeviceSampler(...))')
        if y is not None and y.is_cuda and y.device.index != torch.cuda.current_device():
            # a model / batch on another GPU than the current one (net.to('cuda:1') without torch.cuda.set_device works in the reference):
            # the launches must go to THAT device's stream and read its copy of the library's globals
            with torch.cuda.device(y.device):
                return self.train_on_batch(y)
        if y is None or y.is_cuda:
            N.check_persistent()
        # Module.train() walks every submodule: 1.24 ms of host time per call on C1's ~1 000 modules -- more than the 1.05 ms the whole
        # step takes on the device.  The root's flag answers for the tree on most calls; a CHILD put into eval() on its own (the root stays
        # in training mode) is found by a walk on the first call and every TRAIN_MODE_RECHECK calls after it
        self._calls_since_mode_walk += 1
        if not self.net.training or self._calls_since_mode_walk >= self.TRAIN_MODE_RECHECK:
            if not self.net.training or any(not m_.training for m_ in self.net.modules()):
                if self._g_fb is not None:
                    raise RuntimeError('FlowTrainer: a submodule was switched to eval() after the step was captured in a hipGraph (the graph '
                                       'replays training-mode launches); build a new FlowTrainer')
                self.net.train()
            self._calls_since_mode_walk = 0
        if self._g_fb is not None and self._captured_det != N.deterministic():
            raise RuntimeError('FlowTrainer: _native.deterministic() was switched after the step was captured in a hipGraph -- the graph holds '
                               'the launch shapes of the mode it was captured in; build a new FlowTrainer')
        if self.graph and self._g_fb is None and self._eager_steps >= self.warmup:
            try:
                self._capture(y)
            except Exception as e:                      # keep training eagerly rather than dying on a capture problem
                import warnings
                warnings.warn('hipGraph capture failed (%s: %s); continuing with eager launches' % (type(e).__name__, e))
                self.graph = False
                self._g_fb = self._g_opt = None
                if torch.cuda.is_available():
                    torch.cuda.synchronize()
        if self._g_fb is not None:
            if not self._replicas_synced:               # warmup=0: no eager step ran before the capture
                self._sync_replicas_after_first_step()
            if self._static_y is not None:
                self._static_y.copy_(y, non_blocking=True)
            self._g_fb.replay()                         # (world size 1, or one_graph: the whole step)
            if self._g_opt is not None:
                self.bucket.all_reduce_mean_()
                self._g_opt.replay()
            return self._static_z, self._static_loss
        z, loss = self._forward_backward(y)
        self.bucket.all_reduce_mean_()
        self.optim.step()
        self._eager_steps += 1
        if not self._replicas_synced:
            self._sync_replicas_after_first_step()
        return z.detach(), loss.detach()

    def _sync_replicas_after_first_step(self):
        """The first forward performs the data-dependent ActNorm initialisation (modules.py:238-244) on each replica's OWN
        shard, so the replicas' log_scale / bias differ after step 1.  Data parallelism needs identical replicas:
        rank 0's parameters and buffers win (policy 2); afterwards running statistics evolve per replica
        unless the trainer runs in sync-statistics mode."""
        self._replicas_synced = True
        if self.bucket.collective:
            # every parameter AND buffer: the frozen PLU constants (P, pivots, sign_s), MAF's perm and the running statistics
            # too, so that replicas built from different seeds cannot keep private copies of those
            nfdist.broadcast_parameters(self.net, src=0, group=self.bucket.group, force=True)

    # -- evaluation -----------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def log_py(self, y):
        self.net.eval()
        z, ld = self.net(y)
        return standard_normal_logprob(z) + ld

    @torch.no_grad()
    def sample_y(self, n, dims, generator=None):
        self.net.eval()
        dev = next(self.net.parameters()).device
        z = torch.randn((n, ) + tuple(dims), device=dev, generator=generator)
        y, ld = self.net.backward(z)
        return y, torch.exp(standard_normal_logprob(z) - ld)
