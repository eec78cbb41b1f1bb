"""
Data-parallel path on CPU: world_size-2 gloo processes.  Checks that the flat gradient bucket + one all-reduce
reproduce the single-process gradient of the global batch (for a stack without batch-coupled statistics) and that
sharding / broadcasting behave.  The model here is the ORACLE (tests may use it): the product transforms are GPU-only,
the DP helper is model-agnostic.
"""
import importlib
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = 'normalizing-flows-pytorch_amd'


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Tiny(torch.nn.Module):
    """per-sample independent toy flow: y = z * exp(s) + t, ld = sum s  (no batch statistics)."""

    def __init__(self):
        super().__init__()
        self.s = torch.nn.Parameter(torch.tensor([0.1, -0.2, 0.3]))
        self.t = torch.nn.Parameter(torch.tensor([0.5, 0.0, -0.5]))
        self.frozen = torch.nn.Parameter(torch.ones(2, dtype=torch.int32), requires_grad=False)

    def forward(self, z):
        return z * torch.exp(self.s) + self.t, self.s.sum().expand(z.shape[0])


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    nfdist = importlib.import_module(PKG + '.dist')
    train = importlib.import_module(PKG + '.train')
    r, w, _ = nfdist.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)                   # deliberately different init: broadcast must fix it
    net = _Tiny()
    with torch.no_grad():
        net.s.add_(torch.randn(3) * 0.1)
    nfdist.broadcast_parameters(net)
    bucket = nfdist.GradBucket(net.parameters())
    assert bucket.numel == 6                        # the frozen int32 parameter is skipped
    g = torch.Generator().manual_seed(5)
    y_global = torch.randn(16, 3, generator=g)
    y = nfdist.shard(y_global, rank, world)
    bucket.zero_()
    z, ld = net(y)
    train.nll_loss(z, ld).backward()
    assert net.s.grad.data_ptr() == bucket.flat.data_ptr()      # grads are views of the flat bucket
    bucket.all_reduce_mean_()
    q.put((rank, net.s.detach().tolist(), bucket.flat.tolist()))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_dp_gradients_equal_global_batch():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, s0, g0), (_, s1, g1) = res
    assert s0 == s1                                 # broadcast made the replicas identical
    assert g0 == g1                                 # all-reduce leaves identical buckets
    s0, g0 = torch.tensor(s0), torch.tensor(g0)
    # single-process reference on the GLOBAL batch
    sys.path.insert(0, ROOT)
    train = importlib.import_module(PKG + '.train')
    net = _Tiny()
    with torch.no_grad():
        net.s.copy_(s0)
    y_global = torch.randn(16, 3, generator=torch.Generator().manual_seed(5))
    z, ld = net(y_global)
    train.nll_loss(z, ld).backward()
    want = torch.cat([net.s.grad.view(-1), net.t.grad.view(-1)])
    assert torch.allclose(g0, want, atol=1e-6)


def test_shard_rejects_ragged_batch():
    sys.path.insert(0, ROOT)
    nfdist = importlib.import_module(PKG + '.dist')
    with pytest.raises(ValueError):
        nfdist.shard(torch.zeros(10, 2), 0, 4)
    assert torch.equal(nfdist.shard(torch.arange(8).view(8, 1), 1, 2), torch.arange(4, 8).view(4, 1))


class _InitFlow(torch.nn.Module):
    """toy flow with a DATA-DEPENDENT first-batch initialisation (like ActNorm): replicas see different shards."""

    def __init__(self):
        super().__init__()
        self.log_scale = torch.nn.Parameter(torch.zeros(3))
        self.bias = torch.nn.Parameter(torch.zeros(3))
        self.initialized = False

    def forward(self, z):
        if not self.initialized:
            with torch.no_grad():
                self.log_scale.copy_(torch.log(z.std(0) + 1e-5))
                self.bias.copy_(z.mean(0))
            self.initialized = True
        y = (z - self.bias) / torch.exp(self.log_scale)
        return y, (-self.log_scale.sum()).expand(z.shape[0])


def _trainer_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    nfdist = importlib.import_module(PKG + '.dist')
    train = importlib.import_module(PKG + '.train')
    nfdist.init_from_env(backend='gloo')
    net = _InitFlow()
    tr = train.FlowTrainer(net, lr=1e-2, graph=False)          # CPU: torch Adam over the bucket, same control flow
    g = torch.Generator().manual_seed(9)
    y_global = torch.randn(32, 3, generator=g) * 2 + 1
    y = nfdist.shard(y_global, rank, world)
    losses = []
    for _ in range(3):
        _, loss = tr.train_on_batch(y)
        losses.append(float(loss))
    q.put((rank, net.log_scale.detach().tolist(), net.bias.detach().tolist(), losses))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_trainer_replicas_identical_after_data_dependent_init():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_trainer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, ls0, b0, l0), (_, ls1, b1, l1) = res
    assert ls0 == ls1 and b0 == b1                             # rank 0's initialisation won, updates identical since
    assert l0[0] != l1[0]                                      # (step 1 really did run on different shards)


# ---- parity mode: synchronised batch statistics ------------------------------------------------------------------------------------
class _ToyBNFlow(torch.nn.Module):
    """A batch-coupled toy flow on CPU tensors, built from the PRODUCT's own pieces wherever they run without a GPU: the MLP
    conditioner (conditioners.MLP: weight-normed linears + five training-mode BatchNorm1d, here through dist.sync_batch_norm), a
    flow BatchNorm whose statistics come from dist.global_moments (constants for autograd, biased variance + eps, as
    modules.py:283-307) and an ActNorm-style data-dependent initialisation from the same moments.  The transform arithmetic itself
    is plain torch (the HIP transforms are GPU-only)."""

    def __init__(self, pkg_name):
        super().__init__()
        cond = importlib.import_module(pkg_name + '.conditioners')
        self.nfdist = importlib.import_module(pkg_name + '.dist')
        self.net = cond.MLP(1, 2)
        self.log_scale = torch.nn.Parameter(torch.zeros(2))
        self.bias = torch.nn.Parameter(torch.zeros(2))
        self.a = torch.nn.Parameter(torch.tensor([0.7]))
        self.initialized = False

    def forward(self, z):
        d = self.nfdist
        ld = torch.zeros(z.shape[0])
        if not self.initialized:                         # ActNorm init (modules.py:238-244) over the global batch
            with torch.no_grad():
                mean, var, n = d.global_moments(z)
                self.log_scale.copy_(torch.log(torch.sqrt(var * n / (n - 1.0)) + 1e-5))
                self.bias.copy_(mean)
            self.initialized = True
        z = (z - self.bias) / torch.exp(self.log_scale)
        ld = ld - self.log_scale.sum()
        with torch.no_grad():                            # flow BatchNorm statistics: constants for autograd
            mean, var, _ = d.global_moments(z)
            var = var + 1e-5
        z = (z - mean) / torch.sqrt(var)
        ld = ld - 0.5 * torch.log(var).sum()
        p = self.net(z[:, 1:2])                          # conditioner on the untouched half
        s = torch.tanh(p[:, 1]) * self.a
        y0 = z[:, 0] * torch.exp(s) + p[:, 0]
        return torch.stack([y0, z[:, 1]], dim=1), ld + s


def _sync_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    nfdist = importlib.import_module(PKG + '.dist')
    train = importlib.import_module(PKG + '.train')
    if world > 1:
        nfdist.init_from_env(backend='gloo')
    torch.manual_seed(7)                                 # identical initial weights on every rank
    net = _ToyBNFlow(PKG).train()
    g = torch.Generator().manual_seed(5)
    y_global = torch.randn(64, 2, generator=g) * torch.tensor([0.5, 2.0]) + torch.tensor([1.0, -3.0])
    y = nfdist.shard(y_global, rank, world) if world > 1 else y_global
    trainer = train.FlowTrainer(net, graph=False, fused_adam=False, sync_stats=True)
    out = []
    for _ in range(3):                                   # step 1 initialises, steps 2 and 3 train: parameters must stay in lockstep
        z, loss = trainer.train_on_batch(y)
        out.append((z.detach().numpy().copy(), float(loss), trainer.bucket.flat.detach().numpy().copy()))
    state = {k: v.detach().double().numpy().copy() for k, v in net.state_dict().items()}
    q.put((rank, out, state))                            # numpy, not tensors: shared-memory tensors die with the sender
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def _run(world):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sync_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_sync_statistics_reproduce_the_single_process_global_batch():
    """FlowTrainer(sync_stats=True) over two gloo ranks == one process on the concatenated batch: outputs (the ranks' shards put
    back together), the loss (mean of the per-shard means), the all-reduced flat gradient, and after three optimizer steps
    every parameter and every running statistic."""
    single = _run(1)[0]
    two = _run(2)
    for step in range(3):
        z1, l1, g1 = (torch.from_numpy(single[1][step][0]), single[1][step][1], torch.from_numpy(single[1][step][2]))
        z2 = torch.cat([torch.from_numpy(two[0][1][step][0]), torch.from_numpy(two[1][1][step][0])], dim=0)
        l2 = 0.5 * (two[0][1][step][1] + two[1][1][step][1])
        assert torch.allclose(z2, z1, atol=2e-6, rtol=1e-5), (step, float((z2 - z1).abs().max()))
        assert abs(l2 - l1) < 1e-6 * max(1.0, abs(l1)), (step, l1, l2)
        for r in range(2):
            g2 = torch.from_numpy(two[r][1][step][2])
            assert torch.allclose(g2, g1, atol=1e-6 * max(1.0, float(g1.abs().max())), rtol=1e-4), (step, r, float((g2 - g1).abs().max()))
    # parameters after three Adam steps -- except those whose gradient is ANALYTICALLY zero (a shift or scale in front of a batch
    # normalisation: the toy's ActNorm pair, the conditioner's pre-BatchNorm biases): their gradient is rounding noise, Adam's first
    # steps move them by lr * sign(noise), and no two runs agree on that (the forward values do not depend on them)
    def noise_only(k):
        return k in ('bias', 'log_scale') or (k.endswith('module.bias') and 'out_block' not in k)
    n = 0
    for r in range(2):
        for k, v in single[2].items():
            if noise_only(k):
                continue
            # running means downstream of such a bias inherit its +- lr random walk (3 steps x 1e-4): 1e-3 absolute here; the tight
            # lockstep evidence is z / loss / gradient at every step above
            assert torch.allclose(torch.from_numpy(two[r][2][k]), torch.from_numpy(v), atol=1e-3, rtol=1e-4), (r, k)
            n += 1
    assert n >= 2 * 30


# ---- the N > 1 hipGraph control flow of FlowTrainer (graph A, all-reduce, graph B) on gloo --------------------------------------------
class _OracleBackedGlow(torch.nn.Module):
    """The PRODUCT's Glow((2,), '2d') module tree -- its parameters, buffers and state_dict layout, i.e. exactly what GradBucket,
    the broadcasts and Adam see on C2 -- executed on CPU by the oracle over the LIVE tensors (the HIP transforms are GPU-only;
    tests may use the oracle)."""

    def __init__(self, layers):
        super().__init__()
        from types import SimpleNamespace as NS
        from oracle import models as om
        pkg = importlib.import_module(PKG)
        self.net = pkg.Glow((2, ), '2d', NS(layers=layers, mixtures=None)).net
        sd = {'net.' + k: v for k, v in self.net.named_parameters()}
        sd.update({'net.' + k: v for k, v in self.net.named_buffers()})
        self.ora = om.FlowOracle('glow', (2, ), '2d', layers, sd, training=True)

    def forward(self, y):
        return self.ora.forward(y)


class _RerunGraph:
    """stand-in for a hipGraph on CPU: ``capture`` records the closure WITHOUT leaving a trace (a real capture executes nothing: the
    parameters, buffers, optimizer state and the oracle's ActNorm flags are restored after the run that is needed to obtain the static
    outputs), ``replay`` re-runs it and refreshes the static outputs in place."""
    trainer = None

    def capture(self, fn):
        import copy
        tr = type(self).trainer
        keep = [t.detach().clone() for t in list(tr.net.parameters()) + list(tr.net.buffers())]
        opt = copy.deepcopy(tr.optim.state_dict())
        flat = tr.bucket.flat.clone()
        self.fn = fn
        self.out = fn()
        with torch.no_grad():
            for t, k in zip(list(tr.net.parameters()) + list(tr.net.buffers()), keep):
                t.copy_(k)
            tr.bucket.flat.copy_(flat)
        tr.optim.load_state_dict(opt)
        return self.out

    def replay(self):
        new = self.fn()
        if new is not None:
            for o, n in zip(self.out, new):
                o.copy_(n)


def _graph_worker(rank, world, port, q, use_graph, one_graph=False):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    nfdist = importlib.import_module(PKG + '.dist')
    train = importlib.import_module(PKG + '.train')
    nfdata = importlib.import_module(PKG + '.data')
    nfdist.init_from_env(backend='gloo')
    torch.set_num_threads(2)
    torch.manual_seed(3 + rank)                          # DIFFERENT initial weights per rank: the coalesced broadcast must fix that
    import numpy as np
    np.random.seed(3 + rank)
    net = _OracleBackedGlow(4).train()
    n_coll = nfdist.broadcast_parameters(net)
    tr = train.FlowTrainer(net, lr=1e-3, graph=use_graph, warmup=2, graph_factory=_RerunGraph, one_graph=one_graph)
    _RerunGraph.trainer = tr
    y_global = nfdata.sample('moons', 256, 77)
    y = nfdist.shard(y_global, rank, world)
    losses = []
    for i in range(6):
        if not use_graph and i == 2:
            tr.train_on_batch(y + 0.01 * i)              # the capturing call of the graph run takes one extra eager step on its batch
        z, loss = tr.train_on_batch(y + 0.01 * i)        # a different batch every step: the static input must be refreshed
        losses.append(float(loss))
    params = torch.cat([p.detach().reshape(-1) for p in net.parameters() if p.requires_grad]).numpy().copy()
    q.put((rank, losses, params, n_coll, tr._g_fb is not None, tr._g_opt is not None, int(tr.optim.state_dict()['state'][0]['step'])))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def _run_graph(use_graph, one_graph=False):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_graph_worker, args=(r, 2, port, q, use_graph, one_graph)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_trainer_two_graph_path_matches_eager_over_two_ranks():
    """FlowTrainer(graph=True) at world size 2 (train.py: graph A = zero + forward + NLL + backward, then the flat bucket's
    all-reduce, then graph B = Adam) against FlowTrainer(graph=False) on the same two ranks, shards and batches, with the product's
    own Glow-2D module tree (C2's parameter layout): every loss of six steps and the final parameters must agree, the replicas must
    stay identical, and start-up must need a handful of broadcasts (one per dtype), not one per tensor."""
    import numpy as np
    eager = _run_graph(False)
    graph = _run_graph(True)
    for r in range(2):
        assert graph[r][4] and graph[r][5], 'the two-graph path (graph A + graph B) was not taken at world size 2'
        assert not eager[r][4]
        assert graph[r][3] <= 4, '%d broadcasts at start-up' % graph[r][3]
        assert graph[r][6] == eager[r][6] == 7, (graph[r][6], eager[r][6])   # six calls + the capture's extra step
        assert np.allclose(graph[r][1], eager[r][1], rtol=1e-6, atol=1e-6), (graph[r][1], eager[r][1])
        assert np.allclose(graph[r][2], eager[r][2], rtol=1e-5, atol=1e-6), float(np.abs(graph[r][2] - eager[r][2]).max())
    assert np.array_equal(graph[0][2], graph[1][2]), 'replicas diverged on the graph path'
    assert graph[0][1][0] != graph[1][1][0]              # (the ranks really trained on different shards)


def test_trainer_one_graph_path_matches_eager_over_two_ranks():
    """FlowTrainer(graph=True, one_graph=True) at world size 2: zero + forward + NLL + backward, the flat bucket's all-reduce and Adam
    captured as ONE graph (one replay per step; the collective sits between backward and Adam inside the capture) -- same losses, step
    count and final parameters as the eager two-rank run, identical replicas, and no second graph."""
    import numpy as np
    eager = _run_graph(False)
    graph = _run_graph(True, one_graph=True)
    for r in range(2):
        assert graph[r][4] and not graph[r][5], 'the one-graph path was not taken at world size 2'
        assert graph[r][6] == eager[r][6] == 7, (graph[r][6], eager[r][6])
        assert np.allclose(graph[r][1], eager[r][1], rtol=1e-6, atol=1e-6), (graph[r][1], eager[r][1])
        assert np.allclose(graph[r][2], eager[r][2], rtol=1e-5, atol=1e-6), float(np.abs(graph[r][2] - eager[r][2]).max())
    assert np.array_equal(graph[0][2], graph[1][2]), 'replicas diverged on the one-graph path'


# ---- a ONE-rank group sent down the N > 1 control flow (NF_DP_FORCE_COLLECTIVE=1): what tests/test_gpu_rccl.py runs on RCCL ---------------
def _one_rank_worker(port, q, force, one_graph):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      NF_DP_FORCE_COLLECTIVE='1' if force else '0')
    nfdist = importlib.import_module(PKG + '.dist')
    train = importlib.import_module(PKG + '.train')
    nfdata = importlib.import_module(PKG + '.data')
    nfdist.init_from_env(backend='gloo')                 # initialises a one-rank group only when forced
    torch.set_num_threads(2)
    torch.manual_seed(3)
    import numpy as np
    np.random.seed(3)
    net = _OracleBackedGlow(3).train()
    tr = train.FlowTrainer(net, lr=1e-3, graph=True, warmup=2, graph_factory=_RerunGraph, one_graph=one_graph)
    _RerunGraph.trainer = tr
    y = nfdata.sample('moons', 128, 77)
    losses = [float(tr.train_on_batch(y + 0.01 * i)[1]) for i in range(5)]
    params = torch.cat([p.detach().reshape(-1) for p in net.parameters() if p.requires_grad]).numpy().copy()
    q.put((losses, params, torch.distributed.is_initialized(), tr.bucket.collective, tr._g_fb is not None, tr._g_opt is not None,
           bool(tr._g_whole)))
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def _run_one_rank(force, one_graph=False):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_one_rank_worker, args=(_free_port(), q, force, one_graph))
    p.start()
    res = q.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0
    return res


def test_one_rank_group_takes_the_data_parallel_control_flow_when_forced():
    """NF_DP_FORCE_COLLECTIVE=1 (dist.GradBucket.collective, dist.init_from_env): a one-rank process group runs graph A, the flat bucket's
    all-reduce and graph B -- or, with one_graph, the collective inside the one graph -- and, the one-rank all-reduce being the identity,
    reproduces the plain single-process trainer exactly (CPU: deterministic).  Unforced, one rank initialises no group at all."""
    import numpy as np
    plain = _run_one_rank(False)
    two = _run_one_rank(True)
    one = _run_one_rank(True, one_graph=True)
    assert plain[2:] == (False, False, True, False, True)
    assert two[2:] == (True, True, True, True, False)
    assert one[2:] == (True, True, True, False, True)
    for r in (two, one):
        assert r[0] == plain[0], (r[0], plain[0])
        assert np.array_equal(r[1], plain[1])
