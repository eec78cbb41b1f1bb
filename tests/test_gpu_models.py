"""
End-to-end parity of the product models (HIP transforms + PyTorch-ROCm conditioners) against the golden vectors
captured from the reference: forward (z, log-det), NLL loss, gradients of every parameter, the state the
forward pass mutates, and both inverses.  Needs a real MI355X.
"""
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from oracle import transforms as tf
from tests import _golden as G

pytestmark = pytest.mark.gpu
TOL = 1e-5
DEV = 'cuda'


SLACK = 4.0


def _oracle_gradient_gaps(name):
    """max |fp32 - float64| per parameter gradient of the oracle's own first training step on the golden's weights and batch: how far a
    correct fp32 implementation is from exact arithmetic on this model (empty for the stochastic ResFlow estimators)"""
    from oracle import trajectory as traj
    kind, cls, dims, datatype, layers, mix = G.MODEL_CASES[name]
    if kind == 'resflow':
        return {}
    sd0 = G.group('model_' + name, 'sd0/')
    y = G.group('model_' + name, '')['y']
    out = {}
    recs = []
    for dt in (torch.float32, torch.float64):
        np.random.seed(0)                                  # MADE draws its masks from the global numpy stream (constant for D = 2)
        r, _ = traj.run(kind, dims, datatype, layers, sd0, y, 1, mixtures=mix, dtype=dt)
        recs.append(r[1]['grads'])
    for k, g32 in recs[0].items():
        if k in recs[1]:
            out['net.' + k if not k.startswith('net.') else k] = float((g32.double() - recs[1][k]).abs().max())
    return out


def _cancelling(kind, k):
    """MADE's FIRST masked linear sees one live input (D = 2: mask [[1, 0]] x 32), so every unit's pre-activation is affine in the same
    scalar and the BatchNorm backward makes its gradient orthogonal to that scalar up to eps / (var + eps): the weight gradient is a sum
    whose terms cancel to a few percent of their size.  Two correct fp32 summation orders differ by ~1e-4 of the largest entry there
    (measured: the fp32 oracle itself sits 3.6e-6 .. 3.5e-5 from float64 depending on the host CPU's BLAS; the GPU 1.0e-4)."""
    return kind == 'maf' and k.endswith('.weights.0')


def _build(pkg, name):
    kind, cls, dims, datatype, layers, mix = G.MODEL_CASES[name]
    if not hasattr(pkg, cls):
        pytest.skip('%s not built yet' % cls)
    net = getattr(pkg, cls)(dims, datatype, NS(layers=layers, mixtures=mix, logdet='exact', spnorm_coeff=0.9))
    for m in net.modules():
        if hasattr(m, 'noise_on_cpu'):
            m.noise_on_cpu = True                     # same Hutchinson noise stream as the (CPU) reference
    sd0 = G.group('model_' + name, 'sd0/')
    missing = net.load_state_dict(sd0, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return net.to(DEV), G.group('model_' + name, '', DEV), kind, dims


@pytest.mark.parametrize('name', list(G.MODEL_CASES))
def test_model_golden(pkg, name):
    net, g, kind, dims = _build(pkg, name)
    net.train()
    G.seed_noise(777)
    z, ld = net(g['y'].clone())
    G.assert_close(z, g['train/z'], TOL, what='z')
    G.assert_close(ld, g['train/ld'], TOL, rtol=2e-6, what='ld')
    loss = tf.nll_loss(z, ld)
    G.assert_close(loss, g['train/loss'], TOL * max(1.0, abs(float(g['train/loss'])) / np.prod(dims)), what='loss')
    loss.backward()
    gaps = _oracle_gradient_gaps(name)
    n = 0
    for k, p in net.named_parameters():
        if 'grad/' + k in g:
            want = g['grad/' + k]
            assert p.grad is not None, k
            # pre-BatchNorm biases of MADE: analytically zero gradient, the stored value is cancellation noise
            noise = kind == 'maf' and '.biases.' in k and not k.endswith('.biases.3')
            # 2e-5 of the largest entry + SLACK x the fp32 oracle's own distance from float64 for this tensor (measured below: MADE's
            # training-mode BatchNorm backward is cancellation-heavy, its weight gradients sit ~1e-5 relative from float64 themselves)
            scale = max(1.0, float(want.abs().max()))
            tol = 2e-3 if noise else (1e-4 * scale if _cancelling(kind, k) else 2 * TOL * scale + SLACK * gaps.get(k, 0.0))
            G.assert_close(p.grad, want, tol, what=k)
            n += 1
    assert n > 4
    sd = net.state_dict()
    for k, want in G.group('model_' + name, 'sd1/').items():
        G.assert_close(sd[k].float(), want.float(), 2e-6, what=k)
    tol_inv = 2e-4 if kind in ('flowpp', 'resflow') else TOL
    with torch.no_grad():
        G.seed_noise(778)
        x, ldi = net.backward(g['train/z'].clone())
        G.assert_close(x, g['train/x_inv'], tol_inv, what='train x_inv')
        G.assert_close(ldi, g['train/ld_inv'], 10 * tol_inv, what='train ld_inv')
        net.eval()
        G.seed_noise(779)
        z, ld = net(g['y'].clone())
        G.assert_close(z, g['eval/z'], TOL, what='eval z')
        G.assert_close(ld, g['eval/ld'], TOL, rtol=2e-6, what='eval ld')
        x, ldi = net.backward(g['eval/z'].clone())
        G.assert_close(x, g['eval/x_inv'], tol_inv, what='eval x_inv')
        G.assert_close(ldi, g['eval/ld_inv'], 10 * tol_inv, what='eval ld_inv')


@pytest.mark.parametrize('name', ['glow2d', 'realnvp2d', 'maf2d', 'flowpp2d', 'glow_img', 'resflow2d', 'realnvp_img', 'flowpp_img'])
def test_model_golden_direct_grad_bucket(pkg, name):
    """same gradients when the backward kernels accumulate straight into the flat GradBucket (parameters re-homed
    into one flat buffer, fused NLL) -- the configuration the trainer and bench.py run."""
    import importlib
    nfdist = importlib.import_module(pkg.__name__ + '.dist')
    train = importlib.import_module(pkg.__name__ + '.train')
    net, g, kind, dims = _build(pkg, name)
    bucket = nfdist.GradBucket(net.parameters(), flatten_params=True)
    sd0 = G.group('model_' + name, 'sd0/', DEV)
    for k, p in net.named_parameters():                       # re-homing preserved the values
        assert torch.equal(p.detach(), sd0[k]), k
    net.train()
    gaps = _oracle_gradient_gaps(name)
    for rep in range(2):                                      # twice: zeroing + accumulation semantics
        G.seed_noise(777)
        if rep == 1:
            net.load_state_dict(G.group('model_' + name, 'sd0/'), strict=(kind != 'resflow'))   # SN drops module.weight
            for m in net.modules():
                if hasattr(m, 'initialized'):
                    m.initialized = False
        bucket.zero_()
        z, ld = net(g['y'].clone())
        loss = train.nll_loss(z, ld)
        loss.backward()
        G.assert_close(loss, g['train/loss'], TOL * max(1.0, abs(float(g['train/loss'])) / np.prod(dims)), what='loss')
        for k, p in net.named_parameters():
            if 'grad/' + k in g:
                want = g['grad/' + k]
                noise = kind == 'maf' and '.biases.' in k and not k.endswith('.biases.3')
                scale = max(1.0, float(want.abs().max()))
                tol = 2e-3 if noise else (1e-4 * scale if _cancelling(kind, k) else 2 * TOL * scale + SLACK * gaps.get(k, 0.0))
                G.assert_close(p.grad, want, tol, what='%s (rep %d)' % (k, rep))
                assert p.grad.data_ptr() >= bucket.flat.data_ptr()


def test_trainer_gradient_gather_matches_accumulate(pkg):
    """FlowTrainer's gather of framework-produced gradients (nf_multi_copy instead of one AccumulateGrad add per
    parameter) leaves exactly the same flat gradient bucket as plain accumulation, step after step."""
    import importlib
    train = importlib.import_module(pkg.__name__ + '.train')
    net, g, kind, dims = _build(pkg, 'glow_img')
    net2, _, _, _ = _build(pkg, 'glow_img')
    cond = importlib.import_module(pkg.__name__ + '.conditioners')
    for m in list(net.modules()) + list(net2.modules()):
        if isinstance(m, cond.ConvNet):
            m.fused = False                                   # the MIOpen module path: its parameter gradients come from autograd
    ta = train.FlowTrainer(net, graph=False)
    tb = train.FlowTrainer(net2, graph=False)
    tb._gather = False                                        # reference behaviour: AccumulateGrad into the bucket views
    y = g['y'].clone()
    for step in range(3):
        for t in (ta, tb):
            G.seed_noise(777 + step)
            t.net.train()
            t._forward_backward(y)
        if step >= 1:
            assert ta._indirect, 'an image Glow has convolution / BatchNorm2d parameters produced by autograd'
        # not bit-equal: some backward kernels of the image path reduce through atomics (order varies run to run)
        G.assert_close(ta.bucket.flat, tb.bucket.flat, 1e-5 * float(tb.bucket.flat.abs().max()), rtol=1e-4, what='flat grads, step %d' % step)
        for p in ta.bucket.params:
            assert p.grad is not None and p.grad.data_ptr() >= ta.bucket.flat.data_ptr()


def test_trainer_keeps_direct_gradient_sinks(pkg):
    """parameters whose gradient a hand-written backward writes straight into the bucket must NOT be classified as
    framework-produced (a tensor hook fires with None for them): a vector MAF keeps every sink, and its step stays at
    two launches per flow layer."""
    import importlib
    from types import SimpleNamespace as NS
    train = importlib.import_module(pkg.__name__ + '.train')
    F = importlib.import_module(pkg.__name__ + '.functional')
    torch.manual_seed(0)
    net = pkg.MAF((2, ), 'density', NS(layers=3, mixtures=8)).to(DEV)
    t = train.FlowTrainer(net, graph=False)
    y = torch.randn(512, 2, device=DEV)
    for _ in range(2):
        t.train_on_batch(y)
        assert t._indirect == []
        assert all(F.grad_sink(p) is not None for p in net.parameters())



@pytest.mark.parametrize('dims,B,K', [((3, 32, 32), 16, 2), ((3, 16, 16), 5, 3)])
def test_trainer_deferred_conv_weight_gradients(pkg, dims, B, K, monkeypatch):
    """inside a trainer step the image conditioners' backward launches only the data-gradient passes and the weight-gradient
    passes of all layers run sixteen per launch when the weight-norm backward asks for them (nf_conv_bn_bwd with g_weff =
    NULL + nf_conv_bn_wgrad_multi): same loss and the same flat gradient as the unsplit launches."""
    import copy
    import importlib
    from types import SimpleNamespace as NS
    train = importlib.import_module(pkg.__name__ + '.train')
    fc = importlib.import_module(pkg.__name__ + '.fused_conv')
    torch.manual_seed(7)
    net1 = pkg.Glow(dims, 'image', NS(layers=K, mixtures=8)).to(DEV)
    net2 = copy.deepcopy(net1)
    y = torch.rand(B, *dims, device=DEV)
    t1, t2 = train.FlowTrainer(net1, graph=False), train.FlowTrainer(net2, graph=False)
    queued = []
    real_flush = fc.ConvDefer.flush

    def flush(self):
        queued.append(len(self.layers))
        return real_flush(self)

    monkeypatch.setattr(fc.ConvDefer, 'flush', flush)
    for step in range(3):                                   # step 0 initialises the ActNorms (atomics: replicas re-synced below)
        monkeypatch.setattr(fc, 'CONV_DEFER_ON', True)
        z1, l1 = t1._forward_backward(y)
        monkeypatch.setattr(fc, 'CONV_DEFER_ON', False)
        z2, l2 = t2._forward_backward(y)
        assert not fc.CONV_DEFER.layers and not fc.CONV_DEFER.sums and not fc.CONV_DEFER.active
        if step > 0:
            G.assert_close(z1, z2, 1e-5, rtol=1e-5, what='z, step %d' % step)
            G.assert_close(l1, l2, 1e-5, rtol=1e-5, what='loss, step %d' % step)
            scale = float(t2.bucket.flat.abs().max())
            # (same kernels and operands; the slab sums and the atomics of the batch statistics add in a different order, and a
            #  pre-activation within rounding of zero may take the other side of its ReLU in one of the two replicas)
            bad = ((t1.bucket.flat - t2.bucket.flat).abs() > 2e-5 * max(1.0, scale)).float().mean()
            assert float(bad) <= 2e-3, 'flat gradients differ in %.2e of the entries (step %d)' % (float(bad), step)
            G.assert_close(t1.bucket.flat, t2.bucket.flat, 2e-3 * max(1.0, scale), what='flat grads, step %d' % step)
        t1.optim.step()
        net2.load_state_dict(net1.state_dict())
        t2.bucket.flat_params.copy_(t1.bucket.flat_params)
    n_cond = sum(1 for m in net1.modules() if type(m).__name__ == 'ConvNet')
    assert max(queued) == 6 * n_cond, 'weight-gradient passes were not deferred (queue lengths %r)' % (queued, )


@pytest.mark.parametrize('dims,B,K', [((3, 32, 32), 8, 5), ((3, 16, 16), 64, 6)])
def test_weight_gradient_table_launch_matches_the_sixteen_layer_launches(pkg, dims, B, K, monkeypatch):
    """more than sixteen queued layers of one shape run as ONE launch over a descriptor table in device memory (nf_conv_bn_wgrad_table, round 6:
    config 4 queues 320 hidden layers per resolution) instead of sixteen per launch (nf_conv_bn_wgrad_multi): same kernels and operands,
    another number of slabs per layer -- the flat gradient agrees to the rounding of a differently grouped sum, in eager steps and in a
    captured hipGraph (the table is written by by-value launches: nothing of the host is re-read at replay)."""
    import copy
    import importlib
    from types import SimpleNamespace as NS
    train = importlib.import_module(pkg.__name__ + '.train')
    fc = importlib.import_module(pkg.__name__ + '.fused_conv')
    torch.manual_seed(3)
    net1 = pkg.Glow(dims, 'image', NS(layers=K, mixtures=8)).to(DEV)
    net2 = copy.deepcopy(net1)
    y = torch.rand(B, *dims, device=DEV)
    t1, t2 = train.FlowTrainer(net1, graph=True, warmup=2), train.FlowTrainer(net2, graph=False)
    calls = {'table': 0, 'multi': 0}
    real_call = fc.N.call

    def call(name, *a):
        if name == 'nf_conv_bn_wgrad_table':
            calls['table'] += 1
        elif name == 'nf_conv_bn_wgrad_multi':
            calls['multi'] += 1
        return real_call(name, *a)
    monkeypatch.setattr(fc.N, 'call', call)
    for step in range(5):                                   # steps 0, 1 eager, step 2 captures (+ one eager step), steps 3, 4 replay
        monkeypatch.setattr(fc, 'WGRAD_TABLE', True)
        z1, l1 = t1.train_on_batch(y)
        n_table = calls['table']
        monkeypatch.setattr(fc, 'WGRAD_TABLE', False)
        if step == 2:
            t2.train_on_batch(y)                            # (the capturing call takes one extra eager step on its batch)
        z2, l2 = t2.train_on_batch(y)
        assert calls['table'] == n_table, 'the sixteen-layer form launched a table'
        if step == 0:                                       # ActNorm init by atomics: identical parameters from here on
            net2.load_state_dict(net1.state_dict())
            t2.bucket.flat_params.copy_(t1.bucket.flat_params)
            t2.optim.load_state_dict(t1.optim.state_dict())
            continue
        G.assert_close(l1, l2, 1e-5 * max(1.0, abs(float(l2))), what='loss, step %d' % step)
        scale = float(t2.bucket.flat.abs().max())
        bad = ((t1.bucket.flat - t2.bucket.flat).abs() > 2e-5 * max(1.0, scale)).float().mean()
        assert float(bad) <= 2e-3, 'flat gradients differ in %.2e of the entries (step %d)' % (float(bad), step)
        net2.load_state_dict(net1.state_dict())             # (Adam on rounding-different gradients: re-align the replicas)
        t2.bucket.flat_params.copy_(t1.bucket.flat_params)
        t2.optim.load_state_dict(t1.optim.state_dict())
    assert t1._g_fb is not None and calls['table'] >= 3 and calls['multi'] >= 3, calls


@pytest.mark.parametrize('name,dims,datatype,B,K', [('Glow', (2, ), 'density', 4096, 4), ('Glow', (2, ), 'density', 512, 3),
                                                    ('RealNVP', (2, ), 'density', 256, 4), ('MAF', (2, ), 'density', 2048, 3),
                                                    ('Flowpp', (2, ), 'density', 4096, 3), ('Glow', (3, 16, 16), 'image', 8, 2)])
def test_trainer_hipgraph_replay_matches_eager(pkg, name, dims, datatype, B, K):
    """the whole-step hipGraph (what bench.py times: capture of zero-grad + forward + backward incl. every deferred fold /
    finalize / weight-gradient pass, and of the Adam step) replays to the same training trajectory as eager launches."""
    import copy
    import importlib
    from types import SimpleNamespace as NS
    import numpy as np
    train = importlib.import_module(pkg.__name__ + '.train')
    torch.manual_seed(11)
    np.random.seed(11)
    net1 = getattr(pkg, name)(dims, datatype, NS(layers=K, mixtures=4)).to(DEV)
    net2 = copy.deepcopy(net1)
    te, tg = train.FlowTrainer(net1, graph=False), train.FlowTrainer(net2, graph=True, warmup=3)
    g = torch.Generator().manual_seed(5)
    for step in range(8):
        y = (torch.rand(B, *dims, generator=g) if datatype == 'image' else torch.randn(B, *dims, generator=g) * 0.8).to(DEV)
        if tg._g_fb is None and tg._eager_steps >= tg.warmup:
            te.train_on_batch(y)                            # the capturing call takes one extra (eager) step on its batch
        z1, l1 = te.train_on_batch(y)
        z2, l2 = tg.train_on_batch(y)
        if step == 0:      # the ActNorm init sums by atomics: start both trajectories from identical parameters
            net2.load_state_dict(net1.state_dict())
            tg.bucket.flat_params.copy_(te.bucket.flat_params)
        else:
            # (two trainings: summation-order noise of the atomics passes through Adam, so the bars are wide -- a fold, finalize or
            #  weight-gradient pass missing from the captured graph moves the trajectory by orders of magnitude more)
            G.assert_close(l2, l1, 2e-3 * max(1.0, abs(float(l1))), what='loss, step %d' % step)
            G.assert_close(z2, z1, 2e-2, rtol=2e-2, what='z, step %d' % step)
    assert tg._g_fb is not None, 'the step was never captured'
    scale = float(te.bucket.flat.abs().max())
    bad = ((tg.bucket.flat - te.bucket.flat).abs() > 2e-3 * max(1.0, scale)).float().mean()
    assert float(bad) < 1e-2, 'gradients of the replayed step differ in %.2e of the entries' % float(bad)


@pytest.mark.parametrize('name', ['glow2d', 'realnvp2d', 'maf2d', 'glow_img'])
def test_sync_statistics_mode_matches_goldens(pkg, name):
    """the parity mode of data parallelism (dist.sync_statistics: every batch statistic through global moments + all-reduce, the
    conditioners' BatchNorm layers through dist.sync_batch_norm, layer-by-layer launches) on ONE process must give what the
    reference gives on that batch: forward, loss and every gradient against the goldens.  (Two ranks are covered on CPU / gloo by
    tests/test_dist_cpu.py; with one rank the collectives are no-ops and this pins the arithmetic of the GPU code path.)"""
    import importlib
    nfdist = importlib.import_module(pkg.__name__ + '.dist')
    net, g, kind, dims = _build(pkg, name)
    net.train()
    with nfdist.sync_statistics():
        z, ld = net(g['y'].clone())
        loss = tf.nll_loss(z, ld)
        loss.backward()
    G.assert_close(z, g['train/z'], TOL, what='z')
    G.assert_close(ld, g['train/ld'], TOL, rtol=2e-6, what='ld')
    gaps = _oracle_gradient_gaps(name)
    n = 0
    for k, p in net.named_parameters():
        if 'grad/' + k in g:
            want = g['grad/' + k]
            noise = kind == 'maf' and '.biases.' in k and not k.endswith('.biases.3')
            scale = max(1.0, float(want.abs().max()))
            tol = 2e-3 if noise else (1e-4 * scale if _cancelling(kind, k) else 4 * TOL * scale + SLACK * gaps.get(k, 0.0))   # the conditioners run as rocBLAS / MIOpen + ATen modules in this mode (their summation order): 2.6e-5 measured
            G.assert_close(p.grad, want, tol, what=k)
            n += 1
    assert n > 4
    sd = net.state_dict()
    for k, want in G.group('model_' + name, 'sd1/').items():
        G.assert_close(sd[k].float(), want.float(), 2e-6, what=k)


INV_GRAD_CASES = [
    # kind, class, dims, datatype, layers, mixtures, batch
    ('realnvp', 'RealNVP', (2, ), '2d', 4, None, 64),
    ('glow', 'Glow', (2, ), '2d', 3, None, 64),
    ('flowpp', 'Flowpp', (2, ), '2d', 2, 4, 64),
    ('realnvp', 'RealNVP', (3, 8, 8), 'image', 1, None, 4),
    ('glow', 'Glow', (3, 8, 8), 'image', 1, None, 4),
]


@pytest.mark.parametrize('case', INV_GRAD_CASES, ids=['%s_%s' % (c[0], 'x'.join(str(d) for d in c[2])) for c in INV_GRAD_CASES])
def test_inverse_direction_with_autograd_matches_the_oracle(pkg, case):
    """``net.backward(z)`` with an input that requires grad (and, inside ``differentiable_inverse()``, without one) records the REFERENCE's
    graph of the inverse flow (normalizing-flows-pytorch_amd/inverse_grad.py; flows/coupling.py:114-122,192-210, flows/modules.py:152-155,
    252-256,309-322,484-497): x, the log-det, the gradient of the input and of every parameter against the oracle's autograd through its
    own inverse, in float32 with the float64 gap as slack.  (The gradient stops at an invertible 1x1 convolution in the reference as well --
    its solve runs under no_grad -- so the Glow cases check exactly that: zero input gradient, parameter gradients behind the last solve.)"""
    from oracle import models as om
    from oracle import trajectory as traj
    kind, cls, dims, datatype, layers, mix, B = case
    torch.manual_seed(11)
    np.random.seed(11)
    net = getattr(pkg, cls)(dims, datatype, NS(layers=layers, mixtures=mix))
    with torch.no_grad():                                    # non-trivial statistics / ActNorm parameters without a data-dependent step
        for k, v in net.state_dict().items():
            if k.endswith(('batch_mean', 'running_mean', 'bias')) and v.dim() >= 2:
                v.copy_(torch.randn_like(v) * 0.1)
            if k.endswith(('batch_var', 'running_var')):
                v.copy_(torch.rand_like(v) + 0.5)
            if k.endswith('log_scale') and v.dim() >= 2:
                v.copy_(torch.randn_like(v) * 0.1)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    z = torch.randn((B, ) + dims) * 0.7
    gx, gld = torch.randn((B, ) + dims), torch.randn(B)
    rec = {}
    for dt in (torch.float32, torch.float64):
        ora = om.FlowOracle(kind, dims, datatype, layers, traj.cast_state(sd, dt), mixtures=mix, training=True,
                            actnorm_initialized=True).requires_grad_(True)
        zz = z.to(dt).clone().requires_grad_(True)
        x, ld = ora.backward(zz)
        ((x * gx.to(dt)).sum() + (ld * gld.to(dt)).sum()).backward()
        rec[dt] = (x.detach(), ld.detach(), None if zz.grad is None else zz.grad.detach(),
                   {k: p.grad.detach() for k, p in ora.parameters().items() if p.grad is not None})
    net = net.to(DEV).train()
    for m_ in net.modules():
        if hasattr(m_, 'initialized'):
            m_.initialized = True
    x32, ld32, gz32, gp32 = rec[torch.float32]
    x64, ld64, gz64, gp64 = rec[torch.float64]

    def gap(a, b):
        return float((a.double() - b).abs().max())

    for with_input_grad in (True, False):
        net.zero_grad(set_to_none=True)
        zd = z.to(DEV).clone().requires_grad_(with_input_grad)
        if with_input_grad:
            xd, ldd = net.backward(zd)
        else:
            with pkg.differentiable_inverse():
                xd, ldd = net.backward(zd)
        assert xd.requires_grad or ldd.requires_grad
        ((xd * gx.to(DEV)).sum() + (ldd * gld.to(DEV)).sum()).backward()
        G.assert_close(xd, x32, 1e-4 * max(1.0, float(x32.abs().max())) + SLACK * gap(x32, x64), what='x')          # (Flow++: the bisection bracket)
        G.assert_close(ldd, ld32, 2e-3 * max(1.0, float(ld32.abs().max())) if kind == 'flowpp' else
                       TOL * max(1.0, float(ld32.abs().max())) + SLACK * gap(ld32, ld64), what='log-det')
        if with_input_grad:
            want = gz32 if gz32 is not None else torch.zeros_like(z)
            got = zd.grad if zd.grad is not None else torch.zeros_like(zd)
            loose = 2e-3 if kind == 'flowpp' else TOL
            G.assert_close(got, want, loose * max(1.0, float(want.abs().max())) + SLACK * (gap(gz32, gz64) if gz32 is not None else 0.0), what='g_z')
        named = dict(net.named_parameters())
        n = 0
        for k, g32 in gp32.items():
            kk = k if k in named else 'net.' + k
            p = named[kk]
            if float(g32.abs().max()) == 0.0 and p.grad is None:
                continue
            assert p.grad is not None, kk
            s = max(1.0, float(g32.abs().max()))
            loose = 5e-3 if kind == 'flowpp' else 2 * TOL          # (Flow++: gradients are evaluated at the bracket-limited x)
            G.assert_close(p.grad, g32, loose * s + SLACK * gap(g32, gp64[k]), what='grad ' + kk)
            n += 1
        assert n >= 4, n
    # without a request the sampling kernels run and record nothing
    xs, lds = net.backward(z.to(DEV))
    assert not xs.requires_grad and not lds.requires_grad
    G.assert_close(xs, x32, 1e-4 * max(1.0, float(x32.abs().max())) + SLACK * gap(x32, x64), what='x (sampling kernels)')
