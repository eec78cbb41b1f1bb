"""
Parity of the fused fp32-MFMA conditioner kernels (linear + BatchNorm + ReLU chains: csrc/linear_bn.hip) against the
module-by-module PyTorch path of the same networks and against the oracle's CPU restatement.  Needs a real MI355X.
"""
import copy
import importlib

import numpy as np
import pytest
import torch

from oracle import nets as onets
from tests import _golden as G

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _grad_tol(w):
    return 1e-4 * max(1.0, float(w.abs().max()))       # BatchNorm backward is cancellation-heavy


def _mlp_pair(pkg, in_ch, out_ch):
    cond = importlib.import_module(pkg.__name__ + '.conditioners')
    torch.manual_seed(in_ch * 100 + out_ch)
    ref = cond.MLP(in_ch, out_ch).to(DEV)
    with torch.no_grad():                                  # make BatchNorm affine / running stats non-trivial
        for m in ref.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.3)
                m.running_mean.normal_(0, 0.2)
                m.running_var.uniform_(0.5, 2.0)
    return ref, copy.deepcopy(ref)


@pytest.mark.parametrize('in_ch,out_ch,N', [(1, 2, 4096), (1, 2, 257), (3, 6, 1000), (16, 32, 64), (32, 32, 96)])
@pytest.mark.parametrize('training', [True, False])
def test_fused_mlp_vs_modules(pkg, in_ch, out_ch, N, training):
    ref, fus = _mlp_pair(pkg, in_ch, out_ch)
    ref.train(training)
    fus.train(training)
    g = torch.Generator().manual_seed(N)
    x = (torch.randn(N, in_ch, generator=g) * 0.7).to(DEV)
    gout = torch.randn(N, out_ch, generator=g).to(DEV)
    xr, xf = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    yr = ref.forward_reference(xr)
    yf = fus(xf)
    G.assert_close(yf, yr, 2e-5, rtol=2e-5, what='output')
    yr.backward(gout)
    yf.backward(gout)
    G.assert_close(xf.grad, xr.grad, _grad_tol(xr.grad), what='grad input')
    pr, pf = dict(ref.named_parameters()), dict(fus.named_parameters())
    for k in pr:
        assert pf[k].grad is not None, k
        pre_bn_bias = training and k.endswith('module.bias') and 'out_block' not in k
        tol = 2e-3 + 1e-6 * N if pre_bn_bias else _grad_tol(pr[k].grad)   # analytically-zero gradients: noise only
        G.assert_close(pf[k].grad, pr[k].grad, tol, what='grad ' + k)
    br, bf = dict(ref.named_buffers()), dict(fus.named_buffers())
    for k in br:
        G.assert_close(bf[k].float(), br[k].float(), 2e-6, rtol=1e-5, what='buffer ' + k)


def test_fused_mlp_vs_oracle_cpu(pkg):
    cond = importlib.import_module(pkg.__name__ + '.conditioners')
    torch.manual_seed(0)
    mlp = cond.MLP(1, 2)
    sd = {k: v.clone() for k, v in mlp.state_dict().items()}
    x = torch.randn(512, 1) * 0.5
    want = onets.mlp(x, sd, '', training=True)
    got = mlp.to(DEV).train()(x.to(DEV))
    G.assert_close(got, want, 1e-5, what='mlp vs oracle')
    for k, v in mlp.state_dict().items():
        G.assert_close(v.float(), sd[k].float(), 2e-6, what=k)       # running statistics updated identically


@pytest.mark.parametrize('D,N', [(2, 16384), (2, 100), (5, 333), (8, 64)])
@pytest.mark.parametrize('training', [True, False])
def test_fused_made_pair_vs_modules(pkg, D, N, training):
    torch.manual_seed(D)
    ref = pkg.AutoregressiveTransfrom(D).to(DEV)
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.3)
                m.running_mean.normal_(0, 0.2)
                m.running_var.uniform_(0.5, 2.0)
    fus = copy.deepcopy(ref)
    ref.train(training)
    fus.train(training)
    g = torch.Generator().manual_seed(N)
    z = torch.randn(N, D, generator=g).to(DEV)
    gs, gt = torch.randn(N, D, generator=g).to(DEV), torch.randn(N, D, generator=g).to(DEV)
    zr, zf = z.clone().requires_grad_(True), z.clone().requires_grad_(True)
    np.random.seed(11)
    sr, tr = ref.net_s(zr), ref.net_t(zr)                  # module path (rocBLAS + MIOpen)
    np.random.seed(11)
    sf, tf_ = fus.conditioners(zf)                         # fused path
    G.assert_close(sf, sr, 2e-5, rtol=2e-5, what='s')
    G.assert_close(tf_, tr, 2e-5, rtol=2e-5, what='t')
    torch.autograd.backward([sr, tr], [gs, gt])
    torch.autograd.backward([sf, tf_], [gs, gt])
    G.assert_close(zf.grad, zr.grad, _grad_tol(zr.grad), what='grad z')
    pr, pf = dict(ref.named_parameters()), dict(fus.named_parameters())
    for k in pr:
        if pr[k].grad is None:
            continue
        pre_bn_bias = training and '.biases.' in k and not k.endswith('.biases.3')
        G.assert_close(pf[k].grad, pr[k].grad, 2e-3 + 1e-6 * N if pre_bn_bias else _grad_tol(pr[k].grad),
                       what='grad ' + k)
    br, bf = dict(ref.named_buffers()), dict(fus.named_buffers())
    for k in br:
        G.assert_close(bf[k].float(), br[k].float(), 2e-6, rtol=1e-5, what='buffer ' + k)


@pytest.mark.parametrize('D,K,N', [(2, 8, 65536), (2, 8, 77), (2, 4, 1000), (4, 8, 300), (8, 4, 64)])
def test_fused_flowpp_conditioner_forward(pkg, D, K, N):
    fused = importlib.import_module(pkg.__name__ + '.fused')
    torch.manual_seed(D * 10 + K)
    layer = pkg.MixLogAttnCoupling((D, ), n_mixtures=K).to(DEV)
    with torch.no_grad():
        for m in layer.net.modules():
            if isinstance(m, torch.nn.LayerNorm):
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.3)
    x = (torch.randn(N, D // 2) * 0.8).to(DEV)
    assert fused.flowpp_cond_fusable(layer.net, x)
    with torch.no_grad():
        want = layer.net(x)
        got = fused.flowpp_cond_forward_nograd(layer.net, x)
    G.assert_close(got, want, 2e-5, rtol=2e-5, what='flow++ conditioner forward')


@pytest.mark.parametrize('direct', [False, True])
@pytest.mark.parametrize('D,K,N', [(2, 8, 65536), (2, 8, 77), (2, 4, 1000), (4, 8, 300), (8, 4, 64), (2, 8, 1)])
def test_fused_flowpp_conditioner_backward(pkg, D, K, N, direct):
    """one-launch backward of the gated-attention conditioner against autograd through the module stack
    (flows/coupling.py:142-149): input gradient and every parameter gradient, conv1's V/K rows exactly zero."""
    fused = importlib.import_module(pkg.__name__ + '.fused')
    torch.manual_seed(D * 10 + K)
    layer = pkg.MixLogAttnCoupling((D, ), n_mixtures=K).to(DEV)
    with torch.no_grad():
        for m in layer.net.modules():
            if isinstance(m, torch.nn.LayerNorm):
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.3)
    x = (torch.randn(N, D // 2) * 0.8).to(DEV).requires_grad_(True)
    x2 = x.detach().clone().requires_grad_(True)
    w = torch.randn(N, layer.net[-1].out_features, device=DEV)
    want = layer.net(x)
    (want * w).sum().backward()
    ref = {k: p.grad.clone() for k, p in layer.net.named_parameters()}
    for p in layer.net.parameters():
        p.grad = None
    if direct:
        for p in layer.net.parameters():
            p.grad = torch.zeros_like(p)
            p._nf_direct_grad = True
    got = fused.flowpp_cond_forward(layer.net, x2)
    G.assert_close(got, want, 2e-5, rtol=2e-5, what='forward')
    (got * w).sum().backward()
    G.assert_close(x2.grad, x.grad, _grad_tol(x.grad), what='input grad')
    for k, p in layer.net.named_parameters():
        G.assert_close(p.grad, ref[k], _grad_tol(ref[k]), what='grad ' + k)
    F_ = layer.net[3].filters
    assert torch.count_nonzero(layer.net[3].conv1.weight.grad[:2 * F_]) == 0
    assert torch.count_nonzero(layer.net[3].conv1.bias.grad[:2 * F_]) == 0


@pytest.mark.parametrize('direct', [False, True])
@pytest.mark.parametrize('K,N,odd', [(8, 65536, False), (8, 1001, True), (4, 77, False), (8, 1, True)])
def test_flowpp_coupling_with_next_actnorm(pkg, K, N, odd, direct):
    """[MixLogAttnCoupling, ActNorm] on two features: the Compose peephole runs the next step's ActNorm inside the coupling's
    launches (nf_flowpp_vec_couple_fwd / _bwd); outputs, log-det, input gradient and every parameter gradient against the
    layer-by-layer path (flows/flowpp.py:60-66, modules.py:246-249)."""
    torch.manual_seed(K + N)
    k = pkg.MixLogAttnCoupling((2, ), n_mixtures=K, odd=odd).to(DEV)
    a = pkg.ActNorm((2, )).to(DEV)
    with torch.no_grad():
        a.log_scale.normal_(0, 0.4)
        a.bias.normal_(0, 0.5)
        k.a_log_scale.fill_(0.3)
        k.a_bias.fill_(-0.2)
        for p in k.net[-1].parameters():
            p.normal_(0, 0.2)
    a.initialized = True
    comp = pkg.Compose([k, a]).to(DEV)
    z = (torch.randn(N, 2) * 0.9).to(DEV).requires_grad_(True)
    z2 = z.detach().clone().requires_grad_(True)
    wy, wl = torch.randn(N, 2, device=DEV), torch.randn(N, device=DEV)

    comp.fuse = False
    y, ld = comp(z, torch.zeros(N, device=DEV))
    ((y * wy).sum() + (ld * wl).sum()).backward()
    ref = {n: p.grad.clone() for n, p in comp.named_parameters()}
    for p in comp.parameters():
        p.grad = None
        if direct:
            p.grad = torch.zeros_like(p)
            p._nf_direct_grad = True
    comp.fuse = True
    assert comp._flowpp_pair_at(0, z2)
    y2, ld2 = comp(z2, torch.zeros(N, device=DEV))
    G.assert_close(y2, y, 2e-5, rtol=2e-5, what='pair output')
    G.assert_close(ld2, ld, 2e-5, rtol=2e-5, what='pair log-det')
    ((y2 * wy).sum() + (ld2 * wl).sum()).backward()
    G.assert_close(z2.grad, z.grad, _grad_tol(z.grad), what='input grad')
    for n, p in comp.named_parameters():
        G.assert_close(p.grad, ref[n], _grad_tol(ref[n]), what='grad ' + n)

    # an uninitialised ActNorm keeps its data-dependent init: no fusion
    a.initialized = False
    assert not comp._flowpp_pair_at(0, z2)


@pytest.mark.parametrize('in_ch,out_ch,N', [(1, 2, 4096), (1, 2, 257), (3, 6, 1000), (16, 32, 64), (32, 32, 96),
                                            (1, 2, 1), (2, 4, 16384)])
@pytest.mark.parametrize('training', [True, False])
def test_mlp_chain_forward_vs_modules(pkg, in_ch, out_ch, N, training):
    """persistent single-launch MLP conditioner (csrc/mlp_chain.hip) against the module-by-module path:
    output, running statistics, num_batches_tracked."""
    if N == 1 and training:
        pytest.skip('BatchNorm1d refuses a single training sample')
    fused = importlib.import_module(pkg.__name__ + '.fused')
    ref, fus = _mlp_pair(pkg, in_ch, out_ch)
    ref.train(training)
    fus.train(training)
    x = torch.randn(N, in_ch, device=DEV)
    assert fused.mlp_chain_usable(fus, x)
    with torch.no_grad():
        want = ref.forward_reference(x)
        got, _ = fused.mlp_chain_forward_nograd(fus, x, training)
    G.assert_close(got, want, 2e-5, rtol=2e-5, what='mlp chain forward')
    br, bf = dict(ref.named_buffers()), dict(fus.named_buffers())
    for k in br:
        G.assert_close(bf[k].float(), br[k].float(), 2e-6, rtol=1e-5, what='buffer ' + k)


@pytest.mark.parametrize('direct', [False, True])
@pytest.mark.parametrize('in_ch,out_ch,N', [(1, 2, 4096), (1, 2, 257), (3, 6, 1000), (16, 32, 64), (32, 32, 96),
                                            (2, 4, 16384)])
@pytest.mark.parametrize('training', [True, False])
def test_mlp_chain_vs_modules(pkg, in_ch, out_ch, N, training, direct):
    """persistent MLP conditioner, forward + backward, against autograd through the module stack evaluated on the CPU in
    FLOAT64 (exact-arithmetic yard-stick) -- bar: 2e-5 * max|.| + 4 x the distance of the same module stack in float32 on the
    CPU from that yard-stick (the cancellation in BatchNorm's backward is a property of the problem, measured here, not a
    hand-picked tolerance).  At most two rows may sit on the other side of a ReLU kink (pre-activation within rounding of 0)."""
    fused = importlib.import_module(pkg.__name__ + '.fused')
    ref, fus = _mlp_pair(pkg, in_ch, out_ch)
    fus.train(training)
    g = torch.Generator().manual_seed(N)
    x = torch.randn(N, in_ch, generator=g) * 0.7
    gout = torch.randn(N, out_ch, generator=g)
    cpu = {}
    for dt in (torch.float32, torch.float64):
        m = copy.deepcopy(ref).cpu().to(dt).train(training)
        xr = x.detach().clone().to(dt).requires_grad_(True)
        yr = m.forward_reference(xr)
        yr.backward(gout.to(dt))
        cpu[dt] = (yr.detach(), xr.grad.detach(), {k: p.grad.detach() for k, p in m.named_parameters()})
    y64, gx64, gp64 = cpu[torch.float64]
    y32, gx32, gp32 = cpu[torch.float32]

    def bar(want64, got32):
        return 2e-5 * max(1.0, float(want64.abs().max())) + 4.0 * float((got32.double() - want64).abs().max())

    if direct:
        for p in fus.parameters():
            p.grad = torch.zeros_like(p)
            p._nf_direct_grad = True
    xf = x.detach().clone().to(DEV).requires_grad_(True)
    yf = fused.mlp_forward(fus, xf, chain=True)
    assert float((yf.detach().cpu().double() - y64).abs().max()) <= bar(y64, y32), 'output'
    yf.backward(gout.to(DEV))
    err = (xf.grad.cpu().double() - gx64).abs().max(dim=1).values
    flipped = int((err > bar(gx64, gx32)).sum())
    assert flipped <= 2, 'grad input: %d rows beyond the bar, max abs err %.3e (bar %.3e)' % (flipped, float(err.max()), bar(gx64, gx32))
    pf = dict(fus.named_parameters())
    for k, want in gp64.items():
        assert pf[k].grad is not None, k
        pre_bn_bias = training and k.endswith('module.bias') and 'out_block' not in k
        tol = bar(want, gp32[k]) + flipped * 8.0 / N * max(1.0, float(want.abs().max()))
        if pre_bn_bias:
            tol = max(tol, 2e-3 + 1e-6 * N)                # analytically-zero gradients: rounding noise only
        e = float((pf[k].grad.cpu().double() - want).abs().max())
        assert e <= tol, 'grad %s: max abs err %.3e (bar %.3e, cpu32 itself %.3e)' % (k, e, tol, float((gp32[k].double() - want).abs().max()))


@pytest.mark.parametrize('direct', [False, True])
@pytest.mark.parametrize('D,odd,N', [(2, False, 4096), (2, True, 300), (4, False, 1000), (4, True, 77), (2, False, 16384)])
@pytest.mark.parametrize('training', [True, False])
def test_glow_step_vec_vs_unfused(pkg, D, odd, N, training, direct):
    """the one-launch Glow step (ActNorm -> 1x1 -> affine coupling + MLP) against the three-launch path it replaces:
    output, log-det, input gradient and every parameter gradient (with a log-det gradient in play)."""
    fused = importlib.import_module(pkg.__name__ + '.fused')
    NF = importlib.import_module(pkg.__name__ + '.functional')
    torch.manual_seed(D * 7 + int(odd))

    def make():
        torch.manual_seed(D * 7 + int(odd))
        a, c, k = pkg.ActNorm((D, )), pkg.InvertibleConv1x1(D), pkg.AffineCoupling((D, ), odd=odd)
        mods = torch.nn.ModuleList([a, c, k]).to(DEV)
        with torch.no_grad():
            a.log_scale.normal_(0, 0.2)
            a.bias.normal_(0, 0.3)
            k.s_log_scale.fill_(0.7)
            k.s_bias.fill_(0.1)
            for m in k.net.modules():
                if isinstance(m, torch.nn.BatchNorm1d):
                    m.weight.uniform_(0.5, 1.5)
                    m.bias.normal_(0, 0.3)
                    m.running_mean.normal_(0, 0.2)
                    m.running_var.uniform_(0.5, 2.0)
        a.initialized = True
        mods.train(training)
        return a, c, k, mods

    a1, c1, k1, m1 = make()
    a2, c2, k2, m2 = make()
    g = torch.Generator().manual_seed(N + D)
    z = (torch.randn(N, D, generator=g) * 0.8).to(DEV)
    gy = torch.randn(N, D, generator=g).to(DEV)
    wl = torch.randn(N, generator=g).to(DEV)
    z1, z2 = z.clone().requires_grad_(True), z.clone().requires_grad_(True)
    ld0 = torch.randn(N, generator=g).to(DEV)
    # reference: glow head + multi-launch conditioner + coupling kernel
    h, zc, ld1 = NF.glow_head(z1, ld0.clone(), a1.log_scale, a1.bias, c1.P, c1.L, c1.U, c1.L_mask, c1.U_mask, c1.sign_s,
                              c1.log_s, k1.mode, k1.odd)
    y1, ld1 = NF.affine_coupling(h, fused.mlp_forward(k1.net, zc, chain=False), k1.s_log_scale, k1.s_bias, ld1, k1.mode,
                                 k1.odd)
    ((y1 * gy).sum() + (ld1 * wl).sum()).backward()
    if direct:
        for p in m2.parameters():
            if p.requires_grad:
                p.grad = torch.zeros_like(p)
                p._nf_direct_grad = True
    assert fused.glow_step_vec_usable(z2, k2.net)
    y2, ld2 = fused.glow_step_vec(z2, ld0.clone(), a2, c2, k2)
    G.assert_close(y2, y1, 2e-5, rtol=2e-5, what='y')
    G.assert_close(ld2, ld1, 2e-5, rtol=2e-5, what='log-det')
    ((y2 * gy).sum() + (ld2 * wl).sum()).backward()
    # 16384 rows x 160 ReLU units: now and then ONE pre-activation sits within rounding of zero and the two paths (whose
    # sums are ordered differently) mask it differently -- a legitimate discontinuity, not an error.  Tolerate two such rows.
    big = N >= 16384
    err = (z2.grad - z1.grad).abs().max(dim=1).values
    bad = int((err > _grad_tol(z1.grad)).sum())
    assert bad <= (2 if big else 0), 'grad z: %d rows beyond tolerance, max abs err %.3e' % (bad, float(err.max()))
    p1, p2 = dict(m1.named_parameters()), dict(m2.named_parameters())
    for name, p in p1.items():
        if not p.requires_grad:
            continue
        assert p2[name].grad is not None, name
        pre_bn_bias = training and name.endswith('module.bias') and 'out_block' not in name
        # a bias in front of a BatchNorm has gradient exactly 0; both paths return fp32 cancellation noise of N-term sums
        tol = 5e-3 + 3e-6 * N if pre_bn_bias else _grad_tol(p.grad)
        if big:
            tol = max(tol, 1e-2 * float(p.grad.abs().max()))
        G.assert_close(p2[name].grad, p.grad, tol, what='grad ' + name)
    b1, b2 = dict(m1.named_buffers()), dict(m2.named_buffers())
    for name in b1:
        G.assert_close(b2[name].float(), b1[name].float(), 2e-6, rtol=1e-5, what='buffer ' + name)


def test_zz_persistent_kernels_never_timed_out(pkg):
    """runs last in this file: no bounded spin loop of the persistent kernels gave up during the tests above."""
    assert pkg._native.persistent_timeouts() == 0


@pytest.mark.parametrize('direct', [False, True])
@pytest.mark.parametrize('D,N', [(2, 4096), (2, 300), (4, 1000), (2, 16384), (3, 77)])
def test_maf_step_vec_vs_unfused(pkg, D, N, direct):
    """the one-launch MAF step (flow BatchNorm -> perm -> MADE pair -> affine transform) against the per-layer path."""
    fused = importlib.import_module(pkg.__name__ + '.fused')
    NF = importlib.import_module(pkg.__name__ + '.functional')

    def make():
        torch.manual_seed(D * 11)
        np.random.seed(5)
        bn = pkg.BatchNorm((D, ), affine=False)
        ar = pkg.AutoregressiveTransfrom(D)
        mods = torch.nn.ModuleList([bn, ar]).to(DEV)
        with torch.no_grad():
            bn.running_mean.normal_(0, 0.2)
            bn.running_var.uniform_(0.5, 2.0)
            ar.s_log_scale.fill_(0.6)
            ar.s_bias.fill_(0.1)
            for m in ar.modules():
                if isinstance(m, torch.nn.BatchNorm1d):
                    m.weight.uniform_(0.5, 1.5)
                    m.bias.normal_(0, 0.3)
                    m.running_mean.normal_(0, 0.2)
                    m.running_var.uniform_(0.5, 2.0)
        mods.train(True)
        return bn, ar, mods

    bn1, ar1, m1 = make()
    bn2, ar2, m2 = make()
    g = torch.Generator().manual_seed(N + D)
    z = (torch.randn(N, D, generator=g) * 0.8 + 0.3).to(DEV)
    gy = torch.randn(N, D, generator=g).to(DEV)
    wl = torch.randn(N, generator=g).to(DEV)
    ld0 = torch.randn(N, generator=g).to(DEV)
    z1, z2 = z.clone().requires_grad_(True), z.clone().requires_grad_(True)
    np.random.seed(9)                                        # MADE masks are drawn from the global numpy RNG
    h, ld1 = NF.flowbn_head(z1, ld0.clone(), bn1)
    y1, ld1 = ar1(h, ld1)
    ((y1 * gy).sum() + (ld1 * wl).sum()).backward()
    if direct:
        for p in m2.parameters():
            if p.requires_grad:
                p.grad = torch.zeros_like(p)
                p._nf_direct_grad = True
    assert fused.maf_step_usable(z2, bn2, ar2)
    np.random.seed(9)
    y2, ld2 = fused.maf_step_vec(z2, ld0.clone(), bn2, ar2)
    G.assert_close(y2, y1, 2e-5, rtol=2e-5, what='y')
    G.assert_close(ld2, ld1, 2e-5, rtol=2e-5, what='log-det')
    ((y2 * gy).sum() + (ld2 * wl).sum()).backward()
    big = N >= 16384                                         # see test_glow_step_vec_vs_unfused: rare ReLU-mask flips
    err = (z2.grad - z1.grad).abs().max(dim=1).values
    bad = int((err > _grad_tol(z1.grad)).sum())
    assert bad <= (2 if big else 0), 'grad z: %d rows beyond tolerance, max abs err %.3e' % (bad, float(err.max()))
    p1, p2 = dict(m1.named_parameters()), dict(m2.named_parameters())
    for name, p in p1.items():
        if not p.requires_grad:
            continue
        assert p2[name].grad is not None, name
        pre_bn_bias = '.biases.' in name and not name.endswith('.biases.3')
        # a bias in front of a BatchNorm has gradient exactly 0; both paths return fp32 cancellation noise of N-term sums
        tol = 5e-3 + 3e-6 * N if pre_bn_bias else _grad_tol(p.grad)
        if big:
            tol = max(tol, 1e-2 * float(p.grad.abs().max()))
        G.assert_close(p2[name].grad, p.grad, tol, what='grad ' + name)
    b1, b2 = dict(m1.named_buffers()), dict(m2.named_buffers())
    for name in b1:
        G.assert_close(b2[name].float(), b1[name].float(), 2e-6, rtol=1e-5, what='buffer ' + name)


@pytest.mark.parametrize('gather', [True, False])
@pytest.mark.parametrize('shape,mode_name', [((64, 3, 32, 32), 'checker'), ((64, 12, 16, 16), 'channel'), ((64, 48, 4, 4), 'channel'),
                                             ((5, 6, 16, 16), 'checker'), ((1, 24, 8, 8), 'channel'), ((64, 24, 8, 8), 'checker')])
def test_flowbn_head_one_persistent_launch(pkg, monkeypatch, shape, mode_name, gather):
    """the flow BatchNorm head of image data (statistics + normalise + log-det + running buffers + conditioning-half gather) in ONE
    persistent launch (k_flowbn_head_fused: the workgroups exchange their per-channel sums) against the two launches it replaces and
    against float64: outputs to 1e-5, statistics to 1e-6 relative, the same bits from two runs (no float atomics), backward intact."""
    NF = importlib.import_module(pkg.__name__ + '.functional')
    Nn = pkg._native
    mode = Nn.SPLIT_CHECKER if mode_name == 'checker' else Nn.SPLIT_CHANNEL
    B, C, H, W = shape
    assert int(Nn.load().nf_flowbn_head_fused_ws_floats(B, C, H, W)) > 0
    torch.manual_seed(B * 7 + C)
    x = (torch.randn(shape, device=DEV) * 1.3 + 0.4 * torch.arange(C, device=DEV).view(1, C, 1, 1) / C)
    ld0 = torch.randn(B, device=DEV)

    def run(fused_on):
        monkeypatch.setattr(NF, 'FLOWBN_FUSED', fused_on)
        torch.manual_seed(1)
        bn = pkg.BatchNorm((C, H, W), affine=False).to(DEV).train()
        with torch.no_grad():
            bn.running_mean.normal_(0, 0.2)
            bn.running_var.uniform_(0.5, 2.0)
        xi = x.clone().requires_grad_(True)
        out = NF.flowbn_head(xi, ld0.clone(), bn, mode, False, gather=gather)
        y, ld = out[0], out[-1]
        loss = (y * torch.linspace(-1, 1, y.numel(), device=DEV).view_as(y)).sum() + ld.sum()
        if gather:
            loss = loss + (out[1] ** 2).sum() * 0.5
        loss.backward()
        torch.cuda.synchronize()
        return [t.detach().clone() for t in out] + [xi.grad.clone(), bn.running_mean.clone(), bn.running_var.clone(), bn.batch_mean.clone(), bn.batch_var.clone()]

    a, a2, b = run(True), run(True), run(False)
    for u, v in zip(a, a2):
        assert torch.equal(u, v)                         # (fixed-order sums: bit-reproducible without the ordered mode)
    names = (['y', 'z1c', 'ld'] if gather else ['y', 'ld']) + ['grad x', 'running_mean', 'running_var', 'batch_mean', 'batch_var']
    for n, u, v in zip(names, a, b):
        G.assert_close(u, v, 2e-5, rtol=2e-5, what=n)
    xd = x.double()
    mean, var = xd.mean(dim=(0, 2, 3)), xd.var(dim=(0, 2, 3), unbiased=False)
    G.assert_close(a[-2].reshape(-1), mean.float(), 1e-6, rtol=1e-6, what='batch mean vs float64')
    assert float(((a[-1].double().reshape(-1) - (var + 1e-5)) / var).abs().max()) < 2e-6      # (eps inside: modules.py:287)
    assert Nn.persistent_timeouts() == 0


@pytest.mark.parametrize('direct', [False, True])
@pytest.mark.parametrize('D,odd,N', [(2, False, 256), (2, True, 300), (4, False, 1000), (4, True, 77), (2, False, 4096)])
def test_realnvp_step_vec_vs_unfused(pkg, D, odd, N, direct):
    """the one-launch RealNVP step (flow BatchNorm -> affine coupling + MLP) against the path it replaces."""
    fused = importlib.import_module(pkg.__name__ + '.fused')
    NF = importlib.import_module(pkg.__name__ + '.functional')

    def make():
        torch.manual_seed(D * 13 + int(odd))
        bn, k = pkg.BatchNorm((D, ), affine=False), pkg.AffineCoupling((D, ), odd=odd)
        mods = torch.nn.ModuleList([bn, k]).to(DEV)
        with torch.no_grad():
            bn.running_mean.normal_(0, 0.2)
            bn.running_var.uniform_(0.5, 2.0)
            k.s_log_scale.fill_(0.7)
            k.s_bias.fill_(0.1)
            for m in k.net.modules():
                if isinstance(m, torch.nn.BatchNorm1d):
                    m.weight.uniform_(0.5, 1.5)
                    m.bias.normal_(0, 0.3)
                    m.running_mean.normal_(0, 0.2)
                    m.running_var.uniform_(0.5, 2.0)
        mods.train(True)
        return bn, k, mods

    bn1, k1, m1 = make()
    bn2, k2, m2 = make()
    g = torch.Generator().manual_seed(N + D)
    z = (torch.randn(N, D, generator=g) * 0.8 + 0.2).to(DEV)
    gy = torch.randn(N, D, generator=g).to(DEV)
    wl = torch.randn(N, generator=g).to(DEV)
    ld0 = torch.randn(N, generator=g).to(DEV)
    z1, z2 = z.clone().requires_grad_(True), z.clone().requires_grad_(True)
    h, zc, ld1 = NF.flowbn_head(z1, ld0.clone(), bn1, k1.mode, k1.odd, gather=True)
    y1, ld1 = NF.affine_coupling(h, fused.mlp_forward(k1.net, zc, chain=False), k1.s_log_scale, k1.s_bias, ld1, k1.mode, k1.odd)
    ((y1 * gy).sum() + (ld1 * wl).sum()).backward()
    if direct:
        for p in m2.parameters():
            if p.requires_grad:
                p.grad = torch.zeros_like(p)
                p._nf_direct_grad = True
    assert fused.realnvp_step_vec_usable(z2, bn2, k2.net)
    y2, ld2 = fused.realnvp_step_vec(z2, ld0.clone(), bn2, k2)
    G.assert_close(y2, y1, 2e-5, rtol=2e-5, what='y')
    G.assert_close(ld2, ld1, 2e-5, rtol=2e-5, what='log-det')
    ((y2 * gy).sum() + (ld2 * wl).sum()).backward()
    G.assert_close(z2.grad, z1.grad, _grad_tol(z1.grad), what='grad z')
    p1, p2 = dict(m1.named_parameters()), dict(m2.named_parameters())
    for name, p in p1.items():
        if not p.requires_grad:
            continue
        assert p2[name].grad is not None, name
        pre_bn_bias = name.endswith('module.bias') and 'out_block' not in name
        # a bias in front of a BatchNorm has gradient exactly 0; both paths return fp32 cancellation noise of N-term sums
        tol = 5e-3 + 3e-6 * N if pre_bn_bias else _grad_tol(p.grad)
        G.assert_close(p2[name].grad, p.grad, tol, what='grad ' + name)
    b1, b2 = dict(m1.named_buffers()), dict(m2.named_buffers())
    for name in b1:
        G.assert_close(b2[name].float(), b1[name].float(), 2e-6, rtol=1e-5, what='buffer ' + name)


@pytest.mark.parametrize('mode', [True, 'steps'])
@pytest.mark.parametrize('D,B,K', [(2, 4096, 8), (2, 300, 3), (4, 1000, 4), (2, 16384, 2)])
def test_glow_flow_vec_matches_step_by_step(pkg, D, B, K, mode, monkeypatch):
    """the whole run of vector Glow steps as one autograd node -- one launch per direction (k_glow_flow_fwd / _bwd, mode True)
    or one launch per step with the gradient folds of all steps deferred to one launch (nf_glow_flow_steps_*, mode 'steps')
    -- against the same steps launched one by one: outputs, log-det, every gradient in the flat bucket, BatchNorm running
    statistics."""
    from types import SimpleNamespace as NS
    train = importlib.import_module(pkg.__name__ + '.train')
    fused = importlib.import_module(pkg.__name__ + '.fused')
    torch.manual_seed(D * 1000 + B)
    net1 = pkg.Glow((D, ), 'density', NS(layers=K, mixtures=8)).to(DEV)
    net2 = copy.deepcopy(net1)
    y = (torch.randn(B, D) * 0.7).to(DEV)
    t1, t2 = train.FlowTrainer(net1, graph=False), train.FlowTrainer(net2, graph=False)
    calls = {'n': 0}
    real = fused.glow_flow_vec
    monkeypatch.setattr(fused, 'GLOW_FLOW', mode)           # opt-in paths (NF_GLOW_FLOW=1 / =steps)

    def counted(*a, **k):
        calls['n'] += 1
        return real(*a, **k)

    for step in range(3):                                   # step 0 initialises the ActNorms (no whole-flow launch yet)
        monkeypatch.setattr(fused, 'glow_flow_vec', counted)
        monkeypatch.setattr(fused, 'glow_flow_vec_usable', fused.__dict__['glow_flow_vec_usable'])
        t1.net.train()
        z1, l1 = t1._forward_backward(y)
        n_flow = calls['n']
        monkeypatch.setattr(fused, 'glow_flow_vec_usable', lambda z, steps: False)
        t2.net.train()
        z2, l2 = t2._forward_backward(y)
        monkeypatch.undo()
        monkeypatch.setattr(fused, 'GLOW_FLOW', mode)
        G.assert_close(z1, z2, 2e-5, rtol=2e-5, what='z, step %d' % step)
        G.assert_close(l1, l2, 2e-5, rtol=2e-5, what='loss, step %d' % step)
        scale = float(t2.bucket.flat.abs().max())
        bad = ((t1.bucket.flat - t2.bucket.flat).abs() > 1e-4 * max(1.0, scale)).float().mean()
        # (the biases in front of a BatchNorm have gradient 0: both paths return cancellation noise there; 16384 rows: ReLU flips)
        assert float(bad) <= (5e-3 if B >= 16384 else 1e-3), 'flat gradients differ in %.2e of the entries (step %d)' % (float(bad), step)
        G.assert_close(t1.bucket.flat, t2.bucket.flat, (5e-2 if B >= 16384 else 1e-2) * max(1.0, scale), what='flat grads, step %d' % step)
        b1, b2 = dict(net1.named_buffers()), dict(net2.named_buffers())
        for name in b2:
            G.assert_close(b1[name].float(), b2[name].float(), 2e-6, rtol=1e-5, what='buffer ' + name)
        t1.optim.step()
        # Adam turns noise-level gradient differences (the atomics of the step-0 ActNorm init) into lr-sized parameter
        # differences: keep the two replicas identical instead of comparing two diverging trainings
        net2.load_state_dict(net1.state_dict())
        t2.bucket.flat_params.copy_(t1.bucket.flat_params)
    assert n_flow >= 2, 'the whole-flow launch was never taken'
    assert fused.N.persistent_timeouts() == 0


@pytest.mark.parametrize('training', [False, True])
@pytest.mark.parametrize('D,B,K', [(2, 4096, 8), (4, 333, 3), (2, 16384, 2)])
def test_glow_flow_nograd_matches_step_by_step(pkg, D, B, K, training, monkeypatch):
    """density evaluation under no_grad: the run of vector Glow steps in one launch (evaluation-mode statistics: no exchanges;
    training-mode statistics: small batches) against one launch per step."""
    from types import SimpleNamespace as NS
    fused = importlib.import_module(pkg.__name__ + '.fused')
    torch.manual_seed(D * 10 + B)
    net1 = pkg.Glow((D, ), 'density', NS(layers=K, mixtures=8)).to(DEV)
    with torch.no_grad():
        net1.train()
        net1((torch.randn(max(B, 64), D) * 0.8).to(DEV))
    net2 = copy.deepcopy(net1)
    net1.train(training)
    net2.train(training)
    y = (torch.randn(B, D) * 0.9).to(DEV)
    calls = {'n': 0}
    real = fused.glow_flow_vec_nograd

    def counted(*a, **k):
        calls['n'] += 1
        return real(*a, **k)

    monkeypatch.setattr(fused, 'glow_flow_vec_nograd', counted)
    with torch.no_grad():
        z1, l1 = net1(y)
        monkeypatch.setattr(fused, 'GLOW_FLOW', '0')
        z2, l2 = net2(y)
    expect = (not training) or B <= fused.GLOW_FLOW_AUTO_ROWS
    assert (calls['n'] >= 1) == expect
    G.assert_close(z1, z2, 1e-5, rtol=1e-5, what='z')
    G.assert_close(l1, l2, 1e-5, rtol=1e-5, what='log-det')
    b1, b2 = dict(net1.named_buffers()), dict(net2.named_buffers())
    for name in b2:
        G.assert_close(b1[name].float(), b2[name].float(), 2e-6, rtol=1e-5, what='buffer ' + name)
    assert fused.N.persistent_timeouts() == 0


@pytest.mark.parametrize('training', [False, True])
@pytest.mark.parametrize('D,B,K', [(2, 4096, 8), (2, 300, 3), (4, 1000, 4), (2, 16384, 2), (4, 7, 1)])
def test_glow_inverse_vec_matches_layerwise(pkg, D, B, K, training, monkeypatch):
    """the inverse of a run of vector Glow steps -- one launch per step, or one for the whole run (k_mlp_chain_fwd<1, true>,
    k_glow_flow_inv) -- against the layer-by-layer inverse (ActNorm^-1, lu_solve-style 1x1^-1, coupling^-1 with its MLP):
    samples, log-det, the conditioner's running statistics; and forward(inverse(y)) = y."""
    from types import SimpleNamespace as NS
    fused = importlib.import_module(pkg.__name__ + '.fused')
    torch.manual_seed(D * 100 + B)
    net1 = pkg.Glow((D, ), 'density', NS(layers=K, mixtures=8)).to(DEV)
    with torch.no_grad():
        net1.train()
        net1((torch.randn(max(B, 64), D) * 0.8).to(DEV))     # data-dependent ActNorm init + some running statistics
        for p in net1.parameters():
            if p.requires_grad:
                p.add_(torch.randn_like(p) * 0.05)           # away from the identity-like initialisation
    net2 = copy.deepcopy(net1)
    net1.train(training)
    net2.train(training)
    y = (torch.randn(B, D) * 0.9).to(DEV)
    calls = {'n': 0}
    real = fused.glow_flow_vec_inverse

    def counted(*a, **k):
        calls['n'] += 1
        return real(*a, **k)

    monkeypatch.setattr(fused, 'glow_flow_vec_inverse', counted)
    with torch.no_grad():
        z1, l1 = net1.backward(y.clone())
        monkeypatch.setattr(fused, 'GLOW_INVERSE', False)
        z2, l2 = net2.backward(y.clone())
    assert calls['n'] >= 1, 'the fused inverse was not taken'
    G.assert_close(z1, z2, 2e-5, rtol=2e-5, what='inverse samples')
    G.assert_close(l1, l2, 2e-5, rtol=2e-5, what='inverse log-det')
    b1, b2 = dict(net1.named_buffers()), dict(net2.named_buffers())
    for name in b2:
        G.assert_close(b1[name].float(), b2[name].float(), 2e-6, rtol=1e-5, what='buffer ' + name)
    if not training:
        with torch.no_grad():
            yy, lf = net1(z1)
        G.assert_close(yy, y, 1e-4, rtol=1e-4, what='forward(inverse(y))')
        G.assert_close(lf + l1, torch.zeros_like(lf), 1e-4 * K, what='log-det of the round trip')
    assert fused.N.persistent_timeouts() == 0


@pytest.mark.parametrize('training', [False, True])
@pytest.mark.parametrize('D,B,K', [(2, 256, 8), (4, 1000, 3), (2, 4096, 2), (2, 5, 1)])
def test_realnvp_eval_and_inverse_vec_match_layerwise(pkg, D, B, K, training, monkeypatch):
    """RealNVP runs [flow BatchNorm, AffineCoupling] on vector data: the inverse (one launch per run or per step; training mode:
    batch buffers + batch-statistics conditioner, evaluation mode: running statistics) and the evaluation-mode forward under
    no_grad (one launch, no exchange) against the layer-by-layer path."""
    from types import SimpleNamespace as NS
    fused = importlib.import_module(pkg.__name__ + '.fused')
    torch.manual_seed(D * 100 + B + 3)
    net1 = pkg.RealNVP((D, ), 'density', NS(layers=K, mixtures=8)).to(DEV)
    with torch.no_grad():
        net1.train()
        for _ in range(2):
            net1((torch.randn(max(B, 64), D) * 0.8 + 0.1).to(DEV))          # batch buffers and running statistics off their defaults
        for p in net1.parameters():
            p.add_(torch.randn_like(p) * 0.05)
    net2 = copy.deepcopy(net1)
    net1.train(training)
    net2.train(training)
    y = (torch.randn(B, D) * 0.9).to(DEV)
    calls = {'inv': 0, 'fwd': 0}
    real_inv, real_fwd = fused.realnvp_flow_vec_inverse, fused.realnvp_flow_vec_eval

    def c_inv(*a, **k):
        calls['inv'] += 1
        return real_inv(*a, **k)

    def c_fwd(*a, **k):
        calls['fwd'] += 1
        return real_fwd(*a, **k)

    monkeypatch.setattr(fused, 'realnvp_flow_vec_inverse', c_inv)
    monkeypatch.setattr(fused, 'realnvp_flow_vec_eval', c_fwd)
    with torch.no_grad():
        z1, l1 = net1.backward(y.clone())
        f1, lf1 = net1(y) if not training else (None, None)
        monkeypatch.setattr(fused, 'GLOW_INVERSE', False)
        z2, l2 = net2.backward(y.clone())
        f2, lf2 = net2(y) if not training else (None, None)
    assert calls['inv'] >= 1, 'the fused inverse was not taken'
    G.assert_close(z1, z2, 2e-5, rtol=2e-5, what='inverse samples')
    G.assert_close(l1, l2, 2e-5, rtol=2e-5, what='inverse log-det')
    if not training:
        assert calls['fwd'] >= 1, 'the one-launch evaluation forward was not taken'
        G.assert_close(f1, f2, 1e-5, rtol=1e-5, what='evaluation forward')
        G.assert_close(lf1, lf2, 1e-5, rtol=1e-5, what='evaluation log-det')
        with torch.no_grad():
            yy, lr = net1(z1)
        G.assert_close(yy, y, 1e-4, rtol=1e-4, what='forward(inverse(y))')
        G.assert_close(lr + l1, torch.zeros_like(lr), 1e-4 * K, what='log-det of the round trip')
    b1, b2 = dict(net1.named_buffers()), dict(net2.named_buffers())
    for name in b2:
        G.assert_close(b1[name].float(), b2[name].float(), 2e-6, rtol=1e-5, what='buffer ' + name)
    assert fused.N.persistent_timeouts() == 0


@pytest.mark.parametrize('training', [False, True])
@pytest.mark.parametrize('D,B,K', [(2, 16384, 3), (2, 300, 5), (1, 1000, 2), (3, 4096, 2), (2, 9, 1)])
def test_maf_eval_and_inverse_step_match_layerwise(pkg, D, B, K, training, monkeypatch):
    """MAF steps [flow BatchNorm, AutoregressiveTransfrom]: the inverse in one launch per step (D <= 2: D sequential MADE passes
    inside the kernel; training mode: batch statistics and one running-statistics update per pass) and the evaluation-mode
    forward under no_grad (one launch, no exchange) against the layer-by-layer path."""
    from types import SimpleNamespace as NS
    import numpy as np
    fused = importlib.import_module(pkg.__name__ + '.fused')
    torch.manual_seed(D * 100 + B + 5)
    np.random.seed(D + B)
    net1 = pkg.MAF((D, ), 'density', NS(layers=K, mixtures=8)).to(DEV)
    with torch.no_grad():
        net1.train()
        for _ in range(2):
            net1((torch.randn(max(B, 64), D) * 0.8 + 0.1).to(DEV))
        for p in net1.parameters():
            p.add_(torch.randn_like(p) * 0.05)
    net2 = copy.deepcopy(net1)
    net1.train(training)
    net2.train(training)
    y = (torch.randn(B, D) * 0.9).to(DEV)
    calls = {'inv': 0, 'fwd': 0}
    real_inv, real_fwd = fused.maf_step_inverse, fused.maf_step_eval

    def c_inv(*a, **k):
        calls['inv'] += 1
        return real_inv(*a, **k)

    def c_fwd(*a, **k):
        calls['fwd'] += 1
        return real_fwd(*a, **k)

    monkeypatch.setattr(fused, 'maf_step_inverse', c_inv)
    monkeypatch.setattr(fused, 'maf_step_eval', c_fwd)
    with torch.no_grad():
        np.random.seed(1)
        z1, l1 = net1.backward(y.clone())
        f1, lf1 = net1(y) if not training else (None, None)
        monkeypatch.setattr(fused, 'GLOW_INVERSE', False)
        np.random.seed(1)
        z2, l2 = net2.backward(y.clone())
        f2, lf2 = net2(y) if not training else (None, None)
    assert (calls['inv'] == K) == (D <= 2), 'fused inverse steps taken: %d' % calls['inv']
    G.assert_close(z1, z2, 2e-5, rtol=2e-5, what='inverse samples')
    G.assert_close(l1, l2, 2e-5, rtol=2e-5, what='inverse log-det')
    if not training:
        assert calls['fwd'] == K, 'the one-launch evaluation forward was not taken'
        G.assert_close(f1, f2, 1e-5, rtol=1e-5, what='evaluation forward')
        G.assert_close(lf1, lf2, 1e-5, rtol=1e-5, what='evaluation log-det')
    b1, b2 = dict(net1.named_buffers()), dict(net2.named_buffers())
    for name in b2:
        G.assert_close(b1[name].float(), b2[name].float(), 2e-6, rtol=1e-5, what='buffer ' + name)
    assert fused.N.persistent_timeouts() == 0


@pytest.mark.parametrize('B,K,mix', [(65536, 11, 8), (1000, 3, 4), (40000, 9, 8)])
def test_flowpp_deferred_finalize_matches_per_step(pkg, B, K, mix, monkeypatch):
    """the trainer defers the slab finalizes of the fused Flow++ steps to one launch per eight steps after backward
    (nf_flowpp_vec_step_bwd phase 1 + nf_flowpp_vec_step_finalize): same gradients as the per-step finalize launches."""
    from types import SimpleNamespace as NS
    train = importlib.import_module(pkg.__name__ + '.train')
    fused = importlib.import_module(pkg.__name__ + '.fused')
    torch.manual_seed(B + K)
    net1 = pkg.Flowpp((2, ), 'density', NS(layers=K, mixtures=mix)).to(DEV)
    net2 = copy.deepcopy(net1)
    y = (torch.randn(B, 2) * 0.7).to(DEV)
    t1, t2 = train.FlowTrainer(net1, graph=False), train.FlowTrainer(net2, graph=False)
    queued = []
    real_flush = fused.FlowppDefer.flush

    def flush(self):
        queued.append(len(self.queue))
        return real_flush(self)

    monkeypatch.setattr(fused.FlowppDefer, 'flush', flush)
    for step in range(3):                                   # step 0 initialises the ActNorms
        monkeypatch.setattr(fused, 'FLOWPP_DEFER', True)
        z1, l1 = t1._forward_backward(y)
        monkeypatch.setattr(fused, 'FLOWPP_DEFER', False)
        z2, l2 = t2._forward_backward(y)
        assert not fused.FPP_DEFER.queue and not fused.FPP_DEFER.active
        if step > 0:   # (step 0: the data-dependent ActNorm init sums by atomics, the replicas differ at noise level until synced)
            G.assert_close(z1, z2, 1e-6, rtol=1e-6, what='z, step %d' % step)
            G.assert_close(l1, l2, 1e-6, rtol=1e-6, what='loss, step %d' % step)
            # (identical kernels and slabs; the finalize adds by float atomics in either form)
            G.assert_close(t1.bucket.flat, t2.bucket.flat, 2e-5 * max(1.0, float(t2.bucket.flat.abs().max())), what='flat grads')
        t1.optim.step()
        net2.load_state_dict(net1.state_dict())
        t2.bucket.flat_params.copy_(t1.bucket.flat_params)
    assert max(queued) == K, 'the finalizes were not deferred (queue lengths %r)' % (queued, )


@pytest.mark.parametrize('D,B,K', [(2, 16384, 3), (4, 1000, 4), (2, 100, 18), (3, 4096, 2)])
def test_maf_flow_vec_matches_step_by_step(pkg, D, B, K, monkeypatch):
    """a run of MAF steps as one autograd node whose backward defers the gradient folds of all steps to one launch
    (nf_maf_step_bwd_partial + nf_maf_fold_all; 18 steps: two fold launches) against the same steps one by one: outputs, loss,
    flat gradients, flow-BatchNorm and BatchNorm1d buffers."""
    from types import SimpleNamespace as NS
    import numpy as np
    train = importlib.import_module(pkg.__name__ + '.train')
    fused = importlib.import_module(pkg.__name__ + '.fused')
    torch.manual_seed(D * 1000 + B)
    net1 = pkg.MAF((D, ), 'density', NS(layers=K, mixtures=8)).to(DEV)
    net2 = copy.deepcopy(net1)
    y = (torch.randn(B, D) * 0.7).to(DEV)
    t1, t2 = train.FlowTrainer(net1, graph=False), train.FlowTrainer(net2, graph=False)
    calls = {'n': 0}
    real = fused.maf_flow_vec

    def counted(*a, **k):
        calls['n'] += 1
        return real(*a, **k)

    for step in range(2):
        monkeypatch.setattr(fused, 'MAF_FLOW', True)
        monkeypatch.setattr(fused, 'maf_flow_vec', counted)
        t1.net.train()
        np.random.seed(step)                                # the MADE masks are drawn with numpy's RNG on every forward
        z1, l1 = t1._forward_backward(y)
        monkeypatch.setattr(fused, 'MAF_FLOW', False)
        t2.net.train()
        np.random.seed(step)
        z2, l2 = t2._forward_backward(y)
        monkeypatch.undo()
        G.assert_close(z1, z2, 1e-6, rtol=1e-6, what='z, step %d' % step)
        G.assert_close(l1, l2, 1e-6, rtol=1e-6, what='loss, step %d' % step)
        # (same step bodies; both folds add partial sums by float atomics, whose order is not fixed)
        G.assert_close(t1.bucket.flat, t2.bucket.flat, 2e-5 * max(1.0, float(t2.bucket.flat.abs().max())), what='flat grads')
        b1, b2 = dict(net1.named_buffers()), dict(net2.named_buffers())
        for name in b2:
            G.assert_close(b1[name].float(), b2[name].float(), 1e-6, rtol=1e-6, what='buffer ' + name)
        t1.optim.step()
        net2.load_state_dict(net1.state_dict())
        t2.bucket.flat_params.copy_(t1.bucket.flat_params)
    assert calls['n'] == 2, 'the deferred-fold run was not taken'
    assert fused.N.persistent_timeouts() == 0


@pytest.mark.parametrize('mode', [True, 'steps'])
@pytest.mark.parametrize('D,B,K', [(2, 256, 8), (4, 1000, 3), (2, 4096, 2), (2, 16384, 3)])
def test_realnvp_flow_vec_matches_step_by_step(pkg, D, B, K, mode, monkeypatch):
    """a run of vector RealNVP steps as one autograd node -- one launch per direction (mode True), or one launch per step with
    the gradient folds of all steps deferred to one launch (nf_realnvp_flow_steps_*, mode 'steps') -- against the same steps
    launched one by one (bit-identical step bodies): outputs, loss, flat gradients, flow-BatchNorm and BatchNorm1d buffers."""
    if mode == 'steps' and B <= 256:
        pytest.skip('one or two workgroups fold without a barrier: nothing to defer')
    from types import SimpleNamespace as NS
    train = importlib.import_module(pkg.__name__ + '.train')
    fused = importlib.import_module(pkg.__name__ + '.fused')
    # (seed 2256 puts one pre-activation of the fifth step within rounding of its ReLU kink: the one-workgroup kernel and the step kernels
    # then differ by a mask decision -- percent-level gradients from there on, tools/probes/solo_dbg.py; another draw for that case)
    torch.manual_seed(D * 1000 + B + (1 if (mode is True and D == 2 and B <= 256) else 0))
    net1 = pkg.RealNVP((D, ), 'density', NS(layers=K, mixtures=8)).to(DEV)
    net2 = copy.deepcopy(net1)
    y = (torch.randn(B, D) * 0.7).to(DEV)
    t1, t2 = train.FlowTrainer(net1, graph=False), train.FlowTrainer(net2, graph=False)
    calls = {'n': 0}
    real = fused.realnvp_flow_vec

    def counted(*a, **k):
        calls['n'] += 1
        return real(*a, **k)

    for step in range(2):
        monkeypatch.setattr(fused, 'GLOW_FLOW', mode)
        monkeypatch.setattr(fused, 'realnvp_flow_vec', counted)
        t1.net.train()
        z1, l1 = t1._forward_backward(y)
        monkeypatch.setattr(fused, 'GLOW_FLOW', '0')
        t2.net.train()
        z2, l2 = t2._forward_backward(y)
        monkeypatch.undo()
        # B <= 256, D = 2: the whole-flow launch is the one-workgroup kernel of csrc/flow_solo.hip (transposed tiles, another summation
        # order in every product and statistic) against the 16-row-tile step kernels: fp32 rounding through K steps, not bit identity
        solo = mode is True and D == 2 and B <= 256
        tz = 5e-5 if solo else 1e-6
        G.assert_close(z1, z2, tz, rtol=tz, what='z, step %d' % step)
        G.assert_close(l1, l2, tz, rtol=tz, what='loss, step %d' % step)
        # (same step bodies; with <= 32 workgroups the fold adds by float atomics, whose order is not fixed)
        G.assert_close(t1.bucket.flat, t2.bucket.flat, (1e-5 if mode is True else 5e-5) * (20.0 if solo else 1.0) * max(1.0, float(t2.bucket.flat.abs().max())), what='flat grads')
        b1, b2 = dict(net1.named_buffers()), dict(net2.named_buffers())
        for name in b2:
            G.assert_close(b1[name].float(), b2[name].float(), 2e-5 if solo else 1e-6, rtol=2e-5 if solo else 1e-6, what='buffer ' + name)
        t1.optim.step()
        net2.load_state_dict(net1.state_dict())
        t2.bucket.flat_params.copy_(t1.bucket.flat_params)
    assert calls['n'] == 2, 'the whole-flow launch was not taken'
    assert fused.N.persistent_timeouts() == 0



@pytest.mark.parametrize('mode', [1, 3])
@pytest.mark.parametrize('B,K,seed', [(256, 4, 11), (256, 6, 12), (100, 3, 13), (17, 2, 14), (64, 2, 15)])
def test_realnvp_one_workgroup_kernels_match_the_grid_kernels(pkg, B, K, seed, mode):
    """csrc/flow_solo.hip (the whole batch of a 2-D RealNVP run in ONE workgroup: features x batch tiles, sums over the batch through LDS
    transposition tiles, per-wave weight-gradient partials) in its two settings -- forward only, both directions (the default since
    round 5: the backward reads the BatchNorm inputs the forward stashed, so it cannot run behind the grid forward) --
    against the grid kernels of csrc/mlp_chain.hip on the same weights and batch: outputs, loss, every gradient, the BatchNorm1d and
    flow-BatchNorm buffers; full batches, ragged ones (100 = three full waves + 4 columns, 17) and one that leaves waves empty.  Short
    runs: two fp32 paths drift apart by the same factor per step as either does from float64 (1e-3 at 32 steps, tools/probes/
    solo_dbg.py: the one-workgroup forward is the closer one); the full-depth case is tests/test_gpu_fullsize_parity.py's, against the oracle."""
    from types import SimpleNamespace as NS
    train = importlib.import_module(pkg.__name__ + '.train')
    N = pkg._native
    torch.manual_seed(seed)
    net0 = pkg.RealNVP((2, ), 'density', NS(layers=K, mixtures=8)).to(DEV)
    y = (torch.randn(B, 2) * 0.7).to(DEV)
    outs = []
    try:
        for m in (0, mode):
            N.call('nf_flow_solo_config', m)
            net = copy.deepcopy(net0).train()
            tr = train.FlowTrainer(net, graph=False)
            z, loss = tr._forward_backward(y)
            torch.cuda.synchronize()
            outs.append((z.detach().clone(), float(loss), tr.bucket.flat.detach().clone(), {k: v.detach().clone() for k, v in net.named_buffers()}))
    finally:
        N.call('nf_flow_solo_config', 3)
    (z0, l0, g0, b0), (z1, l1, g1, b1) = outs
    G.assert_close(z1, z0, 5e-5, rtol=5e-5, what='z')
    assert abs(l1 - l0) <= 5e-5 * max(1.0, abs(l0)), (l1, l0)
    G.assert_close(g1, g0, 2e-4 * max(1.0, float(g0.abs().max())), rtol=0.0, what='flat gradients')
    for name in b0:
        G.assert_close(b1[name].float(), b0[name].float(), 2e-5, rtol=2e-5, what='buffer ' + name)
    assert N.persistent_timeouts() == 0
