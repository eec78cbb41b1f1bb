"""
Parity of the HIP kernels (called through the C ABI via the product's functional layer) against the golden
vectors captured from the reference and against the oracle on fresh seeded inputs.  Needs a real MI355X.
Tolerances: 1e-5 fp32 (north_star), bit-exact for index maps.
"""
import importlib

import numpy as np
import pytest
import torch

from oracle import indexmaps as im
from oracle import transforms as tf
from tests import _golden as G

pytestmark = pytest.mark.gpu
TOL = 1e-5
DEV = 'cuda'


@pytest.fixture(scope='module')
def nf(pkg):
    assert torch.cuda.is_available(), 'gpu tests need a GPU'
    pkg._native.load()
    return pkg


def _scaled(want):
    return TOL * max(1.0, float(want.abs().max()))


def _gap(a32, a64):
    """how far the fp32 oracle is from the same computation in float64: the measured yard-stick the bars below add, times SLACK, to the
    1e-5 bar -- two correct fp32 implementations cannot agree better than either agrees with the exact result"""
    return float((a32.detach().double() - a64.detach()).abs().max())


SLACK = 4.0


def _ld_tol(g, a='ld', b='ld0'):
    """1e-5 relative to the size of the log-det increment (a (B,) vector of magnitude up to P*C*|log_scale|)."""
    return TOL * max(1.0, float((g[a] - g[b]).abs().max()))


# ---- index maps: bit exact ------------------------------------------------------------------------------------------
def test_indexmaps_golden(nf):
    NF = nf.functional
    for key in G.keys('indexmaps'):
        parts = key.split('/')
        if parts[0] in ('checker', 'channel', '1d') and parts[-1] == 'z0':
            tag, odd = parts[1], parts[2] == 'odd1'
            z = G.group('indexmaps', 'in/' + tag, DEV)[''].float()
            mode = {'checker': 1, 'channel': 2, '1d': 0}[parts[0]]
            g = G.group('indexmaps', '/'.join(parts[:3]) + '/', DEV)
            z0, z1 = NF.half_gather(z, 0, mode, odd), NF.half_gather(z, 1, mode, odd)
            assert torch.equal(z0, g['z0'].float()) and torch.equal(z1, g['z1'].float()), key
        if parts[0] == 'squeeze2d':
            z = G.group('indexmaps', 'in/' + parts[1], DEV)[''].float()
            want = G.group('indexmaps', key, DEV)[''].float()
            assert torch.equal(NF.squeeze2d(z), want)
            assert torch.equal(NF.unsqueeze2d(want), z)


@pytest.mark.parametrize('dims,mode', [((6, ), 0), ((3, 32, 32), 1), ((12, 16, 16), 1), ((12, 16, 16), 2),
                                       ((48, 8, 8), 2), ((5, 6, 10), 1)])
@pytest.mark.parametrize('odd', [False, True])
def test_indexmaps_vs_oracle(nf, dims, mode, odd):
    NF = nf.functional
    g = torch.Generator().manual_seed(1)
    z = torch.randn((7, ) + dims, generator=g)
    zd = z.to(DEV).requires_grad_(True)
    for which in (0, 1):
        want = im.split(z, mode, odd)[which]
        got = NF.half_gather(zd, which, mode, odd)
        assert torch.equal(got.cpu(), want)
        gh = torch.randn(want.shape, generator=g)
        (gz, ) = torch.autograd.grad(got, zd, gh.to(DEV))
        zr = z.clone().requires_grad_(True)
        (gz_ref, ) = torch.autograd.grad(im.split(zr, mode, odd)[which], zr, gh)
        assert torch.equal(gz.cpu(), gz_ref)
    if len(dims) == 3:
        assert torch.equal(NF.squeeze2d(zd).cpu(), im.squeeze2d(z))
        zs = im.squeeze2d(z)
        assert torch.equal(NF.unsqueeze2d(zs.to(DEV)).cpu(), z)


# ---- affine coupling ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('tag,mode', [('1d', 0), ('1d6', 0), ('checker', 1), ('channel', 2)])
@pytest.mark.parametrize('odd', [False, True])
def test_affine_coupling_golden(nf, tag, mode, odd):
    NF = nf.functional
    g = G.group('ops', 'affine/%s/odd%d/' % (tag, odd), DEV)
    a = torch.tensor([float(g['meta'][0])], device=DEV, requires_grad=True)
    c = torch.tensor([float(g['meta'][1])], device=DEV, requires_grad=True)
    z, params = g['z'].requires_grad_(True), g['params'].requires_grad_(True)
    y, ld = NF.affine_coupling(z, params, a, c, g['ld0'].clone(), mode, odd)
    G.assert_close(y, g['y'], TOL, what='y')
    G.assert_close(ld, g['ld'], _ld_tol(g), what='ld')
    gz, gp, ga, gc = torch.autograd.grad([y, ld], [z, params, a, c], [g['gy'], g['gld']])
    for got, n in [(gz, 'gz'), (gp, 'gparams'), (ga, 'ga'), (gc, 'gc')]:
        G.assert_close(got, g[n], _scaled(g[n]), what=n)
    x, ldi = NF.affine_coupling(g['y'], g['params'], a.detach(), c.detach(), g['ld'].clone(), mode, odd, inverse=True)
    G.assert_close(x, g['x_inv'], TOL, what='x_inv')
    G.assert_close(ldi, g['ld_inv'], _ld_tol(g), what='ld_inv')


@pytest.mark.parametrize('dims,mode,B', [((2, ), 0, 4096), ((3, 32, 32), 1, 8), ((12, 16, 16), 2, 8),
                                         ((48, 8, 8), 1, 5), ((24, 32, 32), 2, 3)])
def test_affine_coupling_vs_oracle(nf, dims, mode, B):
    NF = nf.functional
    g = torch.Generator().manual_seed(11)
    z = torch.randn((B, ) + dims, generator=g)
    half = im.split(z, mode, True)[0]
    pshape = list(half.shape)
    pshape[1] *= 2
    params = torch.randn(pshape, generator=g) * 0.5
    a, c = torch.tensor([0.4]), torch.tensor([-0.1])
    ld0 = torch.randn(B, generator=g)
    gy, gld = torch.randn(z.shape, generator=g), torch.randn(B, generator=g)
    for odd in (False, True):
        leaves = [t.clone().requires_grad_(True) for t in (z, params, a, c)]
        y, ld = tf.affine_coupling(leaves[0], ld0, leaves[1], leaves[2], leaves[3], mode, odd)
        want = torch.autograd.grad([y, ld], leaves, [gy, gld])
        l64 = [t.double().clone().requires_grad_(True) for t in (z, params, a, c)]
        y64, ld64 = tf.affine_coupling(l64[0], ld0.double(), l64[1], l64[2], l64[3], mode, odd)
        want64 = torch.autograd.grad([y64, ld64], l64, [gy.double(), gld.double()])
        dl = [t.clone().to(DEV).requires_grad_(True) for t in (z, params, a, c)]
        yd, ldd = NF.affine_coupling(dl[0], dl[1], dl[2], dl[3], ld0.to(DEV), mode, odd)
        G.assert_close(yd, y, TOL)
        G.assert_close(ldd, ld, TOL * max(1.0, float(ld.detach().abs().max())))
        got = torch.autograd.grad([yd, ldd], dl, [gy.to(DEV), gld.to(DEV)])
        for gg, ww, w64 in zip(got, want, want64):      # 1e-5 of the largest entry + the fp32 oracle's own distance from float64
            G.assert_close(gg, ww, _scaled(ww) + SLACK * _gap(ww, w64))
        xi, ldi = NF.affine_coupling(yd.detach(), dl[1].detach(), dl[2].detach(), dl[3].detach(), ldd.detach().clone(),
                                     mode, odd, inverse=True)
        G.assert_close(xi, z, 2e-5)                      # round trip
        G.assert_close(ldi, ld0, 2e-5 * max(1.0, float(ld.detach().abs().max())))


# ---- ActNorm / flow BatchNorm --------------------------------------------------------------------------------------------
@pytest.mark.parametrize('tag', ['2d', 'img'])
def test_actnorm_golden(nf, tag):
    g = G.group('ops', 'actnorm/%s/' % tag, DEV)
    dims = tuple(g['z'].shape[1:])
    layer = nf.ActNorm(dims).to(DEV)
    z = g['z'].requires_grad_(True)
    y, ld = layer(z, g['ld0'].clone())
    G.assert_close(layer.log_scale, g['log_scale'], 2e-6, what='init log_scale')
    G.assert_close(layer.bias, g['bias'], 2e-6, what='init bias')
    G.assert_close(y, g['y'], TOL)
    G.assert_close(ld, g['ld'], _ld_tol(g))
    gz, gls, gb = torch.autograd.grad([y, ld], [z, layer.log_scale, layer.bias], [g['gy'], g['gld']])
    G.assert_close(gz, g['gz'], TOL)
    G.assert_close(gls, g['glog_scale'], _scaled(g['glog_scale']))
    G.assert_close(gb, g['gbias'], _scaled(g['gbias']))
    x, ldi = layer.backward(g['y'], g['ld'].clone())
    G.assert_close(x, g['x_inv'], TOL)
    G.assert_close(ldi, g['ld_inv'], _ld_tol(g))


@pytest.mark.parametrize('tag', ['2d', 'img'])
def test_flow_bn_golden(nf, tag):
    g0 = G.group('ops', 'flowbn/%s/step0/' % tag, DEV)
    dims = tuple(g0['x'].shape[1:])
    layer = nf.BatchNorm(dims, affine=False).to(DEV)
    layer.train()
    for step in range(2):
        g = G.group('ops', 'flowbn/%s/step%d/' % (tag, step), DEV)
        x = g['x'].requires_grad_(True)
        y, ld = layer(x, g['ld0'].clone())
        for n in ('batch_mean', 'batch_var', 'running_mean', 'running_var'):
            G.assert_close(getattr(layer, n), g[n], 2e-6, what=n)
        G.assert_close(y, g['y'], TOL)
        G.assert_close(ld, g['ld'], _ld_tol(g))
        (gx, ) = torch.autograd.grad([y], [x], [g['gy']])
        G.assert_close(gx, g['gx'], TOL)
        xi, ldi = layer.backward(g['y'], g['ld'].clone())
        G.assert_close(xi, g['x_inv'], TOL)
        G.assert_close(ldi, g['ld_inv'], _ld_tol(g))
    layer.eval()
    g = G.group('ops', 'flowbn/%s/eval/' % tag, DEV)
    y, ld = layer(g['x'], g['ld0'].clone())
    G.assert_close(y, g['y'], TOL)
    G.assert_close(ld, g['ld'], _ld_tol(g))
    xi, ldi = layer.backward(g['y'], g['ld'].clone())
    G.assert_close(xi, g['x_inv'], TOL)
    G.assert_close(ldi, g['ld_inv'], _ld_tol(g))


def test_flow_bn_affine_grads_vs_oracle(nf):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(16, 6, 4, 4, generator=g) * 2 + 1
    lg, beta = torch.randn(1, 6, 1, 1, generator=g) * 0.3, torch.randn(1, 6, 1, 1, generator=g)
    ld0, gy, gld = torch.randn(16, generator=g), torch.randn(x.shape, generator=g), torch.randn(16, generator=g)
    mean, var = tf.flow_bn_stats(x)
    leaves = [t.clone().requires_grad_(True) for t in (x, lg, beta)]
    y, ld = tf.flow_bn(leaves[0], ld0, mean, var, leaves[1], leaves[2])
    want = torch.autograd.grad([y, ld], leaves, [gy, gld])
    layer = nf.BatchNorm((6, 4, 4), affine=True).to(DEV)
    with torch.no_grad():
        layer.log_gamma.copy_(lg)
        layer.beta.copy_(beta)
    xd = x.to(DEV).requires_grad_(True)
    yd, ldd = layer(xd, ld0.to(DEV))
    G.assert_close(yd, y, TOL)
    G.assert_close(ldd, ld, TOL * 10)
    got = torch.autograd.grad([yd, ldd], [xd, layer.log_gamma, layer.beta], [gy.to(DEV), gld.to(DEV)])
    for gg, ww in zip(got, want):
        G.assert_close(gg, ww, _scaled(ww) * 2)


# ---- invertible 1x1 ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('C', [2, 3, 12, 48])
def test_invconv_golden(nf, C):
    g = G.group('ops', 'invconv/%d/' % C, DEV)
    layer = nf.InvertibleConv1x1(C).to(DEV)
    sd = {n: g[n] for n in ('P', 'L', 'U', 'I', 'pivots', 'L_mask', 'U_mask', 'log_s', 'sign_s')}
    layer.load_state_dict(sd)
    z = g['z'].requires_grad_(True)
    y, ld = layer(z, g['ld0'].clone())
    G.assert_close(y, g['y'], TOL)
    G.assert_close(ld, g['ld'], _ld_tol(g))
    gz, gL, gU, gs = torch.autograd.grad([y, ld], [z, layer.L, layer.U, layer.log_s], [g['gy'], g['gld']])
    for got, n in [(gz, 'gz'), (gL, 'gL'), (gU, 'gU'), (gs, 'glog_s')]:
        G.assert_close(got, g[n], _scaled(g[n]), what=n)
    x, ldi = layer.backward(g['y'], g['ld'].clone())
    G.assert_close(x, g['x_inv'], TOL)
    G.assert_close(ldi, g['ld_inv'], _ld_tol(g))


@pytest.mark.parametrize('C,P,B', [(5, 7, 9), (48, 64, 64), (12, 256, 16), (3, 1024, 4), (2, 1, 4096),
                                   (48, 64, 2050), (12, 256, 513), (20, 64, 2100)])   # the last three: 64-pixel-block apply
def test_invconv_vs_oracle(nf, C, P, B):
    NF = nf.functional
    g = torch.Generator().manual_seed(C)
    z = torch.randn(B, C, P, generator=g)
    W = torch.linalg.qr(torch.randn(C, C, generator=g))[0] + 0.05 * torch.randn(C, C, generator=g)
    log_s = torch.randn(C, generator=g) * 0.1
    ld0, gy, gld = torch.randn(B, generator=g), torch.randn(z.shape, generator=g), torch.randn(B, generator=g)
    leaves = [t.clone().requires_grad_(True) for t in (z, W, log_s)]
    y, ld = tf.invconv(leaves[0], ld0, leaves[1], leaves[2])
    want = torch.autograd.grad([y, ld], leaves, [gy, gld])
    dl = [t.clone().to(DEV).requires_grad_(True) for t in (z, W, log_s)]
    yd, ldd = NF.invconv(dl[0], dl[1], ld0.to(DEV), dl[2])
    G.assert_close(yd, y, TOL)
    G.assert_close(ldd, ld, TOL * max(1.0, float(ld.detach().abs().max())))
    got = torch.autograd.grad([yd, ldd], dl, [gy.to(DEV), gld.to(DEV)])
    for gg, ww in zip(got, want):
        G.assert_close(gg, ww, _scaled(ww) * 4)


# ---- logit --------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('eps', [1.0e-5, 0.01])
def test_logit_golden(nf, eps):
    NF = nf.functional
    g = G.group('ops', 'logit/%g/' % eps, DEV)
    x = g['x'].requires_grad_(True)
    y, ld = NF.logit(x, g['ld0'].clone(), eps)
    G.assert_close(y, g['y'], TOL)
    G.assert_close(ld, g['ld'], _ld_tol(g), rtol=2e-6)
    (gx, ) = torch.autograd.grad([y, ld], [x], [g['gy'], g['gld']])
    G.assert_close(gx, g['gx'], TOL, rtol=1e-5)
    xi, ldi = NF.logit(g['yin'].detach(), g['ld0'].clone(), eps, inverse=True)
    G.assert_close(xi, g['x_inv'], TOL)
    G.assert_close(ldi, g['ld_inv'], _ld_tol(g, 'ld_inv', 'ld0'))


def test_empty_batch(nf):
    NF = nf.functional
    z = torch.zeros(0, 2, device=DEV)
    ld = torch.zeros(0, device=DEV)
    y, ld2 = NF.affine_coupling(z, torch.zeros(0, 2, device=DEV), torch.ones(1, device=DEV), torch.zeros(1, device=DEV),
                                ld, 0, False)
    assert y.shape == (0, 2) and ld2.shape == (0, )
    y, _ = NF.logit(torch.zeros(0, 3, 4, 4, device=DEV), ld, 0.01)
    assert y.shape == (0, 3, 4, 4)


def test_cpu_tensor_is_refused(nf):
    with pytest.raises(RuntimeError):
        nf.functional.logit(torch.rand(2, 3), torch.zeros(2), 0.01)


# ---- Flow++ mixture-of-logistics coupling ---------------------------------------------------------------------------------
@pytest.mark.parametrize('tag,mode', [('1d', 0), ('checker', 1), ('channel', 2)])
@pytest.mark.parametrize('odd', [False, True])
def test_mixlog_coupling_golden(nf, tag, mode, odd):
    NF = nf.functional
    g = G.group('ops', 'mixlog/%s/odd%d/' % (tag, odd), DEV)
    a0, c0, K = float(g['meta'][0]), float(g['meta'][1]), int(g['meta'][2])
    a = torch.tensor([a0], device=DEV, requires_grad=True)
    c = torch.tensor([c0], device=DEV, requires_grad=True)
    z, params = g['z'].requires_grad_(True), g['params'].requires_grad_(True)
    y, ld = NF.mixlog_coupling(z, params, a, c, g['ld0'].clone(), K, mode, odd)
    G.assert_close(y, g['y'], TOL, what='y')
    G.assert_close(ld, g['ld'], _ld_tol(g), what='ld')
    grads = torch.autograd.grad([y, ld], [z, params, a, c], [g['gy'], g['gld']])
    for got, n in zip(grads, ['gz', 'gparams', 'ga', 'gc']):
        G.assert_close(got, g[n], _scaled(g[n]), what=n)
    x, ldi = NF.mixlog_coupling(g['y'], g['params'], a.detach(), c.detach(), g['ld'].clone(), K, mode, odd, inverse=True)
    G.assert_close(x, g['x_inv'], 1e-4, what='x_inv (bisection bracket)')
    # The reference stops its bisection at a bracket of 1e-4, so x is only defined to that width and ld_inv = ... - sum logpdf(x) moves
    # with it (|d logpdf / dx| <= ~20 here: 2e-3).  What IS sharp: the log-det the kernel reports must be the analytic one AT ITS OWN
    # x -- restated here in float64 from the oracle's pieces (coupling.py:204-208: affine inverse, logit inverse, then the mixture's
    # log-density at the x the bisection returned).
    G.assert_close(ldi, g['ld_inv'], 2e-3, what='ld_inv (bracket-limited)')
    oc = im.split(g['z'].cpu(), mode, odd)[0].shape[1]
    sections = [oc] * 2 + [oc * K] * 3
    d64 = lambda t: t.detach().cpu().double()                # noqa: E731
    aa, bb, logpi, mu, ss = tf.mixlog_split_params(d64(g['params']), sections, K, d64(a), d64(c))
    y0 = im.split(d64(g['y']), mode, odd)[0]
    u = torch.exp(-aa) * (y0 - bb)
    ld1 = d64(g['ld']) - tf._per_sample_sum(aa)
    _, ld2 = tf.logit_inverse(u, ld1)
    x0 = im.split(d64(x), mode, odd)[0]
    want_ld = ld2 - tf._per_sample_sum(tf._mix_logpdf(x0, logpi, mu, ss))
    n_per = x0[0].numel()                                    # a per-sample log-det is an fp32 sum of ~7 n_per terms of size O(1 .. 10)
    G.assert_close(ldi, want_ld.float(), _scaled(want_ld) + 1.0e-6 * n_per, what='log-det of the inverse at its own x (float64 restatement)')


@pytest.mark.parametrize('B,K', [(999, 8), (1024, 3), (2051, 8), (517, 1), (4096, 5)])
def test_mixlog_row_kernels_match_oracle_and_the_octet_kernels(nf, B, K):
    """the large-batch kernels of the 2-D mixture coupling (csrc/mixlog.hip, round 6: one row per thread, linear-space sums with shared
    transcendentals, LDS-staged 16-byte traffic; they serve batches >= 262 144 rows) forced onto small RAGGED batches by
    nf_mixlog_rows_config(0): forward, backward and inverse against the float64 oracle with the fp32 oracle's own distance as slack, and against
    the one-component-per-lane kernels on the same inputs.  Rows far in the tails (|x - mu| / scale ~ 100: the linear-space density underflows)
    take the kernels' log-space path and must come out as finite and as close."""
    NF = nf.functional
    lib = nf._native.load()
    g = torch.Generator().manual_seed(100 + B + K)
    z = torch.randn(B, 2, generator=g)
    z[::37] *= 40.0                                        # far tails on both sides
    params = torch.randn(B, 2 + 3 * K, generator=g) * 0.7
    sections = [1] * 2 + [K] * 3
    a, c = torch.tensor([0.5]), torch.tensor([0.05])
    ld0 = torch.randn(B, generator=g)
    gy, gld = torch.randn(z.shape, generator=g), torch.randn(B, generator=g)
    for odd in (False, True):
        l64 = [t.double().clone().requires_grad_(True) for t in (z, params, a, c)]
        y64, ld64 = tf.mixlog_coupling(l64[0], ld0.double(), l64[1], sections, K, l64[2], l64[3], 0, odd)
        want64 = torch.autograd.grad([y64, ld64], l64, [gy.double(), gld.double()])
        l32 = [t.clone().requires_grad_(True) for t in (z, params, a, c)]
        y32, ld32 = tf.mixlog_coupling(l32[0], ld0, l32[1], sections, K, l32[2], l32[3], 0, odd)
        want32 = torch.autograd.grad([y32, ld32], l32, [gy, gld])
        res = {}
        for which, lim in (('rows', 0), ('octets', 1 << 40)):
            assert lib.nf_mixlog_rows_config(lim) == 0
            try:
                dl = [t.clone().to(DEV).requires_grad_(True) for t in (z, params, a, c)]
                yd, ldd = NF.mixlog_coupling(dl[0], dl[1], dl[2], dl[3], ld0.to(DEV), K, 0, odd)
                got = torch.autograd.grad([yd, ldd], dl, [gy.to(DEV), gld.to(DEV)])
                xi, ldi = NF.mixlog_coupling(yd.detach(), dl[1].detach(), dl[2].detach(), dl[3].detach(), ldd.detach().clone(), K, 0, odd,
                                             inverse=True)
                torch.cuda.synchronize()
            finally:
                lib.nf_mixlog_rows_config(-1)
            res[which] = (yd.detach(), ldd.detach(), [t.detach() for t in got], xi, ldi)
            assert all(bool(torch.isfinite(t).all()) for t in [yd, ldd, xi, ldi] + list(got)), which
            body = torch.ones(B, dtype=torch.bool)
            body[::37] = False                              # the planted tail rows: the clamped CDF is flat there and the logit's slope is 1 / eps --
            bd = body.to(DEV)                               # they must come out finite (above) and the same from both kernel families (below)
            G.assert_close(yd[bd], y32[body], _scaled(y32.detach()[body]) + SLACK * _gap(y32[body], y64[body]), what=which + ' y')
            G.assert_close(ldd[bd], ld32[body], _scaled(ld32.detach()[body]) + SLACK * _gap(ld32[body], ld64[body]), what=which + ' ld')
            for gg, ww, w64, n in zip(got, want32, want64, ('g_z', 'g_params', 'g_a', 'g_c')):
                if n in ('g_a', 'g_c'):
                    # ONE number each: a sum over the batch of terms that reach 1e3 on the planted tail rows and cancel -- the bar is
                    # relative to the sum of magnitudes a B-term fp32 sum carries, not to the total
                    G.assert_close(gg, ww, _scaled(ww) + SLACK * _gap(ww, w64) + 1.0e-6 * B * 40.0, what=which + ' ' + n)
                else:
                    G.assert_close(gg[bd], ww[body], _scaled(ww[body]) + SLACK * _gap(ww[body], w64[body]), what=which + ' ' + n)
            # round trip where the forward did not clamp the CDF (|logit F| < 9: F inside (1.2e-4, 1 - 1.2e-4); a single narrow component puts
            # ordinary rows outside, and there x is not recoverable by any implementation)
            aa = torch.tanh(params[:, 0]) * a + c
            open_ = body & (((y32.detach()[:, int(odd)] - params[:, 1]) * torch.exp(-aa)).abs() < 9.0)
            assert int(open_.sum()) > B // 2
            G.assert_close(xi[open_.to(DEV)], z[open_], 2e-4, what=which + ' round trip')
        # the two kernel families on identical inputs: the same numbers to rounding (y, ld) and to the bisection bracket (inverse)
        G.assert_close(res['rows'][0], res['octets'][0].cpu(), 1.0e-3 * max(1.0, float(y32.detach().abs().max())), what='rows vs octets y (tails included)')
        G.assert_close(res['rows'][0][body.to(DEV)], res['octets'][0][body.to(DEV)].cpu(), _scaled(y32.detach()[body]) + SLACK * _gap(y32[body], y64[body]), what='rows vs octets y')
        G.assert_close(res['rows'][3][open_.to(DEV)], res['octets'][3][open_.to(DEV)].cpu(), 2e-4, what='rows vs octets inverse')


@pytest.mark.parametrize('dims,mode,B,K', [((2, ), 0, 65536, 8), ((3, 8, 8), 1, 4, 4), ((8, 8, 8), 2, 4, 8)])
def test_mixlog_coupling_vs_oracle(nf, dims, mode, B, K):
    NF = nf.functional
    g = torch.Generator().manual_seed(21)
    z = torch.randn((B, ) + dims, generator=g)
    half = im.split(z, mode, False)[0]
    pshape = list(half.shape)
    oc = pshape[1]
    pshape[1] = oc * (2 + 3 * K)
    params = torch.randn(pshape, generator=g) * 0.7
    sections = [oc] * 2 + [oc * K] * 3
    a, c = torch.tensor([0.5]), torch.tensor([0.05])
    ld0 = torch.randn(B, generator=g)
    gy, gld = torch.randn(z.shape, generator=g), torch.randn(B, generator=g)
    for odd in (False, True):
        leaves = [t.clone().requires_grad_(True) for t in (z, params, a, c)]
        y, ld = tf.mixlog_coupling(leaves[0], ld0, leaves[1], sections, K, leaves[2], leaves[3], mode, odd)
        want = torch.autograd.grad([y, ld], leaves, [gy, gld])
        l64 = [t.double().clone().requires_grad_(True) for t in (z, params, a, c)]
        y64, ld64 = tf.mixlog_coupling(l64[0], ld0.double(), l64[1], sections, K, l64[2], l64[3], mode, odd)
        want64 = torch.autograd.grad([y64, ld64], l64, [gy.double(), gld.double()])
        dl = [t.clone().to(DEV).requires_grad_(True) for t in (z, params, a, c)]
        yd, ldd = NF.mixlog_coupling(dl[0], dl[1], dl[2], dl[3], ld0.to(DEV), K, mode, odd)
        # the logit amplifies the CDF's rounding by 1 / (F (1 - F)) near the tails: measured, not guessed -- the fp32 oracle's own
        # distance from float64 enters the bar
        G.assert_close(yd, y, _scaled(y.detach()) + SLACK * _gap(y, y64))
        G.assert_close(ldd, ld, _scaled(ld.detach()) + SLACK * _gap(ld, ld64))
        got = torch.autograd.grad([yd, ldd], dl, [gy.to(DEV), gld.to(DEV)])
        for gg, ww, w64 in zip(got, want, want64):
            G.assert_close(gg, ww, _scaled(ww) + SLACK * _gap(ww, w64))
        xi, ldi = NF.mixlog_coupling(yd.detach(), dl[1].detach(), dl[2].detach(), dl[3].detach(), ldd.detach().clone(), K,
                                     mode, odd, inverse=True)
        G.assert_close(xi, z, 2e-4)                      # round trip: bisection bracket 6e-5
        G.assert_close(ldi, ld0, 5e-3 * max(1.0, float(ld.detach().abs().max())))


def test_mixlog_bisection_stuck_rule(nf):
    """the 100-iteration regime: an element whose target equals the CDF at the first midpoint never moves its
    bracket, so the whole batch runs 100 iterations (modules.py:205) and converges far below the 25-step bracket."""
    NF = nf.functional
    g = torch.Generator().manual_seed(3)
    B, K = 4096, 8
    params = torch.randn(B, 2 + 3 * K, generator=g) * 0.5
    a, c = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
    z = torch.randn(B, 2, generator=g)
    zd, pd = z.to(DEV), params.to(DEV)
    y, ld = NF.mixlog_coupling(zd, pd, a, c, torch.zeros(B, device=DEV), K, 0, False)
    x25, _ = NF.mixlog_coupling(y, pd, a, c, torch.zeros(B, device=DEV), K, 0, False, inverse=True)
    err25 = float((x25 - zd).abs().max())
    # plant one stuck element: feed the GPU's own CDF value at x = 0 (first midpoint) as the target
    z2 = zd.clone()
    z2[0, 0] = 0.0
    y2, _ = NF.mixlog_coupling(z2, pd, a, c, torch.zeros(B, device=DEV), K, 0, False)
    x100, _ = NF.mixlog_coupling(y2, pd, a, c, torch.zeros(B, device=DEV), K, 0, False, inverse=True)
    err100 = float((x100[1:] - z2[1:]).abs().max())
    assert err25 < 2e-4
    assert err100 <= err25 + 1e-6


@pytest.mark.parametrize('tag', ['K4_2d', 'K8_2d', 'K4_4d'])
def test_mixlogcdf_module_golden(nf, tag):
    """the standalone MixLogCDF module (modules.py:186-212, reference signature) against the reference-captured goldens:
    forward, autograd (through log_softmax as the reference test harness did), and the HIP bisection in BOTH regimes -- the 25-
    iteration one and the 100-iteration one (``x_inv100`` / ``ld_inv100``: one element of ``target100`` is exactly the CPU's CDF at
    the first midpoint, which keeps its bracket open and makes the reference run all 100 steps for the whole batch)."""
    g = G.group('ops', 'mixlogcdf/%s/' % tag, DEV)
    layer = nf.MixLogCDF()
    x = g['x'].clone().requires_grad_(True)
    raw = g['logpi_raw'].clone().requires_grad_(True)
    mu, s = g['mu'].clone().requires_grad_(True), g['s'].clone().requires_grad_(True)
    ld0 = g['ld0'].clone()
    y, ld = layer(x, torch.log_softmax(raw, dim=1), mu, s, ld0)
    assert torch.equal(ld0, g['ld0']), 'the reference returns a NEW log-det tensor here (modules.py:194)'
    G.assert_close(y, g['y'], TOL, what='y')
    G.assert_close(ld, g['ld'], _ld_tol(g), what='ld')
    grads = torch.autograd.grad([y, ld], [x, raw, mu, s], [g['gy'], g['gld']])
    for got, n in zip(grads, ['gx', 'glogpi_raw', 'gmu', 'gs']):
        G.assert_close(got, g[n], _scaled(g[n]), what=n)
    logpi = torch.log_softmax(g['logpi_raw'], dim=1)
    n_per = g['x'][0].numel()
    # 25-iteration regime: the answer is only defined up to the bracket the reference stops at (2000 * 2^-25 = 6e-5)
    xi, ldi = layer.backward(g['target'].clone(), logpi, g['mu'], g['s'], g['ld0'].clone())
    G.assert_close(xi, g['x_inv'], 1e-4, what='x_inv (25-step bracket)')
    G.assert_close(ldi, g['ld_inv'], 2e-3 * n_per, what='ld_inv')
    # 100-iteration regime.  Whether the kernel's own `val == target` tie fires depends on the last bit of its CDF at 0.0, so the
    # planted element may resolve to within the 25-step bracket of the reference's 0.0 instead of exactly 0.0; every other element
    # of the reference's answer is converged to float precision, and the kernel's must be within the bracket of that.
    xi, ldi = layer.backward(g['target100'].clone(), logpi, g['mu'], g['s'], g['ld0'].clone())
    G.assert_close(xi, g['x_inv100'], 1e-4, what='x_inv100')
    G.assert_close(ldi, g['ld_inv100'], 2e-3 * n_per, what='ld_inv100')
    # planted with the KERNEL's own CDF value the tie is exact: the bracket of that element never moves (x = 0.0 exactly) and the
    # whole batch runs 100 steps -- every other element then agrees with the reference's converged answer far below the bracket
    t_own = g['target100'].clone()
    first = (0, ) * t_own.dim()
    zero = torch.zeros_like(g['x'])
    t_own[first] = layer(zero, logpi, g['mu'], g['s'], g['ld0'].clone())[0][first]
    xi, _ = layer.backward(t_own, logpi, g['mu'], g['s'], g['ld0'].clone())
    assert float(xi[first]) == 0.0
    mask = torch.ones_like(xi, dtype=torch.bool)
    mask[first] = False
    assert float((xi - g['x_inv100'])[mask].abs().max()) < 2e-5, 'converged (100-step) answers must agree far below the 25-step bracket'


def test_mixlogcdf_module_vs_oracle_large(nf):
    """B = 65536, K = 8 (C3's mixture shape): forward and gradients against the oracle, round trip through the bisection."""
    g = torch.Generator().manual_seed(5)
    B, K = 65536, 8
    x = torch.randn(B, 1, generator=g)
    raw, mu, s = (torch.randn(B, K, 1, generator=g) * 0.7 for _ in range(3))
    ld0 = torch.randn(B, generator=g)
    gy, gld = torch.randn(B, 1, generator=g), torch.randn(B, generator=g)
    leaves = [t.clone().requires_grad_(True) for t in (x, raw, mu, s)]
    y, ld = tf.mixlogcdf(leaves[0], ld0, torch.log_softmax(leaves[1], dim=1), leaves[2], leaves[3])
    want = torch.autograd.grad([y, ld], leaves, [gy, gld])
    dl = [t.clone().to(DEV).requires_grad_(True) for t in (x, raw, mu, s)]
    layer = nf.MixLogCDF()
    yd, ldd = layer(dl[0], torch.log_softmax(dl[1], dim=1), dl[2], dl[3], ld0.to(DEV))
    G.assert_close(yd, y, TOL)
    G.assert_close(ldd, ld, TOL * max(1.0, float(ld.detach().abs().max())))
    got = torch.autograd.grad([yd, ldd], dl, [gy.to(DEV), gld.to(DEV)])
    for gg, ww in zip(got, want):
        G.assert_close(gg, ww, _scaled(ww), rtol=1e-4)
    with torch.no_grad():
        xi, ldi = layer.backward(yd.detach(), torch.log_softmax(dl[1], dim=1), dl[2].detach(), dl[3].detach(), ldd.detach())
    ok = (yd.detach() > 1e-4) & (yd.detach() < 1 - 1e-4)            # outside, the CDF is flat to fp32 and x is not identifiable
    assert float((xi - dl[0].detach())[ok].abs().max()) < 2e-4
    assert float((ldi - ld0.to(DEV))[ok.reshape(-1)].abs().max()) < 5e-3


def test_persistent_kernel_timeout_is_loud(nf, pkg):
    """a grid exchange that gives up must surface as an exception, not as silently wrong numbers: with a poll budget of zero
    every workgroup of a multi-workgroup persistent launch gives up at once; the sticky host-mapped error word then makes the
    next native call (and FlowTrainer.train_on_batch) raise."""
    import importlib
    from types import SimpleNamespace as NS
    Nn = pkg._native
    nftrain = importlib.import_module(pkg.__name__ + '.train')
    torch.manual_seed(0)
    net = pkg.Glow((2, ), '2d', NS(layers=2)).to(DEV)
    trainer = nftrain.FlowTrainer(net, graph=False)
    y = torch.randn(4096, 2, device=DEV) * 0.5
    trainer.train_on_batch(y)                                    # ActNorm init
    trainer.train_on_batch(y)                                    # fused persistent step kernels, 32 workgroups: fine
    torch.cuda.synchronize()
    Nn.check_persistent()
    assert Nn.persistent_timeouts() == 0
    try:
        Nn.persistent_reset(spin_limit=0)                        # every wait that is not satisfied at once gives up
        with pytest.raises(Nn.PersistentKernelTimeout):
            for _ in range(3):
                trainer.train_on_batch(y)
                torch.cuda.synchronize()
        assert Nn.persistent_timeouts() > 0
    finally:
        Nn.persistent_reset(spin_limit=1 << 22)
    Nn.check_persistent()
    assert Nn.persistent_timeouts() == 0


# ---- MAF autoregressive transform -------------------------------------------------------------------------------------
@pytest.mark.parametrize('D', [2, 5])
def test_ar_transform_golden(nf, D):
    g = G.group('ops', 'ar/%d/' % D, DEV)
    layer = nf.AutoregressiveTransfrom(D)
    layer.load_state_dict(G.group('ops', 'ar/%d/sd/' % D))
    layer = layer.to(DEV).train()
    z = g['z'].requires_grad_(True)
    np.random.seed(1234)
    y, ld = layer(z, g['ld0'].clone())
    G.assert_close(y, g['y'], TOL)
    G.assert_close(ld, g['ld'], TOL)
    names = [n for n, _ in layer.named_parameters()]
    grads = torch.autograd.grad([y, ld], [z] + [p for _, p in layer.named_parameters()], [g['gy'], g['gld']],
                                allow_unused=True)
    G.assert_close(grads[0], g['gz'], _scaled(g['gz']))
    for n, got in zip(names, grads[1:]):
        want = g['grad/' + n]
        got = torch.zeros_like(want) if got is None else got
        # a bias in front of a train-mode BatchNorm has an analytically ZERO gradient (the batch mean removes it):
        # what the reference stores there is fp32 cancellation noise of size ~1e-7 * sum|g_h|, not a value to match
        pre_bn_bias = '.biases.' in n and not n.endswith('.biases.3')
        G.assert_close(got, want, 2e-3 if pre_bn_bias else _scaled(want) * 2, what=n)
    sd = layer.state_dict()
    for k, want in G.group('ops', 'ar/%d/sd_after/' % D).items():
        G.assert_close(sd[k].float(), want.float(), 2e-6, what=k)
    layer.eval()
    with torch.no_grad():
        np.random.seed(99)
        ye, lde = layer(g['z'].detach(), g['ld0'].clone())
        G.assert_close(ye, g['y_eval'], TOL)
        G.assert_close(lde, g['ld_eval'], TOL)
        np.random.seed(99)
        zin = g['y_eval'].clone()
        xi, ldi = layer.backward(zin, g['ld_eval'].clone())
        assert torch.equal(zin, g['y_eval'])             # the caller's tensor is not mutated (appendix D Q5)
        G.assert_close(xi, g['x_inv'], TOL)
        G.assert_close(ldi, g['ld_inv'], TOL)


def test_nll_loss_vs_reference_formula(nf):
    g = torch.Generator().manual_seed(2)
    for shape in [(4096, 2), (64, 3, 32, 32), (7, 5)]:
        z = torch.randn(shape, generator=g)
        ld = torch.randn(shape[0], generator=g)
        zr, lr = z.clone().requires_grad_(True), ld.clone().requires_grad_(True)
        want = tf.nll_loss(zr, lr)
        want.backward()
        zd, ldv = z.to(DEV).requires_grad_(True), ld.to(DEV).requires_grad_(True)
        got = nf.functional.nll_loss(zd, ldv)
        (got * 1.0).backward()
        D = z[0].numel()
        G.assert_close(got, want, 1e-5 * max(1.0, abs(float(want)) / D * D ** 0.5))
        G.assert_close(zd.grad, zr.grad, 1e-7, rtol=1e-5)
        G.assert_close(ldv.grad, lr.grad, 1e-7, rtol=1e-5)


def test_flat_adam_matches_torch_adam(pkg):
    import importlib
    nfdist = importlib.import_module(pkg.__name__ + '.dist')
    train = importlib.import_module(pkg.__name__ + '.train')
    torch.manual_seed(0)
    shapes = [(32, 32), (32, ), (1, ), (2, 32), (1, 2, 1, 1)]
    pa = [torch.nn.Parameter(torch.randn(s, device=DEV)) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    bucket = nfdist.GradBucket(pa, flatten_params=True)
    opt_a = train.FlatAdam(bucket, lr=1e-2, weight_decay=0.01)
    opt_b = torch.optim.Adam(pb, lr=1e-2, weight_decay=0.01)
    for it in range(5):
        bucket.zero_()
        opt_b.zero_grad()
        for a, b in zip(pa, pb):
            g = torch.randn(a.shape, device=DEV)
            a.grad.add_(g)
            b.grad = g.clone()
        opt_a.step()
        opt_b.step()
        for a, b in zip(pa, pb):
            G.assert_close(a, b, 2e-6, rtol=2e-6, what='adam step %d' % it)
    assert int(opt_a.step_count) == 5


# ---- invertible residual block: the training step on HIP (closed-form second derivatives) ------------------------------------------------
@pytest.mark.parametrize('B,D', [(64, 2), (1001, 2), (4096, 4), (37, 3)])
def test_iresblock_training_hip_matches_autograd(nf, B, D):
    """value, input gradient and every parameter gradient of a training-mode InvertibleResLinear: csrc/resmlp.hip (the Russian-
    roulette value on the per-sample Jacobian, the Neumann-series gradient estimator with closed-form second derivatives,
    iresblock.py:59-109) against the same block differentiated by nested PyTorch autograd sweeps in FLOAT64 on the CPU -- the
    formulation the reference uses -- with identical noise and series lengths (bar: 2e-5 of the largest entry + 4 x the distance
    of the float32 CPU evaluation from the float64 one)."""
    import copy
    torch.manual_seed(5)
    blk = nf.InvertibleResLinear(D, D, coeff=0.9).train()
    blk.noise_on_cpu = True
    with torch.no_grad():
        for m in blk.g_fn:
            if isinstance(m, nf.LipSwish):
                m.beta.fill_(0.8 + 0.3 * torch.rand(()).item())
    x = torch.randn(B, D) * 0.7
    ld0 = torch.randn(B) * 0.1
    gy, gld = torch.randn(B, D), torch.full((B, ), -1.0 / B)     # the reference takes the log-det gradient of the FIRST sample for all
    res = {}
    for tag, dev, dt, hip in (('cpu64', 'cpu', torch.float64, False), ('cpu32', 'cpu', torch.float32, False), ('gpu', DEV, torch.float32, True)):
        b2 = copy.deepcopy(blk).to(dev).to(dt)
        b2.hip_training = hip
        torch.manual_seed(123)
        np.random.seed(123)
        if dt == torch.float64:                          # same float32 noise values in every run
            b2._randn_like = lambda t, shape=None: torch.randn(tuple(t.shape) if shape is None else shape).double()
        xx = x.detach().clone().to(dev).to(dt).requires_grad_(True)
        y, ld = b2(xx, ld0.to(dev).to(dt).clone())
        torch.autograd.backward([y, ld], [gy.to(dev).to(dt), gld.to(dev).to(dt)])
        grads = {k: p.grad.detach().cpu().double() for k, p in b2.named_parameters() if p.grad is not None}
        res[tag] = (y.detach().cpu().double(), ld.detach().cpu().double(), xx.grad.detach().cpu().double(), grads)
    ref, c32, gpu = res['cpu64'], res['cpu32'], res['gpu']

    def check(what, got, want, fp32):
        bar = 2e-5 * max(1.0, float(want.abs().max())) + 4.0 * float((fp32 - want).abs().max())
        err = float((got - want).abs().max())
        assert err <= bar, '%s: |gpu - cpu64| %.3e > %.3e (cpu32 itself %.3e)' % (what, err, bar, float((fp32 - want).abs().max()))
    check('y', gpu[0], ref[0], c32[0])
    check('ld', gpu[1], ref[1], c32[1])
    check('grad x', gpu[2], ref[2], c32[2])
    assert set(gpu[3]) == set(ref[3]), (sorted(gpu[3]), sorted(ref[3]))
    for k in ref[3]:
        check('grad ' + k, gpu[3][k], ref[3][k], c32[3][k])


@pytest.mark.parametrize('dims,masking,odd', [((6, ), 'checkerboard', False), ((6, ), 'checkerboard', True), ((12, 8, 8), 'channelwise', False),
                                              ((12, 8, 8), 'channelwise', True)])
def test_additive_coupling(nf, dims, masking, odd):
    """NICE additive coupling (flows/coupling.py:52-79): z0 + net_t(z1), merged; log-det untouched; inverse; gradient of z."""
    torch.manual_seed(4)
    layer = nf.AdditiveCoupling(dims, masking=masking, odd=odd).to(DEV).eval()      # eval: the conditioner's BatchNorm uses constants
    B = 16
    z = torch.randn((B, ) + dims, device=DEV, requires_grad=True)
    ld0 = torch.randn(B, device=DEV)
    y, ld = layer(z, ld0.clone())
    z0, z1 = im.split(z.detach().cpu(), layer.mode, odd)
    with torch.no_grad():
        t = layer.net_t(z1.to(DEV)).cpu()
    want = im.merge(z0 + t, z1, layer.mode, odd, dims)
    G.assert_close(y, want, TOL, what='y')
    assert torch.equal(ld, ld0), 'additive coupling has a unit Jacobian'
    w = torch.randn_like(y)
    (y * w).sum().backward()
    zr = z.detach().clone().requires_grad_(True)                                    # reference gradient: the same net through torch ops
    r0, r1 = layer.squeeze(zr)
    yr0 = r0 + layer.net_t(r1)
    w0, w1 = im.split(w.cpu(), layer.mode, odd)
    ((yr0 * w0.to(DEV)).sum() + (r1 * w1.to(DEV)).sum()).backward()
    G.assert_close(z.grad, zr.grad, _scaled(zr.grad), what='grad z')
    with torch.no_grad():
        x, ldi = layer.backward(y.detach(), ld.clone())
    G.assert_close(x, z, 2 * TOL, what='round trip')
    assert torch.equal(ldi, ld0)


@pytest.mark.parametrize('odd', [False, True])
def test_squeeze2d_layers_incl_odd(nf, odd):
    """Squeeze2d / Unsqueeze2d as layers, with the half swap of ``odd`` (flows/squeeze.py:86-111, :153-189): bit-exact vs the oracle."""
    z = torch.arange(2 * 3 * 4 * 6, dtype=torch.float32).reshape(2, 3, 4, 6)
    ld = torch.zeros(2)
    full = im.squeeze2d(z)                                       # (B, 4 C, H / 2, W / 2) in the odd = False order
    h = full.shape[1] // 2
    want = torch.cat([full[:, h:], full[:, :h]], dim=1) if odd else full
    sq, un = nf.Squeeze2d(odd=odd), nf.Unsqueeze2d(odd=odd)
    got, ld1 = sq(z.to(DEV), ld.to(DEV))
    assert torch.equal(got.cpu(), want) and torch.equal(ld1.cpu(), ld)
    back, _ = sq.backward(got, ld1)
    assert torch.equal(back.cpu(), z)
    f, _ = un(got, ld1)
    assert torch.equal(f.cpu(), z)
    b, _ = un.backward(z.to(DEV), ld.to(DEV))
    assert torch.equal(b.cpu(), want)


def test_weight_norm_multi_ragged_shapes(nf):
    """nf_weight_norm_fwd / _bwd (csrc/weight_norm.hip: 32 columns x 8 row groups per workgroup) on layers whose rows / columns are not
    multiples of the tile, 70 layers in one call (two launches of <= 64), against w = v g / (||v||_dim0 + eps) (flows/weight_norm.py:35-41)
    and its autograd in float64; the accumulate form (gradient sinks) through a second backward."""
    fused = importlib.import_module(nf.__name__ + '.fused')
    torch.manual_seed(3)
    shapes = [(32, 32, 3, 3), (12, 32, 1, 1), (32, 3, 3, 3), (192, 32, 1, 1), (32, 96, 3, 3), (5, 7), (1, 33), (33, 1), (9, 40, 3, 3), (64, 2)]
    shapes = shapes * 7
    eps = 1e-5
    vs = [torch.randn(*s, device=DEV, requires_grad=True) for s in shapes]
    gs = [torch.rand(*((1, ) + s[1:]), device=DEV) + 0.5 for s in shapes]
    for g in gs:
        g.requires_grad_(True)
    tensors = []
    for v, g in zip(vs, gs):
        tensors += [v, g]
    outs = fused._WeightNormMulti.apply(eps, *tensors)
    cot = [torch.randn_like(o) for o in outs]
    grads = torch.autograd.grad(outs, tensors, cot)
    for k, s in enumerate(shapes):
        v64, g64 = vs[k].detach().double().requires_grad_(True), gs[k].detach().double().requires_grad_(True)
        w64 = v64 * g64 / (torch.sqrt((v64 * v64).sum(0, keepdim=True)) + eps)
        gv64, gg64 = torch.autograd.grad(w64, (v64, g64), cot[k].double())
        # the same formula through torch in fp32: its distance from fp64 is the yard-stick (one-row layers differentiate to
        # g_w g eps / den^2 by cancellation: no fp32 evaluation of the formula is within 1e-5 absolute there)
        v32, g32 = vs[k].detach().clone().requires_grad_(True), gs[k].detach().clone().requires_grad_(True)
        w32 = v32 * g32 / (torch.sqrt((v32 * v32).sum(0, keepdim=True)) + eps)
        gv32, gg32 = torch.autograd.grad(w32, (v32, g32), cot[k])
        for got, want, ref32 in ((outs[k], w64.detach(), w32), (grads[2 * k], gv64, gv32), (grads[2 * k + 1], gg64, gg32)):
            assert got.shape == want.shape, (s, got.shape, want.shape)
            bar = TOL * max(1.0, float(want.abs().max())) + SLACK * _gap(ref32, want)
            assert float((got.detach().double() - want).abs().max()) <= bar, s


@pytest.mark.parametrize('K', [288, 864, 32])
def test_bf16_split_is_no_precision_reduction(nf, K):
    """the persistent image-conditioner kernels form every fp32 product as six bf16 x bf16 partial products of three-way
    splits (x = h + m + l exactly), accumulated in fp32 on v_mfma_f32_32x32x16_bf16.  Asserted here, not just probed: on conditioner-like
    operands (weights ~ U(-0.06, 0.06), ReLU-like activations; K = 288 = one 32-channel 3 x 3 layer, 864 = the 96-channel first layer of
    the 4 x 4 level) the split form's max error against float64 is no larger than that of v_mfma_f32_32x32x2_f32 -- the exact fp32
    instruction -- times 1.25 (both are a few ulp of the accumulated magnitude), and both meet the 1e-5 bar relative to max sum|a||b|.
    nf_selftest_gemm32 runs the chain kernels' own nf_cc_split2 / NF_CC_MFMA6 (csrc/conv_chain.hip)."""
    N = nf._native
    worst = []
    for seed in range(8):
        g = torch.Generator().manual_seed(100 + seed)
        A = (torch.rand(32, K, generator=g) * 2 - 1) * 0.06
        Bm = torch.clamp(torch.rand(K, 32, generator=g) * 3 - 1, min=0.0)
        if seed >= 4:                                         # wide dynamic range: normal activations, weights over three decades
            Bm = torch.randn(K, 32, generator=g)
            A = A * torch.pow(10.0, torch.rand(32, K, generator=g) * 3 - 2)
        ref = A.double() @ Bm.double()
        mag = float((A.double().abs() @ Bm.double().abs()).max())
        out = []
        Ad, Bd = A.to(DEV).contiguous(), Bm.to(DEV).contiguous()     # (held: a temporary's block would be handed to the next temporary)
        for mode in (0, 1):
            D = torch.empty(32, 32, device=DEV)
            N.call('nf_selftest_gemm32', N.ptr(Ad), N.ptr(Bd), N.ptr(D), K, mode, N.stream())
            torch.cuda.synchronize()
            out.append(float((D.cpu().double() - ref).abs().max()))
        e_fp32, e_split = out
        worst.append((e_fp32, e_split, mag))
        assert e_split <= 1.25 * e_fp32 + 1e-7 * mag, (K, seed, e_fp32, e_split, mag)
        assert e_split <= 1e-5 * mag and e_fp32 <= 1e-5 * mag, (K, seed, e_fp32, e_split, mag)
    print('K = %d: max |fp32 MFMA - f64| %.3e   max |bf16 x 3 - f64| %.3e' % (K, max(w[0] for w in worst), max(w[1] for w in worst)))


@pytest.mark.parametrize('shape', [(64, 2), (7, 6), (5, 3, 8, 8), (3, 5000)])
def test_sigmoid_tanh_arctanh_modules_match_the_oracle(nf, shape):
    """Sigmoid / Tanh / Arctanh (flows/modules.py:125-183) on csrc/logit.hip: both directions against the oracle's restatement (pinned
    against the live reference in tests/test_oracle_vs_reference.py), values, log-dets and the autograd of the forward directions."""
    from oracle import transforms as tf
    torch.manual_seed(sum(shape))
    x = torch.randn(*shape) * 1.5
    u = torch.rand(*shape)
    t = torch.rand(*shape) * 1.96 - 0.98
    ld0 = torch.randn(shape[0])
    cases = [(nf.Sigmoid(), 'forward', x, lambda a, l: tf.sigmoid(a, l)), (nf.Sigmoid(), 'backward', u, lambda a, l: tf.sigmoid(a, l, inverse=True)),
             (nf.Tanh(), 'forward', x, lambda a, l: tf.tanh(a, l)), (nf.Tanh(), 'backward', t, lambda a, l: tf.tanh(a, l, inverse=True)),
             (nf.Arctanh(), 'forward', t, lambda a, l: tf.tanh(a, l, inverse=True)), (nf.Arctanh(), 'backward', x, lambda a, l: tf.tanh(a, l))]
    n = float(x[0].numel())
    for mod, direction, inp, ora in cases:
        a_g = inp.to(DEV).requires_grad_(direction == 'forward')
        y, ld = getattr(mod, direction)(a_g, ld0.to(DEV).clone())
        # (the oracle in float64: its fp32 autograd of log(1 - tanh^2 x) cancels for |x| > 4, the kernels' analytic derivative does not)
        a_c = inp.double().requires_grad_(direction == 'forward')
        yo, ldo = ora(a_c, ld0.double())
        G.assert_close(y, yo, 2e-6 * max(1.0, float(yo.abs().max())), rtol=2e-6, what='%s.%s y' % (type(mod).__name__, direction))
        G.assert_close(ld, ldo, 2e-6 * n * max(1.0, float(ldo.abs().max()) / n), rtol=1e-5, what='%s.%s ld' % (type(mod).__name__, direction))
        if direction == 'forward':
            wy, wl = torch.randn_like(yo), torch.randn_like(ldo)
            ((y * wy.float().to(DEV)).sum() + (ld * wl.float().to(DEV)).sum()).backward()
            ((yo * wy).sum() + (ldo * wl).sum()).backward()
            G.assert_close(a_g.grad, a_c.grad.float(), 1e-5 * max(1.0, float(a_c.grad.abs().max())), rtol=1e-5, what='%s grad' % type(mod).__name__)


@pytest.mark.parametrize('B,D', [(1, 2), (33, 6), (1000, 2), (5, 64)])
def test_squeeze1d_modules_bit_exact(nf, B, D):
    """Squeeze1d / Unsqueeze1d (flows/squeeze.py:114-151): bit-exact against the oracle's index map, both layers, both directions, odd and
    even; the gradient of the forward direction is the inverse permutation."""
    from oracle import transforms as tf
    z = torch.randn(B, D)
    ld = torch.zeros(B, device=DEV)
    for odd in (False, True):
        zg = z.to(DEV).requires_grad_(True)
        out, ld2 = nf.Squeeze1d(odd).forward(zg, ld)
        assert ld2 is ld
        assert torch.equal(out.detach().cpu(), tf.squeeze1d_layer(z, odd))
        w = torch.randn(B, D)
        (out * w.to(DEV)).sum().backward()
        assert torch.equal(zg.grad.cpu(), tf.squeeze1d_layer(w, odd, inverse=True))
        assert torch.equal(nf.Squeeze1d(odd).backward(z.to(DEV), ld)[0].cpu(), tf.squeeze1d_layer(z, odd, inverse=True))
        assert torch.equal(nf.Unsqueeze1d(odd).forward(z.to(DEV), ld)[0].cpu(), tf.squeeze1d_layer(z, odd, inverse=True))
        assert torch.equal(nf.Unsqueeze1d(odd).backward(z.to(DEV), ld)[0].cpu(), tf.squeeze1d_layer(z, odd))
