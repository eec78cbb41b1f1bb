"""
The oracle against the committed golden vectors captured from the reference (tests/golden/make_goldens.py).
Runs everywhere (CPU only): this is what keeps the oracle pinned on machines without /root/reference.
Tolerance: 1e-5 fp32 (north_star), bit-exact for the index maps, bisection bracket for the Flow++ inverse.
"""
import numpy as np
import pytest
import torch

from oracle import indexmaps as im
from oracle import models as om
from oracle import nets
from oracle import transforms as tf
from tests import _golden as G

TOL = 1e-5


def test_indexmaps_bit_exact():
    for key in G.keys('indexmaps'):
        parts = key.split('/')
        if parts[0] in ('checker', 'channel', '1d') and parts[-1] == 'z0':
            tag, odd = parts[1], parts[2] == 'odd1'
            z = G.group('indexmaps', 'in/' + tag)['']
            mode = {'checker': im.MODE_CHECKER, 'channel': im.MODE_CHANNEL, '1d': im.MODE_1D}[parts[0]]
            g = G.group('indexmaps', '/'.join(parts[:3]) + '/')
            z0, z1 = im.split(z, mode, odd)
            assert torch.equal(z0, g['z0']) and torch.equal(z1, g['z1']), key
            assert torch.equal(im.merge(z0, z1, mode, odd, tuple(z.shape[1:])), z)
        if parts[0] == 'squeeze2d':
            z = G.group('indexmaps', 'in/' + parts[1])['']
            want = G.group('indexmaps', key)['']
            assert torch.equal(im.squeeze2d(z), want)
            assert torch.equal(im.unsqueeze2d(want), z)


@pytest.mark.parametrize('tag,mode', [('1d', im.MODE_1D), ('1d6', im.MODE_1D), ('checker', im.MODE_CHECKER),
                                      ('channel', im.MODE_CHANNEL)])
@pytest.mark.parametrize('odd', [False, True])
def test_affine_coupling(tag, mode, odd):
    g = G.group('ops', 'affine/%s/odd%d/' % (tag, odd))
    a, c = torch.tensor([float(g['meta'][0])], requires_grad=True), torch.tensor([float(g['meta'][1])], requires_grad=True)
    z, params = g['z'].requires_grad_(True), g['params'].requires_grad_(True)
    y, ld = tf.affine_coupling(z, g['ld0'], params, a, c, mode, odd)
    G.assert_close(y, g['y'], TOL, what='y')
    G.assert_close(ld, g['ld'], TOL, what='ld')
    gz, gp, ga, gc = torch.autograd.grad([y, ld], [z, params, a, c], [g['gy'], g['gld']])
    for got, want, n in [(gz, g['gz'], 'gz'), (gp, g['gparams'], 'gparams'), (ga, g['ga'], 'ga'), (gc, g['gc'], 'gc')]:
        G.assert_close(got, want, TOL * max(1.0, float(want.abs().max())), what=n)
    x, ldi = tf.affine_coupling(g['y'], g['ld'], g['params'], a.detach(), c.detach(), mode, odd, inverse=True)
    G.assert_close(x, g['x_inv'], TOL, what='x_inv')
    G.assert_close(ldi, g['ld_inv'], TOL, what='ld_inv')


@pytest.mark.parametrize('tag', ['2d', 'img'])
def test_actnorm(tag):
    g = G.group('ops', 'actnorm/%s/' % tag)
    ls, b = tf.actnorm_init(g['z'])
    G.assert_close(ls, g['log_scale'], 1e-6, what='init log_scale')
    G.assert_close(b, g['bias'], 1e-6, what='init bias')
    z, ls, b = g['z'].requires_grad_(True), g['log_scale'].requires_grad_(True), g['bias'].requires_grad_(True)
    y, ld = tf.actnorm(z, g['ld0'], ls, b)
    G.assert_close(y, g['y'], TOL)
    G.assert_close(ld, g['ld'], TOL)
    gz, gls, gb = torch.autograd.grad([y, ld], [z, ls, b], [g['gy'], g['gld']])
    G.assert_close(gz, g['gz'], TOL)
    G.assert_close(gls, g['glog_scale'], TOL * max(1.0, float(g['glog_scale'].abs().max())))
    G.assert_close(gb, g['gbias'], TOL * max(1.0, float(g['gbias'].abs().max())))
    x, ldi = tf.actnorm(g['y'], g['ld'], g['log_scale'], g['bias'], inverse=True)
    G.assert_close(x, g['x_inv'], TOL)
    G.assert_close(ldi, g['ld_inv'], TOL)


@pytest.mark.parametrize('C', [2, 3, 12, 48])
def test_invconv(C):
    g = G.group('ops', 'invconv/%d/' % C)
    L, U, log_s = g['L'].requires_grad_(True), g['U'].requires_grad_(True), g['log_s'].requires_grad_(True)
    z = g['z'].requires_grad_(True)
    W = tf.invconv_weight(g['P'], L, U, g['I'], g['L_mask'], g['U_mask'], g['sign_s'], log_s)
    y, ld = tf.invconv(z, g['ld0'], W, log_s)
    G.assert_close(y, g['y'], TOL)
    G.assert_close(ld, g['ld'], TOL)
    gz, gL, gU, gs = torch.autograd.grad([y, ld], [z, L, U, log_s], [g['gy'], g['gld']])
    for got, want in [(gz, g['gz']), (gL, g['gL']), (gU, g['gU']), (gs, g['glog_s'])]:
        G.assert_close(got, want, TOL * max(1.0, float(want.abs().max())))
    x, ldi = tf.invconv_inverse(g['y'], g['ld'], g['L'], g['U'], g['L_mask'], g['U_mask'], g['sign_s'], g['log_s'],
                                g['pivots'])
    G.assert_close(x, g['x_inv'], TOL)
    G.assert_close(ldi, g['ld_inv'], TOL)


@pytest.mark.parametrize('tag', ['2d', 'img'])
def test_flow_bn(tag):
    run_mean = run_var = None
    for step in range(2):
        g = G.group('ops', 'flowbn/%s/step%d/' % (tag, step))
        x = g['x'].requires_grad_(True)
        mean, var = tf.flow_bn_stats(x)
        G.assert_close(mean, g['batch_mean'], 1e-6)
        G.assert_close(var, g['batch_var'], 1e-6)
        run_mean = (torch.zeros_like(mean) if run_mean is None else run_mean) * 0.9 + mean * 0.1
        run_var = (torch.ones_like(var) if run_var is None else run_var) * 0.9 + var * 0.1
        G.assert_close(run_mean, g['running_mean'], 1e-6)
        G.assert_close(run_var, g['running_var'], 1e-6)
        zero = torch.zeros_like(mean)
        y, ld = tf.flow_bn(x, g['ld0'], mean, var, zero, zero)
        G.assert_close(y, g['y'], TOL)
        G.assert_close(ld, g['ld'], TOL)
        (gx, ) = torch.autograd.grad([y], [x], [g['gy']])
        G.assert_close(gx, g['gx'], TOL)
        xi, ldi = tf.flow_bn(g['y'], g['ld'], mean, var, zero, zero, inverse=True)
        G.assert_close(xi, g['x_inv'], TOL)
        G.assert_close(ldi, g['ld_inv'], TOL)
    g = G.group('ops', 'flowbn/%s/eval/' % tag)
    zero = torch.zeros_like(run_mean)
    y, ld = tf.flow_bn(g['x'], g['ld0'], run_mean, run_var, zero, zero)
    G.assert_close(y, g['y'], TOL)
    G.assert_close(ld, g['ld'], TOL)
    xi, ldi = tf.flow_bn(g['y'], g['ld'], run_mean, run_var, zero, zero, inverse=True)
    G.assert_close(xi, g['x_inv'], TOL)
    G.assert_close(ldi, g['ld_inv'], TOL)


@pytest.mark.parametrize('eps', [1.0e-5, 0.01])
def test_logit(eps):
    g = G.group('ops', 'logit/%g/' % eps)
    x = g['x'].requires_grad_(True)
    y, ld = tf.logit(x, g['ld0'], eps)
    G.assert_close(y, g['y'], TOL)
    G.assert_close(ld, g['ld'], TOL, rtol=2e-6)
    (gx, ) = torch.autograd.grad([y, ld], [x], [g['gy'], g['gld']])
    G.assert_close(gx, g['gx'], TOL, rtol=1e-5)
    yin = g['yin'].requires_grad_(True)
    xi, ldi = tf.logit_inverse(yin, g['ld0'])
    G.assert_close(xi, g['x_inv'], TOL)
    G.assert_close(ldi, g['ld_inv'], TOL)
    (gyin, ) = torch.autograd.grad([xi, ldi], [yin], [g['gy'], g['gld']])
    G.assert_close(gyin, g['gyin'], TOL)


@pytest.mark.parametrize('tag', ['K4_2d', 'K8_2d', 'K4_4d'])
def test_mixlogcdf(tag):
    g = G.group('ops', 'mixlogcdf/%s/' % tag)
    x, lp = g['x'].requires_grad_(True), g['logpi_raw'].requires_grad_(True)
    mu, s = g['mu'].requires_grad_(True), g['s'].requires_grad_(True)
    logpi = torch.log_softmax(lp, dim=1)
    y, ld = tf.mixlogcdf(x, g['ld0'], logpi, mu, s)
    G.assert_close(y, g['y'], TOL)
    G.assert_close(ld, g['ld'], TOL)
    grads = torch.autograd.grad([y, ld], [x, lp, mu, s], [g['gy'], g['gld']])
    for got, n in zip(grads, ['gx', 'glogpi_raw', 'gmu', 'gs']):
        G.assert_close(got, g[n], TOL * max(1.0, float(g[n].abs().max())), what=n)
    with torch.no_grad():
        xi, ldi, it = tf.mixlogcdf_inverse(g['target'], g['ld0'], logpi, mu, s, return_iters=True)
        assert torch.equal(xi, g['x_inv']) and it in (25, 100)
        G.assert_close(ldi, g['ld_inv'], TOL)
        xi, ldi, it = tf.mixlogcdf_inverse(g['target100'], g['ld0'], logpi, mu, s, return_iters=True)
        assert it == 100 and torch.equal(xi, g['x_inv100'])
        G.assert_close(ldi, g['ld_inv100'], TOL)


@pytest.mark.parametrize('tag,mode,dims', [('1d', im.MODE_1D, (2, )), ('checker', im.MODE_CHECKER, (2, 4, 4)),
                                           ('channel', im.MODE_CHANNEL, (4, 4, 4))])
@pytest.mark.parametrize('odd', [False, True])
def test_mixlog_coupling(tag, mode, dims, odd):
    g = G.group('ops', 'mixlog/%s/odd%d/' % (tag, odd))
    a0, c0, K = float(g['meta'][0]), float(g['meta'][1]), int(g['meta'][2])
    a, c = torch.tensor([a0], requires_grad=True), torch.tensor([c0], requires_grad=True)
    z, params = g['z'].requires_grad_(True), g['params'].requires_grad_(True)
    oc = params.shape[1] // (2 + 3 * K)
    sections = [oc] * 2 + [oc * K] * 3
    y, ld = tf.mixlog_coupling(z, g['ld0'], params, sections, K, a, c, mode, odd)
    G.assert_close(y, g['y'], TOL)
    G.assert_close(ld, g['ld'], TOL)
    grads = torch.autograd.grad([y, ld], [z, params, a, c], [g['gy'], g['gld']])
    for got, n in zip(grads, ['gz', 'gparams', 'ga', 'gc']):
        G.assert_close(got, g[n], TOL * max(1.0, float(g[n].abs().max())), what=n)
    with torch.no_grad():
        x, ldi = tf.mixlog_coupling(g['y'], g['ld'], g['params'], sections, K, a, c, mode, odd, inverse=True)
    G.assert_close(x, g['x_inv'], 1e-4, what='x_inv (bisection bracket)')
    G.assert_close(ldi, g['ld_inv'], 2e-3, what='ld_inv')


class _FixedMasks:
    """replays stored MADE masks instead of drawing them (D > 2 draws are RNG-dependent in the reference)."""

    def __init__(self, flat, D, nh=3, H=32):
        shapes = [(H, D)] + [(H, H)] * (nh - 1) + [(D, H)]
        self.masks, o = [], 0
        for sh in shapes:
            n = sh[0] * sh[1]
            self.masks.append(flat[o:o + n].reshape(sh))
            o += n


@pytest.mark.parametrize('D', [2, 5])
def test_ar_transform(D):
    g = G.group('ops', 'ar/%d/' % D)
    sd = {k[len('sd/'):]: v.clone() for k, v in g.items() if k.startswith('sd/')}
    layer_plan = dict(op='ar', prefix='', dims=(D, ))
    ora = om.FlowOracle('maf', (D, ), None, 0, sd, training=True)
    ms, mt = _FixedMasks(g['masks_s'], D).masks, _FixedMasks(g['masks_t'], D).masks
    seq = iter([ms, mt] * 64)
    ora._made_masks = lambda D_, nh: next(seq)
    if D == 2:                                   # degenerate draw: the oracle's own mask rule must reproduce it
        own = nets.made_masks(2, 3, 32, np.random.RandomState(0))
        assert all(torch.equal(a, b) for a, b in zip(own, ms))
    names = [k for k in sd if sd[k].is_floating_point() and 'running' not in k and k != 'perm']
    for k in names:
        sd[k].requires_grad_(True)
    z = g['z'].requires_grad_(True)
    y, ld = ora._apply(layer_plan, z, g['ld0'], False)
    G.assert_close(y, g['y'], TOL)
    G.assert_close(ld, g['ld'], TOL)
    grads = torch.autograd.grad([y, ld], [z] + [sd[k] for k in names], [g['gy'], g['gld']], allow_unused=True)
    G.assert_close(grads[0], g['gz'], TOL * max(1.0, float(g['gz'].abs().max())))
    for k, got in zip(names, grads[1:]):
        want = g['grad/' + k]
        got = torch.zeros_like(want) if got is None else got
        G.assert_close(got, want, TOL * max(1.0, float(want.abs().max())), what=k)
    for k in sd:
        G.assert_close(sd[k].detach().float(), g['sd_after/' + k].float(), 1e-6, what=k)
    ora.training = False
    del ora._made_masks                          # eval passes: the oracle's OWN mask rule on the same RNG stream
    with torch.no_grad():
        ora.mask_rng = np.random.RandomState(99)     # np.random.seed(99) in make_goldens.py == this legacy stream
        ye, lde = ora._apply(layer_plan, g['z'], g['ld0'], False)
        G.assert_close(ye, g['y_eval'], TOL)
        G.assert_close(lde, g['ld_eval'], TOL)
        ora.mask_rng = np.random.RandomState(99)
        xi, ldi = ora._apply(layer_plan, g['y_eval'], g['ld_eval'], True)
        G.assert_close(xi, g['x_inv'], TOL)
        G.assert_close(ldi, g['ld_inv'], TOL)


@pytest.mark.parametrize('name', list(G.MODEL_CASES))
def test_models(name):
    kind, _, dims, datatype, layers, mix = G.MODEL_CASES[name]
    sd = G.group('model_' + name, 'sd0/')
    g = G.group('model_' + name, '')
    ora = om.FlowOracle(kind, dims, datatype, layers, sd, mixtures=mix, training=True, logdet='exact', spnorm_coeff=0.9)
    ora.requires_grad_(True)
    G.seed_noise(777)
    z, ld = ora.forward(g['y'])
    G.assert_close(z, g['train/z'], TOL)
    G.assert_close(ld, g['train/ld'], TOL, rtol=2e-6)
    loss = tf.nll_loss(z, ld)
    G.assert_close(loss, g['train/loss'], TOL * max(1.0, abs(float(g['train/loss'])) / np.prod(dims)))
    loss.backward()
    n = 0
    for k, v in ora.parameters().items():
        if 'grad/' + k in g:
            want = g['grad/' + k]
            G.assert_close(v.grad, want, 2 * TOL * max(1.0, float(want.abs().max())), what=k)
            n += 1
    assert n > 4
    for k, want in G.group('model_' + name, 'sd1/').items():
        G.assert_close(sd[k].detach().float(), want.float(), 1e-6, what=k)
    ora.requires_grad_(False)
    tol_inv = 2e-4 if kind == 'flowpp' else TOL
    with torch.no_grad():
        G.seed_noise(778)
        x, ldi = ora.backward(g['train/z'])
        G.assert_close(x, g['train/x_inv'], tol_inv)
        G.assert_close(ldi, g['train/ld_inv'], 10 * tol_inv)
        ora.training = False
        G.seed_noise(779)
        z, ld = ora.forward(g['y'])
        G.assert_close(z, g['eval/z'], TOL)
        G.assert_close(ld, g['eval/ld'], TOL, rtol=2e-6)
        x, ldi = ora.backward(g['eval/z'])
        G.assert_close(x, g['eval/x_inv'], tol_inv)
        G.assert_close(ldi, g['eval/ld_inv'], 10 * tol_inv)
