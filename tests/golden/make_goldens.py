"""
Generates the golden fixtures under tests/golden/ by RUNNING THE UPSTREAM REFERENCE
(/root/reference/flows, imported under the alias ``ref_flows``) on seeded inputs.

    python tests/golden/make_goldens.py            # only works where /root/reference exists

The fixtures are data only (inputs, parameters, expected outputs, expected gradients); nothing of the
reference's source is stored.  They travel to the GPU box, the reference does not.

Files (np.savez_compressed, all fp32 unless integer):
  indexmaps.npz   G1   arange tensors through every split / merge / squeeze map
  ops.npz         G2-G8, G10  per-layer forward / inverse / log-det / autograd gradients with stub conditioners
  model_<name>.npz G9  small end-to-end models: initial state_dict, input, (z, ld) train+eval, loss, all grads,
                        mutated state, inverse
  mainloop_<name>.npz G11 main.py's train_on_batch (Adam + StepLR) for 3 steps: per-step batch, z, loss; final state_dict
"""
import importlib
import os
import sys
from types import SimpleNamespace as NS

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from tests._ref import load_reference  # noqa: E402

ref = load_reference()
if ref is None:
    sys.exit('reference not available: goldens can only be generated in the authoring container')
rmod = importlib.import_module('ref_flows.modules')
rsq = importlib.import_module('ref_flows.squeeze')
rcp = importlib.import_module('ref_flows.coupling')
rmaf = importlib.import_module('ref_flows.maf')


def npy(t):
    return t.detach().cpu().numpy().copy()


def gen(seed):
    return torch.Generator().manual_seed(seed)


# ----------------------------------------------------------------------------------------------------------------------
def make_indexmaps():
    out = {}
    for dims in [(3, 4, 4), (12, 4, 4), (2, 6, 4)]:
        z = torch.arange(2 * int(np.prod(dims)), dtype=torch.float32).reshape((2, ) + dims)
        tag = 'x'.join(map(str, dims))
        out['in/%s' % tag] = npy(z).astype(np.int32)
        for odd in (False, True):
            z0, z1 = rsq.checker_split(z, odd)
            out['checker/%s/odd%d/z0' % (tag, odd)] = npy(z0).astype(np.int32)
            out['checker/%s/odd%d/z1' % (tag, odd)] = npy(z1).astype(np.int32)
            assert torch.equal(rsq.checker_merge(z0, z1, odd), z)
            if dims[0] % 2 == 0:
                z0, z1 = rsq.channel_split(z, 1, odd)
                out['channel/%s/odd%d/z0' % (tag, odd)] = npy(z0).astype(np.int32)
                out['channel/%s/odd%d/z1' % (tag, odd)] = npy(z1).astype(np.int32)
        zs, _ = rsq.Squeeze2d()(z, None)
        out['squeeze2d/%s' % tag] = npy(zs).astype(np.int32)
        assert torch.equal(rsq.Unsqueeze2d()(zs, None)[0], z)
    for D in (2, 6):
        z = torch.arange(3 * D, dtype=torch.float32).reshape(3, D)
        out['in/%d' % D] = npy(z).astype(np.int32)
        for odd in (False, True):
            z0, z1 = rsq.squeeze1d(z, odd)
            out['1d/%d/odd%d/z0' % (D, odd)] = npy(z0).astype(np.int32)
            out['1d/%d/odd%d/z1' % (D, odd)] = npy(z1).astype(np.int32)
            assert torch.equal(rsq.unsqueeze1d(z0, z1, odd), z)
    np.savez_compressed(os.path.join(HERE, 'indexmaps.npz'), **out)


# ----------------------------------------------------------------------------------------------------------------------
class _Stub(torch.nn.Module):
    """conditioner stand-in: ignores its input and returns a stored tensor (so the TRANSFORM is isolated)."""

    def __init__(self, params):
        super().__init__()
        self.params = params

    def forward(self, x):
        return self.params


def _grads(outs, gouts, wrt):
    gs = torch.autograd.grad(outs, wrt, gouts, allow_unused=True)
    return [torch.zeros_like(w) if g is None else g for g, w in zip(gs, wrt)]


def make_ops():
    out = {}

    # ---- G2 / G10 affine coupling -------------------------------------------------------------------------------
    for tag, dims, masking, B in [('1d', (2, ), 'checkerboard', 64), ('1d6', (6, ), 'checkerboard', 16),
                                  ('checker', (3, 8, 8), 'checkerboard', 2), ('channel', (12, 4, 4), 'channelwise', 2)]:
        for odd in (False, True):
            g = gen(10 + odd)
            torch.manual_seed(5)
            layer = rcp.AffineCoupling(dims, masking=masking, odd=odd)
            with torch.no_grad():
                layer.s_log_scale.fill_(0.7)
                layer.s_bias.fill_(-0.2)
            z = (torch.randn((B, ) + dims, generator=g) * 0.8).requires_grad_(True)
            z0, _ = layer.squeeze(z)
            pshape = list(z0.shape)
            pshape[1] *= 2
            params = (torch.randn(pshape, generator=g) * 0.7).requires_grad_(True)
            layer.net = _Stub(params)
            ld0 = torch.randn(B, generator=g)
            y, ld = layer(z, ld0.clone())
            gy, gld = torch.randn(y.shape, generator=g), torch.randn(B, generator=g)
            gz, gp, ga, gc = _grads([y, ld], [gy, gld], [z, params, layer.s_log_scale, layer.s_bias])
            x, ldi = layer.backward(y.detach(), ld.detach().clone())
            k = 'affine/%s/odd%d/' % (tag, odd)
            for n, v in [('z', z), ('params', params), ('ld0', ld0), ('y', y), ('ld', ld), ('gy', gy), ('gld', gld),
                         ('gz', gz), ('gparams', gp), ('ga', ga), ('gc', gc), ('x_inv', x), ('ld_inv', ldi)]:
                out[k + n] = npy(v)
            out[k + 'meta'] = np.array([0.7, -0.2], dtype=np.float32)

    # ---- G3 ActNorm ---------------------------------------------------------------------------------------------
    for tag, dims, B in [('2d', (2, ), 64), ('img', (12, 4, 4), 4)]:
        g = gen(20)
        layer = rmod.ActNorm(dims)
        z = (torch.randn((B, ) + dims, generator=g) * 1.7 + 0.3).requires_grad_(True)
        ld0 = torch.randn(B, generator=g)
        y, ld = layer(z, ld0.clone())                              # data-dependent init happens here
        gy, gld = torch.randn(y.shape, generator=g), torch.randn(B, generator=g)
        gz, gls, gb = _grads([y, ld], [gy, gld], [z, layer.log_scale, layer.bias])
        x, ldi = layer.backward(y.detach(), ld.detach().clone())
        k = 'actnorm/%s/' % tag
        for n, v in [('z', z), ('ld0', ld0), ('log_scale', layer.log_scale), ('bias', layer.bias), ('y', y), ('ld', ld),
                     ('gy', gy), ('gld', gld), ('gz', gz), ('glog_scale', gls), ('gbias', gb), ('x_inv', x),
                     ('ld_inv', ldi)]:
            out[k + n] = npy(v)

    # ---- G4 invertible 1x1 --------------------------------------------------------------------------------------
    for C, spatial, B in [(2, (), 64), (3, (4, 4), 3), (12, (4, 4), 2), (48, (2, 2), 2)]:
        g = gen(30 + C)
        torch.manual_seed(30 + C)
        layer = rmod.InvertibleConv1x1(C)
        with torch.no_grad():                                      # move away from the orthogonal init
            layer.L.add_(torch.randn(C, C, generator=g) * 0.05)
            layer.U.add_(torch.randn(C, C, generator=g) * 0.05)
            layer.log_s.add_(torch.randn(C, generator=g) * 0.1)
        z = torch.randn((B, C) + spatial, generator=g).requires_grad_(True)
        ld0 = torch.randn(B, generator=g)
        y, ld = layer(z, ld0.clone())
        gy, gld = torch.randn(y.shape, generator=g), torch.randn(B, generator=g)
        gz, gL, gU, gs = _grads([y, ld], [gy, gld], [z, layer.L, layer.U, layer.log_s])
        x, ldi = layer.backward(y.detach(), ld.detach().clone())
        k = 'invconv/%d/' % C
        for n in ('P', 'L', 'U', 'I', 'pivots', 'L_mask', 'U_mask', 'log_s', 'sign_s'):
            out[k + n] = npy(getattr(layer, n))
        for n, v in [('z', z), ('ld0', ld0), ('y', y), ('ld', ld), ('gy', gy), ('gld', gld), ('gz', gz), ('gL', gL),
                     ('gU', gU), ('glog_s', gs), ('x_inv', x), ('ld_inv', ldi)]:
            out[k + n] = npy(v)

    # ---- G5 flow BatchNorm --------------------------------------------------------------------------------------
    for tag, dims, B in [('2d', (2, ), 64), ('img', (6, 4, 4), 4)]:
        g = gen(40)
        layer = rmod.BatchNorm(dims, affine=False)
        layer.train()
        k = 'flowbn/%s/' % tag
        for step in range(2):
            x = (torch.randn((B, ) + dims, generator=g) * (1.5 + step) - 0.4).requires_grad_(True)
            ld0 = torch.randn(B, generator=g)
            y, ld = layer(x, ld0.clone())
            gy = torch.randn(y.shape, generator=g)
            (gx, ) = _grads([y], [gy], [x])
            xi, ldi = layer.backward(y.detach(), ld.detach().clone())      # train-mode inverse: batch stats
            for n, v in [('x', x), ('ld0', ld0), ('y', y), ('ld', ld), ('gy', gy), ('gx', gx), ('x_inv', xi),
                         ('ld_inv', ldi), ('running_mean', layer.running_mean), ('running_var', layer.running_var),
                         ('batch_mean', layer.batch_mean), ('batch_var', layer.batch_var)]:
                out[k + 'step%d/' % step + n] = npy(v)
        layer.eval()
        x = torch.randn((B, ) + dims, generator=g)
        ld0 = torch.randn(B, generator=g)
        y, ld = layer(x, ld0.clone())
        xi, ldi = layer.backward(y, ld.clone())
        for n, v in [('x', x), ('ld0', ld0), ('y', y), ('ld', ld), ('x_inv', xi), ('ld_inv', ldi)]:
            out[k + 'eval/' + n] = npy(v)

    # ---- G6 Logit -----------------------------------------------------------------------------------------------
    for eps in (1.0e-5, 0.01):
        g = gen(50)
        layer = rmod.Logit(eps)
        x = torch.rand(8, 3, 4, 4, generator=g)
        x.view(-1)[:6] = torch.tensor([0.0, eps / 2, eps, 1.0 - eps, 1.0 - eps / 2, 1.0])
        x.requires_grad_(True)
        ld0 = torch.randn(8, generator=g)
        y, ld = layer(x, ld0.clone())
        gy, gld = torch.randn(y.shape, generator=g), torch.randn(8, generator=g)
        (gx, ) = _grads([y, ld], [gy, gld], [x])
        yin = (torch.randn(8, 3, 4, 4, generator=g) * 4).requires_grad_(True)
        xi, ldi = layer.backward(yin, ld0.clone())
        (gyin, ) = _grads([xi, ldi], [gy, gld], [yin])
        k = 'logit/%g/' % eps
        for n, v in [('x', x), ('ld0', ld0), ('y', y), ('ld', ld), ('gy', gy), ('gld', gld), ('gx', gx), ('yin', yin),
                     ('x_inv', xi), ('ld_inv', ldi), ('gyin', gyin)]:
            out[k + n] = npy(v)

    # ---- G7 MixLogCDF + Flow++ coupling chain -------------------------------------------------------------------
    for K, shape in [(4, (33, 1)), (8, (65, 1)), (4, (2, 4, 4, 4))]:
        g = gen(60 + K)
        B = shape[0]
        C = shape[1:]
        x = (torch.randn(shape, generator=g) * 1.5).requires_grad_(True)
        logpi_raw = torch.randn((B, K) + C, generator=g).requires_grad_(True)
        logpi = torch.log_softmax(logpi_raw, dim=1)
        mu = (torch.randn((B, K) + C, generator=g) * 1.2).requires_grad_(True)
        s = (torch.randn((B, K) + C, generator=g) * 0.5).requires_grad_(True)
        ld0 = torch.randn(B, generator=g)
        layer = rmod.MixLogCDF()
        y, ld = layer(x, logpi, mu, s, ld0.clone())
        gy, gld = torch.randn(y.shape, generator=g), torch.randn(B, generator=g)
        gx, glp, gmu, gs = _grads([y, ld], [gy, gld], [x, logpi_raw, mu, s])
        k = 'mixlogcdf/K%d_%dd/' % (K, len(shape))
        with torch.no_grad():
            target = torch.rand(shape, generator=g) * 0.96 + 0.02
            xi, ldi = layer.backward(target.clone(), logpi, mu, s, ld0.clone())
            # 100-iteration regime: one element whose target is EXACTLY the CDF at the first midpoint (0.0)
            t100 = target.clone()
            first = (0, ) * len(shape)
            t100[first] = torch.exp(rmod.mix_logistic_logcdf(torch.zeros(shape), logpi, mu, s))[first]
            xi100, ldi100 = layer.backward(t100.clone(), logpi, mu, s, ld0.clone())
        for n, v in [('x', x), ('logpi_raw', logpi_raw), ('mu', mu), ('s', s), ('ld0', ld0), ('y', y), ('ld', ld),
                     ('gy', gy), ('gld', gld), ('gx', gx), ('glogpi_raw', glp), ('gmu', gmu), ('gs', gs),
                     ('target', target), ('x_inv', xi), ('ld_inv', ldi), ('target100', t100), ('x_inv100', xi100),
                     ('ld_inv100', ldi100)]:
            out[k + n] = npy(v)

    for tag, dims, masking, B, K in [('1d', (2, ), 'checkerboard', 64, 8), ('checker', (2, 4, 4), 'checkerboard', 2, 4),
                                     ('channel', (4, 4, 4), 'channelwise', 2, 4)]:
        for odd in (False, True):
            g = gen(70 + odd)
            torch.manual_seed(7)
            layer = rcp.MixLogAttnCoupling(dims, masking=masking, odd=odd, n_mixtures=K)
            with torch.no_grad():
                layer.a_log_scale.fill_(0.6)
                layer.a_bias.fill_(0.1)
            z = (torch.randn((B, ) + dims, generator=g) * 0.8).requires_grad_(True)
            z0, _ = layer.squeeze(z)
            pshape = list(z0.shape)
            pshape[1] = sum(layer.sections)
            params = (torch.randn(pshape, generator=g) * 0.7).requires_grad_(True)
            layer.net = _Stub(params)
            ld0 = torch.randn(B, generator=g)
            y, ld = layer(z, ld0.clone())
            gy, gld = torch.randn(y.shape, generator=g), torch.randn(B, generator=g)
            gz, gp, ga, gc = _grads([y, ld], [gy, gld], [z, params, layer.a_log_scale, layer.a_bias])
            with torch.no_grad():
                x, ldi = layer.backward(y.detach(), ld.detach().clone())
            k = 'mixlog/%s/odd%d/' % (tag, odd)
            for n, v in [('z', z), ('params', params), ('ld0', ld0), ('y', y), ('ld', ld), ('gy', gy), ('gld', gld),
                         ('gz', gz), ('gparams', gp), ('ga', ga), ('gc', gc), ('x_inv', x), ('ld_inv', ldi)]:
                out[k + n] = npy(v)
            out[k + 'meta'] = np.array([0.6, 0.1, K], dtype=np.float32)

    # ---- G8 MADE / autoregressive transform ---------------------------------------------------------------------
    for D, B in [(2, 64), (5, 32)]:
        g = gen(80 + D)
        torch.manual_seed(80 + D)
        layer = rmaf.AutoregressiveTransfrom(D)
        with torch.no_grad():
            layer.s_log_scale.fill_(0.5)
            layer.s_bias.fill_(0.05)
        layer.train()
        k = 'ar/%d/' % D
        for n, v in layer.state_dict().items():
            out[k + 'sd/' + n] = npy(v)
        z = (torch.randn(B, D, generator=g) * 0.9).requires_grad_(True)
        ld0 = torch.randn(B, generator=g)
        np.random.seed(1234)                                       # masks are drawn from np.random on every call
        y, ld = layer(z, ld0.clone())
        out[k + 'masks_s'] = np.concatenate([npy(m).reshape(-1) for m in layer.net_s.masks])
        out[k + 'masks_t'] = np.concatenate([npy(m).reshape(-1) for m in layer.net_t.masks])
        gy, gld = torch.randn(y.shape, generator=g), torch.randn(B, generator=g)
        names = [n for n, p in layer.named_parameters()]
        gs = _grads([y, ld], [gy, gld], [z] + [p for _, p in layer.named_parameters()])
        out[k + 'gz'] = npy(gs[0])
        for n, gv in zip(names, gs[1:]):
            out[k + 'grad/' + n] = npy(gv)
        for n, v in layer.state_dict().items():
            out[k + 'sd_after/' + n] = npy(v)
        layer.eval()
        with torch.no_grad():
            np.random.seed(99)
            ye, lde = layer(z.detach(), ld0.clone())
            np.random.seed(99)
            xi, ldi = layer.backward(ye.clone(), lde.clone())
        for n, v in [('z', z), ('ld0', ld0), ('y', y), ('ld', ld), ('gy', gy), ('gld', gld), ('y_eval', ye),
                     ('ld_eval', lde), ('x_inv', xi), ('ld_inv', ldi)]:
            out[k + n] = npy(v)

    np.savez_compressed(os.path.join(HERE, 'ops.npz'), **out)


# ----------------------------------------------------------------------------------------------------------------------
MODELS = [
    # name, class, dims, datatype, layers, mixtures, batch
    ('realnvp2d', 'RealNVP', (2, ), '2d', 2, None, 64),
    ('glow2d', 'Glow', (2, ), '2d', 2, None, 64),
    ('flowpp2d', 'Flowpp', (2, ), '2d', 2, 8, 64),
    ('maf2d', 'MAF', (2, ), '2d', 2, None, 64),
    ('glow_img', 'Glow', (3, 16, 16), 'image', 1, None, 4),
    ('resflow2d', 'ResFlow', (2, ), '2d', 2, None, 64),          # logdet='exact' in eval; seeded noise in training
    # round 3: the image stacks of the two other flows north_star names (realnvp.py:17-47, flowpp.py:17-62)
    ('realnvp_img', 'RealNVP', (3, 16, 16), 'image', 1, None, 4),
    ('flowpp_img', 'Flowpp', (3, 16, 16), 'image', 1, 4, 4),
]


def make_models(only=None):
    for name, cls, dims, datatype, layers, mix, B in MODELS:
        if only and name not in only:
            continue
        torch.manual_seed(100)
        np.random.seed(100)
        net = getattr(ref, cls)(dims, datatype, NS(layers=layers, mixtures=mix, logdet='exact', spnorm_coeff=0.9))
        out = {'meta/dims': np.array(dims), 'meta/layers': np.array(layers), 'meta/mixtures': np.array(mix or 0)}
        for k, v in net.state_dict().items():
            out['sd0/' + k] = npy(v)
        g = gen(101)
        y = torch.rand((B, ) + dims, generator=g) if datatype == 'image' else torch.randn((B, ) + dims, generator=g) * 0.5
        out['y'] = npy(y)
        net.train()
        torch.manual_seed(777)                                     # Hutchinson noise / series length of ResFlow
        np.random.seed(777)
        z, ld = net(y.clone())
        zf = z.reshape(B, -1)
        D = zf.shape[1]
        loss = -torch.mean(-0.5 * (zf * zf).sum(1) - 0.5 * D * np.log(2 * np.pi) + ld)
        mvn = torch.distributions.MultivariateNormal(torch.zeros(D), torch.eye(D))
        loss_ref = -1.0 * torch.mean(mvn.log_prob(zf) + ld)        # main.py:85 verbatim semantics
        assert abs(float(loss) - float(loss_ref)) < 1e-4 * max(1.0, abs(float(loss_ref)))
        loss_ref.backward()
        out['train/z'], out['train/ld'], out['train/loss'] = npy(z), npy(ld), npy(loss_ref)
        for k, p in net.named_parameters():
            if p.grad is not None:
                out['grad/' + k] = npy(p.grad)
        for k, v in net.state_dict().items():                      # only what the forward pass mutated
            if not np.array_equal(npy(v), out['sd0/' + k]):
                out['sd1/' + k] = npy(v)
        with torch.no_grad():
            torch.manual_seed(778)
            np.random.seed(778)
            xt, ldt = net.backward(z.detach().clone())             # train-mode inverse (batch stats)
            out['train/x_inv'], out['train/ld_inv'] = npy(xt), npy(ldt)
            net.eval()
            torch.manual_seed(779)
            np.random.seed(779)
            ze, lde = net(y.clone())
            xe, ldie = net.backward(ze.clone())
            out['eval/z'], out['eval/ld'], out['eval/x_inv'], out['eval/ld_inv'] = npy(ze), npy(lde), npy(xe), npy(ldie)
        np.savez_compressed(os.path.join(HERE, 'model_%s.npz' % name), **out)


# ----------------------------------------------------------------------------------------------------------------------
MAINLOOP = [('glow2d', 'Glow', (2, ), '2d', 2, None, 64), ('realnvp2d', 'RealNVP', (2, ), '2d', 2, None, 64),
            ('glow_img', 'Glow', (3, 16, 16), 'image', 1, None, 4)]
MAINLOOP_STEPS = 3


def make_mainloop():
    """G11: ``Model.train_on_batch`` of the reference's main.py (:78-92) run for a few steps with main.py's own optimizer set-up
    (:56-71, configs/default.yaml: Adam lr 1e-4, betas (0.9, 0.999), StepLR) on the reference's classes.  The drop-in test replays it
    on the shim's classes (dropin/flows) and compares z / loss per step and the final parameters."""
    for name, cls, dims, datatype, layers, mix, B in MAINLOOP:
        torch.manual_seed(100)
        np.random.seed(100)
        net = getattr(ref, cls)(dims, datatype, NS(layers=layers, mixtures=mix, logdet='exact', spnorm_coeff=0.9))
        out = {'meta/dims': np.array(dims), 'meta/layers': np.array(layers), 'meta/steps': np.array(MAINLOOP_STEPS)}
        for k, v in net.state_dict().items():
            out['sd0/' + k] = npy(v)
        D = int(np.prod(dims))
        normal = torch.distributions.MultivariateNormal(torch.zeros(D), torch.eye(D))
        optim = torch.optim.Adam(net.parameters(), lr=1.0e-4, betas=(0.9, 0.999), weight_decay=0.0)
        sched = torch.optim.lr_scheduler.StepLR(optim, step_size=10000, gamma=0.5)
        net.train()
        g = gen(202)
        for s in range(MAINLOOP_STEPS):
            y = torch.rand((B, ) + dims, generator=g) if datatype == 'image' else torch.randn((B, ) + dims, generator=g) * 0.5
            z, ld = net(y.contiguous())
            z = z.view(y.size(0), -1)
            loss = -1.0 * torch.mean(normal.log_prob(z) + ld)
            optim.zero_grad()
            loss.backward()
            optim.step()
            sched.step()
            out['step%d/y' % s], out['step%d/z' % s], out['step%d/loss' % s] = npy(y), npy(z), npy(loss)
        for k, v in net.state_dict().items():
            out['sdN/' + k] = npy(v)
        np.savez_compressed(os.path.join(HERE, 'mainloop_%s.npz' % name), **out)


if __name__ == '__main__':
    import warnings
    warnings.filterwarnings('ignore')
    if sys.argv[1:] == ['mainloop']:
        make_mainloop()
    elif len(sys.argv) > 1:                                        # python make_goldens.py realnvp_img flowpp_img: only these models
        make_models(set(sys.argv[1:]))
    else:
        make_indexmaps()
        make_ops()
        make_models()
        make_mainloop()
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print('%-24s %8.1f KB' % (f, os.path.getsize(os.path.join(HERE, f)) / 1024))
