"""
Residual Flow (invertible residual blocks): flows/iresblock.py:17-301, flows/spectral_norm.py:5-72,
flows/modules.py:215-222 (LipSwish), flows/resflow.py:9-38 -- SURVEY.md section 8(a) row a15.

Status (round 1): the ActNorm in front of every block runs on the HIP path; the residual block itself is restated on
PyTorch-ROCm autograd (the log-det estimators are nested vector-Jacobian products of a 2->32->32->2 LipSwish MLP; no
BASELINE config exercises them).  Same module / parameter names as the reference, same estimator semantics
(Russian-roulette series for training, `exact` / `fixed` / `unbias` for evaluation, fixed-point inverse with the
batch-global exit), same RNG consumption order (np.random.geometric, then a normal draw) so seeded runs are
comparable; `noise_on_cpu = True` draws the Hutchinson noise from the CPU generator (parity tests).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .layers import ActNorm, Compose


class LipSwish(nn.Module):
    def __init__(self):
        super().__init__()
        self.beta = nn.Parameter(torch.ones([1], dtype=torch.float32))

    def forward(self, x):
        return x * torch.sigmoid(self.beta * x) / 1.1


def _l2n(v, eps=1e-12):
    return v / (v.norm() + eps)


class SpectralNorm(nn.Module):
    """one power iteration per call; the weight is scaled down to spectral norm `coeff` only when it exceeds it.
    Parameters / buffers as in the reference: module.{bias, weight_bar} + buffers weight_u, weight_v (the wrapped
    layer's own `weight` stays registered until the first call, exactly like the reference, so checkpoints taken
    before or after a first step both load)."""

    def __init__(self, module, coeff=0.97, eps=1.0e-5):
        super().__init__()
        self.module = module
        self.coeff = coeff
        self.eps = eps
        w = module.weight
        h = w.shape[0]
        u = w.data.new(h).normal_(0, 1)
        v = w.data.new(w.view(h, -1).shape[1]).normal_(0, 1)
        module.register_buffer('weight_u', _l2n(u))
        module.register_buffer('weight_v', _l2n(v))
        module.register_parameter('weight_bar', nn.Parameter(w.data.clone()))

    def effective_weight(self):
        m = self.module
        w = m.weight_bar
        wm = w.view(w.shape[0], -1)
        with torch.no_grad():
            m.weight_v.copy_(_l2n(torch.mv(wm.t(), m.weight_u)))
            m.weight_u.copy_(_l2n(torch.mv(wm, m.weight_v)))
        sigma = m.weight_u.dot(wm.mv(m.weight_v))
        scale = self.coeff / (sigma + self.eps)
        return w * torch.clamp(scale, max=1.0)               # == `w * scale if scale < 1 else w`, without the host sync

    def forward(self, x):
        m = self.module
        if 'weight' in m._parameters:                        # the reference deletes it on its first call
            del m._parameters['weight']
        return F.linear(x, self.effective_weight(), m.bias)


class InvertibleResLinear(nn.Module):
    def __init__(self, in_features, out_features, base_filters=32, n_layers=2, activation='lipswish', coeff=0.97,
                 ftol=1.0e-4, logdet_estimator='unbias'):
        super().__init__()
        if activation != 'lipswish':
            raise NotImplementedError('only the LipSwish residual branch is built by the reference models')
        self.coeff, self.ftol, self.estimator = coeff, ftol, logdet_estimator
        self.noise_on_cpu = False
        dims = [in_features] + [base_filters] * n_layers + [out_features]
        layers = []
        for i, (a, b) in enumerate(zip(dims[:-1], dims[1:])):
            layers.append(SpectralNorm(nn.Linear(a, b), coeff=coeff))
            if i != len(dims) - 2:
                layers.append(LipSwish())
        self.g_fn = nn.Sequential(*layers)

    # ---- noise / estimators -------------------------------------------------------------------------------------------
    def _randn_like(self, t, shape=None):
        shape = tuple(t.shape) if shape is None else shape
        if self.noise_on_cpu:
            return torch.randn(shape).to(t.device)
        return torch.randn(shape, device=t.device, dtype=t.dtype)

    @staticmethod
    def _vjp(g, z, w, create_graph):
        return torch.autograd.grad(g, z, w, create_graph=create_graph, retain_graph=True)[0]

    def _unbias(self, g, z, n_samples, n_exact, create_graph, p=0.5):
        total = 0.0
        for _ in range(n_samples):
            n = n_exact + np.random.geometric(p)
            v = self._randn_like(g)
            w, acc = v, 0.0
            for k in range(1, n + 1):
                w = self._vjp(g, z, w, create_graph)
                acc = acc + (-1) ** (k + 1) * (torch.sum(w * v, dim=1) / (k * (1.0 - p) ** max(0, (k - n_exact) - 1)))
            total = total + acc
        return total / n_samples

    def _neumann(self, g, z, n_exact=1, p=0.5):
        n = n_exact + np.random.geometric(p)
        v = self._randn_like(g)
        w, s = v, v
        with torch.no_grad():
            for k in range(1, n + 1):
                w = self._vjp(g, z, w, False)
                s = s + ((-1) ** k / (1.0 - p) ** max(0, (k - n_exact) - 1)) * w
        return torch.sum(self._vjp(g, z, s, True) * v, dim=1)

    def _fixed(self, g, z, n_samples=4, n_power_series=8):
        v = self._randn_like(g, (g.shape[0], n_samples, g.shape[1]))
        total, w = 0.0, v.clone()
        for k in range(1, n_power_series + 1):
            w = torch.stack([self._vjp(g, z, w[:, i, :], True) for i in range(n_samples)], dim=1)
            total = total + (-1) ** (k + 1) * (torch.einsum('bnd,bnd->bn', w, v) / k)
        return total.mean(dim=1)

    @staticmethod
    def _exact(g, z):
        D = z.shape[1]
        jac = torch.stack([torch.autograd.grad(g[:, i].sum(), z, create_graph=True, retain_graph=True)[0]
                           for i in range(D)], dim=1)
        return torch.logdet(torch.eye(D, device=z.device, dtype=z.dtype) + jac)

    def _estimate(self, g, z):
        if self.training:
            return self._unbias(g, z, 1, 1, True)
        if self.estimator == 'exact':
            return self._exact(g, z)
        if self.estimator == 'fixed':
            return self._fixed(g, z)
        if self.estimator == 'unbias':
            return self._unbias(g, z, 4, 8, False)
        raise Exception('Unknown log-det estimator: %s' % (self.estimator, ))

    # ---- flow surface ---------------------------------------------------------------------------------------------------
    def forward(self, x, log_df_dz):
        params = [p for p in self.g_fn.parameters() if p.requires_grad]
        g, logdet = _ResidualBranch.apply(self, x, *params)
        return x + g, log_df_dz + logdet

    def backward(self, z, log_df_dz):
        x = z.clone()
        with torch.enable_grad():
            for _ in range(100):
                x = x.detach()
                x, prev = z - self.g_fn(x), x
                if bool(torch.all(torch.abs(x - prev) < self.ftol)):     # batch-global exit (iresblock.py:248)
                    break
            x = x.detach().requires_grad_(True)
            logdet = self._estimate(self.g_fn(x), x)
        return x.detach(), log_df_dz - logdet.detach()


class _ResidualBranch(torch.autograd.Function):
    """memory-saving evaluation of (g(x), log-det estimate): the gradient of the log-det comes from the Neumann-series
    surrogate and is computed eagerly in forward, so the estimator's own graph is never kept (iresblock.py:112-185).
    Like the reference, the upstream log-det gradient is taken from the FIRST sample for the whole batch."""

    @staticmethod
    def forward(ctx, block, x, *params):
        ctx.training = block.training
        with torch.enable_grad():
            xd = x.detach().requires_grad_(True)
            g = block.g_fn(xd)
            surrogate = block._neumann(g, xd)
            if block.training:
                grads = torch.autograd.grad(surrogate.sum(), [xd] + list(params), retain_graph=True, allow_unused=True)
                ctx.surrogate_grads = grads
            value = block._estimate(g, xd)
            ctx.g, ctx.xd, ctx.params = g, xd, params
        return g.detach(), value.detach()

    @staticmethod
    def backward(ctx, d_g, d_logdet):
        if not ctx.training:
            raise ValueError('the residual block must be in training mode to be differentiated')
        with torch.enable_grad():
            first = torch.autograd.grad(ctx.g, [ctx.xd] + list(ctx.params), grad_outputs=d_g, allow_unused=True)
        scale = d_logdet[0].detach()
        out = []
        for a, b in zip(first, ctx.surrogate_grads):
            if a is None and b is None:
                out.append(None)
            elif b is None:
                out.append(a)
            elif a is None:
                out.append(b * scale)
            else:
                out.append(a + b * scale)
        return (None, ) + tuple(out)


class ResFlow(nn.Module):
    """flows/resflow.py:9-38: [ActNorm, InvertibleResLinear] x layers, density data only."""

    def __init__(self, dims, datatype=None, cfg=None):
        super().__init__()
        self.dims = tuple(dims)
        self.n_layers = cfg.layers
        if datatype == 'image':
            raise NotImplementedError('Sorry, residual flow for image generation is not supported!')
        layers = []
        for _ in range(self.n_layers):
            layers.append(ActNorm(self.dims))
            layers.append(InvertibleResLinear(self.dims[0], self.dims[0], coeff=cfg.spnorm_coeff,
                                              logdet_estimator=cfg.logdet))
        self.net = Compose(layers)

    def _zero_ld(self, z):
        return torch.zeros(z.size(0), dtype=z.dtype, device=z.device)

    def forward(self, z):
        return self.net(z, self._zero_ld(z))

    def backward(self, z):
        return self.net.backward(z, self._zero_ld(z))
