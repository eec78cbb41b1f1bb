"""
Residual Flow (invertible residual blocks): flows/iresblock.py:17-301, flows/spectral_norm.py:5-72,
flows/modules.py:215-222 (LipSwish), flows/resflow.py:9-38 -- SURVEY.md section 8(a) row a15.

Everything runs on HIP kernels (csrc/resmlp.hip) for D <= 4: the evaluation-mode forward with all three log-det estimators
on the per-sample exact Jacobian of the 2->32->32->2 LipSwish MLP, the fixed-point inverse (flag-gated iteration launches),
spectral normalisation -- and the TRAINING step: the Russian-roulette value and the Neumann-series gradient estimator
(iresblock.py:59-109, :112-185) with the network's second derivatives in closed form (nf_resmlp_train_bwd) instead of nested
autograd sweeps; `hip_training = False` selects the PyTorch-autograd restatement (kept as the parity reference of the tests).
No BASELINE config exercises this family.  Same module / parameter names as the reference, same estimator semantics
(Russian-roulette series for training, `exact` / `fixed` / `unbias` for evaluation, fixed-point inverse with the
batch-global exit), same RNG consumption order (np.random.geometric, then a normal draw) so seeded runs are
comparable; `noise_on_cpu = True` draws the Hutchinson noise from the CPU generator (parity tests).
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _native as N
from .layers import ActNorm, Compose

HIP_MAX_D, SERIES_MAXK, SERIES_MAXS = 4, 64, 4


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
        self.hip_training = True            # training step on csrc/resmlp.hip (closed-form second derivatives); False: PyTorch autograd
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

    # ---- HIP path (evaluation / sampling) -----------------------------------------------------------------------------------
    def _hip_ok(self, x):
        sn = [m for m in self.g_fn if isinstance(m, SpectralNorm)]
        return (x.is_cuda and x.dim() == 2 and x.shape[1] <= HIP_MAX_D and x.dtype == torch.float32 and len(sn) == 3
                and sn[0].module.weight_bar.shape[0] == 32 and sn[1].module.weight_bar.shape == (32, 32))

    def _hip_weights(self, flags=None, it=0):
        """spectral normalisation of the three matrices in ONE launch (power iteration buffers updated in place)."""
        sn = [m for m in self.g_fn if isinstance(m, SpectralNorm)]
        for m in sn:
            if 'weight' in m.module._parameters:
                del m.module._parameters['weight']
        if getattr(self, '_weff', None) is None or self._weff[0].device != sn[0].module.weight_bar.device:
            self._weff = [torch.empty_like(m.module.weight_bar) for m in sn]
        P3 = ctypes.c_void_p * 3
        I3 = ctypes.c_int * 3
        wb = P3(*[m.module.weight_bar.data_ptr() for m in sn])
        uu = P3(*[m.module.weight_u.data_ptr() for m in sn])
        vv = P3(*[m.module.weight_v.data_ptr() for m in sn])
        we = P3(*[t.data_ptr() for t in self._weff])
        rows = I3(*[m.module.weight_bar.shape[0] for m in sn])
        cols = I3(*[m.module.weight_bar.shape[1] for m in sn])
        N.call('nf_spectral_weights', ctypes.addressof(wb), ctypes.addressof(uu), ctypes.addressof(vv), ctypes.addressof(we),
               ctypes.addressof(rows), ctypes.addressof(cols), 3, float(self.coeff), float(sn[0].eps),
               None if flags is None else flags.data_ptr(), it, N.stream())
        acts = [m for m in self.g_fn if isinstance(m, LipSwish)]
        return [self._weff[0], sn[0].module.bias, self._weff[1], sn[1].module.bias, self._weff[2], sn[2].module.bias,
                acts[0].beta, acts[1].beta]

    def _series(self, g_like, training):
        """(mode, noise, coef, n_terms, S): the estimator the reference would pick, noise drawn in its order."""
        if not training and self.estimator == 'exact':
            return 1, None, None, None, 0
        B, D = g_like.shape
        dev = g_like.device
        coef = np.zeros((SERIES_MAXS, SERIES_MAXK), dtype=np.float32)
        if not training and self.estimator == 'fixed':
            S, nts = 4, [8] * 4
            noise = self._randn_like(g_like, (B, S, D))
            for s_ in range(S):
                for k in range(1, 9):
                    coef[s_, k - 1] = (-1) ** (k + 1) / k
        else:
            if training:
                S, n_exact = 1, 1
            elif self.estimator == 'unbias':
                S, n_exact = 4, 8
            else:
                raise Exception('Unknown log-det estimator: %s' % (self.estimator, ))
            nts, vs, p = [], [], 0.5
            for s_ in range(S):
                n = n_exact + np.random.geometric(p)
                vs.append(self._randn_like(g_like))
                n = min(int(n), SERIES_MAXK)
                nts.append(n)
                for k in range(1, n + 1):
                    coef[s_, k - 1] = (-1) ** (k + 1) / (k * (1.0 - p) ** max(0, (k - n_exact) - 1))
            noise = torch.stack(vs, dim=1)
        return (2, noise.contiguous(), torch.from_numpy(coef).to(dev), torch.tensor(nts + [0] * (SERIES_MAXS - len(nts)),
                                                                                  dtype=torch.int32, device=dev), S)

    def _hip_logdet(self, w, x, y, ld, sign, training):
        mode, noise, coef, nts, S = self._series(x, training)
        B, D = x.shape
        N.call('nf_resmlp_fwd', N.ptr(x), *[N.ptr(t.detach()) for t in w], None if y is None else N.ptr(y), N.ptr(ld),
               float(sign), mode, None if noise is None else N.ptr(noise), None if coef is None else N.ptr(coef),
               None if nts is None else nts.data_ptr(), S, B, D, N.stream())

    # ---- flow surface ---------------------------------------------------------------------------------------------------
    def forward(self, x, log_df_dz):
        if not self.training and not torch.is_grad_enabled() and self._hip_ok(x):
            x = x.contiguous()
            w = self._hip_weights()
            np.random.geometric(0.5)                     # the reference evaluates (and discards) the Neumann surrogate first:
            self._randn_like(x)                          # keep the RNG streams aligned with it (iresblock.py:127)
            y = torch.empty_like(x)
            ld = log_df_dz.clone()
            self._hip_logdet(w, x, y, ld, 1.0, False)
            return y, ld
        if self.training and torch.is_grad_enabled() and self.hip_training and self._hip_ok(x):
            sn = [m for m in self.g_fn if isinstance(m, SpectralNorm)]
            acts = [m for m in self.g_fn if isinstance(m, LipSwish)]
            params = [sn[0].module.weight_bar, sn[0].module.bias, sn[1].module.weight_bar, sn[1].module.bias,
                      sn[2].module.weight_bar, sn[2].module.bias, acts[0].beta, acts[1].beta]
            return _ResidualBranchHip.apply(self, x.contiguous(), log_df_dz, *params)
        params = [p for p in self.g_fn.parameters() if p.requires_grad]
        g, logdet = _ResidualBranch.apply(self, x, *params)
        return x + g, log_df_dz + logdet

    def backward(self, z, log_df_dz):
        if self._hip_ok(z):
            z = z.contiguous()
            x = z.clone()
            B, D = z.shape
            flags = torch.zeros(101, dtype=torch.int32, device=z.device)
            for it in range(100):                        # flag-gated single-iteration launches: batch-global exit, no sync
                w = self._hip_weights(flags, it)
                N.call('nf_resmlp_fixed_point_step', N.ptr(z), N.ptr(x), *[N.ptr(t.detach()) for t in w], flags.data_ptr(),
                       it, float(self.ftol), B, D, N.stream())
            w = self._hip_weights()                      # the final g_fn(x) of the reference (one more power iteration)
            ld = log_df_dz.clone()
            self._hip_logdet(w, x, None, ld, -1.0, self.training)
            return x, ld
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


class _ResidualBranchHip(torch.autograd.Function):
    """(x + g(x), log_df_dz + log-det estimate) of a training-mode block on HIP kernels, and its autograd (iresblock.py:112-185):
    forward = spectral normalisation (one launch) + the Russian-roulette value on the per-sample Jacobian (nf_resmlp_fwd);
    backward = nf_resmlp_train_bwd (the Neumann-series gradient estimator with closed-form second derivatives) +
    nf_spectral_weights_bwd.  Noise and series lengths are drawn on the host in the reference's order: first the Neumann
    surrogate's (np.random.geometric, normal draw), then the value estimator's (iresblock.py:127-137)."""

    @staticmethod
    def forward(ctx, block, x, ld, *params):
        w = block._hip_weights()
        B, D = x.shape
        dev = x.device
        p = 0.5
        n1 = min(int(1 + np.random.geometric(p)), SERIES_MAXK)           # log_df_dz_neumann: n_exact = 1
        v1 = block._randn_like(x).contiguous()
        coef1 = np.zeros(SERIES_MAXK, dtype=np.float32)
        for k in range(1, n1 + 1):
            coef1[k - 1] = (-1) ** k / (1.0 - p) ** max(0, (k - 1) - 1)
        y = torch.empty_like(x)
        ld_out = ld.clone()
        block._hip_logdet(w, x, y, ld_out, 1.0, True)                     # draws (n, v) of the value estimator
        ctx.block = block
        ctx.n1 = n1
        ctx.save_for_backward(x, v1, torch.from_numpy(coef1).to(dev), *[t.detach().clone() for t in w])
        return y, ld_out

    @staticmethod
    def backward(ctx, d_y, d_ld):
        block = ctx.block
        x, v1, coef1 = ctx.saved_tensors[:3]
        w = ctx.saved_tensors[3:]
        B, D = x.shape
        d_y, d_ld = d_y.contiguous(), d_ld.contiguous()
        n_par = 32 * D + 32 + 1024 + 32 + D * 32 + D + 2
        from . import workspace as WS
        gp = WS.zeros(n_par, x.device)
        d_x = torch.empty_like(x)
        N.call('nf_resmlp_train_bwd', N.ptr(x), *[N.ptr(t) for t in w], N.ptr(v1), N.ptr(coef1), ctx.n1, N.ptr(d_y), N.ptr(d_ld),
               N.ptr(d_x), N.ptr(gp), B, D, N.stream())
        o = 0
        parts = []
        for n in (32 * D, 32, 1024, 32, D * 32, D, 1, 1):
            parts.append(gp[o:o + n])
            o += n
        gW1, gb1, gW2, gb2, gW3, gb3, gbe1, gbe2 = parts
        sn = [m for m in block.g_fn if isinstance(m, SpectralNorm)]
        gbar = [torch.zeros_like(m.module.weight_bar) for m in sn]
        P3, I3 = ctypes.c_void_p * 3, ctypes.c_int * 3
        wb = P3(*[m.module.weight_bar.data_ptr() for m in sn])
        uu = P3(*[m.module.weight_u.data_ptr() for m in sn])
        vv = P3(*[m.module.weight_v.data_ptr() for m in sn])
        ge = P3(gW1.data_ptr(), gW2.data_ptr(), gW3.data_ptr())
        gb = P3(*[t.data_ptr() for t in gbar])
        rows = I3(*[m.module.weight_bar.shape[0] for m in sn])
        cols = I3(*[m.module.weight_bar.shape[1] for m in sn])
        N.call('nf_spectral_weights_bwd', ctypes.addressof(wb), ctypes.addressof(uu), ctypes.addressof(vv), ctypes.addressof(ge),
               ctypes.addressof(gb), ctypes.addressof(rows), ctypes.addressof(cols), 3, float(block.coeff), float(sn[0].eps), N.stream())
        return (None, d_y + d_x, d_ld, gbar[0], gb1.clone(), gbar[1], gb2.clone(), gbar[2], gb3.clone(), gbe1.clone(), gbe2.clone())


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
