"""
Invertible residual block (Residual Flow) restated functionally over a reference-shaped state_dict.
Reference: flows/iresblock.py:17-301, flows/spectral_norm.py:5-72, flows/modules.py:215-222 (LipSwish),
flows/resflow.py:9-38.

The layer is stochastic (Hutchinson noise v ~ N(0, I), Russian-roulette series length n = n_exact + Geom(p)); this
restatement draws from the SAME global generators in the SAME order as the reference (np.random.geometric, then
torch.randn_like), so identically seeded runs are comparable number for number.

State keys of one block (prefix p): p+'g_fn.{0,2,4}.module.{bias,weight_bar,weight_u,weight_v}', p+'g_fn.{1,3}.beta'.
"""
import numpy as np
import torch


def _l2n(v, eps=1e-12):
    return v / (v.norm() + eps)


def spectral_weight(sd, p, coeff, eps=1.0e-5):
    """one power iteration (buffers updated in place), then clamp the spectral norm to `coeff` only if it is larger
    (flows/spectral_norm.py:26-43)."""
    u, v, w = sd[p + 'weight_u'], sd[p + 'weight_v'], sd[p + 'weight_bar']
    h = w.shape[0]
    wm = w.reshape(h, -1)
    with torch.no_grad():
        v.copy_(_l2n(torch.mv(wm.t().detach(), u)))
        u.copy_(_l2n(torch.mv(wm.detach(), v)))
    sigma = u.dot(wm.mv(v))
    scale = coeff / (sigma + eps)
    return w * scale.expand_as(w) if bool(scale < 1.0) else w


def lipswish(x, beta):
    return x * torch.sigmoid(beta * x) / 1.1                          # modules.py:221-222


def g_fn(x, sd, p, coeff, n_layers=2):
    """Sequential(SN(Linear), LipSwish, SN(Linear), LipSwish, SN(Linear))  (iresblock.py:258-278)"""
    h = x
    for i in range(n_layers + 1):
        q = p + 'g_fn.%d.module.' % (2 * i)
        h = torch.nn.functional.linear(h, spectral_weight(sd, q, coeff), sd[q + 'bias'])
        if i != n_layers:
            h = lipswish(h, sd[p + 'g_fn.%d.beta' % (2 * i + 1)])
    return h


# ---- log-det estimators (iresblock.py:17-109) ------------------------------------------------------------------------
def logdet_exact(g, z):
    D = z.shape[1]
    jac = torch.stack([torch.autograd.grad(g[:, i].sum(), z, create_graph=True, retain_graph=True)[0] for i in range(D)],
                      dim=1)
    return torch.logdet(torch.eye(D) + jac)


def logdet_fixed(g, z, n_samples=1, n_power_series=8):
    v = torch.randn([g.shape[0], n_samples, g.shape[1]])
    total, w = 0.0, v.clone()
    for k in range(1, n_power_series + 1):
        w = torch.stack([torch.autograd.grad(g, z, grad_outputs=w[:, i, :], retain_graph=True, create_graph=True)[0]
                         for i in range(n_samples)], dim=1)
        total = total + (-1) ** (k + 1) * (torch.einsum('bnd,bnd->bn', w, v) / k)
    return torch.mean(total, dim=1)


def logdet_unbias(g, z, n_samples=1, p=0.5, n_exact=1, is_training=True):
    res = 0.0
    for _ in range(n_samples):
        n = n_exact + np.random.geometric(p)
        v = torch.randn_like(g)
        w, acc = v, 0.0
        for k in range(1, n + 1):
            w = torch.autograd.grad(g, z, w, create_graph=is_training, retain_graph=True)[0]
            cdf = (1.0 - p) ** max(0, (k - n_exact) - 1)
            acc = acc + (-1) ** (k + 1) * (torch.sum(w * v, dim=1) / (k * cdf))
        res = res + acc
    return res / n_samples


def logdet_neumann(g, z, n_samples=1, p=0.5, n_exact=1):
    res = 0.0
    for _ in range(n_samples):
        n = n_exact + np.random.geometric(p)
        v = torch.randn_like(g)
        w, s = v, v
        with torch.no_grad():
            for k in range(1, n + 1):
                w = torch.autograd.grad(g, z, w, retain_graph=True)[0]
                cdf = (1.0 - p) ** max(0, (k - n_exact) - 1)
                s = s + ((-1) ** k / cdf) * w
        s = torch.autograd.grad(g, z, s, create_graph=True)[0]
        res = res + torch.sum(s * v, dim=1)
    return res / n_samples


def pick_estimator(training, name):
    """iresblock.py:213-227"""
    if training:
        return lambda g, z: logdet_unbias(g, z, 1, is_training=True)
    if name == 'exact':
        return logdet_exact
    if name == 'fixed':
        return lambda g, z: logdet_fixed(g, z, n_samples=4, n_power_series=8)
    if name == 'unbias':
        return lambda g, z: logdet_unbias(g, z, n_samples=4, n_exact=8, is_training=False)
    raise Exception('Unknown log-det estimator: %s' % (name, ))


def iresblock_forward(x, ld, sd, p, coeff, training, estimator):
    """InvertibleResBlockBase.forward through MemorySavedLogDetEstimator (iresblock.py:112-185, :229-234).

    Gradient semantics of the reference: d(loss)/d(x, theta) = grad(g; dL_dg) + dL_dlogdet[0] * grad(neumann surrogate).
    Restated as a straight-through: the returned log-det carries the VALUE of the chosen estimator and the GRADIENT of
    the Neumann surrogate scaled by the FIRST sample's upstream gradient (the reference's `dL_dlogdetJg[0]`)."""
    names = [k for k in sd if k.startswith(p + 'g_fn.') and (k.endswith('weight_bar') or k.endswith('bias')
                                                              or k.endswith('beta'))]
    with torch.enable_grad():                                          # iresblock.py:123 (works under no_grad callers)
        if not x.requires_grad:
            x = x.detach().requires_grad_(True)
        g = g_fn(x, sd, p, coeff)
        surrogate = logdet_neumann(g, x)                               # RNG: geometric, randn (drawn even in eval)
        value = pick_estimator(training, estimator)(g, x)              # RNG: geometric, randn
    if training and torch.is_grad_enabled():
        logdet = _FirstSampleScaled.apply(surrogate, value.detach())
        return x + g, ld + logdet
    return (x + g).detach(), ld + value.detach()


class _FirstSampleScaled(torch.autograd.Function):
    """forward: `value`; backward: routes g_out[0] (the reference uses dL_dlogdetJg[0] for EVERY sample,
    iresblock.py:169-173) into the surrogate."""

    @staticmethod
    def forward(ctx, surrogate, value):
        return value.clone()

    @staticmethod
    def backward(ctx, g_out):
        return g_out[0].expand_as(g_out).clone(), None


def iresblock_inverse(z, ld, sd, p, coeff, training, estimator, ftol=1.0e-4, n_iters=100):
    """fixed-point inverse with the batch-global exit (iresblock.py:236-255)."""
    x = z.clone()
    with torch.enable_grad():                                          # iresblock.py:241
        for _ in range(n_iters):
            x = x.detach()
            g = g_fn(x, sd, p, coeff)
            x, prev = z - g, x
            if bool(torch.all(torch.abs(x - prev) < ftol)):
                break
        x = x.detach().requires_grad_(True)
        g = g_fn(x, sd, p, coeff)
        logdet = pick_estimator(training, estimator)(g, x)
    return x.detach(), ld - logdet.detach()
