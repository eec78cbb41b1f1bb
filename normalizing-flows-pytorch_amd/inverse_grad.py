"""
The inverse flow WITH an autograd graph (``net.backward(z)`` when the input requires grad, or inside ``differentiable_inverse()``).

The sampling kernels of the engine build no graph: ``Model.sample_y`` (main.py:109-116) never differentiates them and a graph would only
cost memory there.  A caller who does differentiate the inverse direction (a reverse-KL objective, a latent-space optimisation) gets the
reference's graph instead of an exception: every layer's inverse formula as the reference writes it, composed of the engine's own
forward-direction pieces that already carry autograd -- the conditioners (HIP kernels with hand-written backward), the split gathers
(``functional.half_gather``), the space-to-depth maps -- and framework elementwise ops ON THE GPU for the half a dozen arithmetic lines
in between.  No CPU path: tensors stay where they are.

What the reference's OWN graph contains is reproduced, no more:
  * ``InvertibleConv1x1.backward`` solves under ``torch.no_grad()`` (flows/modules.py:485-492): the gradient of its input stops there in
    the reference too; only the log-det term ``- sum(log_s) * pixels`` (outside the no_grad block, :494-495) is differentiable;
  * flow ``BatchNorm`` statistics are buffers (flows/modules.py:309-316): constants of the graph;
  * ``MixLogCDF.backward`` returns the bisection's midpoint, a constant of the graph (flows/modules.py:196-208: ``lo`` / ``hi`` are built
    from ``torch.where`` over constants); the coupling's parameters receive gradient through the affine part, the sigmoid's log-det and
    ``- log pdf(x)`` (flows/coupling.py:192-210, flows/modules.py:209-212).
Not served (NotImplementedError, as before): the MAF inverse (the reference writes columns of ``z`` in place between conditioner calls,
flows/maf.py:109-119: its own graph does not survive that) and the fixed-point inverse of the residual blocks.
"""
import contextlib

import numpy as np
import torch
import torch.nn.functional as F

from . import functional as NF

_FORCED = [False]


@contextlib.contextmanager
def differentiable_inverse():
    """inside this context ``net.backward(z)`` records the graph even when ``z`` itself does not require grad (gradients of the PARAMETERS
    through the inverse direction: reverse-KL training).  Outside it the graph is recorded only when the input requires grad -- the
    reference's ``sample_y`` (main.py:113) runs with autograd on and needs none."""
    _FORCED.append(True)
    try:
        yield
    finally:
        _FORCED.pop()


def wanted(z, log_df_dz):
    return torch.is_grad_enabled() and (_FORCED[-1] or z.requires_grad or log_df_dz.requires_grad)


_MAPS = {}


def _maps(shape, mode, odd, device):
    """flat per-sample positions of the two halves of the split (an arange pushed through the engine's own gather: exact in fp32 for any
    sample below 2^24 elements)"""
    key = (tuple(shape[1:]), int(mode), bool(odd), str(device))
    m = _MAPS.get(key)
    if m is None:
        n = int(np.prod(shape[1:]))
        ar = torch.arange(n, dtype=torch.float32, device=device).reshape((1, ) + tuple(shape[1:]))
        with torch.no_grad():
            i0 = NF.half_gather(ar, 0, mode, odd).reshape(-1).long()
            i1 = NF.half_gather(ar, 1, mode, odd).reshape(-1).long()
        m = _MAPS[key] = (i0, i1)
    return m


def _split(z, mode, odd):
    i0, i1 = _maps(z.shape, mode, odd, z.device)
    flat = z.reshape(z.shape[0], -1)
    from .functional import _half_shape
    hs = _half_shape(z, mode)
    return flat[:, i0].reshape(hs), flat[:, i1].reshape(hs)


def _merge(z0, z1, like, mode, odd):
    i0, i1 = _maps(like.shape, mode, odd, like.device)
    B = like.shape[0]
    out = torch.empty(B, i0.numel() + i1.numel(), dtype=z0.dtype, device=z0.device)
    out = out.index_copy(1, i0, z0.reshape(B, -1)).index_copy(1, i1, z1.reshape(B, -1))
    return out.reshape(like.shape)


def _per_sample(t):
    return t.reshape(t.shape[0], -1).sum(dim=1)


def _pixels(z):
    return int(np.prod(z.shape)) // (z.shape[0] * z.shape[1])


def layer_inverse(layer, y, ld):
    """(x, ld) of ONE layer's inverse with the reference's graph"""
    from . import layers as L
    if isinstance(layer, L.Compose):
        for sub in reversed(layer.layers):
            y, ld = layer_inverse(sub, y, ld)
        return y, ld
    if isinstance(layer, L.Identity):
        return y, ld
    if isinstance(layer, L.Logit):                                   # modules.py:152-155
        return torch.sigmoid(y), ld + _per_sample(y - 2.0 * F.softplus(y))
    if isinstance(layer, L.ActNorm):                                 # modules.py:252-256
        return y * torch.exp(layer.log_scale) + layer.bias, ld + torch.sum(layer.log_scale) * _pixels(y)
    if isinstance(layer, L.BatchNorm):                               # modules.py:309-322 (statistics: buffers)
        mean, var = layer._stats()
        x = (y - layer.beta) / torch.exp(layer.log_gamma)
        x = x * torch.sqrt(var) + mean
        return x, ld + torch.sum(-layer.log_gamma + 0.5 * torch.log(var)) * _pixels(y)
    if isinstance(layer, L.InvertibleConv1x1):                       # modules.py:484-497: the solve is under no_grad there as well
        with torch.no_grad():
            x, _ = layer.backward(y.detach(), torch.zeros(y.shape[0], dtype=y.dtype, device=y.device))
        return x, ld - torch.sum(layer.log_s, dim=0) * _pixels(y)
    if isinstance(layer, (L.Squeeze2d, L.Unsqueeze2d, L.Squeeze1d, L.Unsqueeze1d)):
        x, _ = layer.backward(y, ld)                                 # index maps with autograd of their own (functional._SpaceDepth / _Squeeze1d)
        return x, ld
    if isinstance(layer, L.AffineCoupling):                          # coupling.py:114-122
        y0, y1 = _split(y, layer.mode, layer.odd)
        params = layer.net(y1.contiguous())
        oc = layer.out_chs
        t = params[:, :oc]
        s = torch.tanh(params[:, oc:]) * layer.s_log_scale + layer.s_bias
        x0 = torch.exp(-s) * (y0 - t)
        return _merge(x0, y1, y, layer.mode, layer.odd), ld - _per_sample(s)
    if isinstance(layer, L.AdditiveCoupling):                        # coupling.py:73-79
        y0, y1 = _split(y, layer.mode, layer.odd)
        return _merge(y0 - layer.net_t(y1.contiguous()), y1, y, layer.mode, layer.odd), ld
    if isinstance(layer, L.MixLogAttnCoupling):                      # coupling.py:192-210
        K = layer.n_mixtures
        z0, z1 = _split(y, layer.mode, layer.odd)
        params = layer.conditioner_of(z1.contiguous())
        a, b, logpi, mu, s = torch.split(params, layer.sections, dim=1)
        a = torch.tanh(a) * layer.a_log_scale + layer.a_bias
        B, C = z0.shape[0], tuple(z0.shape[1:])
        logpi = F.log_softmax(logpi.reshape((B, K) + C), dim=1)
        mu, s = mu.reshape((B, K) + C), s.reshape((B, K) + C)
        v = torch.exp(-a) * (z0 - b)
        ld = ld - _per_sample(a)
        ld = ld + _per_sample(v - 2.0 * F.softplus(v))               # Logit.backward, modules.py:152-155
        u = torch.sigmoid(v)
        with torch.no_grad():                                        # the bisection's midpoint: a constant of the reference's graph too
            x, _ = NF.mixlogcdf(u.detach().contiguous(), logpi.detach().contiguous(), mu.detach().contiguous(), s.detach().contiguous(),
                                torch.zeros(B, dtype=y.dtype, device=y.device), inverse=True)
        uu = (x.unsqueeze(1) - mu) * torch.exp(-s)                   # mix_logistic_logpdf, modules.py:64-78
        logpdf = torch.logsumexp(logpi + uu - s - 2.0 * F.softplus(uu), dim=1)
        return _merge(x, z1, y, layer.mode, layer.odd), ld - _per_sample(logpdf)
    raise NotImplementedError('the inverse of %s has no autograd graph in this engine (see inverse_grad.py: the MAF and residual-block '
                              'inverses); detach the input or differentiate the forward direction' % type(layer).__name__)
