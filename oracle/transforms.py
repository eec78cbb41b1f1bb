"""
Floating-point bijectors of the hot path, restated functionally in torch-CPU fp32.
Every function returns NEW tensors (the reference updates ``log_df_dz`` in place in
several layers -- only the returned value is contractual).

``ld`` is the running log-det-Jacobian vector of shape (B,).
"""
import math

import torch
import torch.nn.functional as F

from . import indexmaps as im


def _per_sample_sum(x):
    return x.reshape(x.shape[0], -1).sum(dim=1)


def num_pixels(z):
    """np.prod(z.size()) // (B * C): flows/modules.py:248, :303, :479."""
    n = 1
    for d in z.shape[2:]:
        n *= int(d)
    return n


# ---------------------------------------------------------------------------------------------------------------------
# a1  affine coupling                                                       flows/coupling.py:104-122
# ---------------------------------------------------------------------------------------------------------------------

def affine_scale_shift(params, out_chs, s_log_scale, s_bias):
    """t = params[:, :oc]; s = tanh(params[:, oc:]) * s_log_scale + s_bias   (coupling.py:106-107)"""
    t = params[:, :out_chs]
    s = torch.tanh(params[:, out_chs:]) * s_log_scale + s_bias
    return s, t


def affine_transform(z0, s, t, ld):
    """z0 * exp(s) + t ; ld + sum(s)            (coupling.py:109-110; maf.py:105-106)"""
    return z0 * torch.exp(s) + t, ld + _per_sample_sum(s)


def affine_inverse(y0, s, t, ld):
    """exp(-s) * (y0 - t) ; ld - sum(s)         (coupling.py:119-120)"""
    return torch.exp(-s) * (y0 - t), ld - _per_sample_sum(s)


def affine_coupling(z, ld, params, s_log_scale, s_bias, mode, odd, inverse=False):
    """whole AbstractCoupling.forward/backward given the conditioner OUTPUT (coupling.py:32-43, :104-122).

    ``params`` is what ``self.net(z1)`` returned; the conditioner input is ``split(z)[1]``.
    """
    dims = tuple(z.shape[1:])
    z0, z1 = im.split(z, mode, odd)
    s, t = affine_scale_shift(params, z0.shape[1], s_log_scale, s_bias)
    z0, ld = (affine_inverse if inverse else affine_transform)(z0, s, t, ld)
    return im.merge(z0, z1, mode, odd, dims), ld


# ---------------------------------------------------------------------------------------------------------------------
# a4  ActNorm                                                               flows/modules.py:225-256
# ---------------------------------------------------------------------------------------------------------------------

def actnorm_init(z, eps=1.0e-5):
    """data-dependent initialisation (modules.py:238-244): unbiased std over (B, pixels) per channel."""
    zr = z.reshape(z.shape[0], z.shape[1], -1)
    log_scale = torch.log(torch.std(zr, dim=[0, 2]) + eps)
    bias = torch.mean(zr, dim=[0, 2])
    shape = [1, z.shape[1]] + [1] * (z.dim() - 2)
    return log_scale.reshape(shape), bias.reshape(shape)


def actnorm(z, ld, log_scale, bias, inverse=False):
    P = num_pixels(z)
    if not inverse:                                                # modules.py:246-250
        return (z - bias) / torch.exp(log_scale), ld - torch.sum(log_scale) * P
    return z * torch.exp(log_scale) + bias, ld + torch.sum(log_scale) * P    # modules.py:252-256


# ---------------------------------------------------------------------------------------------------------------------
# a5  invertible 1x1 convolution (PLU)                                      flows/modules.py:441-497
# ---------------------------------------------------------------------------------------------------------------------

def invconv_weight(P, L, U, I, L_mask, U_mask, sign_s, log_s):
    """W = P (L*L_mask + I) (U*U_mask + diag(sign_s exp(log_s)))            (modules.py:471-473)"""
    Lp = L * L_mask + I
    Up = U * U_mask + torch.diag(sign_s * torch.exp(log_s))
    return P @ Lp @ Up


def invconv(z, ld, W, log_s):
    """forward: per-pixel W @ z[b,:,p]; ld + sum(log_s) * P                 (modules.py:475-480)"""
    B, C = z.shape[0], z.shape[1]
    y = torch.matmul(W, z.reshape(B, C, -1)).reshape(z.shape)
    return y, ld + torch.sum(log_s) * num_pixels(z)


def invconv_inverse(y, ld, L, U, L_mask, U_mask, sign_s, log_s, pivots):
    """LU solve with the init-time LAPACK pivots (modules.py:485-495); no grad through it in the reference."""
    B, C = y.shape[0], y.shape[1]
    LU = L * L_mask + U * U_mask + torch.diag(sign_s * torch.exp(log_s))
    with torch.no_grad():
        x = torch.linalg.lu_solve(LU.unsqueeze(0), pivots.unsqueeze(0), y.reshape(B, C, -1))
    return x.reshape(y.shape), ld - torch.sum(log_s) * num_pixels(y)


# ---------------------------------------------------------------------------------------------------------------------
# a6  flow BatchNorm (statistics are constants for autograd)                 flows/modules.py:259-322
# ---------------------------------------------------------------------------------------------------------------------

def flow_bn_stats(x, eps=1.0e-5):
    """biased variance with eps INSIDE the stored value (modules.py:285-287)."""
    xr = x.reshape(x.shape[0], x.shape[1], -1)
    mean = xr.mean(dim=[0, 2], keepdim=True)
    var = (xr - mean).pow(2).mean(dim=[0, 2], keepdim=True) + eps
    shape = [1, x.shape[1]] + [1] * (x.dim() - 2)
    return mean.reshape(shape).detach(), var.reshape(shape).detach()


def flow_bn(x, ld, mean, var, log_gamma, beta, inverse=False):
    P = num_pixels(x)
    if not inverse:                                                # modules.py:300-305
        y = (x - mean) / torch.sqrt(var)
        y = y * torch.exp(log_gamma) + beta
        return y, ld + torch.sum(log_gamma - 0.5 * torch.log(var)) * P
    y = (x - beta) / torch.exp(log_gamma)                          # modules.py:315-320
    y = y * torch.sqrt(var) + mean
    return y, ld + torch.sum(-log_gamma + 0.5 * torch.log(var)) * P


# ---------------------------------------------------------------------------------------------------------------------
# a7  Logit                                                                 flows/modules.py:141-156, :19-32
# ---------------------------------------------------------------------------------------------------------------------

def _log_deriv_sigmoid(x):
    return x - 2.0 * F.softplus(x)                                 # modules.py:19-21


def logit(x, ld, eps):
    xc = torch.clamp(x, eps, 1.0 - eps)                            # modules.py:147
    inner = torch.logit(torch.clamp(xc, 1.0e-8, 1.0 - 1.0e-8))     # modules.py:31 (inner clamp is a no-op in fp32)
    return torch.logit(xc), ld + _per_sample_sum(-_log_deriv_sigmoid(inner))


def logit_inverse(y, ld):
    return torch.sigmoid(y), ld + _per_sample_sum(_log_deriv_sigmoid(y))     # modules.py:152-155


# ---------------------------------------------------------------------------------------------------------------------
# a8 / a9  mixture-of-logistics CDF and its bisection inverse               flows/modules.py:64-97, :186-212
# ---------------------------------------------------------------------------------------------------------------------

def sigmoid(x, ld, inverse=False):
    """Sigmoid.forward / .backward (flows/modules.py:125-138; helpers :19-32)"""
    if not inverse:
        return torch.sigmoid(x), ld + _per_sample_sum(_log_deriv_sigmoid(x))
    x = torch.clamp(x, 1.0e-8, 1.0 - 1.0e-8)                                   # modules.py:135
    y = torch.logit(torch.clamp(x, 1.0e-8, 1.0 - 1.0e-8))                      # log_deriv_logit's own clamp, modules.py:29-32
    return torch.logit(x), ld + _per_sample_sum(-_log_deriv_sigmoid(y))


def tanh(x, ld, inverse=False):
    """Tanh.forward / .backward (flows/modules.py:158-170; deriv_tanh :40-43, deriv_arctanh :58-61); Arctanh is the same pair swapped"""
    if not inverse:
        y = torch.tanh(x)
        return y, ld + _per_sample_sum(torch.log(1.0 - y * y))
    xc = torch.clamp(x, -1.0 + 1.0e-8, 1.0 - 1.0e-8)
    return torch.arctanh(x), ld + _per_sample_sum(torch.log(1.0 / (1.0 - xc * xc)))


def squeeze1d_layer(z, odd=False, inverse=False):
    """Squeeze1d.forward / .backward (flows/squeeze.py:63-83, :114-132)"""
    B, C = z.shape
    if not inverse:
        v = z.view(B, C // 2, 2)
        z0, z1 = v[:, :, 0], v[:, :, 1]
        if odd:
            z0, z1 = z1, z0
        return torch.cat([z0, z1], dim=1)
    z0, z1 = torch.split(z, C // 2, dim=1)
    if odd:
        z0, z1 = z1, z0
    return torch.stack([z0, z1], dim=-1).view(B, -1).contiguous()


def _mix_logpdf(x, logpi, mu, s):
    u = (x.unsqueeze(1) - mu) * torch.exp(-s)                      # modules.py:64-67
    return torch.logsumexp(logpi + (u - s - 2.0 * F.softplus(u)), dim=1)    # modules.py:76-85


def _mix_logcdf(x, logpi, mu, s):
    u = (x.unsqueeze(1) - mu) * torch.exp(-s)                      # modules.py:70-73
    return torch.logsumexp(logpi + F.logsigmoid(u), dim=1)         # modules.py:88-97


def mixlogcdf(x, ld, logpi, mu, s):
    """modules.py:190-194.  logpi is already log-softmaxed over the mixture axis (coupling.py:180)."""
    return torch.exp(_mix_logcdf(x, logpi, mu, s)), ld + _per_sample_sum(_mix_logpdf(x, logpi, mu, s))


def mixlogcdf_inverse(x, ld, logpi, mu, s, return_iters=False):
    """100-step bisection with a BATCH-GLOBAL early exit (modules.py:196-212)."""
    lo = torch.full_like(x, -1.0e3)
    hi = torch.full_like(x, 1.0e3)
    iters = 0
    for _ in range(100):
        iters += 1
        mid = (lo + hi) * 0.5
        val = torch.exp(_mix_logcdf(mid, logpi, mu, s))
        lo = torch.where(val < x, mid, lo)
        hi = torch.where(val > x, mid, hi)
        if bool(torch.all(torch.abs(hi - lo) < 1.0e-4)):
            break
    out = (lo + hi) * 0.5
    ld = ld - _per_sample_sum(_mix_logpdf(out, logpi, mu, s))
    return (out, ld, iters) if return_iters else (out, ld)


def mixlog_split_params(params, sections, n_mixtures, a_log_scale, a_bias):
    """coupling.py:177-182: split the conditioner output into (a, b, logpi, mu, s)."""
    a, b, logpi, mu, s = torch.split(params, sections, dim=1)
    a = torch.tanh(a) * a_log_scale + a_bias
    B = params.shape[0]
    C = tuple(a.shape[1:])
    logpi = F.log_softmax(logpi.reshape(B, n_mixtures, *C), dim=1)
    return a, b, logpi, mu.reshape(B, n_mixtures, *C), s.reshape(B, n_mixtures, *C)


def mixlog_coupling(z, ld, params, sections, n_mixtures, a_log_scale, a_bias, mode, odd, inverse=False,
                    logit_eps=1.0e-5):
    """MixLogAttnCoupling._transform / _inverse_transform given the conditioner output (coupling.py:172-210)."""
    dims = tuple(z.shape[1:])
    z0, z1 = im.split(z, mode, odd)
    a, b, logpi, mu, s = mixlog_split_params(params, sections, n_mixtures, a_log_scale, a_bias)
    if not inverse:
        z0, ld = mixlogcdf(z0, ld, logpi, mu, s)                   # coupling.py:184
        z0, ld = logit(z0, ld, logit_eps)                          # coupling.py:185
        z0 = z0 * torch.exp(a) + b                                 # coupling.py:187
        ld = ld + _per_sample_sum(a)                               # coupling.py:188
    else:
        z0 = torch.exp(-a) * (z0 - b)                              # coupling.py:204
        ld = ld - _per_sample_sum(a)                               # coupling.py:205
        z0, ld = logit_inverse(z0, ld)                             # coupling.py:207
        z0, ld = mixlogcdf_inverse(z0, ld, logpi, mu, s)           # coupling.py:208
    return im.merge(z0, z1, mode, odd, dims), ld


# ---------------------------------------------------------------------------------------------------------------------
# prior / loss of the training harness (pin for bits/dim)                    main.py:49-51, :85
# ---------------------------------------------------------------------------------------------------------------------

def standard_normal_logprob(z):
    """MultivariateNormal(0, I).log_prob on the flattened sample (main.py:49-51, :83-85)."""
    zf = z.reshape(z.shape[0], -1)
    D = zf.shape[1]
    return -0.5 * (zf * zf).sum(dim=1) - 0.5 * D * math.log(2.0 * math.pi)


def nll_loss(z, ld):
    return -1.0 * torch.mean(standard_normal_logprob(z) + ld)      # main.py:85


def bits_per_dim(loss, D):
    return float(loss) / (D * math.log(2.0))
