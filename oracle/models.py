"""
Layer stacks (RealNVP, Glow, Flow++, MAF) restated as a flat *plan* -- a list of
(op, key-prefix, attributes) -- executed over a reference-shaped ``state_dict``.

Reference: flows/glow.py:10-68, flows/realnvp.py:9-63, flows/flowpp.py:9-78,
flows/maf.py:88-148, flows/modules.py:325-339 (Compose).
"""
import numpy as np
import torch

from . import indexmaps as im
from . import iresblock as ires
from . import nets
from . import transforms as tf


def _coupling_mode(dims, masking):
    if len(dims) == 1:
        return im.MODE_1D
    if len(dims) == 3 and masking == 'checkerboard':
        return im.MODE_CHECKER
    if len(dims) == 3 and masking == 'channelwise':
        return im.MODE_CHANNEL
    raise Exception('unsupported combination of masking and dimension: %s, %s' % (masking, str(dims)))  # coupling.py:29


def build_plan(kind, dims, datatype, layers, mixtures=None, spnorm_coeff=0.9, logdet='unbias'):
    """returns the list of layers exactly in ``Compose`` order; index == position in net.layers."""
    dims = tuple(dims)
    plan = []

    def add(op, **attrs):
        plan.append(dict(op=op, prefix='net.layers.%d.' % len(plan), **attrs))

    def flow_step(d, masking, odd):
        if kind == 'realnvp':
            add('flow_bn', dims=d)
            add('affine', dims=d, mode=_coupling_mode(d, masking), odd=odd)
        elif kind == 'glow':
            add('actnorm', dims=d)
            add('invconv', dims=d)
            add('affine', dims=d, mode=_coupling_mode(d, masking), odd=odd)
        elif kind == 'flowpp':
            add('actnorm', dims=d)
            if len(d) == 3:
                add('invconv', dims=d)            # flowpp.py:22-23 (image) vs :66-68 (density: no inv-1x1)
            add('mixlog', dims=d, mode=_coupling_mode(d, masking), odd=odd, mixtures=mixtures)
        else:
            raise ValueError(kind)

    if kind == 'resflow':
        if datatype == 'image':
            raise NotImplementedError('residual flow for images is not supported (resflow.py:17-19)')
        for _ in range(layers):                   # resflow.py:22-28
            add('actnorm', dims=dims)
            add('ires', dims=dims, coeff=spnorm_coeff, logdet=logdet)
        return plan

    if kind == 'maf':
        if datatype == 'image':
            raise NotImplementedError('MAF for images is not supported (maf.py:130-132)')
        for _ in range(layers):                   # maf.py:136-138
            add('flow_bn', dims=dims)
            add('ar', dims=dims)
        return plan

    if datatype == 'image':                       # glow.py:18-51, realnvp.py:17-46, flowpp.py:16-61
        add('logit', eps=0.01)
        mid = dims
        while max(mid[1], mid[2]) > 8:
            for i in range(layers):
                flow_step(mid, 'checkerboard', i % 2 != 0)
            add('squeeze2d')
            mid = (mid[0] * 4, mid[1] // 2, mid[2] // 2)
            for i in range(layers):
                flow_step(mid, 'channelwise', i % 2 != 0)
        for i in range(layers + 1):
            flow_step(mid, 'checkerboard', i % 2 != 0)
        while mid[1] != dims[1] or mid[2] != dims[2]:
            add('unsqueeze2d')
            mid = (mid[0] // 4, mid[1] * 2, mid[2] * 2)
    else:                                         # glow.py:53-58 etc.
        for i in range(layers):
            flow_step(dims, 'checkerboard', i % 2 != 0)
    return plan


class FlowOracle:
    """Executes a plan.  Mutable state lives in ``self.sd`` (tensors named like the reference state_dict)
    plus ``self.actnorm_initialized`` (the reference keeps it as a plain attribute, modules.py:235)."""

    def __init__(self, kind, dims, datatype, layers, sd, mixtures=None, training=True, actnorm_initialized=False,
                 mask_rng=None, spnorm_coeff=0.9, logdet='unbias'):
        self.kind, self.dims, self.datatype = kind, tuple(dims), datatype
        self.plan = build_plan(kind, dims, datatype, layers, mixtures, spnorm_coeff, logdet)
        self.sd = sd
        self.training = training
        self.actnorm_initialized = {L['prefix']: bool(actnorm_initialized) for L in self.plan if L['op'] == 'actnorm'}
        self.mask_rng = mask_rng

    # -- helpers ---------------------------------------------------------------------------------------------------
    def parameters(self):
        """trainable leaves, reference naming (requires_grad as in the reference)."""
        frozen = ('.P', '.I', '.pivots', '.L_mask', '.U_mask', '.sign_s', 'running_mean', 'running_var',
                  'num_batches_tracked', 'batch_mean', 'batch_var', '.perm', 'weight_u', 'weight_v', 'module.weight')
        out = {}
        for k, v in self.sd.items():
            if any(k.endswith(f) for f in frozen) or not v.is_floating_point():
                continue
            if k.endswith('log_gamma') or (k.endswith('.beta') and (k[:-4] + 'log_gamma') in self.sd):
                continue                              # models build BatchNorm(affine=False): buffers
            out[k] = v
        return out

    def requires_grad_(self, flag=True):
        for v in self.parameters().values():
            v.requires_grad_(flag)
        return self

    def _conditioner_in(self, z, L):
        return im.split(z, L['mode'], L['odd'])[1]

    def _affine_params(self, z, L):
        z1 = self._conditioner_in(z, L)
        p = L['prefix'] + 'net.'
        return (nets.mlp if z.dim() == 2 else nets.convnet)(z1, self.sd, p, self.training)

    def _mixlog_sections(self, L):
        d, K = L['dims'], L['mixtures']
        if len(d) == 1:
            oc = d[0] - d[0] // 2 if not L['odd'] else d[0] - (d[0] + 1) // 2     # coupling.py:136-138
        elif L['mode'] == im.MODE_CHECKER:
            oc = d[0] * 2
        else:
            oc = d[0] // 2
        return [oc] * 2 + [oc * K] * 3

    def _made_masks(self, D, num_hidden):
        return nets.made_masks(D, num_hidden, 32, self.mask_rng)

    # -- one layer -------------------------------------------------------------------------------------------------
    def _apply(self, L, z, ld, inverse):
        sd, p, op = self.sd, L['prefix'], L['op']
        if op == 'logit':
            return tf.logit_inverse(z, ld) if inverse else tf.logit(z, ld, L['eps'])
        if op == 'squeeze2d':
            return (im.unsqueeze2d(z) if inverse else im.squeeze2d(z)), ld
        if op == 'unsqueeze2d':
            return (im.squeeze2d(z) if inverse else im.unsqueeze2d(z)), ld
        if op == 'actnorm':
            if not inverse and not self.actnorm_initialized[p]:       # modules.py:238-244 (forward only)
                with torch.no_grad():
                    ls, b = tf.actnorm_init(z)
                    sd[p + 'log_scale'].copy_(ls)
                    sd[p + 'bias'].copy_(b)
                self.actnorm_initialized[p] = True
            return tf.actnorm(z, ld, sd[p + 'log_scale'], sd[p + 'bias'], inverse)
        if op == 'invconv':
            if inverse:
                return tf.invconv_inverse(z, ld, sd[p + 'L'], sd[p + 'U'], sd[p + 'L_mask'], sd[p + 'U_mask'],
                                          sd[p + 'sign_s'], sd[p + 'log_s'], sd[p + 'pivots'])
            W = tf.invconv_weight(sd[p + 'P'], sd[p + 'L'], sd[p + 'U'], sd[p + 'I'], sd[p + 'L_mask'],
                                  sd[p + 'U_mask'], sd[p + 'sign_s'], sd[p + 'log_s'])
            return tf.invconv(z, ld, W, sd[p + 'log_s'])
        if op == 'flow_bn':
            if self.training:
                if not inverse:                                       # modules.py:284-296
                    with torch.no_grad():
                        mean, var = tf.flow_bn_stats(z)
                        sd[p + 'batch_mean'].copy_(mean)
                        sd[p + 'batch_var'].copy_(var)
                        sd[p + 'running_mean'].mul_(0.9).add_(sd[p + 'batch_mean'] * 0.1)
                        sd[p + 'running_var'].mul_(0.9).add_(sd[p + 'batch_var'] * 0.1)
                mean, var = sd[p + 'batch_mean'], sd[p + 'batch_var']
            else:
                mean, var = sd[p + 'running_mean'], sd[p + 'running_var']
            return tf.flow_bn(z, ld, mean, var, sd[p + 'log_gamma'], sd[p + 'beta'], inverse)
        if op == 'affine':
            params = self._affine_params(z, L)
            return tf.affine_coupling(z, ld, params, sd[p + 's_log_scale'], sd[p + 's_bias'], L['mode'], L['odd'],
                                      inverse)
        if op == 'mixlog':
            z1 = self._conditioner_in(z, L)
            params = nets.flowpp_net(z1, sd, p + 'net.', conv=(z.dim() == 4))
            return tf.mixlog_coupling(z, ld, params, self._mixlog_sections(L), L['mixtures'], sd[p + 'a_log_scale'],
                                      sd[p + 'a_bias'], L['mode'], L['odd'], inverse)
        if op == 'ar':
            return self._ar(L, z, ld, inverse)
        if op == 'ires':
            if inverse:
                return ires.iresblock_inverse(z, ld, sd, p, L['coeff'], self.training, L['logdet'])
            if not z.requires_grad:
                z = z.detach().requires_grad_(True)           # the estimators differentiate g w.r.t. its input
            return ires.iresblock_forward(z, ld, sd, p, L['coeff'], self.training, L['logdet'])
        raise ValueError(op)

    def _ar(self, L, z, ld, inverse):
        """AutoregressiveTransfrom (maf.py:88-119): num_hidden=3, base_filters=32 as built by MAF (maf.py:138)."""
        sd, p = self.sd, L['prefix']
        D, nh = L['dims'][0], 3

        def st(x):
            s = torch.tanh(nets.made(x, sd, p + 'net_s.', nh, self.training, self._made_masks(D, nh)))
            s = s * sd[p + 's_log_scale'] + sd[p + 's_bias']
            t = nets.made(x, sd, p + 'net_t.', nh, self.training, self._made_masks(D, nh))
            return s, t

        if not inverse:                                               # maf.py:101-107
            z = torch.mm(z, sd[p + 'perm'])
            s, t = st(z)
            return tf.affine_transform(z, s, t, ld)
        z = z.clone()                                                 # the reference mutates its input (Q5)
        for i in range(D):                                            # maf.py:111-116
            s, t = st(z)
            z[:, i] = ((z - t) * torch.exp(-s))[:, i]
            ld = ld - s[:, i]
        return torch.mm(z, sd[p + 'perm'].t()), ld

    # -- model surface (glow.py:62-68 etc.) ---------------------------------------------------------------------------
    def forward(self, z):
        ld = torch.zeros(z.shape[0], dtype=z.dtype)
        for L in self.plan:
            z, ld = self._apply(L, z, ld, False)
        return z, ld

    def backward(self, z):
        ld = torch.zeros(z.shape[0], dtype=z.dtype)
        for L in reversed(self.plan):
            z, ld = self._apply(L, z, ld, True)
        return z, ld

    def loss(self, y):
        z, ld = self.forward(y)
        return tf.nll_loss(z, ld)


def clone_state(sd, device='cpu'):
    return {k: v.detach().clone().to(device) for k, v in sd.items()}
